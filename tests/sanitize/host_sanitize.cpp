// Sanitizer harness for the GPU-free host code of libsvo_hip.so (tests only): compiled by tests/test_host_sanitizers.py with
// g++ -fsanitize=address,undefined together with host/{lm,linalg,kitti_io,draw,synth}.cpp and run on the CPU
// (GPU AddressSanitizer is not available on the target pool).  Exercises: the image decoders on every file in a
// directory of well-formed and hostile inputs, the poses reader, ATE, the dense Cholesky, the track rasteriser with
// end points far outside the image, the synthetic renderer, and the LM step control (svo_lm_solve) over a small
// least-squares backend with rejections.  Exit code 0 = no sanitizer report, every call returned what it should.
#include <dirent.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "svo.h"

// svo_ba_default_options lives in csrc/ba.hip (not part of this build)
extern "C" void svo_ba_default_options(svo_ba_options* o) {
  o->max_iterations = 50; o->max_time_s = 0.0; o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8; o->initial_radius = 1e4; o->max_features = 400; o->accumulation = 0;
}

namespace {
int fails = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

// Backend for svo_lm_solve: K = 2 poses (one free), one "landmark" scalar block eliminated away: payload1 is that of a
// synthetic quadratic + quartic cost in the 6 pose tangents, so that some steps get rejected.
struct Toy {
  double x[6], cand[6];
  double target[6];
  static double cost_at(const double* x, const double* t) {
    double c = 0;
    for (int i = 0; i < 6; ++i) { const double d = x[i] - t[i]; c += 0.5 * (1 + i) * d * d + 0.25 * d * d * d * d; }
    return c;
  }
};
int toy_linearize(void* u, double radius, int first, double* pay) {
  (void)radius; (void)first;
  Toy* T = static_cast<Toy*>(u);
  const int n = 6;
  memset(pay, 0, sizeof(double) * (n * n + 3 * n + 2));
  for (int i = 0; i < n; ++i) {
    const double d = T->x[i] - T->target[i];
    pay[i * n + i] = (1 + i) + 3 * d * d;      // S
    pay[n * n + n + i] = (1 + i) * d + d * d * d;  // g_c
    pay[n * n + 2 * n + i] = (1 + i) + 3 * d * d;  // diag U
  }
  pay[n * n + 3 * n] = Toy::cost_at(T->x, T->target);
  return 0;
}
int toy_step(void* u, const double* dc, const double* cand_poses7, double radius, const svo_lm_step_ctl* ctl, double* pay2,
             double* pay1_next, double* next_radius, int* next_at_candidate) {
  (void)cand_poses7; (void)radius; (void)pay1_next;
  Toy* T = static_cast<Toy*>(u);
  double step2 = 0, x2 = 0;
  for (int i = 0; i < 6; ++i) { T->cand[i] = T->x[i] + dc[i]; step2 += dc[i] * dc[i]; x2 += T->x[i] * T->x[i]; }
  pay2[0] = Toy::cost_at(T->cand, T->target);
  pay2[1] = 0.0; pay2[2] = 0.0; pay2[3] = 0.0;
  *next_radius = 0.0; *next_at_candidate = 0;
  (void)ctl;
  return 0;
}
int toy_accept(void* u) {
  Toy* T = static_cast<Toy*>(u);
  memcpy(T->x, T->cand, sizeof(T->x));
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  // ---- decoders over a directory of files; the test tells through the file name what to expect ("ok_" / "bad_")
  if (argc > 1) {
    DIR* d = opendir(argv[1]);
    CHECK(d != nullptr);
    std::vector<uint8_t> buf(1 << 20);
    int seen = 0;
    while (d) {
      dirent* e = readdir(d);
      if (!e) break;
      const std::string name = e->d_name;
      if (name.size() < 4 || (name.compare(0, 3, "ok_") && name.compare(0, 4, "bad_"))) continue;
      int w = -1, h = -1;
      const int rc = svo_image_read_gray((std::string(argv[1]) + "/" + name).c_str(), buf.data(), buf.size(), &w, &h);
      if (!name.compare(0, 3, "ok_")) CHECK(rc == SVO_OK && w > 0 && h > 0 && (size_t)w * h <= buf.size());
      else CHECK(rc != SVO_OK);
      ++seen;
    }
    if (d) closedir(d);
    CHECK(seen >= 10);
    // capacity smaller than the image
    int w = 0, h = 0;
    std::vector<uint8_t> tiny(8);
    CHECK(svo_image_read_gray((std::string(argv[1]) + "/ok_gray.png").c_str(), tiny.data(), tiny.size(), &w, &h) != SVO_OK);
    // poses file: 3 full rows and a truncated fourth
    std::vector<double> rt(12 * 8);
    int n = -1;
    CHECK(svo_kitti_read_poses((std::string(argv[1]) + "/poses.txt").c_str(), rt.data(), 8, &n) == SVO_OK && n == 3);
    CHECK(svo_kitti_read_poses((std::string(argv[1]) + "/poses.txt").c_str(), rt.data(), 2, &n) == SVO_OK && n == 2);
    CHECK(svo_kitti_read_poses((std::string(argv[1]) + "/does_not_exist.txt").c_str(), rt.data(), 8, &n) != SVO_OK);
  }
  // ---- ATE
  {
    double a[30], b[30], rmse = -1;
    for (int i = 0; i < 10; ++i) { a[3 * i] = i; a[3 * i + 1] = 0.1 * i * i; a[3 * i + 2] = sin(0.3 * i); }
    for (int i = 0; i < 10; ++i) { b[3 * i] = -a[3 * i + 1] + 5; b[3 * i + 1] = a[3 * i] - 2; b[3 * i + 2] = a[3 * i + 2] + 1; }  // rotated + shifted
    CHECK(svo_ate_rmse(a, b, 10, 0, &rmse) == SVO_OK && rmse < 1e-9);
    CHECK(svo_ate_rmse(a, b, 2, 0, &rmse) != SVO_OK);
  }
  // ---- dense Cholesky (panel-blocked, every n up to beyond a window's 6 (K - 1))
  for (int n = 1; n <= 40; ++n) {
    std::vector<double> A((size_t)n * n, 0.0), x(n), bb(n);
    for (int i = 0; i < n; ++i) {
      x[i] = 0.1 * (i + 1);
      for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = 1.0 / (1 + i + j) + (i == j ? n : 0);
    }
    for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += A[(size_t)i * n + j] * x[j]; bb[i] = s; }
    CHECK(svo_cholesky_solve(A.data(), bb.data(), n) == SVO_OK);
    for (int i = 0; i < n; ++i) CHECK(fabs(bb[i] - x[i]) < 1e-9);
  }
  {
    double A[4] = {1, 2, 2, 1}, b2[2] = {1, 1};
    CHECK(svo_cholesky_solve(A, b2, 2) == SVO_ERR_NUMERIC);
  }
  // ---- rasteriser with arrows far outside the image / non-finite end points
  {
    const int W = 97, H = 41;
    std::vector<uint8_t> g((size_t)W * H, 50), rgb((size_t)W * H * 3, 0);
    const float from[] = {10, 10, -5000, 3, 50, 20, 3, 3, 96, 40};
    const float to[] = {40, 30, 90000, -70000, NAN, 5, 3, 3, 1e30f, -1e30f};
    CHECK(svo_draw_track(g.data(), W, H, W, from, to, 5, rgb.data()) == SVO_OK);
    CHECK(svo_draw_track(g.data(), W, H, W - 1, from, to, 5, rgb.data()) != SVO_OK);
  }
  // ---- synthetic renderer (odd sizes)
  {
    svo_synth_params sp;
    svo_synth_default_params(&sp, 67, 45);
    std::vector<uint8_t> L(67 * 45), R(67 * 45);
    CHECK(svo_synth_render(&sp, 3, L.data(), R.data()) == SVO_OK);
    double rt[12];
    CHECK(svo_synth_pose(&sp, 3, rt) == SVO_OK);
  }
  // ---- LM step control over the toy backend: converges, rejects some steps, never reads out of bounds
  {
    Toy T;
    for (int i = 0; i < 6; ++i) { T.x[i] = 0; T.target[i] = (i % 2 ? -1.0 : 1.0) * (0.3 + 0.2 * i); }
    double poses[14] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    svo_lm_ops ops = {&T, toy_linearize, toy_step, toy_accept};
    svo_ba_summary sum;
    svo_lm_stats st;
    svo_ba_options opt;
    svo_ba_default_options(&opt);
    opt.initial_radius = 1e-2;  // small trust region first: exercises the radius growth path
    CHECK(svo_lm_solve(2, poses, &ops, &opt, &sum, &st) == SVO_OK);
    CHECK(sum.iterations > 2 && sum.final_cost < 1e-6 * (sum.initial_cost + 1e-30) + 1e-9);
    for (int i = 0; i < 6; ++i) CHECK(fabs(T.x[i] - T.target[i]) < 1e-3);
    CHECK(svo_lm_solve(0, poses, &ops, &opt, &sum, &st) == SVO_ERR_INVALID);
    int acc = -1; double nr = 0;
    CHECK(svo_lm_decide_step(10.0, 1.0, 100.0, 2.0, 9.0, 0.0, &acc, &nr) == SVO_OK && acc == 1 && nr == 100.0 / (1.0 / 3.0));
    CHECK(svo_lm_decide_step(10.0, 1.0, 100.0, 2.0, 11.0, 0.0, &acc, &nr) == SVO_OK && acc == 0 && nr == 50.0);
  }
  if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
  printf("host sanitize ok\n");
  return 0;
}
