// Pipeline group: ImageProcessor::process + BundleAdjuster::bundle_adjust (reference src/image_processor.cpp:18-163,
// src/feature_tracker.cpp:18-67, driver rule src/vo_node.cpp:141-148) for SEVERAL stereo streams ("lanes") driven by ONE host
// thread.  Every lane runs the reference's per-frame state machine — exactly the statements of host/pipeline.cpp's
// ImageProcessor::process, cut at the points where the host needs a number from the GPU — and an event loop advances all
// lanes: whatever stage several lanes reach in the same pass goes out as ONE launch (csrc/group_kernels.h, blockIdx.y =
// lane) and the solves of the lanes that hit a keyframe together are one ba_lm_kernel launch.  Lanes never wait for each
// other: a lane whose frame is no keyframe starts tracking its next frame while another lane is still in its PnP.
//
// Why (measured in round 2, profiles/r02_exp_launch_rate.txt, r02_exp_placement.txt, r02_kernel_stats_*): eight streams
// as eight host threads on 16-24 hardware queues pay 30-90 us per launch -> completion round trip (7-11 us with at most
// four queues) and their small kernels run 3-5x their solo time.  Here: one driver thread, four HIP streams (tracking +
// front end, keyframe chain, two for solves), a small pool of host workers that only assemble bundle-adjustment problems.
// Results per lane are bit for bit those of its own svo_pipeline (tests/test_group.py).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "chain_math.h"
#include "det_trig.h"
#include "group_kernels.h"
#include "ref_constants.h"
#include "svo.h"

namespace {

// cv::Rodrigues on a CV_32F rvec with declared arithmetic (host/det_trig.h): the same bits on the host, on the device and in the oracle
inline void rodrigues_f(const float* rv, float* R9) { svo_det_rodrigues_f(rv, R9); }

// Eigen::Quaternionf(Matrix3f) (src/image_processor.cpp:92), float arithmetic, row-major m.
void quat_from_R(const float* m, float* q /*wxyz*/) {
  float t = m[0] + m[4] + m[8];
  if (t > 0.f) {
    t = sqrtf(t + 1.0f);
    q[0] = 0.5f * t;
    t = 0.5f / t;
    q[1] = (m[7] - m[5]) * t; q[2] = (m[2] - m[6]) * t; q[3] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
    q[1 + i] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[1 + j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[1 + k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
}

enum LaneState { L_IDLE = 0, L_TRACK_WAIT, L_NEED_SOLVE, L_PNP_WAIT, L_TRI_WAIT, L_DONE };
enum BaState { BA_NONE = 0, BA_ASSEMBLING, BA_READY, BA_INFLIGHT, BA_HOST_SOLVING, BA_HOST_DONE };
enum Word { W_TRACK = 0, W_PNP, W_TRI, W_COUNT };
enum Counter { C_LK = 0, C_PNP, C_TRI, C_COUNT };

struct Lane {
  int lk_line = 0, chain_line = 0;  // the tracking / keyframe-chain line (stream) its launch in flight went to
  // ---- tracker state (FeatureTracker): feature set double-buffered on the device, mirrored in pinned memory
  float* d_xy[2] = {nullptr, nullptr}; float* d_init[2] = {nullptr, nullptr}; long long* d_ids[2] = {nullptr, nullptr};
  float* d_fwd = nullptr; float* d_par = nullptr; uint8_t* d_keep = nullptr;
  float* h_xy[2] = {nullptr, nullptr}; long long* h_ids[2] = {nullptr, nullptr};  // pinned mirrors of the kept set (double-buffered with the device set)
  float* h_kf_xy = nullptr; long long* h_kf_ids = nullptr;  // pinned: the feature set a keyframe hands to the tracker (src/image_processor.cpp:148-162)
  int* h_n = nullptr; float* h_av = nullptr;
  int cur = 0, n = 0, n_initial = 0;
  bool from_host = false;           // the next track reads its features from h_kf_* (tracker (re)initialised by a keyframe)
  const uint8_t* last_pyr = nullptr;
  const uint8_t* last_l0 = nullptr;  // level 0 of last_pyr: inside it, or (within the batch that produced it) the caller's image read in place
  uint8_t* d_own_pyr = nullptr;      // the lane's private clone of its last image's pyramid, used when a whole batch went by without tracking (see process_batch)
  // ---- PnP
  float4* d_store = nullptr; unsigned store_mask = 0;  // device-resident landmark store of the lane, keyed by feature id (get_world_points, src/bundle_adjuster.cpp:159-163)
  double* d_hyp_pose = nullptr; int* d_hyp_count = nullptr; unsigned long long* d_hyp_mask = nullptr;
  double* d_out = nullptr; int* d_nin = nullptr; int* d_inl = nullptr; float* d_trk_xy = nullptr;
  bool pnp_all = false;  // the lane's next PnP launch computes every hypothesis (the first few did not settle the adaptive cap)
  int* h_best = nullptr; int* h_bad = nullptr; double* h_out = nullptr; int* h_nin = nullptr; int* h_inl = nullptr;
  // ---- dedup / sparse stereo / triangulation
  float* d_disp = nullptr;
  int* h_tri_cnt = nullptr; float* h_tri_xy = nullptr; float* h_tri_xyz = nullptr;
  // ---- hand-over words and arrival counters
  int* words = nullptr;             // pinned, W_COUNT words 64 bytes apart
  int seq[W_COUNT] = {};
  unsigned* d_arrive = nullptr;     // device, C_COUNT counters 64 bytes apart (monotone)
  unsigned arrive_total[C_COUNT] = {0, 0, 0};
  // ---- graph + solve (BundleAdjuster)
  svo_ba* ba = nullptr;
  std::atomic<int> ba_state{BA_NONE};
  svo_ba_summary ba_summary{};
  int ba_rc = 0;
  int ba_launch = 0, ba_line = 0;  // which solve launch of the group carries this lane's solve, on which solve line
  unsigned long long ba_ready_seq = 0;  // when its assembled solve was first seen waiting for a launch (0: none waits); the order of admission
  bool has_keyframe = false;
  double solved_pose[7] = {1, 0, 0, 0, 0, 0, 0};
  int last_iterations = 0;
  // ---- ImageProcessor state
  float rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};
  // ---- the frame in flight
  int state = L_IDLE;
  bool queued = false;              // the stage the lane waits for has not been launched yet (its completion word still shows the previous launch)
  int frame = 0;                    // index into the batch
  int pending_from = -1;            // first frame of the batch whose pose waits for the running solve
  bool first_keyframe = false;      // the triangulation in flight is frame 0's (src/image_processor.cpp:30-58)
  SvoChainRec* d_chain = nullptr;   // what the lane's PnP launch leaves for the stereo + triangulation launch queued behind it (host/chain_math.h)
  double t_q = 0.0;                 // when the lane entered the queue of its next stage (microseconds of this call): the gather policy
  bool fused = false;               // the triangulation in flight rides behind a PnP launch: its results are consumed together with PnP's
  int m_tracked = 0, num_inliers = 0, best = -1;
  float rmat[9], quat[4];
  std::vector<long long> kf_tracked_ids; std::vector<float> kf_tracked_xy;  // the keyframe under construction
  std::vector<long long> ids64; std::vector<int64_t> new_ids;
};

struct Pool {  // host workers: they only assemble bundle-adjustment problems (and run the rare host-driven solve)
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::pair<Lane*, int>> jobs;  // (lane, 0: assemble / 1: host-driven solve)
  bool quit = false;
  int device = 0;
  void start(int n, int dev) {
    device = dev;
    for (int i = 0; i < n; ++i) th.emplace_back([this] { run(); });
  }
  void post(Lane* l, int what) {
    { std::lock_guard<std::mutex> g(mu); jobs.emplace_back(l, what); }
    cv.notify_one();
  }
  void run() {
    (void)hipSetDevice(device);
    for (;;) {
      std::pair<Lane*, int> j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return quit || !jobs.empty(); });
        if (quit && jobs.empty()) return;
        j = jobs.front(); jobs.pop_front();
      }
      Lane* l = j.first;
      if (j.second == 0) {
        l->ba_rc = svo_ba_solve_prepare(l->ba);
        l->ba_state.store(BA_READY, std::memory_order_release);
      } else {
        l->ba_rc = svo_ba_solve_finish(l->ba, &l->ba_summary);
        l->ba_state.store(BA_HOST_DONE, std::memory_order_release);
      }
    }
  }
  void stop() {
    { std::lock_guard<std::mutex> g(mu); quit = true; }
    cv.notify_all();
    for (auto& t : th) t.join();
    th.clear();
  }
};

}  // namespace

struct svo_pipeline_group {
  svo_ctx* ctx = nullptr;
  svo_pipeline_params prm{};
  int n_lanes = 0, max_batch = 0;
  int counted_lanes = 0;  // what this group added to the process-wide lane count (svo_ba_note_group_lanes)
  float K[9];
  std::vector<Lane*> lanes;
  std::vector<void*> dev_allocs, pin_allocs;
  // bus lines: a lane always rides the same tracking line and the same keyframe-chain line (stream order keeps its
  // consecutive stages coherent), solves take whichever solve line is free
  static constexpr int MAX_LINES = 8;
  int n_lk = 1, n_chain = 1, n_ba = 2;
  int n_cmp = 0;  // compact lines: st_ba[n_ba ..): solves the admission budget refuses leave at once in the one-workgroup form, on lines of their own
  int xcd_chunks = SVO_XCD_CHUNKS;
  bool xcd_map = true, xcd_map_tri = true;  // the XCD-aware item -> workgroup map of the tracking / stereo launches (group_kernels.h; SVO_GROUP_LK_XCD=0, SVO_GROUP_TRI_XCD=0: blockIdx = (item, lane))
  // SVO_TIMING: host time of the group thread inside the per-keyframe graph calls (ns), and keyframes seen
  double t_get_points = 0, t_add_keyframe = 0, t_finish = 0, t_loop = 0; long n_kf = 0; bool timing = false;
  double gather_us = 0.0;      // > 0: a stage's launch waits up to this long for the other lanes of its line that are still on their way
  double lk_overlap_us = 0.0;  // > 0: a second tracking line may depart once every launch in flight is at least this old (it is in its tail then)
  double lk_t0[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // departure time of the launch in flight on each tracking line (us since the call began)
  hipStream_t st_lk[MAX_LINES + 1] = {}, st_chain[MAX_LINES + 1] = {}, st_ba[MAX_LINES] = {};  // [MAX_LINES]: the express lines (see process_batch)
  bool express = false;
  int ba_launch_id = 0;
  unsigned long long ba_ready_counter = 0;
  // batch-wide front-end outputs
  float* d_corners = nullptr; int* d_ncorners = nullptr; uint8_t* d_pyr[2] = {nullptr, nullptr}; int pyr_cur = 0;
  size_t pyr_stride = 0;
  int* h_counts = nullptr;  // pinned: n_lanes x max_batch corner counts + status
  Pool pool;
  int pnp_iterations = 0, mask_words_cap = 0;
  // statistics of the last batch (launches by kind and the lanes they carried)
  long launches[6] = {0, 0, 0, 0, 0, 0}, lanes_carried[6] = {0, 0, 0, 0, 0, 0};
  // host-pointer / streaming entry: two pinned staging slots, their device twins, a copy stream (allocated on first use)
  uint8_t* h_stage[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [slot][left / right]
  uint8_t* d_stage[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  hipStream_t st_copy = nullptr;
  hipEvent_t ev_up[2] = {nullptr, nullptr};
  int up_batch[2] = {0, 0};  // frames per lane of the upload in flight / completed in the slot (0: none)
};

namespace {

int fail(svo_pipeline_group* g, const char* what, hipError_t e) {
  g->ctx->err = std::string("pipeline group: ") + what + ": " + hipGetErrorString(e);
  return SVO_ERR_HIP;
}

template <typename T>
int dev_alloc(svo_pipeline_group* g, T** p, size_t count) {
  const hipError_t e = hipMalloc((void**)p, sizeof(T) * (count ? count : 1));
  if (e != hipSuccess) return fail(g, "hipMalloc", e);
  g->dev_allocs.push_back(*p);
  return SVO_OK;
}

int word_ready(const Lane* l, int w) { return __atomic_load_n(&l->words[16 * w], __ATOMIC_ACQUIRE) == l->seq[w]; }

SvoPublish make_pub(Lane* l, int w, int counter, int nblocks) {
  SvoPublish p;
  p.word = &l->words[16 * w];
  p.seq = ++l->seq[w];
  if (counter >= 0) {
    l->arrive_total[counter] += (unsigned)nblocks;
    p.arrive = l->d_arrive + 16 * counter;
    p.target = l->arrive_total[counter];
  }
  return p;
}

void fill_pending(Lane* l, svo_frame_result* res, int upto) {
  if (l->pending_from < 0) return;
  for (int j = l->pending_from; j < upto; ++j) {
    memcpy(res[j].pose7, l->solved_pose, sizeof(res[j].pose7));
    if (j == l->pending_from && res[j].is_keyframe) res[j].ba_iterations = l->last_iterations;
  }
  l->pending_from = -1;
}

// join the lane's solve (BundleAdjuster::bundle_adjust's tail, src/bundle_adjuster.cpp:146-155; host/pipeline.cpp run_bundle_adjust)
int finish_solve(svo_pipeline_group* g, Lane* l) {
  const int st = l->ba_state.load(std::memory_order_acquire);
  l->ba_ready_seq = 0;  // (no solve of this lane waits for a launch any more)
  if (st == BA_NONE) return SVO_OK;
  int rc = SVO_OK;
  const auto tf0 = std::chrono::steady_clock::now();
  if (st == BA_INFLIGHT) rc = svo_ba_solve_finish(l->ba, &l->ba_summary);
  else rc = l->ba_rc;  // BA_HOST_DONE
  if (g->timing) g->t_finish += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tf0).count();
  l->ba_state.store(BA_NONE, std::memory_order_release);
  if (rc) return rc;
  l->last_iterations = l->ba_summary.iterations;
  double p[7];
  if ((rc = svo_ba_get_pose(l->ba, -1, p))) return rc;
  for (int i = 0; i < 7; ++i) l->solved_pose[i] = (double)(float)p[i];  // Keyframe holds Quaternionf / Vector3f (:146-153)
  (void)g;
  return SVO_OK;
}

// After a failed batch: launches of other lanes may still be in flight (they read the caller's images and write the pinned
// mirrors and completion words), and arrive_total[] / seq[] were advanced for launches that may never have run.  Drain every
// line, then bring device counters, host totals and completion words back to a common zero.
void quiesce_after_error(svo_pipeline_group* g) {
  for (int i = 0; i <= svo_pipeline_group::MAX_LINES; ++i) {
    if (g->st_lk[i]) (void)hipStreamSynchronize(g->st_lk[i]);
    if (g->st_chain[i]) (void)hipStreamSynchronize(g->st_chain[i]);
    if (i < svo_pipeline_group::MAX_LINES && g->st_ba[i]) (void)hipStreamSynchronize(g->st_ba[i]);
  }
  for (Lane* l : g->lanes) {
    (void)hipMemset(l->d_arrive, 0, sizeof(unsigned) * 16 * C_COUNT);
    for (int c = 0; c < C_COUNT; ++c) l->arrive_total[c] = 0;
    for (int w = 0; w < W_COUNT; ++w) { l->seq[w] = 0; l->words[16 * w] = 0; }
    if (l->h_bad) *l->h_bad = 0;  // a "foreign landmark-store entry" report must not outlive the batch it failed
    l->fused = false; l->pnp_all = false;
    l->queued = false;
  }
  (void)hipDeviceSynchronize();  // (the fills ran on the null stream)
}

}  // namespace

extern "C" void svo_pipeline_group_destroy(svo_pipeline_group* g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  g->pool.stop();
  for (Lane* l : g->lanes) {
    if (l->ba_state.load() == BA_INFLIGHT) { svo_ba_summary s; (void)svo_ba_solve_finish(l->ba, &s); }
  }
  if (g->timing && g->n_kf)
    fprintf(stderr, "[svo group] %d lanes, %ld keyframes; group thread per keyframe (us): get_world_points %.1f, add_keyframe %.1f, join + write-back of the solve %.1f\n",
            g->n_lanes, g->n_kf, 1e-3 * g->t_get_points / g->n_kf, 1e-3 * g->t_add_keyframe / g->n_kf, 1e-3 * g->t_finish / g->n_kf);
  (void)hipStreamSynchronize(g->ctx->stream);
  for (Lane* l : g->lanes) { if (l->ba) { svo_ba_destroy(l->ba); l->ba = nullptr; } }  // before the lines they work on (svo_ba_use_stream) go
  if (g->counted_lanes) svo_ba_note_group_lanes(-g->counted_lanes);
  for (int i = 0; i <= svo_pipeline_group::MAX_LINES; ++i) {
    if (g->st_lk[i] && g->st_lk[i] != g->ctx->stream) { (void)hipStreamSynchronize(g->st_lk[i]); (void)hipStreamDestroy(g->st_lk[i]); }
    if (g->st_chain[i]) { (void)hipStreamSynchronize(g->st_chain[i]); (void)hipStreamDestroy(g->st_chain[i]); }
    if (i < svo_pipeline_group::MAX_LINES && g->st_ba[i]) { (void)hipStreamSynchronize(g->st_ba[i]); (void)hipStreamDestroy(g->st_ba[i]); }
  }
  if (g->st_copy) { (void)hipStreamSynchronize(g->st_copy); (void)hipStreamDestroy(g->st_copy); }
  for (int sl = 0; sl < 2; ++sl) {
    if (g->ev_up[sl]) (void)hipEventDestroy(g->ev_up[sl]);
    for (int e = 0; e < 2; ++e) { if (g->h_stage[sl][e]) (void)hipHostFree(g->h_stage[sl][e]); if (g->d_stage[sl][e]) (void)hipFree(g->d_stage[sl][e]); }
  }
  for (Lane* l : g->lanes) { if (l->ba) svo_ba_destroy(l->ba); delete l; }
  for (void* p : g->dev_allocs) (void)hipFree(p);
  for (void* p : g->pin_allocs) (void)hipHostFree(p);
  delete g;
}

extern "C" int svo_pipeline_group_create(svo_ctx* ctx, svo_pipeline_group** out, const svo_pipeline_params* p, int n_lanes) {
  if (!ctx || !out || !p) return SVO_ERR_INVALID;
  *out = nullptr;
  SVO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  SVO_REQUIRE(ctx, n_lanes >= 1 && n_lanes <= SVO_MAX_LANES, "pipeline_group_create: 1..64 lanes");
  SVO_REQUIRE(ctx, p->width >= 32 && p->height >= 32 && p->width <= ctx->lim.max_width && p->height <= ctx->lim.max_height,
              "pipeline_group_create: image size outside the context limits");
  SVO_REQUIRE(ctx, p->max_corners >= 4 && p->max_corners <= ctx->lim.max_corners && p->max_features >= 4 &&
                       p->max_features <= ctx->lim.max_features, "pipeline_group_create: feature counts outside the context limits");
  SVO_REQUIRE(ctx, p->window_size >= 1 && p->window_size <= 63, "pipeline_group_create: window size must be 1..63");
  SVO_REQUIRE(ctx, ctx->lim.max_batch >= n_lanes, "pipeline_group_create: svo_limits.max_batch must hold lanes x frames per call");
  {
    // A group drives 1 + chain + solve lines (HIP streams); the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues
    // (default 4, read when the runtime initialises).  With fewer queues than lines a 1 ms solve launch shares a queue with
    // tracking launches and serialises them: measured 9.9 k against 16+ k frames/s for 48 lanes.
    static std::atomic<bool> warned{false};
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    const int nq = q && *q ? atoi(q) : 4;
    if (n_lanes > 4 && nq < 8 && !warned.exchange(true))
      fprintf(stderr, "[svo] pipeline group of %d lanes with GPU_MAX_HW_QUEUES=%d: its launches will share hardware queues and serialise; "
                      "export GPU_MAX_HW_QUEUES=16 before the first HIP call (INTEGRATION.md)\n", n_lanes, nq);
  }
  svo_pipeline_group* g = new svo_pipeline_group();
  g->ctx = ctx; g->prm = *p; g->n_lanes = n_lanes;
  g->max_batch = ctx->lim.max_batch / n_lanes;
  const float K[9] = {(float)p->cam.focal, 0.f, (float)p->cam.cx, 0.f, (float)p->cam.focal, (float)p->cam.cy, 0.f, 0.f, 1.f};  // src/vo_node.cpp:104-108
  memcpy(g->K, K, sizeof(K));
  g->pnp_iterations = svo_ref::PNP_ITERATIONS;
  const size_t mc = (size_t)p->max_corners, mf = (size_t)ctx->lim.max_features, B = (size_t)g->max_batch, S = (size_t)n_lanes;
  const int words = (int)((mf + 63) / 64);
  g->mask_words_cap = words;
  int rc = SVO_OK;
  auto chk = [&](hipError_t e, const char* what) { if (rc == SVO_OK && e != hipSuccess) rc = fail(g, what, e); };
  {
    auto knob = [](const char* name, int dflt, int hi) { const char* e = getenv(name); int v = e && *e ? atoi(e) : dflt; return v < 1 ? 1 : (v > hi ? hi : v); };
    g->n_lk = knob("SVO_GROUP_LK_LINES", 1, svo_pipeline_group::MAX_LINES);
    g->timing = getenv("SVO_TIMING") != nullptr;
    { const char* e = getenv("SVO_GROUP_GATHER_US"); g->gather_us = e && *e ? std::max(0.0, atof(e)) : 0.0; }
    { const char* e = getenv("SVO_GROUP_LK_OVERLAP_US"); g->lk_overlap_us = e && *e ? std::max(0.0, atof(e)) : 0.0; }
    g->n_chain = knob("SVO_GROUP_CHAIN_LINES", 2, svo_pipeline_group::MAX_LINES);
    { const char* e = getenv("SVO_GROUP_LK_XCD"); g->xcd_map = !(e && *e && atoi(e) == 0); }
    { const char* e = getenv("SVO_GROUP_TRI_XCD"); g->xcd_map_tri = !(e && *e && atoi(e) == 0); }
    { const char* e = getenv("SVO_GROUP_XCD_CHUNKS"); if (e && *e) g->xcd_chunks = std::max(8, std::min(64, atoi(e) / 8 * 8)); }  // developer experiments
    g->n_ba = knob("SVO_GROUP_BA_LINES", 4, svo_pipeline_group::MAX_LINES);
    { const char* e = getenv("SVO_GROUP_COMPACT_LINES"); g->n_cmp = e && *e ? std::max(0, std::min(atoi(e), svo_pipeline_group::MAX_LINES - g->n_ba)) : 0; }
    g->st_lk[0] = ctx->stream;
    // experiment knob: the tracker's launches on `keep` of every 32 CUs only (the rest stays free for the short kernels of the
    // keyframe chains and the solves)
    int keep = 32;
    { const char* e = getenv("SVO_GROUP_LK_CU_KEEP"); if (e && *e) keep = std::max(4, std::min(32, atoi(e))); }
    for (int i = 0; i < g->n_lk; ++i) {
      if (i == 0 && keep == 32) continue;
      if (keep < 32) {
        uint32_t mask[8];
        for (int w = 0; w < 8; ++w) mask[w] = (uint32_t)((1ull << keep) - 1ull);
        chk(hipExtStreamCreateWithCUMask(&g->st_lk[i], 8, mask), "stream");
      } else {
        chk(hipStreamCreateWithFlags(&g->st_lk[i], hipStreamNonBlocking), "stream");
      }
    }
    // experiment knob SVO_GROUP_CHAIN_PRIORITY=1: the short kernels of the keyframe chains (world-point upload, PnP, dedup,
    // stereo + triangulation) on high-priority streams.  Under the group load a 2 us kernel takes 76 us on average (it waits
    // for wavefront slots behind the tracker's thousands of workgroups) — but measured 15.1-15.2 k against 16.6-16.7 k
    // frames/s with plain streams: off by default.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const char* pe = getenv("SVO_GROUP_CHAIN_PRIORITY");
    const bool chain_hi = pe && *pe && atoi(pe) != 0 && prio_hi != prio_lo;
    for (int i = 0; i < g->n_chain; ++i) {
      if (chain_hi) chk(hipStreamCreateWithPriority(&g->st_chain[i], hipStreamNonBlocking, prio_hi), "stream");
      else chk(hipStreamCreateWithFlags(&g->st_chain[i], hipStreamNonBlocking), "stream");
    }
    {
      // SVO_GROUP_BA_PRIORITY=low|high: the solve lines as streams of another priority class — the runtime keeps one pool of hardware
      // queues per class, so they can never share a queue with (and block) a tracking or keyframe-chain launch
      const char* be = getenv("SVO_GROUP_BA_PRIORITY");
      const int ba_prio = (be && be[0] == 'l') ? prio_lo : ((be && be[0] == 'h') ? prio_hi : 0);
      for (int i = 0; i < g->n_ba + g->n_cmp; ++i) {
        if (ba_prio != 0 && prio_hi != prio_lo) chk(hipStreamCreateWithPriority(&g->st_ba[i], hipStreamNonBlocking, ba_prio), "stream");
        else chk(hipStreamCreateWithFlags(&g->st_ba[i], hipStreamNonBlocking), "stream");
      }
    }
    { const char* e = getenv("SVO_GROUP_EXPRESS"); g->express = e && *e && atoi(e) != 0; }  // off by default: measured 17.1 k against 18.3 k frames/s on the bench
    if (g->express) {
      chk(hipStreamCreateWithFlags(&g->st_lk[svo_pipeline_group::MAX_LINES], hipStreamNonBlocking), "stream");
      chk(hipStreamCreateWithFlags(&g->st_chain[svo_pipeline_group::MAX_LINES], hipStreamNonBlocking), "stream");
    }
  }
  g->pyr_stride = svo_k_pyramid_bytes(p->width, p->height);
  if (!rc) rc = dev_alloc(g, &g->d_corners, 2 * mc * S * B);
  if (!rc) {
    // The runtime binds a HIP stream to one of its GPU_MAX_HW_QUEUES hardware queues when the stream is first USED (the least
    // referenced queue at that moment); streams that share a hardware queue serialise.  With several groups warming up on their
    // own threads the order of first use — and with it which streams end up sharing a queue — differed from run to run (the same
    // command gave 22 k or 41 k frames/s at 128 lanes in 4 groups, profiles/r05_exp_lanes_groups.txt).  Touch every line once, here,
    // on the creating thread, in a fixed order: the binding becomes a function of the creation order alone.  SVO_GROUP_TOUCH=0: off.
    const char* e = getenv("SVO_GROUP_TOUCH");
    if (!(e && *e && atoi(e) == 0)) {
      hipStream_t order[3 * (svo_pipeline_group::MAX_LINES + 1)];
      int no = 0;
      if (e && atoi(e) == 2) {  // (experiment: tracking and chain lines first)
        for (int i = 0; i < g->n_lk; ++i) order[no++] = g->st_lk[i];
        for (int i = 0; i < g->n_chain; ++i) order[no++] = g->st_chain[i];
        for (int i = 0; i < g->n_ba + g->n_cmp; ++i) order[no++] = g->st_ba[i];
      } else {
        for (int i = 0; i < g->n_ba + g->n_cmp; ++i) order[no++] = g->st_ba[i];
        for (int i = 0; i < g->n_chain; ++i) order[no++] = g->st_chain[i];
        for (int i = 0; i < g->n_lk; ++i) order[no++] = g->st_lk[i];
      }
      for (int i = 0; i < no && !rc; ++i) {
        chk(hipMemsetAsync(g->d_corners, 0, 16, order[i]), "hipMemsetAsync");
        chk(hipStreamSynchronize(order[i]), "hipStreamSynchronize");
      }
    }
  }
  if (!rc) rc = dev_alloc(g, &g->d_ncorners, S * B);
  if (!rc) rc = dev_alloc(g, &g->d_pyr[0], g->pyr_stride * S * B);
  if (!rc) rc = dev_alloc(g, &g->d_pyr[1], g->pyr_stride * S * B);
  if (!rc) {
    void* hp = nullptr;
    chk(hipHostMalloc(&hp, sizeof(int) * (S * B + 16), hipHostMallocDefault), "hipHostMalloc");
    if (hp) { g->pin_allocs.push_back(hp); g->h_counts = (int*)hp; memset(hp, 0, sizeof(int) * (S * B + 16)); }
  }
  for (int li = 0; li < n_lanes && !rc; ++li) {
    Lane* l = new Lane();
    g->lanes.push_back(l);
    for (int b = 0; b < 2 && !rc; ++b) {
      rc = dev_alloc(g, &l->d_xy[b], 2 * mf);
      if (!rc) rc = dev_alloc(g, &l->d_init[b], 2 * mf);
      if (!rc) rc = dev_alloc(g, &l->d_ids[b], mf);
    }
    if (!rc) rc = dev_alloc(g, &l->d_fwd, 2 * mf);
    if (!rc) rc = dev_alloc(g, &l->d_par, mf);
    if (!rc) rc = dev_alloc(g, &l->d_keep, mf);
    {
      // ids advance by at most max_features per keyframe: room for 256 keyframes of survival, power of two; an entry found
      // under another id is reported by the PnP launch (never used silently)
      size_t cap = 1, keyframes = 256;
      if (const char* e = getenv("SVO_GROUP_STORE_KEYFRAMES")) keyframes = (size_t)std::max(1, atoi(e));  // test hook: a store too small must fail loudly (tests/test_group.py)
      while (cap < keyframes * mf) cap <<= 1;
      l->store_mask = (unsigned)(cap - 1);
      if (!rc) rc = dev_alloc(g, &l->d_store, cap);
      if (!rc) chk(hipMemset(l->d_store, 0xFF, sizeof(float4) * cap), "hipMemset");
    }
    if (!rc) rc = dev_alloc(g, &l->d_hyp_pose, 7 * (size_t)g->pnp_iterations);
    if (!rc) rc = dev_alloc(g, &l->d_hyp_count, (size_t)g->pnp_iterations);
    if (!rc) rc = dev_alloc(g, &l->d_hyp_mask, (size_t)g->pnp_iterations * words);
    if (!rc) rc = dev_alloc(g, &l->d_out, 8);
    if (!rc) rc = dev_alloc(g, &l->d_nin, 16);
    if (!rc) rc = dev_alloc(g, &l->d_inl, mf);
    if (!rc) rc = dev_alloc(g, &l->d_trk_xy, 2 * mf);
    if (!rc) rc = dev_alloc(g, &l->d_chain, 1);
    if (!rc) rc = dev_alloc(g, &l->d_disp, mc);
    if (!rc) rc = dev_alloc(g, &l->d_own_pyr, g->pyr_stride);
    if (!rc) rc = dev_alloc(g, &l->d_arrive, 16 * C_COUNT);
    if (!rc) chk(hipMemset(l->d_arrive, 0, sizeof(unsigned) * 16 * C_COUNT), "hipMemset");
    if (rc) break;
    // one pinned arena per lane
    const size_t f2 = sizeof(float) * 2 * mf, i8 = sizeof(long long) * mf;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 127) & ~(size_t)127; return o; };
    const size_t o_xy0 = take(f2), o_xy1 = take(f2), o_id0 = take(i8), o_id1 = take(i8), o_kxy = take(f2), o_kid = take(i8), o_n = take(64),
                 o_cnt = take(64), o_out = take(64), o_nin = take(64),
                 o_inl = take(sizeof(int) * mf), o_tc = take(64), o_txy = take(sizeof(float) * 2 * mc), o_txyz = take(sizeof(float) * 3 * mc),
                 o_words = take(64 * W_COUNT);
    void* hp = nullptr;
    chk(hipHostMalloc(&hp, off, hipHostMallocDefault), "hipHostMalloc");
    if (rc) break;
    g->pin_allocs.push_back(hp);
    memset(hp, 0, off);  // completion words are compared by equality with a sequence number: never start from recycled bytes
    char* h = (char*)hp;
    l->h_xy[0] = (float*)(h + o_xy0); l->h_xy[1] = (float*)(h + o_xy1); l->h_ids[0] = (long long*)(h + o_id0); l->h_ids[1] = (long long*)(h + o_id1);
    l->h_kf_xy = (float*)(h + o_kxy); l->h_kf_ids = (long long*)(h + o_kid);
    l->h_n = (int*)(h + o_n); l->h_av = (float*)(h + o_n + 16);
    l->h_best = (int*)(h + o_cnt); l->h_bad = (int*)(h + o_cnt) + 1; l->h_out = (double*)(h + o_out); l->h_nin = (int*)(h + o_nin);
    l->h_inl = (int*)(h + o_inl); l->h_tri_cnt = (int*)(h + o_tc); l->h_tri_xy = (float*)(h + o_txy); l->h_tri_xyz = (float*)(h + o_txyz);
    l->words = (int*)(h + o_words);
    svo_ba_options opt;
    svo_ba_default_options(&opt);
    opt.max_features = p->max_features;       // src/bundle_adjuster.hpp:75
    opt.max_iterations = p->ba_max_iterations;
    opt.max_time_s = p->ba_max_time_s;        // src/bundle_adjuster.cpp:11
    const int max_obs = (p->window_size + 1) * p->max_features + 64;
    rc = svo_ba_create(ctx, &l->ba, p->window_size, &p->cam, &opt, max_obs, max_obs);
    if (!rc) rc = svo_ba_attach_store(l->ba, l->d_store, l->store_mask);
    // the lane's adjuster works on the group's solve lines (no stream of its own: see the hardware-queue note above)
    { const char* e = getenv("SVO_GROUP_OWN_BA_STREAMS"); if (!rc && !(e && *e && atoi(e) != 0)) rc = svo_ba_use_stream(l->ba, g->st_ba[li % g->n_ba]); }
    // (the window solves keep the wide form unless SVO_BA_FORM=compact: measured in round 5, one workgroup per solve costs a lane
    // ~10x the solve latency and halves the frame rate at 48 lanes — profiles/r05_exp_compact_lanes.txt; the compact form serves as
    // the overflow of the admission budget, SVO_BA_OVERFLOW, and as the re-run of a solve that gave up)
  }
  if (rc) { svo_pipeline_group_destroy(g); return rc; }
  {
    const char* e = getenv("SVO_GROUP_WORKERS");
    // one worker per lane up to 8: lanes in phase reach their keyframes together, and a problem that waits for a worker
    // delays its solve — measured with 2 workers for 8 lanes: 400 us from add_keyframe to the solve's launch, 90 of them assembly
    int nw = e ? atoi(e) : (n_lanes < 8 ? n_lanes : 8);
    if (nw < 1) nw = 1;
    if (nw > 8) nw = 8;
    g->pool.start(nw, ctx->device);
  }
  (void)hipDeviceSynchronize();  // the fills above ran on the null stream, the group's lines are non-blocking streams: nothing may still be in flight
  g->counted_lanes = n_lanes;
  svo_ba_note_group_lanes(n_lanes);
  *out = g;
  return SVO_OK;
}

extern "C" int svo_pipeline_group_reset(svo_pipeline_group* g) {
  if (!g) return SVO_ERR_INVALID;
  (void)hipSetDevice(g->ctx->device);
  int rc_join = SVO_OK;
  for (Lane* l : g->lanes) {
    const int rcf = finish_solve(g, l);
    if (rcf && !rc_join) rc_join = rcf;
    svo_ba_reset(l->ba);
    l->has_keyframe = false; l->n = l->n_initial = 0; l->from_host = false; l->last_pyr = nullptr; l->last_l0 = nullptr; l->cur = 0;
    l->rvec[0] = l->rvec[1] = l->rvec[2] = l->tvec[0] = l->tvec[1] = l->tvec[2] = 0.f;
    l->state = L_IDLE; l->queued = false; l->pending_from = -1; l->last_iterations = 0;
    const double id7[7] = {1, 0, 0, 0, 0, 0, 0};
    memcpy(l->solved_pose, id7, sizeof(id7));
    // feature ids restart at 0 (svo_ba_reset): no entry of the landmark store may survive under an old id, and a reported
    // foreign entry is forgotten with the stream that produced it
    if (l->h_bad) *l->h_bad = 0;
    l->fused = false; l->pnp_all = false;
    // (on the context's stream and waited for below: hipMemset on device memory may return before the fill has run, and the group's
    // lines are non-blocking streams — a fill that landed behind the first solve of the next batch wiped its entries; seen under rocprofv3)
    static const int fill_mode = [] { const char* e = getenv("SVO_GROUP_RESET_FILL"); return e ? atoi(e) : 0; }();  // experiment: 1 = null stream, not waited for (before 954d023); 2 = no fill
    if (fill_mode == 1) { if (l->d_store && hipMemset(l->d_store, 0xFF, sizeof(float4) * ((size_t)l->store_mask + 1)) != hipSuccess && !rc_join) rc_join = SVO_ERR_HIP; }
    else if (fill_mode == 0)
    if (l->d_store && hipMemsetAsync(l->d_store, 0xFF, sizeof(float4) * ((size_t)l->store_mask + 1), g->ctx->stream) != hipSuccess && !rc_join) rc_join = SVO_ERR_HIP;
  }
  if (hipStreamSynchronize(g->ctx->stream) != hipSuccess && !rc_join) rc_join = SVO_ERR_HIP;
  if (rc_join) { quiesce_after_error(g); return rc_join; }  // a solve that was in flight did not come back cleanly
  return SVO_OK;
}

extern "C" int svo_pipeline_group_lanes(const svo_pipeline_group* g) { return g ? g->n_lanes : 0; }

extern "C" int svo_pipeline_group_get_tracked(svo_pipeline_group* g, int lane, int64_t* ids, float* xy, int capacity, int* n) {
  if (!g || !n || lane < 0 || lane >= g->n_lanes) return SVO_ERR_INVALID;
  const Lane* l = g->lanes[lane];
  *n = l->n;
  const float* sx = l->from_host ? l->h_kf_xy : l->h_xy[l->cur];
  const long long* si = l->from_host ? l->h_kf_ids : l->h_ids[l->cur];
  for (int i = 0; i < l->n && i < capacity; ++i) {
    if (ids) ids[i] = (int64_t)si[i];
    if (xy) { xy[2 * i] = sx[2 * i]; xy[2 * i + 1] = sx[2 * i + 1]; }
  }
  return SVO_OK;
}

extern "C" int svo_pipeline_group_solve_work(svo_pipeline_group* g, double* out4, int reset) {
  if (!g || !out4) return SVO_ERR_INVALID;
  out4[0] = out4[1] = out4[2] = out4[3] = 0.0;
  for (Lane* l : g->lanes) {
    double w[4];
    svo_ba_work(l->ba, w, reset);
    for (int i = 0; i < 4; ++i) out4[i] += w[i];
  }
  return SVO_OK;
}

extern "C" int svo_pipeline_group_last_stats(const svo_pipeline_group* g, long* launches6, long* lanes6) {
  if (!g || !launches6 || !lanes6) return SVO_ERR_INVALID;
  for (int i = 0; i < 6; ++i) { launches6[i] = g->launches[i]; lanes6[i] = g->lanes_carried[i]; }
  return SVO_OK;
}

extern "C" int svo_pipeline_group_process_batch_dev(svo_pipeline_group* g, const uint8_t* left, const uint8_t* right, size_t lane_stride,
                                                    int batch, svo_frame_result* results) {
  if (!g) return SVO_ERR_INVALID;
  svo_ctx* ctx = g->ctx;
  SVO_REQUIRE(ctx, left && right && results && batch >= 1 && batch <= g->max_batch, "pipeline_group_process_batch: bad arguments");
  const int W = g->prm.width, H = g->prm.height, S = g->n_lanes, mc = g->prm.max_corners;
  const size_t istride = (size_t)W * H;
  SVO_REQUIRE(ctx, lane_stride >= istride * (size_t)batch, "pipeline_group_process_batch: lanes overlap");
  ctx->err.clear();
  SVO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  for (int i = 0; i < 6; ++i) { g->launches[i] = 0; g->lanes_carried[i] = 0; }

  // ---- a1 on every frame of every lane + the pyramids (the reference detects on every frame, src/image_processor.cpp:22)
  g->pyr_cur ^= 1;  // the previous batch's pyramids stay valid: a lane's last tracked image lives there (src/feature_tracker.cpp:66)
  uint8_t* pyr = g->d_pyr[g->pyr_cur];
  int rc = SVO_OK;
  // ... but a lane that neither tracked nor made a keyframe during the WHOLE previous batch (every frame below MIN_DETECTED
  // corners, C-7; with batch = 1 one blank frame is enough) still holds its last image in the buffer that is rebuilt now:
  // the reference keeps a clone of it (src/feature_tracker.cpp:14,66), so the lane takes a private copy first (same stream
  // as the rebuild: ordered in front of it).
  {
    const uint8_t* lo = pyr;
    const uint8_t* hi = pyr + g->pyr_stride * (size_t)g->n_lanes * (size_t)g->max_batch;
    for (Lane* l : g->lanes) {
      if (!l->last_pyr || l->last_pyr < lo || l->last_pyr >= hi) continue;
      SVO_HIP_CHECK(ctx, hipMemcpyAsync(l->d_own_pyr, l->last_pyr, g->pyr_stride, hipMemcpyDeviceToDevice, ctx->stream));
      l->last_pyr = l->d_own_pyr;  // level 0 was cloned into the pyramid at the end of the batch that produced it
      l->last_l0 = l->d_own_pyr;
    }
  }
  if (lane_stride == istride * (size_t)batch) {
    rc = svo_corner_detect_batch_dev(ctx, left, S * batch, W, H, W, istride, mc, g->prm.quality, (double)g->prm.min_feature_distance, g->d_corners, g->d_ncorners);
    // level 0 of the pyramids is NOT copied: the tracker reads the caller's images in place (Pyr::l0) — except every lane's
    // last frame, which the next batch tracks from when the caller's buffer may hold other frames
    if (!rc) rc = svo_k_build_pyramid(ctx, left, S * batch, W, H, W, istride, pyr, g->pyr_stride, true);
    if (!rc) rc = svo_k_pyramid_level0(ctx, left + (size_t)(batch - 1) * istride, S, W, H, W, lane_stride, pyr + (size_t)(batch - 1) * g->pyr_stride,
                                       (size_t)batch * g->pyr_stride, st);
  } else {
    for (int l = 0; l < S && !rc; ++l) {
      rc = svo_corner_detect_batch_dev(ctx, left + l * lane_stride, batch, W, H, W, istride, mc, g->prm.quality, (double)g->prm.min_feature_distance,
                                       g->d_corners + (size_t)l * batch * 2 * mc, g->d_ncorners + (size_t)l * batch);
      if (!rc) rc = svo_k_build_pyramid(ctx, left + l * lane_stride, batch, W, H, W, istride, pyr + (size_t)l * batch * g->pyr_stride, g->pyr_stride, true);
      if (!rc) rc = svo_k_pyramid_level0(ctx, left + l * lane_stride + (size_t)(batch - 1) * istride, 1, W, H, W, istride,
                                         pyr + ((size_t)l * batch + (size_t)(batch - 1)) * g->pyr_stride, g->pyr_stride, st);
    }
  }
  if (rc) return rc;
  g->launches[5] += 7; g->lanes_carried[5] += 7 * S;
  int* hc = g->h_counts;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(hc, g->d_ncorners, sizeof(int) * S * batch, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(hc + S * batch, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (hc[S * batch]) {
    (void)hipMemsetAsync(ctx->d_status, 0, sizeof(int), st);
    ctx->err = (hc[S * batch] & 8) ? "corner detection: more raw local maxima than the streaming pass's list holds (raw_cap: width x height / 4 per image under a 1 GiB budget)"
                         : "corner detection exceeded a workspace bound (svo_limits.max_candidates)";
    return SVO_ERR_CAPACITY;
  }

  auto RES = [&](int lane, int f) -> svo_frame_result& { return results[(size_t)lane * batch + f]; };
  auto IMG = [&](const uint8_t* base, int lane, int f) { return base + lane * lane_stride + (size_t)f * istride; };
  auto PYR = [&](int lane, int f) { return pyr + ((size_t)lane * batch + f) * g->pyr_stride; };
  // level 0 of frame f's pyramid: the caller's image, except the batch's last frame (copied into the pyramid above)
  auto L0 = [&](int lane, int f) -> const uint8_t* { return f == batch - 1 ? PYR(lane, f) : IMG(left, lane, f); };
  auto DET = [&](int lane, int f) { return g->d_corners + ((size_t)lane * batch + f) * 2 * mc; };
  for (int li = 0; li < S; ++li) {
    Lane* l = g->lanes[li];
    l->frame = 0; l->state = L_IDLE; l->queued = false; l->pending_from = -1;
    memset(&RES(li, 0), 0, sizeof(svo_frame_result) * batch);
  }

  // per-pass submission lists
  std::vector<int> q_track, q_pnp, q_tri, q_ba;
  int error = SVO_OK;
  const int MODEL = svo_pnp_model_points();
  // SVO_GROUP_TRACE=1: (microseconds, lane, event) of this call on stderr — where a lane's time goes
  static const bool trace_on = getenv("SVO_GROUP_TRACE") != nullptr;
  // the keyframe chain without a host turn between PnP and stereo + triangulation (round 5); SVO_GROUP_CHAIN_FUSED=0: one launch,
  // one host turn each, as before — same results either way (tests/test_group.py)
  static const bool chain_fused = [] { const char* e = getenv("SVO_GROUP_CHAIN_FUSED"); return !(e && *e && atoi(e) == 0); }();
  struct Ev { double us; int lane; const char* what; int arg; };
  std::vector<Ev> evs;
  const auto t_begin = std::chrono::steady_clock::now();
  auto EV = [&](int lane, const char* what, int arg) {
    if (trace_on) evs.push_back({std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(), lane, what, arg});
  };

  auto now_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(); };
  // frame finished (is_keyframe etc. already in the result): move on
  auto frame_done = [&](int li) {
    Lane* l = g->lanes[li];
    svo_frame_result* res = &RES(li, 0);
    const int i = l->frame;
    if (l->has_keyframe && !(res[i].is_keyframe) && l->pending_from < 0) l->pending_from = i;  // no solve started here: the pose is the last solved one (src/vo_node.cpp:146-150)
    l->frame = i + 1;
    l->state = l->frame < batch ? L_IDLE : L_DONE;
  };

  // src/image_processor.cpp:137-162 once the triangulated new features are on the host: keyframe, solve, tracker
  auto keyframe_tail = [&](int li) -> int {
    Lane* l = g->lanes[li];
    svo_frame_result* res = &RES(li, 0);
    const int i = l->frame;
    const int m_new = *l->h_tri_cnt;
    double pose7[7];
    if (l->first_keyframe) {
      const double id7[7] = {1, 0, 0, 0, 0, 0, 0};  // :41 Vector3f zero, Quaternionf identity
      memcpy(pose7, id7, sizeof(id7));
      l->kf_tracked_ids.clear(); l->kf_tracked_xy.clear();
    } else {
      pose7[0] = l->quat[0]; pose7[1] = l->quat[1]; pose7[2] = l->quat[2]; pose7[3] = l->quat[3];
      pose7[4] = l->tvec[0]; pose7[5] = l->tvec[1]; pose7[6] = l->tvec[2];
    }
    const int nt = (int)l->kf_tracked_ids.size();
    l->new_ids.assign((size_t)(m_new > 0 ? m_new : 1), 0);
    int kept = 0;
    std::vector<int64_t> tid(l->kf_tracked_ids.begin(), l->kf_tracked_ids.end());
    const auto ta0 = std::chrono::steady_clock::now();
    int rc2 = svo_ba_add_keyframe(l->ba, pose7, tid.data(), l->kf_tracked_xy.data(), nt, l->h_tri_xy, l->h_tri_xyz, m_new, l->new_ids.data(), &kept);  // :144
    if (g->timing) { g->t_add_keyframe += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - ta0).count(); g->n_kf++; }
    if (rc2) return rc2;
    // the previous solve was joined before the graph was edited: its poses are final now
    fill_pending(l, res, i);
    l->pending_from = i;
    l->has_keyframe = true;
    l->ba_state.store(BA_ASSEMBLING, std::memory_order_release);  // src/vo_node.cpp:147, started here so that the assembly overlaps the next frames
    g->pool.post(l, 0);
    EV(li, "assembly_posted", i);
    // tracker->init(left, tracked + new, ids) :148-162 — the next track reads these pinned arrays in place
    int n = nt + kept;
    if (n > ctx->lim.max_features) n = ctx->lim.max_features;
    for (int k = 0; k < nt && k < n; ++k) { l->h_kf_xy[2 * k] = l->kf_tracked_xy[2 * k]; l->h_kf_xy[2 * k + 1] = l->kf_tracked_xy[2 * k + 1]; l->h_kf_ids[k] = l->kf_tracked_ids[k]; }
    for (int k = nt; k < n; ++k) { l->h_kf_xy[2 * k] = l->h_tri_xy[2 * (k - nt)]; l->h_kf_xy[2 * k + 1] = l->h_tri_xy[2 * (k - nt) + 1]; l->h_kf_ids[k] = (long long)l->new_ids[k - nt]; }
    l->n = l->n_initial = n;
    l->from_host = true;
    l->last_pyr = PYR(li, i);  // clone of the image, :14 (the batch buffers are double-buffered: it stays valid through the next batch)
    l->last_l0 = L0(li, i);
    if (l->first_keyframe) { l->tvec[0] = l->tvec[1] = l->tvec[2] = 0.f; l->rvec[0] = l->rvec[1] = l->rvec[2] = 0.f; }  // :54-55
    res[i].is_keyframe = 1;
    res[i].n_new = kept;
    frame_done(li);
    return SVO_OK;
  };

  // after PnP (:84-134): keyframe pose, inlier copy; then dedup + sparse stereo + triangulation go out
  // PnP's verdict and pose out of the pinned mirrors: 0 = go on, -1 = error (set) or "launch again with all hypotheses" (queued)
  auto consume_pnp = [&](int li) -> int {
    Lane* l = g->lanes[li];
    if (*l->h_bad) {
      ctx->err = "pipeline group: a tracked feature's entry of the landmark store belongs to another id (store capacity exceeded?)";
      error = SVO_ERR_CAPACITY;
      return -1;
    }
    l->best = *l->h_best;  // the launch's own RANSAC bookkeeping (csrc/pnp.hip pnp_group_kernel); -1: no model
    if (l->best == -2) {     // the adaptive cap stayed above the hypotheses of the first launch: all of them now
      l->pnp_all = true;
      l->state = L_PNP_WAIT;
      q_pnp.push_back(li); l->queued = true; l->t_q = now_us();
      return -1;
    }
    if (l->best >= 0) {
      double rv[3];
      svo_det_rvec_from_quat(l->h_out, rv);  // declared arithmetic (host/det_trig.h)
      for (int k = 0; k < 3; ++k) { l->rvec[k] = (float)rv[k]; l->tvec[k] = (float)l->h_out[4 + k]; }
      l->num_inliers = *l->h_nin;
    }
    return 0;
  };
  auto after_pnp = [&](int li, bool queue_tri = true) {
    Lane* l = g->lanes[li];
    svo_frame_result* res = &RES(li, 0);
    res[l->frame].n_inliers = l->num_inliers;
    rodrigues_f(l->rvec, l->rmat);   // :84-85
    quat_from_R(l->rmat, l->quat);   // :87-92
    const float* txy = l->h_xy[l->cur];
    const long long* tids = l->h_ids[l->cur];
    l->kf_tracked_ids.resize(l->num_inliers); l->kf_tracked_xy.resize(2 * (size_t)l->num_inliers);
    for (int k = 0; k < l->num_inliers; ++k) {  // :104-108
      const int idx = l->h_inl[k];
      l->kf_tracked_ids[k] = tids[idx];
      l->kf_tracked_xy[2 * k] = txy[2 * idx]; l->kf_tracked_xy[2 * k + 1] = txy[2 * idx + 1];
    }
    l->first_keyframe = false;
    if (queue_tri) { q_tri.push_back(li); l->queued = true; l->t_q = now_us(); l->state = L_TRI_WAIT; }
  };

  // the keyframe path once the previous solve is joined (:71-82)
  auto continue_keyframe = [&](int li) -> int {
    Lane* l = g->lanes[li];
    const int m = l->n;
    l->m_tracked = m;
    l->num_inliers = 0; l->best = -1;
    // :72 get_world_points: nothing to do on the host — the PnP launch reads the tracked features' world points from the lane's
    // device-resident landmark store, which the (joined) solve of the previous keyframe has written
    if (m >= MODEL) { q_pnp.push_back(li); l->queued = true; l->t_q = now_us(); l->state = L_PNP_WAIT; }
    else after_pnp(li);
    return SVO_OK;
  };

  // the keyframe gate after tracking (:63-65)
  auto after_track = [&](int li) -> int {
    Lane* l = g->lanes[li];
    svo_frame_result& r = RES(li, l->frame);
    const float av = *l->h_av;
    const float pl = (float)(1.0 - (double)((float)l->n / (float)l->n_initial));  // src/feature_tracker.cpp:64
    r.n_tracked = l->n; r.av_parallax = av; r.percent_lost = pl;
    if (av <= g->prm.parallax_thresh && (double)pl < svo_ref::KEYFRAME_PERCENT_LOST) { frame_done(li); return SVO_OK; }
    if (l->ba_state.load(std::memory_order_acquire) != BA_NONE) { l->state = L_NEED_SOLVE; EV(li, "need_solve", l->ba_state.load()); return SVO_OK; }
    return continue_keyframe(li);
  };

  auto start_frame = [&](int li) -> int {
    Lane* l = g->lanes[li];
    const int i = l->frame;
    svo_frame_result& r = RES(li, i);
    const int n_det = hc[li * batch + i];
    r.n_detected = n_det;
    if (n_det < svo_ref::MIN_DETECTED) { frame_done(li); return SVO_OK; }  // :23-25
    if (!l->has_keyframe) {  // :30-58
      l->first_keyframe = true;
      l->num_inliers = 0;
      q_tri.push_back(li); l->queued = true; l->t_q = now_us();
      l->state = L_TRI_WAIT;
      return SVO_OK;
    }
    if (l->n <= 0) {  // nothing to track: the kernels would see n = 0 (host/pipeline.cpp: av_parallax = 0)
      l->n = 0; *l->h_av = 0.f;
      return after_track(li);
    }
    q_track.push_back(li); l->queued = true; l->t_q = now_us();
    l->state = L_TRACK_WAIT;
    return SVO_OK;
  };

  unsigned idle_spins = 0;
  auto last_progress = std::chrono::steady_clock::now();
  double busy_us = 0.0;  // wall time of the loop passes that did something: how much of the call the driving thread was occupied
  for (;;) {
    bool all_done = true, progressed = false;
    const auto t_pass = std::chrono::steady_clock::now();
    // ---- advance every lane as far as the host alone can
    for (int li = 0; li < S && !error; ++li) {
      Lane* l = g->lanes[li];
      bool again = true;
      while (again && !error) {
        again = false;
        switch (l->state) {
          case L_IDLE:
            error = start_frame(li);
            progressed = true;
            again = l->state == L_IDLE;  // the frame ended on the host (too few corners / no keyframe gate): next frame
            break;
          case L_TRACK_WAIT:
            if (!l->queued && word_ready(l, W_TRACK)) {
              EV(li, "track_done", l->frame);
              l->cur ^= 1; l->from_host = false;
              l->n = *l->h_n;
              l->last_pyr = PYR(li, l->frame);  // :66
              l->last_l0 = L0(li, l->frame);
              error = after_track(li);
              progressed = true; again = l->state == L_IDLE;
            }
            break;
          case L_NEED_SOLVE: {
            const int bs = l->ba_state.load(std::memory_order_acquire);
            if ((bs == BA_INFLIGHT && svo_ba_solve_poll(l->ba)) || bs == BA_HOST_DONE) {
              EV(li, "solve_joined_for_pnp", l->frame);
              error = finish_solve(g, l);
              if (!error) error = continue_keyframe(li);
              progressed = true; again = l->state == L_IDLE;
            }
            break;
          }
          case L_PNP_WAIT:
            if (!l->queued && word_ready(l, W_PNP)) {
              EV(li, "pnp_done", l->frame);
              progressed = true;
              const int verdict = consume_pnp(li);
              if (verdict < 0) break;        // error, or all hypotheses are needed: queued again
              after_pnp(li, true);
            }
            break;
          case L_TRI_WAIT:
            if (!l->queued && word_ready(l, W_TRI)) {
              EV(li, "tri_done", l->frame);
              progressed = true;
              if (l->fused) {  // the launch rode behind the lane's PnP launch: PnP's results first (they are complete: same stream, earlier)
                l->fused = false;
                const int verdict = consume_pnp(li);
                if (verdict < 0) break;      // (all hypotheses needed: the triangulation did nothing; both go out again)
                after_pnp(li, false);
              }
              error = keyframe_tail(li);
              again = l->state == L_IDLE;
            }
            break;
          default: break;
        }
      }
      if (l->state != L_DONE) all_done = false;
    }
    if (error) break;

    // ---- one launch per stage for the lanes that have reached it.  Launch discipline: a stage's kernels go to an in-order
    // stream, so a launch for ONE lane right now would make the lane that is ready ten microseconds later queue behind it
    // for a whole kernel duration (tracking: 150-200 us) — measured: 8 lanes, 110 track launches for 120 lane-frames, 3,000
    // frames/s.  Instead a stage is launched only while none of its launches is in flight; lanes that become ready meanwhile
    // ride the next launch together (a bus, not taxis).  Nobody waits for a lane that is not ready.
    // EXPRESS lines (experiment, SVO_GROUP_EXPRESS=1; off by default): a call ends when its slowest lane ends, and lanes differ
    // (a stream whose windows take 11 LM iterations per solve next to streams that take 4: the other lanes of the group idle for
    // a fifth of the call).  A lane that has fallen behind the group departs at once on an express line of its stage (one more
    // stream each for tracking and the keyframe chain) instead of waiting for the bus.  Lanes are independent and every stage
    // starts only after the host has seen the previous one complete, so which stream carries a launch changes no result.
    // Measured: no gain (17.1 k against 18.3 k frames/s) — what holds a late lane back is the start of its SOLVES, see below.
    constexpr int XL = svo_pipeline_group::MAX_LINES;
    bool lk_busy[XL + 1] = {}, chain_busy[XL + 1] = {};
    double mean_frame = 0.0;
    int n_running = 0;
    for (int li = 0; li < S; ++li) {
      const Lane* l = g->lanes[li];
      if (l->state != L_DONE) { mean_frame += l->frame; ++n_running; }
      if (l->queued) continue;
      lk_busy[l->lk_line] |= l->state == L_TRACK_WAIT;
      chain_busy[l->chain_line] |= l->state == L_PNP_WAIT || l->state == L_TRI_WAIT;
    }
    mean_frame = n_running ? mean_frame / n_running : 0.0;
    auto laggard = [&](int li) { return g->express && n_running > 2 && (double)g->lanes[li]->frame + 1.0 < mean_frame; };
    // the lanes of `q` that ride line `line` (of `n_lines`; XL: the express line takes the laggards), removed from q
    auto take_line = [&](std::vector<int>& q, int line, int n_lines) {
      std::vector<int> mine;
      for (size_t k = 0; k < q.size();) {
        if (line == XL ? laggard(q[k]) : q[k] % n_lines == line) { mine.push_back(q[k]); q.erase(q.begin() + k); } else ++k;
      }
      return mine;
    };
    const double t_now_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
    // GATHER (SVO_GROUP_GATHER_US > 0): a launch lasts as long as its slowest item whatever it carries, so a bus that leaves with
    // four of the line's lanes while others are a hundred microseconds away costs a whole launch more.  A stage's launch therefore
    // waits while a lane of its line is NEAR (in the launch in flight of the stage before it), at most gather_us per waiting lane.
    auto ripe = [&](const std::vector<int>& q, int line, int n_lines, int near_state) {
      if (g->gather_us <= 0.0 || line == XL) return true;
      int mine = 0, near = 0;
      bool waited = false;
      for (int li : q) if (n_lines <= 1 || li % n_lines == line) { ++mine; waited = waited || t_now_us - g->lanes[li]->t_q >= g->gather_us; }
      if (!mine || waited) return true;
      for (int li = 0; li < S; ++li) if ((n_lines <= 1 || li % n_lines == line) && g->lanes[li]->state == near_state && !g->lanes[li]->queued) ++near;
      return near == 0;
    };
    bool tails_only = true;  // every tracking launch in flight is old enough to be in its tail (a few straggling wavefronts)
    for (int line = 0; line < g->n_lk; ++line) tails_only = tails_only && (!lk_busy[line] || t_now_us - g->lk_t0[line] >= g->lk_overlap_us);
    for (int pass = g->express ? -1 : 0; pass < g->n_lk && !error; ++pass) {
      const int line = pass < 0 ? XL : pass;  // the express line first: what it takes no longer waits for a regular line
      if (lk_busy[line]) continue;
      std::vector<int> q_now;
      if (g->lk_overlap_us > 0 && line != XL) {  // dynamic lines: the whole queue departs on a free line, but only next to tails
        if (!tails_only) break;
        if (!ripe(q_track, 0, 1, L_TRI_WAIT)) break;
        q_now.swap(q_track);
      } else {
        if (!ripe(q_track, line, g->n_lk, L_TRI_WAIT)) continue;
        q_now = take_line(q_track, line, g->n_lk);
      }
      if (q_now.empty()) continue;
      if (line != XL) g->lk_t0[line] = t_now_us;
      for (int li : q_now) g->lanes[li]->lk_line = line;
      SvoLkLanes a;
      a.w = W; a.h = H;
      int gx = 1;
      for (int li : q_now) gx = std::max(gx, g->lanes[li]->n);
      a.map.per_chunk = 0; a.map.total = 0; a.map.chunks = 0;
      if (g->xcd_map) {
        int counts[SVO_MAX_LANES], j = 0;
        for (int li : q_now) counts[j++] = g->lanes[li]->n;
        svo_xcd_map_fill(a.map, counts, j, g->xcd_chunks);
      }
      int k = 0;
      for (int li : q_now) {
        Lane* l = g->lanes[li];
        SvoLkLane& x = a.lane[k++];
        const int nxt = l->cur ^ 1;
        x.pyr_prev = l->last_pyr; x.pyr_next = PYR(li, l->frame);
        x.l0_prev = l->last_l0; x.l0_next = L0(li, l->frame);
        x.xy = l->from_host ? l->h_kf_xy : l->d_xy[l->cur];
        x.init_xy = l->from_host ? l->h_kf_xy : l->d_init[l->cur];
        x.ids = l->from_host ? l->h_kf_ids : l->d_ids[l->cur];
        x.n = l->n;
        x.fwd = l->d_fwd; x.keep = l->d_keep; x.parallax = l->d_par;
        x.kept_xy = l->d_xy[nxt]; x.init_dst = l->d_init[nxt]; x.ids_dst = l->d_ids[nxt];
        x.host_xy = l->h_xy[nxt]; x.host_ids = l->h_ids[nxt]; x.host_n = l->h_n; x.host_av = l->h_av;
        l->arrive_total[C_LK] += (unsigned)(a.map.per_chunk > 0 ? std::max(1, l->n) : gx);
        x.arrive = l->d_arrive + 16 * C_LK; x.arrive_target = l->arrive_total[C_LK];
        x.word = &l->words[16 * W_TRACK]; x.seq = ++l->seq[W_TRACK];
        l->queued = false;
      }
      if ((error = svo_kg_track(ctx, g->st_lk[line], a, k, gx))) break;
      for (int li : q_now) EV(li, "track_launch", k);
      g->launches[0]++; g->lanes_carried[0] += k;
      progressed = true;
    }
    if (error) break;
    for (int pass = g->express ? -1 : 0; pass < g->n_chain && !error; ++pass) {
      const int line = pass < 0 ? XL : pass;
      if (chain_busy[line]) continue;
      hipStream_t stc = g->st_chain[line];
      const std::vector<int> h_now = ripe(q_pnp, line, g->n_chain, L_TRACK_WAIT) ? take_line(q_pnp, line, g->n_chain) : std::vector<int>();
      std::vector<int> t_now = take_line(q_tri, line, g->n_chain);
      for (int li : h_now) g->lanes[li]->chain_line = line;
      for (int li : t_now) g->lanes[li]->chain_line = line;
    if (!h_now.empty()) {
      SvoPnpLanes a;
      int k = 0;
      for (int li : h_now) {
        Lane* l = g->lanes[li];
        const int m = l->m_tracked;
        SvoPnpLane& x = a.lane[k++];
        x.ids = l->d_ids[l->cur]; x.store = l->d_store; x.store_mask = l->store_mask;
        x.xy = l->d_xy[l->cur]; x.n = m;
        x.f = (double)g->K[0]; x.cx = (double)g->K[2]; x.cy = (double)g->K[5];
        // rvec/tvec are CV_32F in/out, the solver works in double (host/pipeline.cpp; csrc/pnp.hip svo_k_pnp)
        const double rv[3] = {l->rvec[0], l->rvec[1], l->rvec[2]};
        svo_det_quat_from_rvec(rv, x.q0);  // declared arithmetic (host/det_trig.h)
        for (int c = 0; c < 3; ++c) x.t0[c] = (double)l->tvec[c];
        x.thr2 = (double)svo_ref::PNP_REPROJ_ERROR * (double)svo_ref::PNP_REPROJ_ERROR;
        x.confidence = svo_ref::PNP_CONFIDENCE; x.iterations = g->pnp_iterations;
        x.launched = l->pnp_all ? g->pnp_iterations : svo_kg_pnp_first(g->pnp_iterations);
        l->pnp_all = false;
        const int wgs = svo_kg_pnp_workgroups(x.launched);
        x.hyp_pose = l->d_hyp_pose; x.hyp_count = l->d_hyp_count; x.hyp_mask = l->d_hyp_mask; x.mask_words = svo_div_up(m, 64);
        x.out_pose = l->d_out; x.inliers = l->d_inl; x.n_inliers = l->d_nin; x.inlier_xy = l->d_trk_xy;
        x.host_pose = l->h_out; x.host_inliers = l->h_inl; x.host_nin = l->h_nin; x.host_best = l->h_best; x.host_bad = l->h_bad;
        const SvoPublish pb = make_pub(l, W_PNP, C_PNP, wgs);
        x.arrive = pb.arrive; x.arrive_target = pb.target; x.word = pb.word; x.seq = pb.seq;
        // fused chain: the launch leaves M / the inlier count / "more hypotheses first" for the stereo launch right behind it
        x.chain = chain_fused ? l->d_chain : nullptr;
        for (int c = 0; c < 3; ++c) { x.prev_rvec[c] = l->rvec[c]; x.prev_tvec[c] = l->tvec[c]; }
        x.cam_f = g->K[0]; x.cam_cx = g->K[2]; x.cam_cy = g->K[5]; x.cam_b = (float)g->prm.cam.baseline;
        l->queued = false;
      }
      if ((error = svo_kg_pnp(ctx, stc, a, k))) break;
      for (int li : h_now) EV(li, "pnp_launch", k);
      g->launches[1]++; g->lanes_carried[1] += k;
      progressed = true;
      if (chain_fused) for (int li : h_now) { g->lanes[li]->fused = true; t_now.push_back(li); }
    }
    if (!t_now.empty()) {
      // dedup (:113-128) for the lanes that tracked, then sparse StereoBM + triangulation (:137-142, :34-39 for frame 0)
      SvoStereoTriLanes t;
      t.w = W; t.h = H; t.stride = W; t.ndisp = svo_ref::STEREO_NUM_DISPARITIES; t.block = svo_ref::STEREO_BLOCK_SIZE;
      int kt = 0, gxt = 1;
      t.map.per_chunk = 0; t.map.total = 0; t.map.chunks = 0;
      {
        int counts[SVO_MAX_LANES], j = 0;
        for (int li : t_now) {
          Lane* l = g->lanes[li];
          const int n_det = hc[li * batch + l->frame];
          gxt = std::max(gxt, n_det);
          counts[j++] = n_det;
        }
        if (g->xcd_map_tri) svo_xcd_map_fill(t.map, counts, j, g->xcd_chunks);
      }
      for (int li : t_now) {
        Lane* l = g->lanes[li];
        const int i = l->frame, n_det = hc[li * batch + i];
        float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        SvoStereoTriLane& x = t.lane[kt++];
        x.xy = DET(li, i); x.n_dev = nullptr;
        x.trk = nullptr; x.n_trk = 0; x.min_d = 0.f;
        x.chain = nullptr;
        if (l->fused) {  // behind the lane's PnP launch: M and the inlier count are read from its record on the device
          x.chain = l->d_chain; x.trk = l->d_trk_xy; x.min_d = g->prm.min_feature_distance;
        } else if (!l->first_keyframe) {
          // dedup (:113-128) inside the same launch: every corner's workgroup tests it against the tracked inliers first
          x.trk = l->d_trk_xy; x.n_trk = l->num_inliers; x.min_d = g->prm.min_feature_distance;
          if (x.n_trk <= 0) x.trk = nullptr;  // no inliers (C-9): nothing to keep away from
          // hmat = [R^T | -R^T t]  :130-134 (float Mats; the product accumulates in double)
          memset(pose, 0, sizeof(pose));
          for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) pose[4 * r + c] = l->rmat[3 * c + r];
            double s = 0.0;
            for (int c = 0; c < 3; ++c) s += (double)(-l->rmat[3 * c + r]) * (double)l->tvec[c];
            pose[4 * r + 3] = (float)s;
          }
          pose[15] = 1.f;
        }
        x.left = IMG(left, li, i); x.right = IMG(right, li, i);
        x.n_max = n_det; x.disp = l->d_disp;
        x.M = svo_k_reprojection_matrix(pose, g->K[0], g->K[2], g->K[5], (float)g->prm.cam.baseline);  // :178-189
        x.kept_xy = l->h_tri_xy; x.xyz = l->h_tri_xyz; x.n_kept = l->h_tri_cnt;
        x.pub = make_pub(l, W_TRI, C_TRI, t.map.per_chunk > 0 ? std::max(1, n_det) : gxt);
        l->state = L_TRI_WAIT;
        l->queued = false;
      }
      if ((error = svo_kg_stereo_triangulate(ctx, stc, t, kt, gxt))) break;
      for (int li : t_now) EV(li, "tri_launch", kt);
      g->launches[3]++; g->lanes_carried[3] += kt;
      progressed = true;
    }
    }
    if (error) break;
    // ---- solves.  A finished solve is joined right away (its admission is handed back: lanes that are already through
    // their frames would otherwise keep the budget while others wait for it); the results are the same whenever they are read.
    for (int li = 0; li < S && !error; ++li) {
      Lane* l = g->lanes[li];
      const int bs = l->ba_state.load(std::memory_order_acquire);
      if (l->state != L_NEED_SOLVE && ((bs == BA_INFLIGHT && svo_ba_solve_poll(l->ba)) || bs == BA_HOST_DONE)) { EV(li, "solve_joined", l->frame); error = finish_solve(g, l); progressed = true; }
    }
    if (error) break;
    // everything that has been assembled goes out as one launch once no lane is still assembling
    {
      int assembling = 0;
      q_ba.clear();
      for (int li = 0; li < S; ++li) {
        const int bs = g->lanes[li]->ba_state.load(std::memory_order_acquire);
        assembling += bs == BA_ASSEMBLING;
        if (bs == BA_READY) {
          q_ba.push_back(li);
          if (!g->lanes[li]->ba_ready_seq) g->lanes[li]->ba_ready_seq = ++g->ba_ready_counter;
        }
      }
      // who is offered to the admission first (it may run out of budget behind any of them): a lane whose next keyframe already
      // waits for this solve, then the solve that has waited longest — in lane order the high lanes of a large group starved
      // and became the stragglers of the call (profiles/r04_group_sweep.txt: 56 and 64 lanes)
      std::sort(q_ba.begin(), q_ba.end(), [&](int a, int b) {
        const Lane* la = g->lanes[a]; const Lane* lb = g->lanes[b];
        const bool wa = la->state == L_NEED_SOLVE, wb = lb->state == L_NEED_SOLVE;
        if (wa != wb) return wa;
        return la->ba_ready_seq < lb->ba_ready_seq;
      });
      bool ba_line_busy[svo_pipeline_group::MAX_LINES] = {};
      for (int li = 0; li < S; ++li) {
        const Lane* l = g->lanes[li];
        if (l->ba_state.load(std::memory_order_acquire) == BA_INFLIGHT && !svo_ba_solve_poll(l->ba)) ba_line_busy[l->ba_line] = true;
      }
      int free_line = -1;
      for (int i = 0; i < g->n_ba; ++i) if (!ba_line_busy[i]) { free_line = i; break; }
      // (rounds 3-4 held ready solves back while ANY lane was still assembling, to launch them together: measured in round 4,
      // a solve then left 1.1 ms after its assembly was posted on average — as long as it runs — and 3 ms at the 90th
      // percentile; SVO_GROUP_BA_WAIT_ASSEMBLY=1 restores that)
      static const bool wait_assembly = [] { const char* e = getenv("SVO_GROUP_BA_WAIT_ASSEMBLY"); return e && *e && atoi(e) != 0; }();
      if (!q_ba.empty() && (assembling == 0 || !wait_assembly) && free_line >= 0) {
        svo_ba* bas[SVO_MAX_LANES];
        std::vector<int> cand;
        for (int li : q_ba) {
          Lane* l = g->lanes[li];
          if (l->ba_rc == 1) { l->ba_ready_seq = 0; l->ba_state.store(BA_NONE, std::memory_order_release); continue; }  // nothing to solve
          if (l->ba_rc) { error = l->ba_rc; break; }
          bas[cand.size()] = l->ba;
          cand.push_back(li);
        }
        if (error) break;
        if (!cand.empty()) {
          unsigned long long mask = 0;
          static const bool host_solves = [] { const char* e = getenv("SVO_GROUP_HOST_SOLVES"); return e && *e && atoi(e) != 0; }();  // test hook: every window down the host-driven path (the landmark store is then filled by the scatter launch)
          const int launched = host_solves ? 0 : svo_ba_solve_launch(bas, (int)cand.size(), g->st_ba[free_line], &mask);
          if (launched > 0) { g->launches[4]++; g->lanes_carried[4] += launched; progressed = true; }
          ++g->ba_launch_id;
          int not_taken = -1;  // the first lane the launch skipped: not eligible, or not admitted right now
          for (int k = 0; k < (int)cand.size(); ++k) {
            if (!((mask >> k) & 1ull)) { if (not_taken < 0) not_taken = cand[k]; continue; }
            Lane* l = g->lanes[cand[k]];
            EV(cand[k], "ba_launch", launched);
            l->ba_launch = g->ba_launch_id; l->ba_line = free_line; l->ba_ready_seq = 0; l->ba_state.store(BA_INFLIGHT, std::memory_order_release);
          }
          if (not_taken >= 0 && g->n_cmp > 0 && !host_solves) {
            // compact lines (SVO_GROUP_COMPACT_LINES, round 5): what the admission budget refused leaves at once in the one-workgroup
            // form — no budget, 3-4x the latency — on a line of its own, so that the wide launches' lines stay free
            int cline = -1;
            for (int i = g->n_ba; i < g->n_ba + g->n_cmp; ++i) if (!ba_line_busy[i]) { cline = i; break; }
            if (cline >= 0) {
              svo_ba* cb[SVO_MAX_LANES];
              int cl[SVO_MAX_LANES], nc = 0;
              for (int k = 0; k < (int)cand.size(); ++k) if (!((mask >> k) & 1ull)) { cb[nc] = bas[k]; cl[nc++] = cand[k]; (void)svo_ba_set_solve_form(bas[k], 1); }
              unsigned long long cmask = 0;
              const int went = svo_ba_solve_launch(cb, nc, g->st_ba[cline], &cmask);
              for (int k = 0; k < nc; ++k) (void)svo_ba_set_solve_form(cb[k], -1);
              if (went > 0) {
                ++g->ba_launch_id;
                g->launches[4]++; g->lanes_carried[4] += went; progressed = true;
                for (int k = 0; k < nc; ++k) {
                  if (!((cmask >> k) & 1ull)) continue;
                  Lane* l = g->lanes[cl[k]];
                  EV(cl[k], "ba_launch_compact", went);
                  l->ba_launch = g->ba_launch_id; l->ba_line = cline; l->ba_ready_seq = 0; l->ba_state.store(BA_INFLIGHT, std::memory_order_release);
                }
                not_taken = -1;
                for (int k = 0; k < nc; ++k) if (!((cmask >> k) & 1ull)) { not_taken = cl[k]; break; }
              }
            }
          }
          if (not_taken >= 0) {
            // if nothing of this group is in flight that could free the admission budget (or the problem is simply not
            // eligible for the device-resident solve), the lane is solved by the host-driven loop on a worker; otherwise it
            // is offered again when a solve of this group has been joined
            bool inflight = false;
            for (int li = 0; li < S; ++li) inflight |= g->lanes[li]->ba_state.load(std::memory_order_acquire) == BA_INFLIGHT;
            if (!inflight) {
              Lane* l = g->lanes[not_taken];
              l->ba_ready_seq = 0;
              // SVO_GROUP_STALLED_COMPACT=1 (experiment, round 5): the compact device-resident form instead of the host-driven loop on a
              // worker.  Measured slower — 29 k against 41 k frames/s at 128 lanes (profiles/r05_exp_lanes_groups.txt): a compact solve
              // holds its solve line for ~3 ms, the worker's host-driven solve holds none.
              static const bool stalled_compact = [] { const char* e = getenv("SVO_GROUP_STALLED_COMPACT"); return e && *e && atoi(e) != 0; }();
              bool went = false;
              if (stalled_compact && !host_solves) {
                svo_ba* one = l->ba;
                (void)svo_ba_set_solve_form(one, 1);
                went = svo_ba_solve_launch(&one, 1, g->st_ba[free_line], nullptr) == 1;
                (void)svo_ba_set_solve_form(one, -1);
                if (went) {
                  EV(not_taken, "ba_launch_compact", 1);
                  l->ba_launch = ++g->ba_launch_id; l->ba_line = free_line; l->ba_state.store(BA_INFLIGHT, std::memory_order_release);
                  g->launches[4]++; g->lanes_carried[4]++;
                }
              }
              if (!went) {
                l->ba_state.store(BA_HOST_SOLVING, std::memory_order_release);
                g->pool.post(l, 1);
              }
              progressed = true;
            }
          }
        }
      }
    }
    if (all_done) break;
    if (!progressed) {
      __builtin_ia32_pause();
      if (++idle_spins > 2000u && (idle_spins & 255u) == 0) sched_yield();
      if ((idle_spins & 0xFFFFu) == 0) {  // never hang: a launch that does not come back within seconds is reported with the lanes' states
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - last_progress).count();
        if (waited > 5.0) {
          std::string msg = "pipeline group: no progress for 5 s; lanes (state/frame/solve):";
          for (int li = 0; li < S; ++li) {
            char b[64];
            snprintf(b, sizeof(b), " %d/%d/%d%s", g->lanes[li]->state, g->lanes[li]->frame, g->lanes[li]->ba_state.load(), g->lanes[li]->queued ? "q" : "");
            msg += b;
          }
          ctx->err = msg;
          error = SVO_ERR_HIP;
          break;
        }
      }
    } else {
      idle_spins = 0;
      last_progress = std::chrono::steady_clock::now();
      busy_us += std::chrono::duration<double, std::micro>(last_progress - t_pass).count();
    }
  }
  g->launches[2] += (long)busy_us;
  g->lanes_carried[2] += (long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();

  // ---- end of the batch: join every solve, fill the poses that waited for it
  for (int li = 0; li < S; ++li) {
    Lane* l = g->lanes[li];
    for (;;) {  // a lane whose assembly is still running, or whose solve is not launched yet
      const int bs = l->ba_state.load(std::memory_order_acquire);
      if (bs == BA_NONE || bs == BA_INFLIGHT || bs == BA_HOST_DONE) break;
      if (bs == BA_READY) {
        if (l->ba_rc == 1) { l->ba_state.store(BA_NONE); break; }
        if (l->ba_rc) { if (!error) error = l->ba_rc; l->ba_state.store(BA_NONE); break; }
        svo_ba* one = l->ba;
        if (svo_ba_solve_launch(&one, 1, g->st_ba[0], nullptr) == 1) { l->ba_launch = ++g->ba_launch_id; l->ba_line = 0; l->ba_state.store(BA_INFLIGHT); g->launches[4]++; g->lanes_carried[4]++; }
        else { l->ba_rc = svo_ba_solve_finish(l->ba, &l->ba_summary); l->ba_state.store(BA_HOST_DONE); }
        break;
      }
      __builtin_ia32_pause();
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - last_progress).count() > 60.0) {
        if (!error) { error = SVO_ERR_HIP; ctx->err = "pipeline group: a bundle-adjustment worker did not come back"; }
        break;
      }
    }
    const int rcf = finish_solve(g, l);
    if (rcf && !error) error = rcf;
    svo_frame_result* res = &RES(li, 0);
    if (l->pending_from < 0) l->pending_from = batch;
    fill_pending(l, res, batch);
  }
  // A lane whose last tracked image is NOT the batch's last frame (its later frames had too few corners to track, C-7)
  // still reads that image's level 0 from the caller's buffer: clone it into its pyramid now (src/feature_tracker.cpp:14,66),
  // before the caller may reuse the buffer.
  {
    bool copied = false;
    for (int li = 0; li < S && !error; ++li) {
      Lane* l = g->lanes[li];
      if (!l->last_pyr || l->last_l0 == l->last_pyr || !l->last_l0) continue;
      if (svo_k_pyramid_level0(ctx, l->last_l0, 1, W, H, W, istride, const_cast<uint8_t*>(l->last_pyr), g->pyr_stride, st)) { error = SVO_ERR_HIP; break; }
      l->last_l0 = l->last_pyr;
      copied = true;
    }
    if (copied && hipStreamSynchronize(st) != hipSuccess && !error) { error = SVO_ERR_HIP; ctx->err = "pipeline group: pyramid clone failed"; }
  }
  if (trace_on) {
    static std::mutex trace_mu;  // the groups of a process print their calls one after the other
    std::lock_guard<std::mutex> lk(trace_mu);
    for (const Ev& e : evs) fprintf(stderr, "[svo group] %10.1f lane %2d %-22s %d\n", e.us, e.lane, e.what, e.arg);
    fprintf(stderr, "[svo group] %10.1f end\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
  }
  if (error) { quiesce_after_error(g); return error; }
  if (!ctx->err.empty()) return SVO_ERR_HIP;
  return SVO_OK;
}

// ---- host-pointer / streaming entry (include/svo.h): the reference's images are host cv::Mat copies made in the image
// callback (src/vo_node.cpp:70-73) and consumed frame by frame (:141-143)
namespace {
int ensure_staging(svo_pipeline_group* g) {
  if (g->st_copy) return SVO_OK;
  svo_ctx* ctx = g->ctx;
  SVO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t)g->n_lanes * (size_t)g->max_batch * (size_t)g->prm.width * (size_t)g->prm.height;
  // only what is still missing: a call that failed half way (out of pinned memory) leaves st_copy null and comes back here
  for (int sl = 0; sl < 2; ++sl) {
    for (int e = 0; e < 2; ++e) {
      if (!g->h_stage[sl][e]) SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&g->h_stage[sl][e], bytes, hipHostMallocDefault));
      if (!g->d_stage[sl][e]) SVO_HIP_CHECK(ctx, hipMalloc((void**)&g->d_stage[sl][e], bytes));
    }
    if (!g->ev_up[sl]) SVO_HIP_CHECK(ctx, hipEventCreateWithFlags(&g->ev_up[sl], hipEventDisableTiming));
  }
  SVO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&g->st_copy, hipStreamNonBlocking));
  return SVO_OK;
}
}  // namespace

extern "C" int svo_pipeline_group_staging(svo_pipeline_group* g, int slot, uint8_t** left, uint8_t** right, size_t* lane_stride) {
  if (!g || slot < 0 || slot > 1 || !left || !right) return SVO_ERR_INVALID;
  const int rc = ensure_staging(g);
  if (rc) return rc;
  *left = g->h_stage[slot][0]; *right = g->h_stage[slot][1];
  if (lane_stride) *lane_stride = (size_t)g->max_batch * (size_t)g->prm.width * (size_t)g->prm.height;
  return SVO_OK;
}

extern "C" int svo_pipeline_group_upload(svo_pipeline_group* g, int slot, int batch) {
  if (!g || slot < 0 || slot > 1) return SVO_ERR_INVALID;
  svo_ctx* ctx = g->ctx;
  SVO_REQUIRE(ctx, batch >= 1 && batch <= g->max_batch, "pipeline_group_upload: batch outside 1..max_batch");
  int rc = ensure_staging(g);
  if (rc) return rc;
  SVO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t istride = (size_t)g->prm.width * (size_t)g->prm.height, lane_bytes = (size_t)g->max_batch * istride;
  for (int e = 0; e < 2; ++e) {
    if (batch == g->max_batch)  // one contiguous run
      SVO_HIP_CHECK(ctx, hipMemcpyAsync(g->d_stage[slot][e], g->h_stage[slot][e], lane_bytes * (size_t)g->n_lanes, hipMemcpyHostToDevice, g->st_copy));
    else                        // the first `batch` frames of every lane
      SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(g->d_stage[slot][e], lane_bytes, g->h_stage[slot][e], lane_bytes, (size_t)batch * istride, (size_t)g->n_lanes,
                                          hipMemcpyHostToDevice, g->st_copy));
  }
  SVO_HIP_CHECK(ctx, hipEventRecord(g->ev_up[slot], g->st_copy));
  g->up_batch[slot] = batch;
  return SVO_OK;
}

extern "C" int svo_pipeline_group_process_uploaded(svo_pipeline_group* g, int slot, svo_frame_result* results) {
  if (!g || slot < 0 || slot > 1) return SVO_ERR_INVALID;
  svo_ctx* ctx = g->ctx;
  SVO_REQUIRE(ctx, g->st_copy && g->up_batch[slot] > 0, "pipeline_group_process_uploaded: nothing was uploaded into this slot");
  SVO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  SVO_HIP_CHECK(ctx, hipEventSynchronize(g->ev_up[slot]));  // started a whole batch ago in a streaming loop: long complete
  const int batch = g->up_batch[slot];
  g->up_batch[slot] = 0;
  const size_t lane_stride = (size_t)g->max_batch * (size_t)g->prm.width * (size_t)g->prm.height;
  return svo_pipeline_group_process_batch_dev(g, g->d_stage[slot][0], g->d_stage[slot][1], lane_stride, batch, results);
}

extern "C" int svo_pipeline_group_process_batch(svo_pipeline_group* g, const uint8_t* left, const uint8_t* right, size_t lane_stride,
                                                int batch, svo_frame_result* results) {
  if (!g) return SVO_ERR_INVALID;
  svo_ctx* ctx = g->ctx;
  SVO_REQUIRE(ctx, left && right && results && batch >= 1 && batch <= g->max_batch, "pipeline_group_process_batch: bad arguments");
  const size_t istride = (size_t)g->prm.width * (size_t)g->prm.height;
  SVO_REQUIRE(ctx, lane_stride >= istride * (size_t)batch, "pipeline_group_process_batch: lanes overlap");
  int rc = ensure_staging(g);
  if (rc) return rc;
  const size_t stage_stride = (size_t)g->max_batch * istride;
  for (int l = 0; l < g->n_lanes; ++l) {
    memcpy(g->h_stage[0][0] + (size_t)l * stage_stride, left + (size_t)l * lane_stride, (size_t)batch * istride);
    memcpy(g->h_stage[0][1] + (size_t)l * stage_stride, right + (size_t)l * lane_stride, (size_t)batch * istride);
  }
  if ((rc = svo_pipeline_group_upload(g, 0, batch))) return rc;
  return svo_pipeline_group_process_uploaded(g, 0, results);
}
