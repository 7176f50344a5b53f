// oracle: cv::calcOpticalFlowPyrLK(prev, next, pts, out, status, err, Size(21,21), 3,
//   TermCriteria(COUNT+EPS, 30, 0.01), 0, 1e-2)  — src/feature_tracker.cpp:23-26,32-35 —
// and FeatureTracker::track_features' survivor filter — src/feature_tracker.cpp:38-64.
// TEST INFRASTRUCTURE ONLY.  The LK part is PARITY UNPINNED vs OpenCV (restates SURVEY.md
// Appendix A.3); the survivor filter follows first-party reference source.
//
// Declared arithmetic (what the HIP kernel repeats bit for bit):
//   * pyramid: pyrDown = separable [1 4 6 4 1], REFLECT_101, (sum+128)>>8, size (w+1)/2 x (h+1)/2.
//   * image outside its bounds: REFLECT_101 (OpenCV pads each level by winSize with that border);
//     Scharr derivative ([3 10 3] x [-1 0 1], int16) uses REFLECT_101 for its own taps inside the
//     image and is 0 outside the image (BORDER_CONSTANT padding of the derivative buffer).
//   * bilinear weights are 14-bit integers, cvRound = round-half-even; patches are int16
//     (I descaled by 2^9, derivatives by 2^14).
//   * the window sums A11,A12,A22,b1,b2 are accumulated EXACTLY in int64 (OpenCV uses float or
//     int32-SIMD partial sums depending on the build; an exact sum is order independent, which is
//     what makes CPU/GPU index sets bit-identical), converted int64 -> double -> float, then
//     scaled by 2^-20.  The 2x2 solve is f32 with no FMA contraction.
#include <cmath>
#include <cstring>
#include <vector>

#include "ora_constants.h"
#include "svo_oracle.h"

namespace {
const int kWin = ora_k::kLkWin, kHalf = kWin / 2, kLevels = ora_k::kLkMaxLevel + 1, kMaxIter = ora_k::kLkMaxIterations;
const float kMinEigThreshold = ora_k::kLkMinEigThreshold;
const double kEps = ora_k::kLkEpsilon;
const float kFltScale = 1.0f / (float)(1 << 20);
const float kFltEpsilon = 1.1920928955078125e-7f;

inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

struct Level { const uint8_t* p; int w, h; };

inline int px(const Level& L, int x, int y) { return L.p[(size_t)reflect101(y, L.h) * L.w + reflect101(x, L.w)]; }

// Scharr derivative at integer position; zero outside the image.
inline void scharr(const Level& L, int x, int y, int* ix, int* iy) {
  if (x < 0 || x >= L.w || y < 0 || y >= L.h) { *ix = 0; *iy = 0; return; }
  // vertical pass at columns x-1, x, x+1 (column index reflected inside the image)
  int t0[3], t1[3];
  for (int k = -1; k <= 1; ++k) {
    const int xc = reflect101(x + k, L.w);
    const int a = L.p[(size_t)reflect101(y - 1, L.h) * L.w + xc];
    const int b = L.p[(size_t)y * L.w + xc];
    const int c = L.p[(size_t)reflect101(y + 1, L.h) * L.w + xc];
    t0[k + 1] = (a + c) * 3 + b * 10;
    t1[k + 1] = c - a;
  }
  *ix = (int16_t)(t0[2] - t0[0]);
  *iy = (int16_t)((t1[2] + t1[0]) * 3 + t1[1] * 10);
}

inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }
inline int cv_round(float v) { return (int)std::lrintf(v); }
inline int cv_floor(float v) { return (int)std::floor(v); }

void pyr_down(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh) {
  static const int k[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      int s = 0;
      for (int j = 0; j < 5; ++j) {
        const int yy = reflect101(2 * y + j - 2, sh);
        int r = 0;
        for (int i = 0; i < 5; ++i) r += k[i] * src[(size_t)yy * sstride + reflect101(2 * x + i - 2, sw)];
        s += k[j] * r;
      }
      dst[(size_t)y * dw + x] = (uint8_t)((s + 128) >> 8);
    }
}

struct Pyramid {
  std::vector<uint8_t> buf;
  Level lv[kLevels];
};

void build(const uint8_t* img, int w, int h, int stride, Pyramid& P) {
  P.buf.resize(ora_pyramid_bytes(w, h, kLevels));
  ora_build_pyramid(img, w, h, stride, kLevels, P.buf.data());
  size_t off = 0;
  int lw = w, lh = h;
  for (int l = 0; l < kLevels; ++l) {
    P.lv[l] = {P.buf.data() + off, lw, lh};
    off += (size_t)lw * lh;
    lw = (lw + 1) / 2;
    lh = (lh + 1) / 2;
  }
}

// One point through all levels.  Returns status.
uint8_t lk_point(const Pyramid& A, const Pyramid& B, float px0, float py0, float* ox, float* oy) {
  uint8_t status = 1;
  float nx = 0, ny = 0;
  int16_t Iw[kWin * kWin], dIx[kWin * kWin], dIy[kWin * kWin];
  for (int level = kLevels - 1; level >= 0; --level) {
    const Level& I = A.lv[level];
    const Level& J = B.lv[level];
    const float sc = (float)(1.0 / (double)(1 << level));
    float pxl = px0 * sc, pyl = py0 * sc;
    if (level == kLevels - 1) { nx = pxl; ny = pyl; } else { nx = nx * 2.0f; ny = ny * 2.0f; }
    pxl -= (float)kHalf; pyl -= (float)kHalf;
    const int ipx = cv_floor(pxl), ipy = cv_floor(pyl);
    if (ipx < -kWin || ipx >= I.w || ipy < -kWin || ipy >= I.h) {
      if (level == 0) status = 0;
      continue;
    }
    float a = pxl - (float)ipx, b = pyl - (float)ipy;
    int iw00 = cv_round((1.f - a) * (1.f - b) * (float)(1 << 14));
    int iw01 = cv_round(a * (1.f - b) * (float)(1 << 14));
    int iw10 = cv_round((1.f - a) * b * (float)(1 << 14));
    int iw11 = (1 << 14) - iw00 - iw01 - iw10;
    int64_t sA11 = 0, sA12 = 0, sA22 = 0;
    // (kWin+1)^2 grid of pixel values and Scharr derivatives, then the bilinear patches
    int gI[(kWin + 1) * (kWin + 1)], gX[(kWin + 1) * (kWin + 1)], gY[(kWin + 1) * (kWin + 1)];
    for (int y = 0; y <= kWin; ++y)
      for (int x = 0; x <= kWin; ++x) {
        const int o = y * (kWin + 1) + x;
        gI[o] = px(I, ipx + x, ipy + y);
        scharr(I, ipx + x, ipy + y, &gX[o], &gY[o]);
      }
    for (int y = 0; y < kWin; ++y)
      for (int x = 0; x < kWin; ++x) {
        const int o = y * (kWin + 1) + x, o1 = o + kWin + 1;
        const int ival = descale(gI[o] * iw00 + gI[o + 1] * iw01 + gI[o1] * iw10 + gI[o1 + 1] * iw11, 9);
        const int ixv = descale(gX[o] * iw00 + gX[o + 1] * iw01 + gX[o1] * iw10 + gX[o1 + 1] * iw11, 14);
        const int iyv = descale(gY[o] * iw00 + gY[o + 1] * iw01 + gY[o1] * iw10 + gY[o1 + 1] * iw11, 14);
        Iw[y * kWin + x] = (int16_t)ival;
        dIx[y * kWin + x] = (int16_t)ixv;
        dIy[y * kWin + x] = (int16_t)iyv;
        sA11 += (int64_t)(ixv * ixv);
        sA12 += (int64_t)(ixv * iyv);
        sA22 += (int64_t)(iyv * iyv);
      }
    const float A11 = (float)(double)sA11 * kFltScale;
    const float A12 = (float)(double)sA12 * kFltScale;
    const float A22 = (float)(double)sA22 * kFltScale;
    float D = A11 * A22 - A12 * A12;
    const float dif = A11 - A22;
    const float minEig = (A22 + A11 - std::sqrt(dif * dif + 4.f * A12 * A12)) / (float)(2 * kWin * kWin);
    if (minEig < kMinEigThreshold || D < kFltEpsilon) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;
    float outx = nx, outy = ny;  // nextPts[ptidx] = nextPt is stored before halfWin is subtracted
    nx -= (float)kHalf; ny -= (float)kHalf;
    float pdx = 0, pdy = 0;
    for (int j = 0; j < kMaxIter; ++j) {
      const int inx = cv_floor(nx), iny = cv_floor(ny);
      if (inx < -kWin || inx >= J.w || iny < -kWin || iny >= J.h) {
        if (level == 0) status = 0;
        break;
      }
      a = nx - (float)inx; b = ny - (float)iny;
      iw00 = cv_round((1.f - a) * (1.f - b) * (float)(1 << 14));
      iw01 = cv_round(a * (1.f - b) * (float)(1 << 14));
      iw10 = cv_round((1.f - a) * b * (float)(1 << 14));
      iw11 = (1 << 14) - iw00 - iw01 - iw10;
      int64_t sb1 = 0, sb2 = 0;
      for (int y = 0; y < kWin; ++y)
        for (int x = 0; x < kWin; ++x) {
          const int X = inx + x, Y = iny + y;
          const int diff = descale(px(J, X, Y) * iw00 + px(J, X + 1, Y) * iw01 + px(J, X, Y + 1) * iw10 +
                                       px(J, X + 1, Y + 1) * iw11, 9) - Iw[y * kWin + x];
          sb1 += (int64_t)(diff * dIx[y * kWin + x]);
          sb2 += (int64_t)(diff * dIy[y * kWin + x]);
        }
      const float b1 = (float)(double)sb1 * kFltScale;
      const float b2 = (float)(double)sb2 * kFltScale;
      const float dx = (A12 * b2 - A22 * b1) * D;
      const float dy = (A12 * b1 - A11 * b2) * D;
      nx += dx; ny += dy;
      outx = nx + (float)kHalf; outy = ny + (float)kHalf;
      if ((double)dx * (double)dx + (double)dy * (double)dy <= kEps * kEps) break;
      if (j > 0 && std::fabs(dx + pdx) < kEps && std::fabs(dy + pdy) < kEps) {
        outx -= dx * 0.5f; outy -= dy * 0.5f;
        break;
      }
      pdx = dx; pdy = dy;
    }
    nx = outx; ny = outy;  // nextPts[ptidx] carried to the next level
    if (status && level == 0) {
      // final bounds check done when err is requested (the reference passes `err`)
      const float fx = nx - (float)kHalf, fy = ny - (float)kHalf;
      const int ix = cv_floor(fx), iy = cv_floor(fy);
      if (ix < -kWin || ix >= J.w || iy < -kWin || iy >= J.h) status = 0;
    }
  }
  *ox = nx; *oy = ny;
  return status;
}
}  // namespace

extern "C" size_t ora_pyramid_bytes(int w, int h, int levels) {
  size_t s = 0;
  for (int l = 0; l < levels; ++l) { s += (size_t)w * h; w = (w + 1) / 2; h = (h + 1) / 2; }
  return s;
}

extern "C" void ora_build_pyramid(const uint8_t* img, int w, int h, int stride, int levels,
                                  uint8_t* out) {
  for (int y = 0; y < h; ++y) std::memcpy(out + (size_t)y * w, img + (size_t)y * stride, w);
  uint8_t* prev = out;
  int pw = w, ph = h;
  for (int l = 1; l < levels; ++l) {
    uint8_t* cur = prev + (size_t)pw * ph;
    const int cw = (pw + 1) / 2, ch = (ph + 1) / 2;
    pyr_down(prev, pw, ph, pw, cur, cw, ch);
    prev = cur; pw = cw; ph = ch;
  }
}

extern "C" void ora_lk_track(const uint8_t* prev, const uint8_t* next, int w, int h, int stride,
                             const float* xy, int n, float* out_xy, uint8_t* status) {
  Pyramid A, B;
  build(prev, w, h, stride, A);
  build(next, w, h, stride, B);
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n; ++i)
    status[i] = lk_point(A, B, xy[2 * i], xy[2 * i + 1], &out_xy[2 * i], &out_xy[2 * i + 1]);
}

// FeatureTracker::track_features, src/feature_tracker.cpp:18-67.
//   :23-26  forward LK  last_image -> image
//   :31-36  backward LK image -> last_image from the forward result
//   :44-47  keep iff status1 && status2 && norm(old - back) < 2   (cv::norm on Point2f: double;
//           evaluated as dx*dx+dy*dy < 4.0 in double — same decision, avoids a double sqrt)
//   :48-55  parallax = sqrt(dx^2+dy^2) (float) vs the keyframe position; drop if > 200
//   :59,63  av_parallax = (sequential float sum over kept) / n_all      (SURVEY C-2)
extern "C" int ora_track_features(const uint8_t* prev, const uint8_t* next, int w, int h,
                                  int stride, const float* xy, const float* initial_xy, int n,
                                  float* kept_xy, int* kept_index, float* av_parallax) {
  Pyramid A, B;
  build(prev, w, h, stride, A);
  build(next, w, h, stride, B);
  std::vector<float> fwd(2 * (size_t)n), back(2 * (size_t)n);
  std::vector<uint8_t> s1(n), s2(n);
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n; ++i) {
    s1[i] = lk_point(A, B, xy[2 * i], xy[2 * i + 1], &fwd[2 * i], &fwd[2 * i + 1]);
    s2[i] = lk_point(B, A, fwd[2 * i], fwd[2 * i + 1], &back[2 * i], &back[2 * i + 1]);
  }
  int m = 0;
  float sum = 0.f;
  for (int i = 0; i < n; ++i) {
    if (!s1[i] || !s2[i]) continue;
    const float ex = xy[2 * i] - back[2 * i], ey = xy[2 * i + 1] - back[2 * i + 1];
    if (!((double)ex * (double)ex + (double)ey * (double)ey < ora_k::kFbMaxDistance * ora_k::kFbMaxDistance)) continue;
    const float dx = fwd[2 * i] - initial_xy[2 * i];
    const float dy = fwd[2 * i + 1] - initial_xy[2 * i + 1];
    const float parallax = std::sqrt(dx * dx + dy * dy);
    if (parallax > ora_k::kMaxParallax) continue;
    kept_xy[2 * m] = fwd[2 * i];
    kept_xy[2 * m + 1] = fwd[2 * i + 1];
    kept_index[m] = i;
    sum += parallax;
    ++m;
  }
  *av_parallax = n > 0 ? sum / (float)n : 0.f;
  return m;
}
