// a1 — Shi-Tomasi corner detection, the cv::goodFeaturesToTrack call at
// reference src/image_processor.cpp:22 (semantics: SURVEY.md Appendix A.1; the exact float
// operation order is the one declared in oracle/ora_corner.cpp and repeated here bit for bit).
//
// Three launches per batch of frames (grid.z / blockIdx.x = frame).  The detection path (round 3) does not write the f32
// response map at all: corner_response_nms_kernel keeps a three-row window of responses in registers, takes the 3x3
// non-maximum test there and appends only local maxima above a running lower bound of the threshold (4A bytes written and
// 4A read per image become a list of a few thousand 8-byte entries); corner_threshold_kernel then applies the exact
// threshold quality * max.  The two-launch form below (response map + NMS over the map) remains for the response tap
// (svo_corner_response) and as SVO_CORNER_TWO_PASS=1:
//   corner_response_kernel : streaming, wavefront = 62-column strip, lane = column: Sobel -> products -> 3x3 box
//                            (double, DPP neighbour moves + 3-row register window) -> min eigenvalue map (f32) +
//                            per-image max (order-preserving uint atomicMax).  HBM: reads ~1.1 A bytes, writes 4A.
//   corner_nms_kernel      : threshold at quality*max, 3x3 non-max suppression, wave-aggregated
//                            append of (value key << 32 | raster index) candidates.  Reads 4A.
//   corner_select_kernel   : one 1024-thread workgroup per image.  The reference's greedy
//                            "strongest first, reject within minDistance" scan is the lexicographically
//                            first maximal independent set under the total order (value desc, raster
//                            index desc); it is computed in parallel by monotone fixed-point rounds
//                            (a candidate is accepted once every stronger neighbour is rejected, rejected
//                            once any stronger neighbour is accepted) over a cell-binned candidate list,
//                            then the accepted set is bitonic-sorted in LDS and truncated to maxCorners.
//                            The result is identical to the sequential scan, including the truncation,
//                            because acceptance of a candidate depends only on stronger candidates.
#include "common.h"

namespace {
constexpr int SEL_THREADS = 1024;
constexpr int SEL_MAX_ACCEPT = 16384;    // accepted corners held in LDS for the final sort (aliases the key cache)
constexpr int SEL_CAP_K = 12288;         // candidates whose keys/state are cached in LDS during the rounds
constexpr int SEL_CAP_C = 8192;          // cell-table entries cached in LDS
constexpr int SEL_LDS_BYTES = SEL_CAP_K * 8 + SEL_CAP_C * 2 + SEL_CAP_K + SEL_CAP_K * 2;  // keys | cell table (u16) | state | blocker (u16): 151,552 B of the CU's 160 KiB (>= SEL_MAX_ACCEPT * 8)
constexpr int NC_STRIDE = 32;            // per-image candidate counters live on separate 128-B lines

__device__ __forceinline__ unsigned f32_key(float v) {
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// Candidate key = (response key << 32) | (y << 16 | x).  The low word orders candidates exactly like the raster index
// y * W + x (x < W <= 65535), which is the tie-break of the total order, and decodes without integer divisions.
__device__ __forceinline__ unsigned pack_xy(int x, int y) { return ((unsigned)y << 16) | (unsigned)x; }
__device__ __forceinline__ int key_x(unsigned lo) { return (int)(lo & 0xFFFFu); }
__device__ __forceinline__ int key_y(unsigned lo) { return (int)(lo >> 16); }
template <typename T>
__device__ __forceinline__ T ld_l2(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void st_l2(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}  // namespace

// Streaming form: one wavefront owns a strip of 62 output columns x RS_ROWS rows and walks it top to bottom.
// lane = image column (one halo column on each side), pixel rows are read once (1 byte per lane, one 64-byte
// request per row and wave, RS_PF rows ahead), neighbours come from wave-shift DPP moves, the 3x3 box sum is a
// horizontal 3-sum (DPP) followed by a 3-row sliding window in registers: no LDS, every pixel read ~1.1 times,
// every response written once.  The box sum adds nine float products whose exponents span < 2^29, so the double
// sum is exact in any order (oracle/ora_corner.cpp adds them row by row): reordering keeps the result bit-identical.
// Image borders: Sobel reads BORDER_REFLECT_101 pixels; the box filter reflects COVARIANCE coordinates (a position
// outside the image takes the covariance computed at its reflection), i.e. cov(-1) := cov(1), cov(n) := cov(n-2).
namespace {
constexpr int RS_COLS = 62;   // output columns per wavefront (64 lanes minus one halo column per side)
constexpr int RS_ROWS = 48;   // output rows per wavefront (measured 12/24/32/48/64/96/192: 138/84/71/65/66/79/132 us per 16 KITTI frames;
                              // six instead of three pixel rows in flight: no change)
constexpr int RS_PF = 3;      // pixel rows in flight (= the unroll factor: the three-row windows rotate by renaming)

__device__ __forceinline__ int wave_from_lower(int v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, false); }   // lane i <- lane i-1 (wave_shr:1)
__device__ __forceinline__ int wave_from_upper(int v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, false); }   // lane i <- lane i+1 (wave_shl:1)
__device__ __forceinline__ float wave_from_lower(float v) { return __int_as_float(wave_from_lower(__float_as_int(v))); }
__device__ __forceinline__ float wave_from_upper(float v) { return __int_as_float(wave_from_upper(__float_as_int(v))); }
}  // namespace

__global__ __launch_bounds__(256) void corner_response_kernel(const uint8_t* __restrict__ imgs, int W, int H,
                                                              int row_stride, size_t image_stride,
                                                              float* __restrict__ eig,
                                                              unsigned* __restrict__ maxkey) {
  const int b = blockIdx.z;
  const uint8_t* img = imgs + (size_t)b * image_stride;
  const int lane = threadIdx.x & 63;
  const int ncs = (W + RS_COLS - 1) / RS_COLS, nrs = (H + RS_ROWS - 1) / RS_ROWS;
  const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: strip bounds stay scalar
  if (wid >= ncs * nrs) return;  // whole wavefront
  const int cs = wid % ncs, rs = wid / ncs;
  const int x = cs * RS_COLS - 1 + lane;             // image column of this lane (may be -1 or >= W)
  const int y0 = rs * RS_ROWS, y1 = min(y0 + RS_ROWS, H);
  const int xr = reflect101(x, W);
  // lanes 0 / 63 have no lower / upper neighbour lane: they fetch that pixel themselves
  const int xe = reflect101(lane == 0 ? x - 1 : x + 1, W);
  const bool edge = lane == 0 || lane == 63;
  const bool fix_lo = x == -1, fix_hi = x == W;      // covariance taken from the reflected column (lane + 2 / lane - 2)
  const bool border_wave = cs == 0 || (cs + 1) * RS_COLS - 1 + 1 >= W;  // wave-uniform: some lane is outside [0, W)
  const double scale = 1.0 / (4.0 * 3.0 * 255.0);
  const float k1 = (float)(1.0 * scale), k0 = (float)(2.0 * scale);
  float* E = eig + (size_t)b * W * H;
  const bool out_lane = lane >= 1 && lane <= RS_COLS && x < W;

  // covariance rows needed: j in [j0, j1]; pixel rows j0-1 .. j1+1 (reflected)
  const int j0 = y0 > 0 ? y0 - 1 : 0, j1 = y1 < H ? y1 : H - 1;
  auto load_row = [&](int py, int& c, int& e) {
    const uint8_t* row = img + (size_t)reflect101(py, H) * row_stride;
    c = row[xr];
    e = edge ? (int)row[xe] : 0;
  };
  // per pixel row: rdx = r - l, rdy = m k0 + (l + r) k1 (the row terms of the separable Sobel pair)
  auto row_terms = [&](int c, int e, float& rdx, float& rdy) {
    int l = wave_from_lower(c), r = wave_from_upper(c);
    if (lane == 0) l = e;
    if (lane == 63) r = e;
    rdx = (float)(r - l);
    rdy = (float)c * k0 + (float)(l + r) * k1;
  };
  const int p_first = j0 - 1;            // first pixel row
  const int p_last = j1 + 1;             // last pixel row
  unsigned lmax = 0u;
  auto emit = [&](int y, const double* top, const double* mid, const double* bot) {
    const float a = (float)((top[0] + mid[0]) + bot[0]) * 0.5f;
    const float bq = (float)((top[1] + mid[1]) + bot[1]);
    const float c = (float)((top[2] + mid[2]) + bot[2]) * 0.5f;
    const float d = a - c;
    const float e = (a + c) - sqrtf(d * d + bq * bq);
    if (out_lane) {
      E[(size_t)y * W + x] = e;
      const unsigned k = f32_key(e);
      lmax = k > lmax ? k : lmax;
    }
  };
  // One step consumes pixel row p (already in c/e), refills that queue slot with row p + 3, completes covariance row
  // j = p - 1 and output row j - 1.  The three-row windows (row terms A,B,C; horizontal sums HA,HB,HC) rotate by
  // renaming: the loop is unrolled by three and each copy is called with permuted arguments, so nothing is moved.
  auto step = [&](int p, int& c, int& e, float& dxA, float& dyA, float& dxB, float& dyB, float& dxC, float& dyC, double* HA,
                  double* HB, double* HC) {
    const int cc = c, ee = e;
    if (p + RS_PF <= p_last) load_row(p + RS_PF, c, e);
    row_terms(cc, ee, dxC, dyC);
    const int j = p - 1;
    if (j < j0) return;
    const float dx = (dxA + dxC) * k1 + dxB * k0;
    const float dy = dyC - dyA;
    float xx = dx * dx, xy = dx * dy, yy = dy * dy;
    if (border_wave) {                    // covariance coordinates reflect: column -1 := column 1, column W := column W-2
      const int src = fix_lo ? lane + 2 : lane - 2;
      const float x2 = __shfl(xx, src), y2 = __shfl(xy, src), z2 = __shfl(yy, src);
      if (fix_lo || fix_hi) { xx = x2; xy = y2; yy = z2; }
    }
    HC[0] = ((double)wave_from_lower(xx) + (double)xx) + (double)wave_from_upper(xx);
    HC[1] = ((double)wave_from_lower(xy) + (double)xy) + (double)wave_from_upper(xy);
    HC[2] = ((double)wave_from_lower(yy) + (double)yy) + (double)wave_from_upper(yy);
    const int y = j - 1;                  // output row whose window (j-2, j-1, j) is now complete
    if (y < y0) return;
    if (y == 0) emit(0, HC, HB, HC);      // covariance row -1 := row 1 (first strip only)
    else emit(y, HA, HB, HC);
  };
  int c0, e0, c1, e1, c2, e2;
  load_row(p_first, c0, e0); load_row(p_first + 1, c1, e1); load_row(p_first + 2, c2, e2);
  float dx0 = 0.f, dy0 = 0.f, dx1 = 0.f, dy1 = 0.f, dx2 = 0.f, dy2 = 0.f;
  double h0[3] = {0, 0, 0}, h1[3] = {0, 0, 0}, h2[3] = {0, 0, 0};
  int p = p_first;
  for (; p + 2 <= p_last; p += 3) {
    step(p, c0, e0, dx1, dy1, dx2, dy2, dx0, dy0, h1, h2, h0);
    step(p + 1, c1, e1, dx2, dy2, dx0, dy0, dx1, dy1, h2, h0, h1);
    step(p + 2, c2, e2, dx0, dy0, dx1, dy1, dx2, dy2, h0, h1, h2);
  }
  // 0..2 leftover rows; then the last strip emits row H-1 with covariance row H := row H-2 (hp = sums of the row
  // before the last covariance row, hl = the last one; copied by value in wave-uniform branches: no scratch)
  double hp[3] = {h1[0], h1[1], h1[2]}, hl[3] = {h2[0], h2[1], h2[2]};
  if (p <= p_last) {
    step(p, c0, e0, dx1, dy1, dx2, dy2, dx0, dy0, h1, h2, h0);
    for (int q = 0; q < 3; ++q) { hp[q] = h2[q]; hl[q] = h0[q]; }
    ++p;
    if (p <= p_last) {
      step(p, c1, e1, dx2, dy2, dx0, dy0, dx1, dy1, h2, h0, h1);
      for (int q = 0; q < 3; ++q) { hp[q] = h0[q]; hl[q] = h1[q]; }
    }
  }
  if (j1 == H - 1 && y1 == H && H - 1 >= y0) emit(H - 1, hp, hl, hp);
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = __shfl_xor(lmax, off);
    lmax = o > lmax ? o : lmax;
  }
  if (lane == 0 && lmax) atomicMax(&maxkey[b], lmax);
}


// ---- response + non-maximum suppression in one streaming pass -----------------------------------------------------------
// The NMS predicate of corner_nms_kernel — v = (c > thr ? c : 0) equals the maximum of the thresholded 3x3 neighbourhood,
// for 1 <= x <= W-2, 1 <= y <= H-2 — holds exactly when c > thr and c >= every one of its eight RAW neighbours (a neighbour
// at or below the threshold is below c anyway), so the local-maximum test does not need the threshold, and the threshold
// quality * max (known only when the whole image is done) is applied afterwards to the few candidates that survive.  While
// streaming, a candidate is dropped early when it is not above quality * (the image's maximum SO FAR): that bound only grows
// towards the final threshold, so nothing the final test would keep is lost (which raw candidates are recorded depends on
// timing; the thresholded SET does not, and corner_select_kernel orders by key).
// Geometry: as corner_response_kernel with two halo columns per side (responses are valid on lanes 1..62, the test needs
// lanes +-1: 60 output columns per wavefront) and one extra response row above and below the strip.
constexpr int RN_HALO = 2, RN_COLS = 64 - 2 * RN_HALO;
constexpr int RN_BUF = 320;   // raw candidates a wavefront collects in LDS before it claims list space (one atomic per flush)
constexpr int NC_RAW = 16;    // word of an image's counter line that counts its raw candidates

__global__ __launch_bounds__(256) void corner_response_nms_kernel(const uint8_t* __restrict__ imgs, int W, int H, int row_stride,
                                                                  size_t image_stride, unsigned* __restrict__ maxkey, double quality,
                                                                  unsigned long long* __restrict__ raw, size_t raw_stride /* entries per image */,
                                                                  int* __restrict__ ncand, int* __restrict__ status) {
  __shared__ unsigned long long sBuf[4][RN_BUF];
  const int b = blockIdx.z;
  const uint8_t* img = imgs + (size_t)b * image_stride;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ncs = (W + RN_COLS - 1) / RN_COLS, nrs = (H + RS_ROWS - 1) / RS_ROWS;
  const int wid = blockIdx.x * 4 + wave;
  if (wid >= ncs * nrs) return;  // whole wavefront
  const int cs = wid % ncs, rs = wid / ncs;
  const int x = cs * RN_COLS - RN_HALO + lane;      // image column of this lane (may be < 0 or >= W)
  const int y0 = rs * RS_ROWS, y1 = min(y0 + RS_ROWS, H);               // rows this wavefront decides
  const int ey0 = y0 > 0 ? y0 - 1 : 0, ey1 = y1 < H ? y1 + 1 : H;       // response rows it computes: [ey0, ey1)
  const int xr = reflect101(x, W);
  const int xe = reflect101(lane == 0 ? x - 1 : x + 1, W);
  const bool edge = lane == 0 || lane == 63;
  const bool fix_lo = x == -1, fix_hi = x == W;
  const bool border_wave = cs == 0 || cs * RN_COLS - RN_HALO + 63 >= W;
  const double scale = 1.0 / (4.0 * 3.0 * 255.0);
  const float k1 = (float)(1.0 * scale), k0 = (float)(2.0 * scale);
  const bool test_lane = lane >= RN_HALO && lane < RN_HALO + RN_COLS && x >= 1 && x <= W - 2;
  unsigned long long* list = raw + (size_t)b * raw_stride;
  int* count = &ncand[b * NC_STRIDE + NC_RAW];
  unsigned long long* buf = sBuf[wave];
  int nbuf = 0;  // wave-uniform
  auto flush = [&]() {
    if (nbuf == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int base = 0;
    if (lane == 0) base = atomicAdd(count, nbuf);
    base = __builtin_amdgcn_readfirstlane(base);
    for (int i = lane; i < nbuf; i += 64) {
      if ((size_t)(base + i) < raw_stride) list[base + i] = buf[i];
      else atomicOr(status, 8);  // the RAW list (timing dependent: filtered against a running maximum), not the candidate list
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    nbuf = 0;
  };

  const int j0 = ey0 > 0 ? ey0 - 1 : 0, j1 = ey1 < H ? ey1 : H - 1;
  auto load_row = [&](int py, int& c, int& e) {
    const uint8_t* row = img + (size_t)reflect101(py, H) * row_stride;
    c = row[xr];
    e = edge ? (int)row[xe] : 0;
  };
  auto row_terms = [&](int c, int e, float& rdx, float& rdy) {
    int l = wave_from_lower(c), r = wave_from_upper(c);
    if (lane == 0) l = e;
    if (lane == 63) r = e;
    rdx = (float)(r - l);
    rdy = (float)c * k0 + (float)(l + r) * k1;
  };
  const int p_first = j0 - 1, p_last = j1 + 1;
  unsigned lmax = 0u;
  float thr_lb = (float)((double)key_f32(ld_l2(&maxkey[b])) * quality);  // maxkey starts at 0: key_f32(0) is a NaN pattern
  if (!(thr_lb >= 0.0f)) thr_lb = 0.0f;
  float eA = 0.f, eB = 0.f;  // responses of the two rows above the one being emitted
  auto emit = [&](int y, const double* top, const double* mid, const double* bot) {
    const float a = (float)((top[0] + mid[0]) + bot[0]) * 0.5f;
    const float bq = (float)((top[1] + mid[1]) + bot[1]);
    const float c = (float)((top[2] + mid[2]) + bot[2]) * 0.5f;
    const float d = a - c;
    const float e = (a + c) - sqrtf(d * d + bq * bq);
    if (lane >= 1 && lane <= 62 && x >= 0 && x < W && y >= y0 && y < y1) {  // every pixel belongs to exactly one wavefront's (rows, lanes 2..61) — lanes 1 / 62 only feed the tests
      if (lane >= RN_HALO && lane < RN_HALO + RN_COLS) { const unsigned k = f32_key(e); lmax = k > lmax ? k : lmax; }
    }
    // row y - 1 can be decided now: its 3x3 neighbourhood is (eA, eB, e) x lanes -1 / 0 / +1
    const int yt = y - 1;
    if (yt >= y0 && yt < y1 && yt >= 1 && yt <= H - 2) {  // wave-uniform
      const float m0 = fmaxf(fmaxf(wave_from_lower(eA), eA), wave_from_upper(eA));
      const float m2 = fmaxf(fmaxf(wave_from_lower(e), e), wave_from_upper(e));
      const float m1 = fmaxf(wave_from_lower(eB), wave_from_upper(eB));
      const bool cand = test_lane && eB > thr_lb && eB >= fmaxf(fmaxf(m0, m1), m2);
      const unsigned long long mask = __ballot(cand);
      if (mask) {
        if (cand) buf[nbuf + __popcll(mask & ((1ull << lane) - 1ull))] = ((unsigned long long)f32_key(eB) << 32) | pack_xy(x, yt);
        nbuf += __popcll(mask);
        if (nbuf > RN_BUF - 64) {
          flush();
          // a fresher lower bound of the threshold (the image's maximum so far, this wavefront's included)
          unsigned wm = lmax;
          for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(wm, off); wm = o > wm ? o : wm; }
          const unsigned gm = ld_l2(&maxkey[b]);
          const float lb = (float)((double)key_f32(gm > wm ? gm : wm) * quality);
          if (lb > thr_lb) thr_lb = lb;
        }
      }
    }
    eA = eB; eB = e;
  };
  auto step = [&](int p, int& c, int& e, float& dxA, float& dyA, float& dxB, float& dyB, float& dxC, float& dyC, double* HA,
                  double* HB, double* HC) {
    const int cc = c, ee = e;
    if (p + RS_PF <= p_last) load_row(p + RS_PF, c, e);
    row_terms(cc, ee, dxC, dyC);
    const int j = p - 1;
    if (j < j0) return;
    const float dx = (dxA + dxC) * k1 + dxB * k0;
    const float dy = dyC - dyA;
    float xx = dx * dx, xy = dx * dy, yy = dy * dy;
    if (border_wave) {
      const int src = fix_lo ? lane + 2 : lane - 2;
      const float x2 = __shfl(xx, src), y2 = __shfl(xy, src), z2 = __shfl(yy, src);
      if (fix_lo || fix_hi) { xx = x2; xy = y2; yy = z2; }
    }
    HC[0] = ((double)wave_from_lower(xx) + (double)xx) + (double)wave_from_upper(xx);
    HC[1] = ((double)wave_from_lower(xy) + (double)xy) + (double)wave_from_upper(xy);
    HC[2] = ((double)wave_from_lower(yy) + (double)yy) + (double)wave_from_upper(yy);
    const int y = j - 1;
    if (y < ey0) return;
    if (y == 0) emit(0, HC, HB, HC);
    else emit(y, HA, HB, HC);
  };
  int c0, e0, c1, e1, c2, e2;
  load_row(p_first, c0, e0); load_row(p_first + 1, c1, e1); load_row(p_first + 2, c2, e2);
  float dx0 = 0.f, dy0 = 0.f, dx1 = 0.f, dy1 = 0.f, dx2 = 0.f, dy2 = 0.f;
  double h0[3] = {0, 0, 0}, h1[3] = {0, 0, 0}, h2[3] = {0, 0, 0};
  int p = p_first;
  for (; p + 2 <= p_last; p += 3) {
    step(p, c0, e0, dx1, dy1, dx2, dy2, dx0, dy0, h1, h2, h0);
    step(p + 1, c1, e1, dx2, dy2, dx0, dy0, dx1, dy1, h2, h0, h1);
    step(p + 2, c2, e2, dx0, dy0, dx1, dy1, dx2, dy2, h0, h1, h2);
  }
  double hp[3] = {h1[0], h1[1], h1[2]}, hl[3] = {h2[0], h2[1], h2[2]};
  if (p <= p_last) {
    step(p, c0, e0, dx1, dy1, dx2, dy2, dx0, dy0, h1, h2, h0);
    for (int q = 0; q < 3; ++q) { hp[q] = h2[q]; hl[q] = h0[q]; }
    ++p;
    if (p <= p_last) {
      step(p, c1, e1, dx2, dy2, dx0, dy0, dx1, dy1, h2, h0, h1);
      for (int q = 0; q < 3; ++q) { hp[q] = h0[q]; hl[q] = h1[q]; }
    }
  }
  if (j1 == H - 1 && ey1 == H && H - 1 >= ey0) emit(H - 1, hp, hl, hp);
  flush();
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = __shfl_xor(lmax, off);
    lmax = o > lmax ? o : lmax;
  }
  if (lane == 0 && lmax) atomicMax(&maxkey[b], lmax);
}

// The exact threshold on the raw local maxima: keeps c > (float)(max * quality), as corner_nms_kernel does.
__global__ __launch_bounds__(256) void corner_threshold_kernel(const unsigned long long* __restrict__ raw, size_t raw_stride,
                                                               const unsigned* __restrict__ maxkey, double quality,
                                                               unsigned long long* __restrict__ cand, int* __restrict__ ncand, int cap,
                                                               int* __restrict__ status) {
  const int b = blockIdx.z;
  const int n = min(ncand[b * NC_STRIDE + NC_RAW], (int)raw_stride);
  const float thr = (float)((double)key_f32(maxkey[b]) * quality);
  __shared__ int sCount, sBase;
  for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {  // workgroup-uniform trip count
    if (threadIdx.x == 0) sCount = 0;
    __syncthreads();
    const int i = i0 + (int)threadIdx.x;
    unsigned long long e = 0;
    bool keep = false;
    if (i < n) {
      e = raw[(size_t)b * raw_stride + i];
      keep = key_f32((unsigned)(e >> 32)) > thr;
    }
    int local = 0;
    if (keep) local = atomicAdd(&sCount, 1);
    __syncthreads();
    if (threadIdx.x == 0 && sCount > 0) sBase = atomicAdd(&ncand[b * NC_STRIDE], sCount);
    __syncthreads();
    if (keep) {
      const int pos = sBase + local;
      if (pos < cap) cand[(size_t)b * cap + pos] = e;
      else atomicOr(status, 1);
    }
    __syncthreads();
  }
}

constexpr int NMS_ROWS = 8;   // pixel rows per thread: a workgroup covers 64 x 32 pixels (one pixel per thread left the
                              // kernel bound by workgroup turnover: 30 k workgroups, 90 us per 16 frames)
__global__ __launch_bounds__(256) void corner_nms_kernel(const float* __restrict__ eig, int W, int H,
                                                         const unsigned* __restrict__ maxkey, double quality,
                                                         unsigned long long* __restrict__ cand,
                                                         int* __restrict__ ncand, int cap, int* __restrict__ status) {
  const int b = blockIdx.z;
  const float* E = eig + (size_t)b * W * H;
  const float maxv = key_f32(maxkey[b]);
  const float thr = (float)((double)maxv * quality);
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int yb = blockIdx.y * (4 * NMS_ROWS) + (threadIdx.x >> 6);
  __shared__ int sCount, sBase;
  if (threadIdx.x == 0) sCount = 0;
  __syncthreads();
  float val[NMS_ROWS];
  unsigned found = 0u;  // bit k: row yb + 4 k holds a candidate
#pragma unroll
  for (int k = 0; k < NMS_ROWS; ++k) {
    const int y = yb + 4 * k;
    val[k] = 0.f;
    if (x < 1 || x > W - 2 || y < 1 || y > H - 2) continue;
    const float c = E[(size_t)y * W + x];
    const float v = c > thr ? c : 0.0f;
    if (v == 0.0f) continue;
    float m = v;
#pragma unroll
    for (int j = -1; j <= 1; ++j)
#pragma unroll
      for (int i = -1; i <= 1; ++i) {
        const float nv = E[(size_t)(y + j) * W + (x + i)];
        const float t = nv > thr ? nv : 0.0f;
        m = t > m ? t : m;
      }
    if (v == m) { val[k] = v; found |= 1u << k; }
  }
  // one global atomic per workgroup that has candidates (a per-thread returning atomic on one word
  // serialises at ~88/us: measured 550 us for 16 frames before this change)
  int local = 0;
  if (found) local = atomicAdd(&sCount, __popc(found));
  __syncthreads();
  if (threadIdx.x == 0 && sCount > 0) sBase = atomicAdd(&ncand[b * NC_STRIDE], sCount);
  __syncthreads();
  if (found) {
    int pos = sBase + local;
#pragma unroll
    for (int k = 0; k < NMS_ROWS; ++k) {
      if (!((found >> k) & 1u)) continue;
      if (pos < cap)
        cand[(size_t)b * cap + pos] = ((unsigned long long)f32_key(val[k]) << 32) | pack_xy(x, yb + 4 * k);
      else
        atomicOr(status, 1);
      ++pos;
    }
  }
}

// One workgroup per image.  All cross-wave global traffic uses L2-scope (sc1) loads/stores.
__global__ __launch_bounds__(SEL_THREADS) void corner_select_kernel(
    const unsigned long long* __restrict__ cand, const int* __restrict__ ncand, int cap, int W, int H,
    float min_distance, int max_corners, int* __restrict__ cell_count, int* __restrict__ cell_start, int max_cells,
    unsigned long long* __restrict__ sorted, uint8_t* __restrict__ state_g, float* __restrict__ out_xy,
    int* __restrict__ out_n, int* __restrict__ status) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  unsigned long long* sKeys = reinterpret_cast<unsigned long long*>(lds_raw);
  __shared__ int sPart[SEL_THREADS];
  __shared__ int sFlag, sCount;
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned long long* C = cand + (size_t)b * cap;
  int n = ncand[b * NC_STRIDE];
  if (n > cap) n = cap;
  float* oxy = out_xy + (size_t)b * max_corners * 2;
  if (min_distance < 1.0f) {
    // no distance constraint: plain top-K by key
    if (n > SEL_MAX_ACCEPT) { if (tid == 0) { atomicOr(status, 2); out_n[b] = 0; } return; }
    int np2 = 1; while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += SEL_THREADS) sKeys[i] = i < n ? C[i] : 0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < np2; i += SEL_THREADS) {
          const int l = i ^ j;
          if (l > i) {
            const unsigned long long a = sKeys[i], c = sKeys[l];
            const bool desc = (i & k) == 0;
            if (desc ? a < c : a > c) { sKeys[i] = c; sKeys[l] = a; }
          }
        }
        __syncthreads();
      }
    const int m = n < max_corners ? n : max_corners;
    for (int i = tid; i < m; i += SEL_THREADS) {
      const unsigned idx = (unsigned)(sKeys[i] & 0xffffffffu);
      oxy[2 * i] = (float)key_x(idx); oxy[2 * i + 1] = (float)key_y(idx);
    }
    if (tid == 0) out_n[b] = m;
    return;
  }
  const int cell = (int)rintf(min_distance);
  const int gw = (W + cell - 1) / cell, gh = (H + cell - 1) / cell;
  const int ncell = gw * gh;
  int* cc = cell_count + (size_t)b * max_cells;
  int* cs = cell_start + (size_t)b * (max_cells + 1);
  unsigned long long* S = sorted + (size_t)b * cap;
  uint8_t* st = state_g + (size_t)b * cap;
  if (ncell > max_cells) { if (tid == 0) { atomicOr(status, 4); out_n[b] = 0; } return; }
  // The rounds below re-read keys, states and the cell table many times; from L2 every read is a ~0.7 us
  // round trip (measured 950 us for this kernel), so when they fit they live in LDS.
  unsigned long long* keysL = reinterpret_cast<unsigned long long*>(lds_raw);
  unsigned short* cellL = reinterpret_cast<unsigned short*>(lds_raw + (size_t)SEL_CAP_K * 8);  // positions <= n <= 12288
  uint8_t* stateL = lds_raw + (size_t)SEL_CAP_K * 8 + (size_t)SEL_CAP_C * 2;
  unsigned short* blkL = reinterpret_cast<unsigned short*>(lds_raw + (size_t)SEL_CAP_K * 8 + (size_t)SEL_CAP_C * 2 + SEL_CAP_K);
  const bool use_lds = n <= SEL_CAP_K && ncell + 1 <= SEL_CAP_C;
  if (use_lds) {
    // Binning entirely in LDS (round 4): histogram, exclusive scan and scatter never touch HBM — the global form below wrote
    // every key and state byte through to memory as a scattered partial line (475 KB per frame for a result of <= 12 KB).
    // The per-cell counters borrow the state + blocker arrays (3 SEL_CAP_K bytes >= 4 SEL_CAP_C), which are initialised afterwards.
    static_assert(3 * SEL_CAP_K >= 4 * SEL_CAP_C, "the cell counters must fit the state + blocker arrays");
    int* cntL = reinterpret_cast<int*>(stateL);
    for (int i = tid; i < ncell; i += SEL_THREADS) cntL[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += SEL_THREADS) {
      const unsigned idx = (unsigned)(C[i] & 0xffffffffu);
      atomicAdd(&cntL[(key_y(idx) / cell) * gw + key_x(idx) / cell], 1);
    }
    __syncthreads();
    {
      const int chunk = (ncell + SEL_THREADS - 1) / SEL_THREADS;
      const int lo = tid * chunk, hi = (lo + chunk < ncell) ? lo + chunk : ncell;
      int s = 0;
      for (int i = lo; i < hi; ++i) s += cntL[i];
      sPart[tid] = s;
      __syncthreads();
      for (int off = 1; off < SEL_THREADS; off <<= 1) {
        int v = tid >= off ? sPart[tid - off] : 0;
        __syncthreads();
        sPart[tid] += v;
        __syncthreads();
      }
      int run = sPart[tid] - s;
      for (int i = lo; i < hi; ++i) { cellL[i] = (unsigned short)run; run += cntL[i]; }
      if (tid == 0) cellL[ncell] = (unsigned short)n;
    }
    __syncthreads();
    // scatter into cell order (order inside a cell is irrelevant: every decision below uses the key)
    for (int i = tid; i < n; i += SEL_THREADS) {
      const unsigned long long k = C[i];
      const unsigned idx = (unsigned)(k & 0xffffffffu);
      const int cid = (key_y(idx) / cell) * gw + key_x(idx) / cell;
      const int slot = atomicSub(&cntL[cid], 1) - 1;
      keysL[(int)cellL[cid] + slot] = k;
    }
    __syncthreads();
    for (int i = tid; i < n; i += SEL_THREADS) { stateL[i] = 0; blkL[i] = 0xFFFFu; }
    __syncthreads();
  } else {
  // (a) histogram of candidates per cell
  for (int i = tid; i < ncell; i += SEL_THREADS) st_l2(&cc[i], 0);
  __syncthreads();
  for (int i = tid; i < n; i += SEL_THREADS) {
    const unsigned idx = (unsigned)(C[i] & 0xffffffffu);
    const int cx = key_x(idx) / cell, cy = key_y(idx) / cell;
    atomicAdd(&cc[cy * gw + cx], 1);
  }
  __syncthreads();
  // (b) exclusive scan -> cell_start
  {
    const int chunk = (ncell + SEL_THREADS - 1) / SEL_THREADS;
    const int lo = tid * chunk, hi = (lo + chunk < ncell) ? lo + chunk : ncell;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += ld_l2(&cc[i]);
    sPart[tid] = s;
    __syncthreads();
    for (int off = 1; off < SEL_THREADS; off <<= 1) {
      int v = tid >= off ? sPart[tid - off] : 0;
      __syncthreads();
      sPart[tid] += v;
      __syncthreads();
    }
    int run = sPart[tid] - s;
    for (int i = lo; i < hi; ++i) { const int c = ld_l2(&cc[i]); st_l2(&cs[i], run); run += c; }
    if (tid == 0) st_l2(&cs[ncell], n);
  }
  __syncthreads();
  // (c) scatter into cell order (order inside a cell is irrelevant: every decision below uses the key)
  for (int i = tid; i < n; i += SEL_THREADS) {
    const unsigned long long k = C[i];
    const unsigned idx = (unsigned)(k & 0xffffffffu);
    const int cid = (key_y(idx) / cell) * gw + key_x(idx) / cell;
    const int slot = atomicSub(&cc[cid], 1) - 1;
    const int pos = ld_l2(&cs[cid]) + slot;
    st_l2(&S[pos], k);
    st_l2(&st[pos], (uint8_t)0);
  }
  __syncthreads();
  }
  auto KEY = [&](int i) -> unsigned long long { return use_lds ? keysL[i] : ld_l2(&S[i]); };
  auto STATE = [&](int i) -> uint8_t { return use_lds ? ((volatile uint8_t*)stateL)[i] : ld_l2(&st[i]); };
  auto SET_STATE = [&](int i, uint8_t v) { if (use_lds) ((volatile uint8_t*)stateL)[i] = v; else st_l2(&st[i], v); };
  auto CELL = [&](int i) -> int { return use_lds ? cellL[i] : ld_l2(&cs[i]); };
  // (d) monotone fixed-point rounds.  state: 0 undecided, 1 accepted, 2 rejected.
  // The dependency chains of the bench images are ~100 links deep (8,144 candidates, 1,897 accepted) and a wavefront
  // pays for its slowest lane, so what matters is the cost of re-examining a pending candidate.  Each pending candidate
  // keeps a resume point (u16 in LDS: neighbour row << 14 | index of its first still-undecided stronger neighbour, the
  // "blocker").  Neighbours scanned before the blocker are final (weaker, too far, or stronger-and-rejected), so a
  // re-examination reads the blocker's state — accepted: reject at once; undecided: nothing to do — and otherwise
  // resumes the scan behind it: every neighbour is visited once per candidate in total.  A register bit mask of the
  // thread's pending candidates skips the decided ones without touching LDS.  Same fixed point as the plain rounds.
  const float md2 = min_distance * min_distance;
  unsigned pend = 0u;   // bit k <-> candidate tid + k * SEL_THREADS (n <= 12288 when the LDS tables are in use: k < 12)
  if (use_lds)
    for (int k = 0; tid + k * SEL_THREADS < n; ++k) pend |= 1u << k;
  for (int round = 0; round < 4096; ++round) {
    if (tid == 0) sFlag = 0;
    __syncthreads();
    // Decisions are monotone and LDS is coherent inside the workgroup, so a thread may re-examine its undecided
    // candidates several times between two barriers: dependency chains resolve without paying a barrier per link.
    int pending = 1;
    if (use_lds) {
      for (int rep = 0; rep < 256 && pend; ++rep) {
        unsigned m = pend;
        while (m) {
          const int k = __builtin_ctz(m);
          m &= m - 1u;
          const int i = tid + k * SEL_THREADS;
          const unsigned enc = ((volatile unsigned short*)blkL)[i];
          int r0 = 0, pstart = -1;
          if (enc != 0xFFFFu) {
            const int pb = (int)(enc & 0x3FFFu);
            const uint8_t sb = ((volatile uint8_t*)stateL)[pb];
            if (sb == 0) continue;                                                   // blocker still undecided
            if (sb == 1) { ((volatile uint8_t*)stateL)[i] = 2; pend &= ~(1u << k); continue; }   // blocker accepted
            r0 = (int)(enc >> 14); pstart = pb + 1;                                  // blocker rejected: resume behind it
          }
          const unsigned long long kk = keysL[i];
          const unsigned idx = (unsigned)(kk & 0xffffffffu);
          const int x = key_x(idx), y = key_y(idx);
          const int cx = x / cell, cy = y / cell;
          const int x1 = cx > 0 ? cx - 1 : 0, x2 = cx < gw - 1 ? cx + 1 : gw - 1;
          const int y1 = cy > 0 ? cy - 1 : 0, y2 = cy < gh - 1 ? cy + 1 : gh - 1;
          int decided = 1;  // 1 accepted unless the scan finds otherwise
          for (int r = r0; y1 + r <= y2 && decided == 1; ++r) {
            const int yy = y1 + r;
            const int p0 = cellL[yy * gw + x1], p1 = cellL[yy * gw + x2 + 1];  // cells x1..x2 of a row are contiguous
            for (int p = (r == r0 && pstart >= 0) ? pstart : p0; p < p1; ++p) {
              const unsigned long long km = keysL[p];
              if (km <= kk) continue;  // only stronger candidates matter (keys are unique)
              const unsigned im = (unsigned)(km & 0xffffffffu);
              const float dx = (float)(x - key_x(im)), dy = (float)(y - key_y(im));
              if (!(dx * dx + dy * dy < md2)) continue;
              const uint8_t sm = ((volatile uint8_t*)stateL)[p];
              if (sm == 1) { decided = 2; break; }
              if (sm == 0) { blkL[i] = (unsigned short)((r << 14) | p); decided = 0; break; }
            }
          }
          if (decided) { ((volatile uint8_t*)stateL)[i] = (uint8_t)decided; pend &= ~(1u << k); }
        }
      }
      pending = pend != 0u;
    } else {
    for (int rep = 0; rep < 16 && pending; ++rep) {
    pending = 0;
    for (int i = tid; i < n; i += SEL_THREADS) {
      if (STATE(i) != 0) continue;
      const unsigned long long k = KEY(i);
      const unsigned idx = (unsigned)(k & 0xffffffffu);
      const int x = key_x(idx), y = key_y(idx);
      const int cx = x / cell, cy = y / cell;
      const int x1 = cx > 0 ? cx - 1 : 0, x2 = cx < gw - 1 ? cx + 1 : gw - 1;
      const int y1 = cy > 0 ? cy - 1 : 0, y2 = cy < gh - 1 ? cy + 1 : gh - 1;
      bool rejected = false, blocked = false;
      for (int yy = y1; yy <= y2 && !rejected; ++yy) {
        const int p0 = CELL(yy * gw + x1), p1 = CELL(yy * gw + x2 + 1);  // cells x1..x2 of a row are contiguous
        for (int p = p0; p < p1; ++p) {
          const unsigned long long km = KEY(p);
          if (km <= k) continue;  // only stronger candidates matter (keys are unique)
          const unsigned im = (unsigned)(km & 0xffffffffu);
          const float dx = (float)(x - key_x(im)), dy = (float)(y - key_y(im));
          if (!(dx * dx + dy * dy < md2)) continue;
          const uint8_t sm = STATE(p);
          if (sm == 1) { rejected = true; break; }
          if (sm == 0) blocked = true;
        }
      }
      if (rejected) SET_STATE(i, (uint8_t)2);
      else if (!blocked) SET_STATE(i, (uint8_t)1);
      else pending = 1;
    }
    }
    }
    if (pending) sFlag = 1;
    __syncthreads();
    const int again = sFlag;
    __syncthreads();
    if (!again) break;
  }
  // (e) gather accepted keys into LDS, sort descending, truncate
  if (tid == 0) sCount = 0;
  __syncthreads();
  // accepted keys go through the (now idle) candidate buffer: sKeys aliases the LDS key cache
  int A, np2 = 1;
  if (use_lds) {
    // in LDS: every thread takes its accepted keys into registers, then the list is rebuilt at the front of the key cache
    // (sKeys aliases it) — no trip through the candidate buffer in HBM
    constexpr int PER = SEL_CAP_K / SEL_THREADS;
    unsigned long long mine[PER];
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = tid + r * SEL_THREADS;
      const bool acc = i < n && stateL[i] == 1;
      mine[r] = acc ? keysL[i] : 0ull;
      cnt += acc;
    }
    __syncthreads();
    int pos = cnt ? atomicAdd(&sCount, cnt) : 0;
    __syncthreads();
    A = sCount;
    while (np2 < A) np2 <<= 1;
#pragma unroll
    for (int r = 0; r < PER; ++r) if (mine[r]) sKeys[pos++] = mine[r];  // a key is never 0: its response is above the threshold
    for (int i = A + tid; i < np2; i += SEL_THREADS) sKeys[i] = 0ull;
    __syncthreads();
  } else {
    unsigned long long* Cw = const_cast<unsigned long long*>(C);
    for (int i = tid; i < n; i += SEL_THREADS) {
      if (STATE(i) == 1) {
        const int p = atomicAdd(&sCount, 1);
        if (p < SEL_MAX_ACCEPT) st_l2(&Cw[p], KEY(i));
      }
    }
    __syncthreads();
    A = sCount;
    if (A > SEL_MAX_ACCEPT) { if (tid == 0) { atomicOr(status, 2); out_n[b] = 0; } return; }
    while (np2 < A) np2 <<= 1;
    for (int i = tid; i < np2; i += SEL_THREADS) sKeys[i] = i < A ? ld_l2(&Cw[i]) : 0ull;
    __syncthreads();
  }
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += SEL_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = sKeys[i], c = sKeys[l];
          const bool desc = (i & k) == 0;
          if (desc ? a < c : a > c) { sKeys[i] = c; sKeys[l] = a; }
        }
      }
      __syncthreads();
    }
  const int m = A < max_corners ? A : max_corners;
  for (int i = tid; i < m; i += SEL_THREADS) {
    const unsigned idx = (unsigned)(sKeys[i] & 0xffffffffu);
    oxy[2 * i] = (float)key_x(idx); oxy[2 * i + 1] = (float)key_y(idx);
  }
  if (tid == 0) out_n[b] = m;
}

// ----------------------------------------------------------------------------- host side
static int corner_launch(svo_ctx* ctx, const uint8_t* imgs, int batch, int W, int H, int row_stride,
                         size_t image_stride, int max_corners, double quality, double min_distance,
                         float* xy, int* n) {
  SVO_REQUIRE(ctx, W <= 65535 && H <= 65535, "corner_detect: image sides above 65535 (candidate keys pack y << 16 | x)");
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_maxkey, 0, sizeof(unsigned) * batch, st));
  SVO_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_ncand, 0, sizeof(int) * NC_STRIDE * batch, st));
  static const bool two_pass = [] { const char* e = getenv("SVO_CORNER_TWO_PASS"); return e && *e && atoi(e) != 0; }();
  if (two_pass) {
    { const int rce = svo_ensure_eig(ctx, (size_t)ctx->lim.max_batch); if (rce) return rce; }
    {
      SvoProfScope prof(ctx, SVO_PROF_CORNER_RESPONSE);
      hipLaunchKernelGGL(corner_response_kernel, dim3(svo_div_up(svo_div_up(W, RS_COLS) * svo_div_up(H, RS_ROWS), 4), 1, batch), dim3(256), 0, st,
                         imgs, W, H, row_stride, image_stride, ctx->d_eig, ctx->d_maxkey);
    }
    SvoProfScope prof(ctx, SVO_PROF_CORNER_NMS);
    hipLaunchKernelGGL(corner_nms_kernel, dim3(svo_div_up(W, 64), svo_div_up(H, 4 * NMS_ROWS), batch), dim3(256), 0, st,
                       ctx->d_eig, W, H, ctx->d_maxkey, quality, ctx->d_cand, ctx->d_ncand, ctx->lim.max_candidates,
                       ctx->d_status);
  } else {
    // the raw local maxima of an image: their own list (ctx->raw_cap entries per image; an overflow sets status bit 1)
    unsigned long long* raw = ctx->d_raw;
    const size_t raw_stride = ctx->raw_cap;
    {
      SvoProfScope prof(ctx, SVO_PROF_CORNER_RESPONSE);
      hipLaunchKernelGGL(corner_response_nms_kernel, dim3(svo_div_up(svo_div_up(W, RN_COLS) * svo_div_up(H, RS_ROWS), 4), 1, batch), dim3(256), 0, st,
                         imgs, W, H, row_stride, image_stride, ctx->d_maxkey, quality, raw, raw_stride, ctx->d_ncand, ctx->d_status);
    }
    SvoProfScope prof(ctx, SVO_PROF_CORNER_NMS);
    hipLaunchKernelGGL(corner_threshold_kernel, dim3(32, 1, batch), dim3(256), 0, st, raw, raw_stride, ctx->d_maxkey, quality, ctx->d_cand,
                       ctx->d_ncand, ctx->lim.max_candidates, ctx->d_status);
  }
  static bool attr_set = false;
  if (!attr_set) {
    SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)corner_select_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SEL_LDS_BYTES));
    attr_set = true;
  }
  SvoProfScope ps(ctx, SVO_PROF_CORNER_SELECT);
  hipLaunchKernelGGL(corner_select_kernel, dim3(batch), dim3(SEL_THREADS), SEL_LDS_BYTES, st, ctx->d_cand,
                     ctx->d_ncand, ctx->lim.max_candidates, W, H, (float)min_distance, max_corners,
                     ctx->d_cell_count, ctx->d_cell_start, ctx->max_cells, ctx->d_sorted, ctx->d_state, xy, n,
                     ctx->d_status);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

static int corner_check_args(svo_ctx* ctx, const void* img, int batch, int W, int H, int row_stride,
                             int max_corners, const void* xy, const void* n) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, img && xy && n, "corner_detect: null buffer");
  SVO_REQUIRE(ctx, W >= 3 && H >= 3 && W <= ctx->lim.max_width && H <= ctx->lim.max_height && row_stride >= W,
              "corner_detect: image size outside the limits given to svo_create");
  SVO_REQUIRE(ctx, batch >= 1 && batch <= ctx->lim.max_batch, "corner_detect: batch outside limits");
  SVO_REQUIRE(ctx, max_corners >= 1, "corner_detect: max_corners must be >= 1");
  return SVO_OK;
}

static int corner_status(svo_ctx* ctx) {
  int* h = (int*)ctx->h_pinned;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(h, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (*h) {
    const int s = *h;
    (void)hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream);
    ctx->err = (s & 1)   ? "corner_detect: more NMS candidates than svo_limits.max_candidates"
               : (s & 8) ? "corner_detect: more raw local maxima than the streaming pass's list holds (svo_ctx raw_cap: width x height / 4 per image under a 1 GiB budget)"
               : (s & 2) ? "corner_detect: more accepted corners than the select kernel holds (16384)"
                         : "corner_detect: min-distance grid larger than the workspace";
    return SVO_ERR_CAPACITY;
  }
  return SVO_OK;
}

extern "C" int svo_corner_detect_batch_dev(svo_ctx* ctx, const uint8_t* imgs, int batch, int width, int height,
                                           int row_stride, size_t image_stride, int max_corners, double quality,
                                           double min_distance, float* xy, int* n) {
  int rc = corner_check_args(ctx, imgs, batch, width, height, row_stride, max_corners, xy, n);
  if (rc) return rc;
  return corner_launch(ctx, imgs, batch, width, height, row_stride, image_stride, max_corners, quality,
                       min_distance, xy, n);
}

extern "C" int svo_corner_detect(svo_ctx* ctx, const uint8_t* img, int width, int height, int row_stride,
                                 int max_corners, double quality, double min_distance, float* xy, int* n) {
  int rc = corner_check_args(ctx, img, 1, width, height, row_stride, max_corners, xy, n);
  if (rc) return rc;
  SvoScratch s(ctx);
  uint8_t* d_img = s.take<uint8_t>((size_t)width * height);
  float* d_xy = s.take<float>(2 * (size_t)max_corners);
  int* d_n = s.take<int>(1);
  if (!d_img || !d_xy || !d_n) { ctx->err = "corner_detect: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(d_img, width, img, row_stride, width, height, hipMemcpyHostToDevice, st));
  rc = corner_launch(ctx, d_img, 1, width, height, width, (size_t)width * height, max_corners, quality,
                     min_distance, d_xy, d_n);
  if (rc) return rc;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(n, d_n, sizeof(int), hipMemcpyDeviceToHost, st));
  rc = corner_status(ctx);
  if (rc) return rc;
  if (*n > 0) SVO_HIP_CHECK(ctx, hipMemcpy(xy, d_xy, sizeof(float) * 2 * (size_t)*n, hipMemcpyDeviceToHost));
  return SVO_OK;
}

extern "C" int svo_corner_response(svo_ctx* ctx, const uint8_t* img, int width, int height, int row_stride,
                                   float* eig) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, img && eig, "corner_response: null buffer");
  SVO_REQUIRE(ctx, width >= 3 && height >= 3 && width <= ctx->lim.max_width && height <= ctx->lim.max_height,
              "corner_response: image size outside limits");
  SvoScratch s(ctx);
  uint8_t* d_img = s.take<uint8_t>((size_t)width * height);
  if (!d_img) { ctx->err = "corner_response: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(d_img, width, img, row_stride, width, height, hipMemcpyHostToDevice, st));
  { const int rce = svo_ensure_eig(ctx, 1); if (rce) return rce; }
  SVO_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_maxkey, 0, sizeof(unsigned), st));
  hipLaunchKernelGGL(corner_response_kernel, dim3(svo_div_up(svo_div_up(width, RS_COLS) * svo_div_up(height, RS_ROWS), 4), 1, 1), dim3(256), 0,
                     st, d_img, width, height, width, (size_t)width * height, ctx->d_eig, ctx->d_maxkey);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(eig, ctx->d_eig, sizeof(float) * (size_t)width * height, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return SVO_OK;
}
