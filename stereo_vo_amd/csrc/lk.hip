// a3 — pyramidal Lucas-Kanade: cv::calcOpticalFlowPyrLK(prev, next, pts, out, status, err, Size(21,21), 3,
// TermCriteria(COUNT+EPS,30,0.01), 0, 1e-2) at reference src/feature_tracker.cpp:23-26,32-35, and the
// survivor / parallax filter of FeatureTracker::track_features, src/feature_tracker.cpp:38-64.
// Semantics: SURVEY.md Appendix A.3 with the arithmetic declared in oracle/ora_lk.cpp (14-bit bilinear
// weights, int16 patches, EXACT int64 window sums => the wave reduction order cannot change a bit,
// f32 2x2 solve without FMA contraction).
//
//   pyr_copy_kernel / pyr_down_kernel : 4-level pyramid, [1 4 6 4 1]^2, REFLECT_101, (s+128)>>8.
//   lk_fb_kernel  : one 256-thread workgroup (4 wavefronts) per feature.  Per level the 24x24 source patch and a 32x32
//                   region of the target image are staged in LDS once (the region is restaged only if the
//                   window walks out of it, so iterations run barrier-free out of LDS); source patch in
//                   LDS, Scharr derivatives are formed on the fly (no derivative image ever hits HBM),
//                   the 21x21 int16 template/gradient patches live in LDS, the 2x2 normal matrix and the
//                   per-iteration mismatch vector are wave-reduced with DPP shuffles.  Forward and
//                   backward tracking and the reference's keep/drop predicate are fused in one launch.
//   track_compact_kernel : stable compaction + the sequential f32 parallax sum (order matters).
#include "kernels.h"
#include "group_kernels.h"
#include "tail_device.h"

namespace {
#include "ref_constants.h"
#ifndef SVO_LK_MAX_ITER
#define SVO_LK_MAX_ITER svo_ref::LK_MAX_ITERATIONS
#endif
constexpr int WIN = svo_ref::LK_WIN, HALF = WIN / 2, LEVELS = svo_ref::LK_MAX_LEVEL + 1, MAX_ITER = SVO_LK_MAX_ITER;
static_assert(WIN == 21 && LEVELS == 4, "the LDS layout and the exact-sum bounds below are derived for a 21x21 window, 4 levels");
constexpr int G = WIN + 1;      // 22: bilinear needs one extra row/col
constexpr int RP = WIN + 3;     // 24: Scharr needs one more on each side
constexpr float MIN_EIG = svo_ref::LK_MIN_EIG_THRESHOLD;
constexpr double LK_EPS = svo_ref::LK_EPSILON;
constexpr float FLT_SCALE = 1.0f / (float)(1 << 20);
constexpr float FLT_EPS = 1.1920928955078125e-7f;

// A pyramid is addressed arithmetically (base pointer + level-0 size); runtime-indexed arrays of level
// pointers would live in scratch memory (measured: 136 B/lane of scratch, 30 MB of spill writes per launch).
struct Pyr {
  const uint8_t* base;
  const uint8_t* l0;  // level 0: the pyramid buffer's own copy, or the caller's image read in place (tight rows)
  int w0, h0;
};

__device__ __forceinline__ Pyr make_pyr(const uint8_t* base, int w, int h, const uint8_t* l0 = nullptr) { return Pyr{base, l0 ? l0 : base, w, h}; }

__device__ __forceinline__ void pyr_level(const Pyr& P, int level, const uint8_t*& p, int& w, int& h) {
  if (level == 0) { p = P.l0; w = P.w0; h = P.h0; return; }
  size_t off = 0;
  w = P.w0; h = P.h0;
  for (int l = 0; l < level; ++l) {
    off += (size_t)w * h;
    w = (w + 1) / 2; h = (h + 1) / 2;
  }
  p = P.base + off;
}

__device__ __forceinline__ int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

// Exact wave-wide integer sum without LDS traffic.  Bounds (8-bit images): |I|,|J| <= 255*32 = 8160 after the
// 2^-9 descale, Scharr |g| <= 16*255 = 4080, so one product is < 2^25 (8160*4080 = 33,292,800; 4080^2 < 2^24),
// a thread's partial over its PPT <= 7 pixels is < 2^28 and the sum over EIGHT lanes (half a DPP row) is
// 7 * 8 * 33,292,800 = 1,864,396,800 < 2^31: three DPP steps are exact in int32.  The eight half-row totals are then
// added as two 16-bit-split halves (see below).  Integer addition is associative, so the result does not depend on the
// order (this is what lets the oracle use a plain sequential int64 sum).
__device__ __forceinline__ long long wave_sum_i64(int v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);  // row_half_mirror: every lane holds its 8-lane total (< 2^31)
  // The eight 8-lane totals need 34 bits.  v = hi * 65536 + lo with 0 <= lo < 65536, |hi| <= 2^15: both halves are summed
  // across the wave in int32 by three more DPP steps (row_mirror, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows
  // 2 and 3; lane 63 ends up with the wave total) — 12 fewer v_readlane and 24 fewer dependent scalar adds per sum than
  // reading the eight partials out one by one, on the critical path of every LK iteration.
  int lo = v & 0xFFFF, hi = v >> 16;
  lo += __builtin_amdgcn_mov_dpp(lo, 0x140, 0xf, 0xf, true);  // row_mirror: lanes of a row hold the row total
  hi += __builtin_amdgcn_mov_dpp(hi, 0x140, 0xf, 0xf, true);
  lo += __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  hi += __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false);
  lo += __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  hi += __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false);
  const long long l = __builtin_amdgcn_readlane(lo, 63), h = __builtin_amdgcn_readlane(hi, 63);
  return h * 65536 + l;
}

// The same exact sum S, delivered as (float)(double)S — the conversion the reference arithmetic applies — without 64-bit
// or f64 operations: S = h * 65536 + l with |h| < 2^19 and 0 <= l < 2^19, so (float)h * 65536 and (float)l are both exact
// and ONE IEEE f32 addition of two exact operands is the correctly rounded value of S, which is what rounding the
// (exactly representable) double gives.
__device__ __forceinline__ float wave_sum_f32(int v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);
  v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);
  int lo = v & 0xFFFF, hi = v >> 16;
  lo += __builtin_amdgcn_mov_dpp(lo, 0x140, 0xf, 0xf, true);
  hi += __builtin_amdgcn_mov_dpp(hi, 0x140, 0xf, 0xf, true);
  lo += __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false);
  hi += __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false);
  lo += __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false);
  hi += __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false);
  const int l = __builtin_amdgcn_readlane(lo, 63), h = __builtin_amdgcn_readlane(hi, 63);
  return (float)h * 65536.0f + (float)l;
}

#ifndef SVO_LK_THREADS
#define SVO_LK_THREADS 64
#endif
typedef short lk_s2 __attribute__((ext_vector_type(2)));
// (lo, hi) as two int16 in one register
__device__ __forceinline__ lk_s2 pack_s2(int lo, int hi) { return __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u)); }

// Threads per feature.  64 (default): ONE wavefront owns a feature — 7 window pixels per lane, all sums by DPP, no
// workgroup barrier anywhere (LDS traffic of a single wave is ordered; a wave-level fence replaces s_barrier), one
// feature per 64-thread workgroup.  256 / 128: the wavefronts of a workgroup share one feature and meet at a barrier
// per iteration (measured on MI355X: 247 us per 731-feature launch with 256).
constexpr int LKT = SVO_LK_THREADS;
constexpr int NW = LKT / 64;
constexpr int FPB = LKT == 64 ? 1 : 256 / LKT;   // features per workgroup.  One: the dispatcher then places single waves, which it balances
                                               // over the SIMDs better than 4-wave workgroups when other streams' kernels share the GPU
                                               // (lk_fb under 8 streams: 238 -> 217 us per launch, +2 % frames/s; alone no difference)
constexpr int PPT = (21 * 21 + LKT - 1) / LKT;  // window pixels per thread
static_assert(LKT == 64 || LKT == 128 || LKT == 256, "threads per feature");
static_assert(PPT * 8 * 33292800ll < 2147483647ll, "half-row sums must stay exact in int32");
constexpr int RM = 5;            // margin of the staged target region around the 22x22 window
constexpr int RS = G + 2 * RM;   // 32

struct LkShared {
  alignas(4) uint8_t raw[RP * RP];      // source patch of the template image (reflect-101 staged)
  alignas(4) uint8_t jreg[RS * RS + 8];  // target-image region; restaged only when the window leaves it (+8: the aligned-dword reads of the last row)
  // template / gradient patches and cross-wave partials of the variants that share a feature between wavefronts; the
  // one-wavefront form keeps its template in registers (1.6 KB instead of 6.2 KB of LDS per feature: sixteen resident
  // tracker wavefronts no longer hold 100 KB of a CU's LDS that the solve kernel's workgroups are waiting for)
  short Iw[LKT == 64 ? 2 : WIN * WIN], dIx[LKT == 64 ? 2 : WIN * WIN], dIy[LKT == 64 ? 2 : WIN * WIN];
  short gx[LKT == 64 ? 2 : G * G], gy[LKT == 64 ? 2 : G * G];
  long long red[2][3][NW];   // cross-wave partials (double-buffered: one barrier per reduction point); unused with one wave
};

// The threads of ONE feature meet here.  One wavefront per feature: LDS operations of a wave complete in order, so a
// workgroup-scope fence (the compiler's s_waitcnt) plus a wave barrier (no reordering across it) is all it takes.
__device__ __forceinline__ void lk_sync() {
  if (NW == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  } else {
    __syncthreads();
  }
}

// Exact 64-bit sums of NV per-thread int32 partials over the wavefronts of the feature.
template <int NV>
__device__ __forceinline__ void block_sum_split(const int* v, long long* out, LkShared& S, int& phase) {
  if (NW == 1) {
#pragma unroll
    for (int k = 0; k < NV; ++k) out[k] = wave_sum_i64(v[k]);
    lk_sync();  // callers rely on the barrier inside (it also publishes LDS written before the reduction)
    return;
  }
  const int wave = (threadIdx.x % LKT) >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const long long t = wave_sum_i64(v[k]);
    if (lane == 0) S.red[phase][k][wave] = t;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    long long t = S.red[phase][k][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += S.red[phase][k][w];
    out[k] = t;
  }
  phase ^= 1;
}

// One feature through all pyramid levels; every lane of the wave returns the same values.
__device__ uint8_t lk_point_wave(const Pyr& A, const Pyr& B, float px0, float py0, float* ox, float* oy, LkShared& S);

__device__ uint8_t lk_point(const Pyr& A, const Pyr& B, float px0, float py0, float* ox, float* oy, LkShared& S) {
  if (NW == 1) return lk_point_wave(A, B, px0, py0, ox, oy, S);
  const int lane = threadIdx.x % LKT;  // 0..LKT-1: index over the threads that share this feature
  int phase = 0;
  uint8_t status = 1;
  float nx = 0.f, ny = 0.f;
  for (int level = LEVELS - 1; level >= 0; --level) {
    const uint8_t *Ip, *Jp;
    int Iw_, Ih_, Jw_, Jh_;
    pyr_level(A, level, Ip, Iw_, Ih_);
    pyr_level(B, level, Jp, Jw_, Jh_);
    const float sc = (float)(1.0 / (double)(1 << level));
    float pxl = px0 * sc, pyl = py0 * sc;
    if (level == LEVELS - 1) { nx = pxl; ny = pyl; } else { nx = nx * 2.0f; ny = ny * 2.0f; }
    pxl -= (float)HALF; pyl -= (float)HALF;
    const int ipx = (int)floorf(pxl), ipy = (int)floorf(pyl);
    if (ipx < -WIN || ipx >= Iw_ || ipy < -WIN || ipy >= Ih_) {
      if (level == 0) status = 0;
      continue;
    }
    float a = pxl - (float)ipx, b = pyl - (float)ipy;
    int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << 14));
    int iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << 14));
    int iw10 = __float2int_rn((1.f - a) * b * (float)(1 << 14));
    int iw11 = (1 << 14) - iw00 - iw01 - iw10;
    lk_sync();
    // stage the (WIN+3)^2 raw patch: tile (r,c) <-> image (ipy-1+r, ipx-1+c), reflect-101
    for (int i = lane; i < RP * RP; i += LKT) {
      const int r = i / RP, c = i % RP;
      S.raw[i] = Ip[__mul24(reflect101(ipy - 1 + r, Ih_), Iw_) + reflect101(ipx - 1 + c, Iw_)];  // 24-bit factors: full-rate multiply
    }
    lk_sync();
    // Scharr at the (WIN+1)^2 grid; zero outside the image (BORDER_CONSTANT derivative padding)
    for (int i = lane; i < G * G; i += LKT) {
      const int r = i / G, c = i % G;
      const int X = ipx + c, Y = ipy + r;
      int vx = 0, vy = 0;
      if (X >= 0 && X < Iw_ && Y >= 0 && Y < Ih_) {
        const uint8_t* t = &S.raw[r * RP + c];  // top-left of the 3x3 neighbourhood (centre at r+1,c+1)
        const int t00 = t[0], t01 = t[1], t02 = t[2];
        const int t10 = t[RP], t11 = t[RP + 1], t12 = t[RP + 2];
        const int t20 = t[2 * RP], t21 = t[2 * RP + 1], t22 = t[2 * RP + 2];
        const int s0 = (t00 + t20) * 3 + t10 * 10, s2 = (t02 + t22) * 3 + t12 * 10;
        const int d0 = t20 - t00, d1 = t21 - t01, d2 = t22 - t02;
        vx = (short)(s2 - s0);
        vy = (short)((d2 + d0) * 3 + d1 * 10);
        (void)t11;
      }
      S.gx[i] = (short)vx; S.gy[i] = (short)vy;
    }
    lk_sync();
    int pA[3] = {0, 0, 0};  // PPT pixels per thread, each product < 2^24.1: fits int32
    for (int i = lane; i < WIN * WIN; i += LKT) {
      const int r = i / WIN, c = i % WIN;
      const int o = r * G + c, o1 = o + G;
      const int rr = (r + 1) * RP + (c + 1);
      // every factor fits 24 signed bits (pixels <= 255, weights <= 2^14, |gradients| <= 4080): v_mul_i32_i24 is a
      // full-rate instruction, the generic 32-bit multiply is quarter rate
      const int ival = descale(__mul24(S.raw[rr], iw00) + __mul24(S.raw[rr + 1], iw01) + __mul24(S.raw[rr + RP], iw10) + __mul24(S.raw[rr + RP + 1], iw11), 9);
      const int ixv = descale(__mul24(S.gx[o], iw00) + __mul24(S.gx[o + 1], iw01) + __mul24(S.gx[o1], iw10) + __mul24(S.gx[o1 + 1], iw11), 14);
      const int iyv = descale(__mul24(S.gy[o], iw00) + __mul24(S.gy[o + 1], iw01) + __mul24(S.gy[o1], iw10) + __mul24(S.gy[o1 + 1], iw11), 14);
      S.Iw[i] = (short)ival; S.dIx[i] = (short)ixv; S.dIy[i] = (short)iyv;
      pA[0] += __mul24(ixv, ixv);
      pA[1] += __mul24(ixv, iyv);
      pA[2] += __mul24(iyv, iyv);
    }
    long long sA[3];
    block_sum_split<3>(pA, sA, S, phase);  // the barrier inside also publishes Iw/dIx/dIy
    const long long sA11 = sA[0], sA12 = sA[1], sA22 = sA[2];
    const float A11 = (float)(double)sA11 * FLT_SCALE;
    const float A12 = (float)(double)sA12 * FLT_SCALE;
    const float A22 = (float)(double)sA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float dif = A11 - A22;
    const float minEig = (A22 + A11 - sqrtf(dif * dif + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
    if (minEig < MIN_EIG || D < FLT_EPS) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;
    float outx = nx, outy = ny;
    nx -= (float)HALF; ny -= (float)HALF;
    float pdx = 0.f, pdy = 0.f;
    // window-pixel offsets of this thread inside the staged region (PPT pixels per thread)
    int woff[PPT], wi[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int i = lane + LKT * u;
      wi[u] = i < WIN * WIN ? i : -1;
      woff[u] = i < WIN * WIN ? (i / WIN) * RS + (i % WIN) : 0;
    }
    int rx0 = 0, ry0 = 0;
    bool staged = false;
    for (int j = 0; j < MAX_ITER; ++j) {
      const int inx = (int)floorf(nx), iny = (int)floorf(ny);
      if (inx < -WIN || inx >= Jw_ || iny < -WIN || iny >= Jh_) {
        if (level == 0) status = 0;
        break;
      }
      a = nx - (float)inx; b = ny - (float)iny;
      iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << 14));
      iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << 14));
      iw10 = __float2int_rn((1.f - a) * b * (float)(1 << 14));
      iw11 = (1 << 14) - iw00 - iw01 - iw10;
      if (!staged || inx < rx0 || inx > rx0 + 2 * RM || iny < ry0 || iny > ry0 + 2 * RM) {  // wave-uniform
        rx0 = inx - RM; ry0 = iny - RM;
        lk_sync();
        if (rx0 >= 0 && ry0 >= 0 && rx0 + RS <= Jw_ && ry0 + RS <= Jh_) {
          for (int i = lane; i < RS * RS; i += LKT) S.jreg[i] = Jp[__mul24(ry0 + i / RS, Jw_) + rx0 + i % RS];
        } else {
          for (int i = lane; i < RS * RS; i += LKT)
            S.jreg[i] = Jp[__mul24(reflect101(ry0 + i / RS, Jh_), Jw_) + reflect101(rx0 + i % RS, Jw_)];
        }
        lk_sync();
        staged = true;
      }
      const int ob = (iny - ry0) * RS + (inx - rx0);
      int pb[2] = {0, 0};
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        if (wi[u] >= 0) {
          const int o = ob + woff[u];
          const int diff = descale(__mul24(S.jreg[o], iw00) + __mul24(S.jreg[o + 1], iw01) + __mul24(S.jreg[o + RS], iw10) + __mul24(S.jreg[o + RS + 1], iw11), 9) - S.Iw[wi[u]];
          pb[0] += __mul24(diff, S.dIx[wi[u]]);   // |diff| <= 8160, |gradient| <= 4080
          pb[1] += __mul24(diff, S.dIy[wi[u]]);
        }
      }
      long long sb[2];
      block_sum_split<2>(pb, sb, S, phase);
      const long long sb1 = sb[0], sb2 = sb[1];
      const float b1 = (float)(double)sb1 * FLT_SCALE;
      const float b2 = (float)(double)sb2 * FLT_SCALE;
      const float dx = (A12 * b2 - A22 * b1) * D;
      const float dy = (A12 * b1 - A11 * b2) * D;
      nx += dx; ny += dy;
      outx = nx + (float)HALF; outy = ny + (float)HALF;
      if ((double)dx * (double)dx + (double)dy * (double)dy <= LK_EPS * LK_EPS) break;
      if (j > 0 && (double)fabsf(dx + pdx) < LK_EPS && (double)fabsf(dy + pdy) < LK_EPS) {
        outx -= dx * 0.5f; outy -= dy * 0.5f;
        break;
      }
      pdx = dx; pdy = dy;
    }
    nx = outx; ny = outy;
    if (status && level == 0) {
      const float fx = nx - (float)HALF, fy = ny - (float)HALF;
      const int ix = (int)floorf(fx), iy = (int)floorf(fy);
      if (ix < -WIN || ix >= Jw_ || iy < -WIN || iy >= Jh_) status = 0;
    }
  }
  *ox = nx; *oy = ny;
  return status;
}

// ---- one wavefront per feature (LKT == 64) ------------------------------------------------------------------------
// The reference's convergence tests compare in double.  (double)|x| < 0.01: the largest float below the double 0.01 is
// (float)0.01 itself (0.00999999977...), so the test is |x| <= 0.01f exactly.  dx^2 + dy^2 <= 1e-4 in double (products of
// floats are exact there, one rounding in the sum): decided in f32 when the f32 value is clear of the threshold by more
// than its own error (3 roundings, < 4e-7 relative), and in f64 — the reference expression itself — only in between.
__device__ __forceinline__ bool below_eps(float x) {
  static_assert(LK_EPS == 0.01, "the float threshold below is derived for epsilon = 0.01");
  return fabsf(x) <= 0.01f;
}
__device__ __forceinline__ bool step_below_eps(float dx, float dy) {
  const float s = dx * dx + dy * dy;
  if (s < 0.9999e-4f) return true;
  if (s > 1.0001e-4f) return false;
  return (double)dx * (double)dx + (double)dy * (double)dy <= LK_EPS * LK_EPS;
}

// An N x N byte tile of an image level -> LDS, by one wavefront, in ONE memory round trip: every lane issues all of its
// loads before it waits for any (the previous per-byte loop waited for each load — the compiler keeps a load and the LDS
// store that depends on it together — which made a 32 x 32 restage cost 16 dependent trips to L2 / HBM; features that
// drift restage up to 38 times per launch and were the launch's tail).  Inside the image: unaligned dword loads (N / 4 per
// row); at the border: byte loads through reflect-101.  `dst` is dword-aligned, CAP its capacity in bytes.
template <int N, int CAP>
__device__ __forceinline__ void stage_tile(uint8_t* dst, const uint8_t* __restrict__ img, int w, int h, int x0, int y0, int lane) {
  static_assert(N % 4 == 0 && N * N <= CAP, "tile rows are whole dwords");
  constexpr int DW = N / 4, NDW = DW * N, PER = (NDW + 63) / 64;
  if (x0 >= 0 && y0 >= 0 && x0 + N <= w && y0 + N <= h) {  // wave-uniform
    uint32_t v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = lane + 64 * k;
      v[k] = 0;
      if (NDW % 64 == 0 || i < NDW) __builtin_memcpy(&v[k], img + __mul24(y0 + i / DW, w) + x0 + 4 * (i % DW), 4);
    }
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = lane + 64 * k;
      if (NDW % 64 == 0 || i < NDW) d32[i] = v[k];
    }
  } else {
    constexpr int NB = N * N, PERB = (NB + 63) / 64;
    uint8_t v[PERB];
#pragma unroll
    for (int k = 0; k < PERB; ++k) {
      const int i = lane + 64 * k;
      v[k] = 0;
      if (NB % 64 == 0 || i < NB) v[k] = img[__mul24(reflect101(y0 + i / N, h), w) + reflect101(x0 + i % N, w)];
    }
#pragma unroll
    for (int k = 0; k < PERB; ++k) {
      const int i = lane + 64 * k;
      if (NB % 64 == 0 || i < NB) dst[i] = v[k];
    }
  }
}

// Lane = (window row r, third of the row): 63 lanes own 7 consecutive pixels of one row each.  The template (patch value
// and both derivatives of its 7 pixels) lives in REGISTERS for all iterations of a level; the Scharr derivatives are
// formed straight from the staged source patch (no derivative grid in LDS, no extra synchronisation); an iteration reads
// its 2 x 8 target bytes as six aligned LDS dwords (v_alignbyte + bit-field extracts) instead of 28 byte reads, and every
// horizontal neighbour is reused.  All sums are exact integers (see wave_sum_i64), so the pixel-to-lane mapping and the
// order of additions cannot change a bit of the result: same numbers as the oracle's sequential int64 loops.
__device__ uint8_t lk_point_wave(const Pyr& A, const Pyr& B, float px0, float py0, float* ox, float* oy, LkShared& S) {
  const int lane = threadIdx.x & 63;
  const bool act = lane < 63;
  const int r = act ? lane / 3 : 0, c0 = act ? 7 * (lane - 3 * (lane / 3)) : 0;
  int phase = 0;
  uint8_t status = 1;
  float nx = 0.f, ny = 0.f;
  for (int level = LEVELS - 1; level >= 0; --level) {
    const uint8_t *Ip, *Jp;
    int Iw_, Ih_, Jw_, Jh_;
    pyr_level(A, level, Ip, Iw_, Ih_);
    pyr_level(B, level, Jp, Jw_, Jh_);
    const float sc = (float)(1.0 / (double)(1 << level));
    float pxl = px0 * sc, pyl = py0 * sc;
    if (level == LEVELS - 1) { nx = pxl; ny = pyl; } else { nx = nx * 2.0f; ny = ny * 2.0f; }
    pxl -= (float)HALF; pyl -= (float)HALF;
    const int ipx = (int)floorf(pxl), ipy = (int)floorf(pyl);
    if (ipx < -WIN || ipx >= Iw_ || ipy < -WIN || ipy >= Ih_) {
      if (level == 0) status = 0;
      continue;
    }
    float a = pxl - (float)ipx, b = pyl - (float)ipy;
    int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << 14));
    int iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << 14));
    int iw10 = __float2int_rn((1.f - a) * b * (float)(1 << 14));
    int iw11 = (1 << 14) - iw00 - iw01 - iw10;
    lk_sync();
    // stage the (WIN+3)^2 raw patch: tile (r,c) <-> image (ipy-1+r, ipx-1+c), reflect-101
    stage_tile<RP, sizeof(S.raw)>(S.raw, Ip, Iw_, Ih_, ipx - 1, ipy - 1, lane);
    lk_sync();
    // template of my 7 pixels: rows r..r+3, columns c0..c0+9 of the patch
    int tI[7], tX[7], tY[7];
    int pA[3] = {0, 0, 0};
    {
      int t[4][10];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 10; ++i) t[j][i] = S.raw[(r + j) * RP + c0 + i];
      // Scharr at grid rows r, r+1 and columns c0..c0+7; zero outside the image (BORDER_CONSTANT derivative padding)
      int gx[2][8], gy[2][8];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int X = ipx + c0 + i, Y = ipy + r + j;
          const int s0 = (t[j][i] + t[j + 2][i]) * 3 + t[j + 1][i] * 10, s2 = (t[j][i + 2] + t[j + 2][i + 2]) * 3 + t[j + 1][i + 2] * 10;
          const int d0 = t[j + 2][i] - t[j][i], d1 = t[j + 2][i + 1] - t[j][i + 1], d2 = t[j + 2][i + 2] - t[j][i + 2];
          const bool in = X >= 0 && X < Iw_ && Y >= 0 && Y < Ih_;
          gx[j][i] = in ? (int)(short)(s2 - s0) : 0;
          gy[j][i] = in ? (int)(short)((d2 + d0) * 3 + d1 * 10) : 0;
        }
#pragma unroll
      for (int p = 0; p < 7; ++p) {
        const int ival = descale(__mul24(t[1][p + 1], iw00) + __mul24(t[1][p + 2], iw01) + __mul24(t[2][p + 1], iw10) + __mul24(t[2][p + 2], iw11), 9);
        const int ixv = descale(__mul24(gx[0][p], iw00) + __mul24(gx[0][p + 1], iw01) + __mul24(gx[1][p], iw10) + __mul24(gx[1][p + 1], iw11), 14);
        const int iyv = descale(__mul24(gy[0][p], iw00) + __mul24(gy[0][p + 1], iw01) + __mul24(gy[1][p], iw10) + __mul24(gy[1][p + 1], iw11), 14);
        // the patches are int16 in the reference arithmetic (oracle: (short) stores)
        // (what an iteration subtracts from the target's bilinear value, folded into its rounding: ((v + 256) >> 9) - t == (v + 256 - 512 t) >> 9)
        tI[p] = (1 << 8) - ((int)(short)ival << 9); tX[p] = (int)(short)ixv; tY[p] = (int)(short)iyv;
        if (act) {
          pA[0] += __mul24(tX[p], tX[p]);
          pA[1] += __mul24(tX[p], tY[p]);
          pA[2] += __mul24(tY[p], tY[p]);
        }
      }
    }
    // the template gradients as 16-bit pairs for the iteration's packed dot products (zero in the idle lane)
    lk_s2 tXp[4], tYp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tXp[q] = pack_s2(act ? tX[2 * q] : 0, act && 2 * q + 1 < 7 ? tX[2 * q + 1] : 0);
      tYp[q] = pack_s2(act ? tY[2 * q] : 0, act && 2 * q + 1 < 7 ? tY[2 * q + 1] : 0);
    }
    long long sA[3];
    block_sum_split<3>(pA, sA, S, phase);
    const float A11 = (float)(double)sA[0] * FLT_SCALE;
    const float A12 = (float)(double)sA[1] * FLT_SCALE;
    const float A22 = (float)(double)sA[2] * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float dif = A11 - A22;
    const float minEig = (A22 + A11 - sqrtf(dif * dif + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
    if (minEig < MIN_EIG || D < FLT_EPS) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;
    float outx = nx, outy = ny;
    nx -= (float)HALF; ny -= (float)HALF;
    float pdx = 0.f, pdy = 0.f;
    int rx0 = 0, ry0 = 0;
    bool staged = false;
    const uint32_t* jw = reinterpret_cast<const uint32_t*>(S.jreg);
    for (int j = 0; j < MAX_ITER; ++j) {
      // wave-uniform values: on the scalar unit, so that the window tests below are scalar compares and plain branches
      const int inx = __builtin_amdgcn_readfirstlane((int)floorf(nx)), iny = __builtin_amdgcn_readfirstlane((int)floorf(ny));
      if (inx < -WIN || inx >= Jw_ || iny < -WIN || iny >= Jh_) {
        if (level == 0) status = 0;
        break;
      }
      a = nx - (float)inx; b = ny - (float)iny;
      iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << 14));
      iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << 14));
      iw10 = __float2int_rn((1.f - a) * b * (float)(1 << 14));
      iw11 = (1 << 14) - iw00 - iw01 - iw10;
      if (!staged || inx < rx0 || inx > rx0 + 2 * RM || iny < ry0 || iny > ry0 + 2 * RM) {  // wave-uniform
        rx0 = inx - RM; ry0 = iny - RM;
        lk_sync();
        stage_tile<RS, RS * RS>(S.jreg, Jp, Jw_, Jh_, rx0, ry0, lane);
        lk_sync();
        staged = true;
      }
      // my two target rows, 8 bytes each, from three aligned dwords per row
      const int base = (iny - ry0 + r) * RS + (inx - rx0) + c0;
      const int wi = base >> 2, sh = base & 3;
      const uint32_t u0 = jw[wi], u1 = jw[wi + 1], u2 = jw[wi + 2];
      const uint32_t v0 = jw[wi + RS / 4], v1 = jw[wi + RS / 4 + 1], v2 = jw[wi + RS / 4 + 2];
      const uint32_t tl = __builtin_amdgcn_alignbyte(u1, u0, sh), th = __builtin_amdgcn_alignbyte(u2, u1, sh);
      const uint32_t bl = __builtin_amdgcn_alignbyte(v1, v0, sh), bh = __builtin_amdgcn_alignbyte(v2, v1, sh);
      // Packed 16-bit dot products (v_dot2c_i32_i16: a.lo b.lo + a.hi b.hi + c, exact in int32): the bilinear value of a pixel
      // is two of them over (pixel, right neighbour) x (weight pair), the two mismatch sums four each over pixel pairs —
      // 14 + 8 + 4 packs instead of 7 x 14 multiplies and adds, and the 16 byte extracts become 14 byte permutes.  Integer
      // arithmetic: the same numbers in any grouping (weights <= 2^14, |diff| <= 8160, |gradient| <= 4080 all fit int16).
      const lk_s2 wtop = pack_s2(iw00, iw01), wbot = pack_s2(iw10, iw11);
      int diff[8];
#pragma unroll
      for (int p = 0; p < 7; ++p) {
        const uint32_t sel = 0x0c000c00u | ((uint32_t)(p + 1) << 16) | (uint32_t)p;  // (byte p, 0, byte p + 1, 0) of the 8-byte row
        const lk_s2 tp = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(th, tl, sel)), bp = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(bh, bl, sel));
        diff[p] = __builtin_amdgcn_sdot2(tp, wtop, __builtin_amdgcn_sdot2(bp, wbot, tI[p], false), false) >> 9;  // descale(., 9) - template, see tI
      }
      diff[7] = 0;
      int pb[2] = {0, 0};
#pragma unroll
      for (int q = 3; q >= 0; --q) {
        const lk_s2 d = pack_s2(diff[2 * q], diff[2 * q + 1]);
        pb[0] = __builtin_amdgcn_sdot2(d, tXp[q], pb[0], false);   // the template pairs of the idle 64th lane are zero
        pb[1] = __builtin_amdgcn_sdot2(d, tYp[q], pb[1], false);
      }
      // (round 5 also measured, both bit-identical and both WITHOUT any change in launch duration or frames/s, hence not kept
      // (profiles/r05_exp_lanes_groups_honest.txt, sweeps v and ab): the two sums finished in LDS — three DPP steps, then two ds_add_u64
      // of eight lanes, 119 instead of 132 VALU instructions per iteration; and the bilinear values above by full-rate f32 FMAs on an
      // f32 copy of the staged region (every value a multiple of 2^-9 below 2^14: exact) — 7 v_pk_fma_f32 + 14 v_fmac_f32 +
      // 7 v_cvt_flr_i32_f32 for the 14 permutes, 14 dot products, 4 byte alignments and 7 shifts: 464 instead of 549 SIMD cycles per
      // iteration at the measured issue rates (tools/exp/issue_rate.hip) — once the copy's row pitch was odd; at pitch 32 the 21 rows
      // of a column met in two LDS banks and the form ran 20 % slower.  A tracking launch lasts as long as its slowest feature
      // (~650 us against ~80 us for the average wavefront): cheaper instructions shorten the average, not the launch.)
      const float b1 = wave_sum_f32(pb[0]) * FLT_SCALE;
      const float b2 = wave_sum_f32(pb[1]) * FLT_SCALE;
      const float dx = (A12 * b2 - A22 * b1) * D;
      const float dy = (A12 * b1 - A11 * b2) * D;
      nx += dx; ny += dy;
      outx = nx + (float)HALF; outy = ny + (float)HALF;
      if (step_below_eps(dx, dy)) break;
      if (j > 0 && ((int)below_eps(dx + pdx) & (int)below_eps(dy + pdy))) {  // one branch, not two
        outx -= dx * 0.5f; outy -= dy * 0.5f;
        break;
      }
      pdx = dx; pdy = dy;
    }
    nx = outx; ny = outy;
    if (status && level == 0) {
      const float fx = nx - (float)HALF, fy = ny - (float)HALF;
      const int ix = (int)floorf(fx), iy = (int)floorf(fy);
      if (ix < -WIN || ix >= Jw_ || iy < -WIN || iy >= Jh_) status = 0;
    }
  }
  *ox = nx; *oy = ny;
  return status;
}
}  // namespace

__global__ __launch_bounds__(256) void pyr_copy_kernel(const uint8_t* __restrict__ imgs, int w, int h, int row_stride,
                                                       size_t image_stride, uint8_t* __restrict__ pyr, size_t pyr_stride) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  pyr[(size_t)blockIdx.z * pyr_stride + (size_t)y * w + x] = imgs[(size_t)blockIdx.z * image_stride + (size_t)y * row_stride + x];
}

// src_imgs != null (level 1 of a pyramid whose level 0 stays in the caller's images): the source is image z of that array
__global__ __launch_bounds__(256) void pyr_down_kernel(uint8_t* __restrict__ pyr, size_t pyr_stride, size_t src_off, int sw,
                                                       int sh, size_t dst_off, int dw, int dh, const uint8_t* __restrict__ src_imgs,
                                                       size_t src_image_stride, int src_row_stride) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const uint8_t* src = src_imgs ? src_imgs + (size_t)blockIdx.z * src_image_stride : pyr + (size_t)blockIdx.z * pyr_stride + src_off;
  const int k[5] = {1, 4, 6, 4, 1};
  int s = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const uint8_t* row = src + (size_t)reflect101(2 * y + j - 2, sh) * (src_imgs ? src_row_stride : sw);
    int r = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) r += k[i] * row[reflect101(2 * x + i - 2, sw)];
    s += k[j] * r;
  }
  pyr[(size_t)blockIdx.z * pyr_stride + dst_off + (size_t)y * dw + x] = (uint8_t)((s + 128) >> 8);
}

// ---- levels 1..3 from ONE read of level 0 (round 5) ---------------------------------------------------------------------------
// pyr_down_kernel runs once per level and reads its source with 25 byte loads per output pixel from workgroups that land on all
// eight XCDs: 1.8 A of level-0 fetches per pair plus the re-read of levels 1 and 2 (profiles/r04_traffic.json: 1.05 MB per pair
// raw for 0.62 MB = 1.33 A of compulsory traffic).  Here a workgroup owns a 128 x 64 tile of level 1 (64 x 32 of level 2,
// 32 x 16 of level 3): it stages the 277 x 149 region of level 0 that its level-3 pixels depend on in LDS (dword loads), forms
// the 137 x 73 region of level 1 and the 67 x 35 region of level 2 there — the halo of the next level is recomputed, never read
// back — and writes its own tiles of the three levels.  Workgroups are dealt XCD-aware: an image's 15 tiles are consecutive
// workgroups of ONE XCD, so the halo rows a neighbour re-reads come from that XCD's L2.  Integer arithmetic ((s + 128) >> 8 of
// the [1 4 6 4 1]^2 sums, REFLECT_101 at every level): the same bytes as pyr_down_kernel whatever the tiling.
constexpr int PB_T1W = 128, PB_T1H = 64;                  // a workgroup's own tile of level 1
constexpr int PB_R0W = 2 * PB_T1W + 21, PB_R0H = 2 * PB_T1H + 21;   // level-0 region: 277 x 149
constexpr int PB_R1W = PB_T1W + 9, PB_R1H = PB_T1H + 9;             // level-1 region: 137 x 73
constexpr int PB_R2W = PB_T1W / 2 + 3, PB_R2H = PB_T1H / 2 + 3;     // level-2 region: 67 x 35
constexpr int PB_P0 = 280, PB_P1 = 140, PB_P2 = 68;                 // LDS pitches (bytes)
__global__ __launch_bounds__(256) void pyr_build_kernel(const uint8_t* __restrict__ imgs, size_t image_stride, int row_stride, int w0, int h0,
                                                        uint8_t* __restrict__ pyr, size_t pyr_stride, int tiles_x, int tiles_y, int n_images) {
  __shared__ __align__(16) uint8_t s0[PB_R0H * PB_P0];
  __shared__ __align__(16) uint8_t s1[PB_R1H * PB_P1];
  __shared__ __align__(16) uint8_t s2[PB_R2H * PB_P2];
  const int per_img = tiles_x * tiles_y, total = per_img * n_images;
  // the hardware deals workgroup L to XCD L % 8: workgroups 8 k + x, k = 0, 1, ... (one XCD's share) take consecutive tiles
  const int per_xcd = (total + 7) >> 3, v = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (v >= total) return;
  const int img = v / per_img, t = v - img * per_img, ty = t / tiles_x, tx = t - ty * tiles_x;
  const int w1 = (w0 + 1) >> 1, h1 = (h0 + 1) >> 1, w2 = (w1 + 1) >> 1, h2 = (h1 + 1) >> 1, w3 = (w2 + 1) >> 1, h3 = (h2 + 1) >> 1;
  const int x0lo = max(0, 2 * PB_T1W * tx - 14), x0hi = min(w0 - 1, 2 * PB_T1W * tx + 2 * PB_T1W + 6);
  const int y0lo = max(0, 2 * PB_T1H * ty - 14), y0hi = min(h0 - 1, 2 * PB_T1H * ty + 2 * PB_T1H + 6);
  const int x1lo = max(0, PB_T1W * tx - 6), x1hi = min(w1 - 1, PB_T1W * tx + PB_T1W + 2);
  const int y1lo = max(0, PB_T1H * ty - 6), y1hi = min(h1 - 1, PB_T1H * ty + PB_T1H + 2);
  const int x2lo = max(0, PB_T1W / 2 * tx - 2), x2hi = min(w2 - 1, PB_T1W / 2 * tx + PB_T1W / 2);
  const int y2lo = max(0, PB_T1H / 2 * ty - 2), y2hi = min(h2 - 1, PB_T1H / 2 * ty + PB_T1H / 2);
  const uint8_t* src = imgs + (size_t)img * image_stride;
  uint8_t* dst = pyr + (size_t)img * pyr_stride;
  const size_t off1 = (size_t)w0 * h0, off2 = off1 + (size_t)w1 * h1, off3 = off2 + (size_t)w2 * h2;
  // ---- level 0 -> LDS: dwords (unaligned: rows are w0 bytes apart), the last few bytes of a row one by one
  {
    const int rw = x0hi - x0lo + 1, rh = y0hi - y0lo + 1, dw = rw >> 2;
    for (int i = threadIdx.x; i < rh * dw; i += 256) {
      const int r = i / dw, c = i - r * dw;
      uint32_t val;
      __builtin_memcpy(&val, src + (size_t)(y0lo + r) * row_stride + x0lo + 4 * c, 4);
      *reinterpret_cast<uint32_t*>(&s0[r * PB_P0 + 4 * c]) = val;
    }
    const int tail = rw - 4 * dw;
    for (int i = threadIdx.x; i < rh * tail; i += 256) {
      const int r = i / tail, c = 4 * dw + (i - r * tail);
      s0[r * PB_P0 + c] = src[(size_t)(y0lo + r) * row_stride + x0lo + c];
    }
  }
  __syncthreads();
  const int k5[5] = {1, 4, 6, 4, 1};
  // ---- level 1 region (own tile to HBM as it is formed)
  {
    const int rw = x1hi - x1lo + 1, rh = y1hi - y1lo + 1;
    for (int i = threadIdx.x; i < rw * rh; i += 256) {
      const int ry = i / rw, rx = i - ry * rw, x = x1lo + rx, y = y1lo + ry;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint8_t* row = &s0[(reflect101(2 * y + j - 2, h0) - y0lo) * PB_P0];
        int r = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) r += k5[q] * row[reflect101(2 * x + q - 2, w0) - x0lo];
        sum += k5[j] * r;
      }
      const uint8_t o = (uint8_t)((sum + 128) >> 8);
      s1[ry * PB_P1 + rx] = o;
      if (x >= PB_T1W * tx && x < PB_T1W * tx + PB_T1W && y >= PB_T1H * ty && y < PB_T1H * ty + PB_T1H) dst[off1 + (size_t)y * w1 + x] = o;
    }
  }
  __syncthreads();
  // ---- level 2 region
  {
    const int rw = x2hi - x2lo + 1, rh = y2hi - y2lo + 1;
    for (int i = threadIdx.x; i < rw * rh; i += 256) {
      const int ry = i / rw, rx = i - ry * rw, x = x2lo + rx, y = y2lo + ry;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint8_t* row = &s1[(reflect101(2 * y + j - 2, h1) - y1lo) * PB_P1];
        int r = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) r += k5[q] * row[reflect101(2 * x + q - 2, w1) - x1lo];
        sum += k5[j] * r;
      }
      const uint8_t o = (uint8_t)((sum + 128) >> 8);
      s2[ry * PB_P2 + rx] = o;
      if (x >= PB_T1W / 2 * tx && x < PB_T1W / 2 * tx + PB_T1W / 2 && y >= PB_T1H / 2 * ty && y < PB_T1H / 2 * ty + PB_T1H / 2) dst[off2 + (size_t)y * w2 + x] = o;
    }
  }
  __syncthreads();
  // ---- level 3: own tile only
  {
    const int x3lo = PB_T1W / 4 * tx, y3lo = PB_T1H / 4 * ty, rw = min(w3, x3lo + PB_T1W / 4) - x3lo, rh = min(h3, y3lo + PB_T1H / 4) - y3lo;
    for (int i = threadIdx.x; i < rw * rh; i += 256) {
      const int ry = i / rw, rx = i - ry * rw, x = x3lo + rx, y = y3lo + ry;
      int sum = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint8_t* row = &s2[(reflect101(2 * y + j - 2, h2) - y2lo) * PB_P2];
        int r = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) r += k5[q] * row[reflect101(2 * x + q - 2, w2) - x2lo];
        sum += k5[j] * r;
      }
      dst[off3 + (size_t)y * w3 + x] = (uint8_t)((sum + 128) >> 8);
    }
  }
}

__global__ __launch_bounds__(LKT * FPB) void lk_kernel(const uint8_t* __restrict__ pyrA, const uint8_t* __restrict__ pyrB, int w,
                                                      int h, const float* __restrict__ xy, int n, float* __restrict__ out,
                                                      uint8_t* __restrict__ status) {
  __shared__ LkShared SS[FPB];
  LkShared& S = SS[threadIdx.x / LKT];
  const int f = blockIdx.x * FPB + threadIdx.x / LKT;
  if (f >= n) return;  // whole wavefronts leave (FPB > 1 only with one wavefront per feature: no workgroup barrier follows)
  const Pyr A = make_pyr(pyrA, w, h), B = make_pyr(pyrB, w, h);
  float ox, oy;
  const uint8_t s = lk_point(A, B, xy[2 * f], xy[2 * f + 1], &ox, &oy, S);
  if (threadIdx.x % LKT == 0) { out[2 * f] = ox; out[2 * f + 1] = oy; status[f] = s; }
}

// forward + backward + keep predicate (src/feature_tracker.cpp:44-55)
__global__ __launch_bounds__(LKT * FPB) void lk_fb_kernel(const uint8_t* __restrict__ pyrA, const uint8_t* __restrict__ pyrB, int w,
                                                         int h, const float* __restrict__ xy, const float* __restrict__ init_xy,
                                                         const int* __restrict__ n_dev, int n_host, float* __restrict__ fwd,
                                                         uint8_t* __restrict__ keep, float* __restrict__ parallax) {
  __shared__ LkShared SS[FPB];
  LkShared& S = SS[threadIdx.x / LKT];
  const int n = n_dev ? *n_dev : n_host;
  const int f = blockIdx.x * FPB + threadIdx.x / LKT;
  if (f >= n) return;
  const Pyr A = make_pyr(pyrA, w, h), B = make_pyr(pyrB, w, h);
  const float x0 = xy[2 * f], y0 = xy[2 * f + 1];
  float fx, fy, bx = 0.f, by = 0.f;
  const uint8_t s1 = lk_point(A, B, x0, y0, &fx, &fy, S);
  uint8_t s2 = 0;
  if (s1) s2 = lk_point(B, A, fx, fy, &bx, &by, S);
  if (threadIdx.x % LKT == 0) {
    uint8_t k = 0;
    float par = 0.f;
    if (s1 && s2) {
      const float ex = x0 - bx, ey = y0 - by;
      if ((double)ex * (double)ex + (double)ey * (double)ey < svo_ref::FB_MAX_DISTANCE * svo_ref::FB_MAX_DISTANCE) {  // norm(old - back) < 2
        const float dx = fx - init_xy[2 * f], dy = fy - init_xy[2 * f + 1];
        par = sqrtf(dx * dx + dy * dy);
        k = !(par > svo_ref::MAX_PARALLAX);
      }
    }
    fwd[2 * f] = fx; fwd[2 * f + 1] = fy; keep[f] = k; parallax[f] = par;
  }
}

// ---- stream-batched form (group_kernels.h): blockIdx.y = lane, blockIdx.x = feature; the same lk_point calls and the
// same keep predicate as lk_fb_kernel, then the lane's LAST wavefront to arrive compacts the lane — the stable
// compaction and the sequential f32 parallax sum of track_compact_kernel (src/feature_tracker.cpp:44-64, SURVEY C-2) by
// one wavefront: ballot / popcount per 64 features; the sum adds every feature in order (+0.0f for a dropped one leaves
// a running sum that is >= +0 bit-identical), 64 scalar lane reads per round.  Per-feature results cross workgroups
// inside the launch: written through, read at the coherence point (tail_device.h); the compacted set goes out with plain
// stores behind ONE system-scope release fence of the compacting wavefront, then the completion word.
// Register budget: the natural 123 VGPRs.  Measured with 80 (amdgpu_waves_per_eu(6, 6): three tracker wavefronts instead of two
// fit next to a 256-VGPR wavefront of a resident solve): +1.5 % frames/s, but the 22 spilled values of the per-level set-up
// cost 9 MB of scratch traffic per stereo pair (3.7 -> 12.8 MB for this kernel, profiles/r03_traffic.json of that build).
// (four wavefronts per SIMD are pinned: the kernel sits at 126 VGPRs, and a build that needed 176 cost 13 % of the frame rate)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void lk_fb_group_kernel(SvoLkLanes g) {
  static_assert(LKT == 64 && FPB == 1, "the stream-batched tracker is written for one wavefront per feature");
  __shared__ LkShared S;
  __shared__ int sLast;
  int li, f;
  if (!svo_xcd_map_item(g.map, li, f)) return;
  const SvoLkLane& a = g.lane[li];
  const int n = a.n, lane = threadIdx.x;
  if (f < n) {
    const Pyr A = make_pyr(a.pyr_prev, g.w, g.h, a.l0_prev), B = make_pyr(a.pyr_next, g.w, g.h, a.l0_next);
    const float x0 = a.xy[2 * f], y0 = a.xy[2 * f + 1];
    // forward, then backward from the forward result: ONE instance of the tracker's code serves both directions (the two inlined
    // copies were 31 KB of code next to the 71 KB of the solve kernel on the same instruction cache)
    float fx = 0.f, fy = 0.f, bx = 0.f, by = 0.f, px = x0, py = y0;
    uint8_t s1 = 0, s2 = 0;
#pragma nounroll
    for (int dir = 0; dir < 2; ++dir) {
      const Pyr P0 = dir ? B : A, P1 = dir ? A : B;
      float ox, oy;
      const uint8_t st = lk_point(P0, P1, px, py, &ox, &oy, S);
      if (dir == 0) { s1 = st; fx = ox; fy = oy; px = ox; py = oy; if (!st) break; }
      else { s2 = st; bx = ox; by = oy; }
    }
    if (lane == 0) {
      uint8_t k = 0;
      float par = 0.f;
      if (s1 && s2) {
        const float ex = x0 - bx, ey = y0 - by;
        if ((double)ex * (double)ex + (double)ey * (double)ey < svo_ref::FB_MAX_DISTANCE * svo_ref::FB_MAX_DISTANCE) {  // norm(old - back) < 2
          const float dx = fx - a.init_xy[2 * f], dy = fy - a.init_xy[2 * f + 1];
          par = sqrtf(dx * dx + dy * dy);
          k = !(par > svo_ref::MAX_PARALLAX);
        }
      }
      svo_wt_store(&a.fwd[2 * f], fx); svo_wt_store(&a.fwd[2 * f + 1], fy);
      svo_wt_store(&a.keep[f], k); svo_wt_store(&a.parallax[f], par);
    }
  }
  if (!svo_last_arrival(a.arrive, a.arrive_target, &sLast)) return;
  svo_latency_critical();
  int base = 0;
  float sum = 0.f;
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int i = c0 + lane;
    const bool k = i < n && svo_coherent_load(&a.keep[i]) != 0;
    const float par = k ? svo_coherent_load(&a.parallax[i]) : 0.0f;
    const unsigned long long mask = __ballot(k);
    if (k) {
      const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
      const float fx = svo_coherent_load(&a.fwd[2 * i]), fy = svo_coherent_load(&a.fwd[2 * i + 1]);
      // plain stores, one release fence below: scalar write-through stores are one fabric write each (thousands per lane
      // and frame: measured 200 us per track), the fence is one L2 write-back per lane and frame
      a.kept_xy[2 * slot] = fx; a.kept_xy[2 * slot + 1] = fy;
      a.init_dst[2 * slot] = a.init_xy[2 * i]; a.init_dst[2 * slot + 1] = a.init_xy[2 * i + 1];  // the tracker's per-feature state follows the feature (C-1: old ids)
      const long long id = a.ids[i];
      a.ids_dst[slot] = id;
      a.host_xy[2 * slot] = fx; a.host_xy[2 * slot + 1] = fy;
      a.host_ids[slot] = id;
    }
    base += __popcll(mask);
#pragma unroll
    for (int t = 0; t < 64; ++t) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(par), t));
  }
  if (lane == 0) {
    *a.host_n = base;
    *a.host_av = n > 0 ? sum / (float)n : 0.f;
  }
  // release at system scope (write-back, no invalidate), then the word: the pinned mirrors are for the host, the device set
  // for the lane's next stage, which the host may launch on ANOTHER stream as soon as it sees the word — i.e. before this
  // launch (other lanes still tracking) has ended
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  if (lane == 0) __hip_atomic_store(a.word, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Stable compaction of the kept features; av_parallax = (sequential float sum over kept)/n (SURVEY C-2).
__global__ __launch_bounds__(1024) void track_compact_kernel(const float* __restrict__ fwd, const uint8_t* __restrict__ keep,
                                                             const float* __restrict__ parallax, const int* __restrict__ n_dev,
                                                             int n_host, float* __restrict__ kept_xy, int* __restrict__ kept_index,
                                                             int* __restrict__ n_kept, float* __restrict__ av_parallax,
                                                             SvoTrackCarry carry) {
  svo_latency_critical();
  __shared__ int sWave[16];
  // parallax of the kept features, +0.0f for the dropped ones, in feature order: the summing lane adds every entry — adding
  // +0.0f leaves a running sum that is >= +0 bit-identical — so its loop is one dependent add per feature and nothing else
  __shared__ __align__(16) float sPar[4096];
  const int n = n_dev ? *n_dev : n_host;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const bool k = i < n && keep[i];
    if (i < 4096) sPar[i] = k ? parallax[i] : 0.0f;  // staged in the same memory round trip as the flags
    const unsigned long long mask = __ballot(k);
    if (lane == 0) sWave[wave] = __popcll(mask);
    __syncthreads();
    int off = 0, total = 0;
    for (int w = 0; w < 16; ++w) { const int c = sWave[w]; if (w < wave) off += c; total += c; }
    if (k) {
      const int slot = base + off + __popcll(mask & ((1ull << lane) - 1ull));
      kept_xy[2 * slot] = fwd[2 * i]; kept_xy[2 * slot + 1] = fwd[2 * i + 1];
      kept_index[slot] = i;
      if (carry.init_src) {  // the tracker's per-feature state follows the feature (C-1: old ids)
        carry.init_dst[2 * slot] = carry.init_src[2 * i]; carry.init_dst[2 * slot + 1] = carry.init_src[2 * i + 1];
        const long long id = carry.ids_src[i];
        carry.ids_dst[slot] = id;
        if (carry.host_xy) { carry.host_xy[2 * slot] = fwd[2 * i]; carry.host_xy[2 * slot + 1] = fwd[2 * i + 1]; carry.host_ids[slot] = id; }
      }
    }
    base += total;
    __syncthreads();
  }
  // sequential f32 sum in feature order (the reference's loop order decides the rounding, SURVEY C-2)
  float sum = 0.f;
  for (int c0 = 0; c0 < n; c0 += 4096) {
    if (c0 > 0) {  // beyond the first 4096 features: restage
      __syncthreads();
      for (int i = threadIdx.x; i < 4096 && c0 + i < n; i += 1024) sPar[i] = keep[c0 + i] ? parallax[c0 + i] : 0.0f;
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const int m = n - c0 < 4096 ? n - c0 : 4096;
      int i = 0;
      for (; i + 32 <= m; i += 32) {  // 8 LDS reads in flight, then 32 dependent adds
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const float4*>(&sPar[i + 4 * u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) { sum += q[u].x; sum += q[u].y; sum += q[u].z; sum += q[u].w; }
      }
      for (; i < m; ++i) sum += sPar[i];
    }
  }
  if (threadIdx.x == 0) {
    const float av = n > 0 ? sum / (float)n : 0.f;
    *n_kept = base;
    *av_parallax = av;
    if (carry.host_n) {  // pinned host words: the keyframe gate reads them, no D2H blit
      *carry.host_n = base;
      *carry.host_av = av;
    }
  }
  svo_publish_block(carry.pub);
}

// ----------------------------------------------------------------------------- host side
size_t svo_k_pyramid_bytes(int w, int h) {
  size_t s = 0;
  for (int l = 0; l < LEVELS; ++l) { s += (size_t)w * h; w = (w + 1) / 2; h = (h + 1) / 2; }
  return s;
}

// level0_in_place: the pyramid's level-0 slot is left unwritten (the tracker reads the caller's image instead, see Pyr::l0);
// levels 1.. are the same bytes either way.
int svo_k_build_pyramid(svo_ctx* ctx, const uint8_t* imgs, int batch, int w, int h, int row_stride, size_t image_stride,
                        uint8_t* pyr, size_t pyr_stride, bool level0_in_place) {
  hipStream_t st = ctx->stream;
  if (!level0_in_place)
    hipLaunchKernelGGL(pyr_copy_kernel, dim3(svo_div_up(w, 64), svo_div_up(h, 4), batch), dim3(256), 0, st, imgs, w, h,
                       row_stride, image_stride, pyr, pyr_stride);
  // levels 1..3 from one read of level 0 (pyr_build_kernel); the per-level launches stay for images too small for its tiling to make
  // sense, for other level counts, and behind SVO_PYR_PER_LEVEL=1 (A/B, tests)
  static const bool per_level = [] { const char* e = getenv("SVO_PYR_PER_LEVEL"); return e && *e && atoi(e) != 0; }();
  if (LEVELS == 4 && !per_level && w >= 64 && h >= 64) {
    const int w1 = (w + 1) / 2, h1 = (h + 1) / 2, tiles_x = svo_div_up(w1, PB_T1W), tiles_y = svo_div_up(h1, PB_T1H);
    const long long total = (long long)tiles_x * tiles_y * batch;
    if (total > 0 && total < (1ll << 30)) {
      SvoProfScope prof(ctx, SVO_PROF_PYR_DOWN);
      hipLaunchKernelGGL(pyr_build_kernel, dim3((unsigned)(8 * ((total + 7) / 8))), dim3(256), 0, st, imgs, image_stride, row_stride, w, h, pyr, pyr_stride,
                         tiles_x, tiles_y, batch);
      SVO_HIP_CHECK(ctx, hipGetLastError());
      return SVO_OK;
    }
  }
  size_t src_off = 0;
  int sw = w, sh = h;
  for (int l = 1; l < LEVELS; ++l) {
    const size_t dst_off = src_off + (size_t)sw * sh;
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    SvoProfScope prof(ctx, SVO_PROF_PYR_DOWN);
    const bool from_images = level0_in_place && l == 1;
    hipLaunchKernelGGL(pyr_down_kernel, dim3(svo_div_up(dw, 64), svo_div_up(dh, 4), batch), dim3(256), 0, st, pyr, pyr_stride,
                       src_off, sw, sh, dst_off, dw, dh, from_images ? imgs : (const uint8_t*)nullptr, image_stride, row_stride);
    src_off = dst_off; sw = dw; sh = dh;
  }
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

// level 0 of `batch` pyramids from their images (what svo_k_build_pyramid skips with level0_in_place)
int svo_k_pyramid_level0(svo_ctx* ctx, const uint8_t* imgs, int batch, int w, int h, int row_stride, size_t image_stride, uint8_t* pyr,
                         size_t pyr_stride, hipStream_t st) {
  hipLaunchKernelGGL(pyr_copy_kernel, dim3(svo_div_up(w, 64), svo_div_up(h, 4), batch), dim3(256), 0, st, imgs, w, h, row_stride, image_stride,
                     pyr, pyr_stride);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_k_lk(svo_ctx* ctx, const uint8_t* pyr_prev, const uint8_t* pyr_next, int w, int h, const float* xy, int n,
             float* out_xy, uint8_t* status) {
  if (n <= 0) return SVO_OK;
  hipLaunchKernelGGL(lk_kernel, dim3(svo_div_up(n, FPB)), dim3(LKT * FPB), 0, ctx->stream, pyr_prev, pyr_next, w, h, xy, n, out_xy, status);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_k_track(svo_ctx* ctx, const uint8_t* pyr_prev, const uint8_t* pyr_next, int w, int h, const float* xy,
                const float* initial_xy, const int* n_dev, int n_max, float* fwd_xy, uint8_t* keep_flag, float* parallax,
                float* kept_xy, int* kept_index, int* n_kept, float* av_parallax, const SvoTrackCarry* carry) {
  if (n_max > 0) {
    SvoProfScope prof(ctx, SVO_PROF_LK_FB);
    hipLaunchKernelGGL(lk_fb_kernel, dim3(svo_div_up(n_max, FPB)), dim3(LKT * FPB), 0, ctx->stream, pyr_prev, pyr_next, w, h, xy, initial_xy, n_dev,
                       n_max, fwd_xy, keep_flag, parallax);
  }
  hipLaunchKernelGGL(track_compact_kernel, dim3(1), dim3(1024), 0, ctx->stream, fwd_xy, keep_flag, parallax, n_dev, n_max,
                     kept_xy, kept_index, n_kept, av_parallax, carry ? *carry : SvoTrackCarry{});
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_kg_track(svo_ctx* ctx, hipStream_t st, const SvoLkLanes& lanes, int n_lanes, int grid_x) {
  SvoProfScope prof(ctx, SVO_PROF_LK_FB, st);
  if (lanes.map.per_chunk > 0) hipLaunchKernelGGL(lk_fb_group_kernel, dim3(lanes.map.grid()), dim3(64), 0, st, lanes);
  else hipLaunchKernelGGL(lk_fb_group_kernel, dim3(grid_x, n_lanes), dim3(64), 0, st, lanes);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

static int lk_check(svo_ctx* ctx, const void* a, const void* b, int w, int h, int stride) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, a && b, "lk: null image");
  SVO_REQUIRE(ctx, w >= 8 && h >= 8 && w <= ctx->lim.max_width && h <= ctx->lim.max_height && stride >= w,
              "lk: image size outside limits");
  return SVO_OK;
}

extern "C" int svo_build_pyramid(svo_ctx* ctx, const uint8_t* img, int width, int height, int row_stride, uint8_t* levels,
                                 size_t levels_bytes) {
  int rc = lk_check(ctx, img, levels, width, height, row_stride);
  if (rc) return rc;
  const size_t pb = svo_k_pyramid_bytes(width, height);
  SVO_REQUIRE(ctx, levels_bytes >= pb, "build_pyramid: output buffer too small");
  SvoScratch s(ctx);
  uint8_t* dI = s.take<uint8_t>((size_t)width * height);
  uint8_t* dP = s.take<uint8_t>(pb);
  if (!dI || !dP) { ctx->err = "build_pyramid: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(dI, width, img, row_stride, width, height, hipMemcpyHostToDevice, st));
  rc = svo_k_build_pyramid(ctx, dI, 1, width, height, width, (size_t)width * height, dP, pb);
  if (rc) return rc;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(levels, dP, pb, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return SVO_OK;
}

// uploads both images and builds both pyramids in the workspace
static int lk_upload(svo_ctx* ctx, SvoScratch& s, const uint8_t* prev, const uint8_t* next, int w, int h, int stride,
                     uint8_t** pA, uint8_t** pB) {
  const size_t pb = svo_k_pyramid_bytes(w, h);
  uint8_t* dI = s.take<uint8_t>(2 * (size_t)w * h);
  uint8_t* dP = s.take<uint8_t>(2 * pb);
  if (!dI || !dP) { ctx->err = "lk: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(dI, w, prev, stride, w, h, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(dI + (size_t)w * h, w, next, stride, w, h, hipMemcpyHostToDevice, st));
  int rc = svo_k_build_pyramid(ctx, dI, 2, w, h, w, (size_t)w * h, dP, pb);
  if (rc) return rc;
  *pA = dP; *pB = dP + pb;
  return SVO_OK;
}

extern "C" int svo_lk_track(svo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int width, int height, int row_stride,
                            const float* xy, int n, float* out_xy, uint8_t* status) {
  int rc = lk_check(ctx, prev, next, width, height, row_stride);
  if (rc) return rc;
  SVO_REQUIRE(ctx, n >= 0 && (n == 0 || (xy && out_xy && status)), "lk_track: null buffer");
  if (n == 0) return SVO_OK;
  SvoScratch s(ctx);
  uint8_t *pA, *pB;
  rc = lk_upload(ctx, s, prev, next, width, height, row_stride, &pA, &pB);
  if (rc) return rc;
  float* dxy = s.take<float>(2 * (size_t)n);
  float* dout = s.take<float>(2 * (size_t)n);
  uint8_t* dst = s.take<uint8_t>(n);
  if (!dxy || !dout || !dst) { ctx->err = "lk_track: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dxy, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
  rc = svo_k_lk(ctx, pA, pB, width, height, dxy, n, dout, dst);
  if (rc) return rc;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(out_xy, dout, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(status, dst, n, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return SVO_OK;
}

extern "C" int svo_track_features(svo_ctx* ctx, const uint8_t* prev, const uint8_t* next, int width, int height,
                                  int row_stride, const float* xy, const float* initial_xy, int n, float* kept_xy,
                                  int* kept_index, int* n_kept, float* av_parallax) {
  int rc = lk_check(ctx, prev, next, width, height, row_stride);
  if (rc) return rc;
  SVO_REQUIRE(ctx, n >= 0 && n_kept && av_parallax && (n == 0 || (xy && initial_xy && kept_xy && kept_index)),
              "track_features: null buffer");
  *n_kept = 0; *av_parallax = 0.f;
  if (n == 0) return SVO_OK;
  SvoScratch s(ctx);
  uint8_t *pA, *pB;
  rc = lk_upload(ctx, s, prev, next, width, height, row_stride, &pA, &pB);
  if (rc) return rc;
  float* dxy = s.take<float>(2 * (size_t)n);
  float* dinit = s.take<float>(2 * (size_t)n);
  float* dfwd = s.take<float>(2 * (size_t)n);
  float* dpar = s.take<float>(n);
  float* dkept = s.take<float>(2 * (size_t)n);
  int* dkidx = s.take<int>(n);
  uint8_t* dkeep = s.take<uint8_t>(n);
  int* dn = s.take<int>(1);
  float* dav = s.take<float>(1);
  if (!dxy || !dinit || !dfwd || !dpar || !dkept || !dkidx || !dkeep || !dn || !dav) {
    ctx->err = "track_features: workspace too small";
    return SVO_ERR_CAPACITY;
  }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dxy, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dinit, initial_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
  rc = svo_k_track(ctx, pA, pB, width, height, dxy, dinit, nullptr, n, dfwd, dkeep, dpar, dkept, dkidx, dn, dav);
  if (rc) return rc;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(n_kept, dn, sizeof(int), hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(av_parallax, dav, sizeof(float), hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (*n_kept > 0) {
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(kept_xy, dkept, sizeof(float) * 2 * (size_t)*n_kept, hipMemcpyDeviceToHost, st));
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(kept_index, dkidx, sizeof(int) * (size_t)*n_kept, hipMemcpyDeviceToHost, st));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  return SVO_OK;
}
