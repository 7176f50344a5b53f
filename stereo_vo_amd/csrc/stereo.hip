// a7 — SAD block-matching disparity: cv::StereoBM::create(48, 21)->compute(L, R) + convertTo(CV_32F, 1/16)
// at reference src/image_processor.cpp:173-176 (semantics: SURVEY.md Appendix A.2; XSOBEL prefilter
// cap 31, minDisparity 0, textureThreshold 10, uniquenessRatio 15, no speckle / L-R check).
// All integer arithmetic => bit-exact by construction, any summation order.
//
//   stereo_at_kernel     : what the pipeline uses.  The reference samples the dense map only at the
//                          <=300 feature pixels (src/image_processor.cpp:193) and every step of the block
//                          matcher is local, so evaluating it at those pixels alone is exactly
//                          equivalent (SURVEY C-8).  One 256-thread workgroup per feature: the raw
//                          23x23 (left) and 23x70 (right) patches are staged in LDS, prefiltered in LDS,
//                          and the 48 SADs are accumulated by 4 waves (lane = disparity, wave = row
//                          group) with LDS integer atomics.  No intermediate image touches HBM.
//   stereo_prefilter_kernel + stereo_dense_kernel : the drop-in for StereoBM::compute (full CV_16S map),
//                          separable running-window SADs kept in LDS (see the kernel).
#include "common.h"
#include "group_kernels.h"
#include "ref_constants.h"
#include "tail_device.h"

namespace {
constexpr int CAP = 31, TEXTURE_THRESHOLD = 10, UNIQUENESS_RATIO = 15;
constexpr int MAX_NDISP = 64, MAX_BLOCK = 21;

__device__ __forceinline__ int pf_row(int y, int H) {
  if (y < 0) return H > 1 ? 1 : 0;
  if (y >= H) return H > 1 ? H - 2 : 0;
  return y;
}

// XSOBEL prefilter value at (x,y) from a raw image (global or LDS accessor).
template <typename Load>
__device__ __forceinline__ int prefilter_at(Load I, int x, int y, int W, int H) {
  if (x <= 0 || x >= W - 1) return CAP;
  if ((H & 1) && y == H - 1) return CAP;  // leftover odd row
  const int y0 = pf_row(y - 1, H), y2 = pf_row(y + 1, H);
  const int v = (I(x + 1, y0) - I(x - 1, y0)) + 2 * (I(x + 1, y) - I(x - 1, y)) + (I(x + 1, y2) - I(x - 1, y2));
  return min(max(v, -CAP), CAP) + CAP;
}

// Winner selection + uniqueness + sub-pixel from sad[-1..ndisp] (index i = ndisp-1-d). Returns CV_16S value.
__device__ __forceinline__ int bm_select(int* s /* points at index 0, s[-1] and s[ndisp] writable */, int ndisp,
                                         int tsum) {
  if (tsum < TEXTURE_THRESHOLD) return -16;
  int minsad = 0x7fffffff, mind = -1;
  for (int i = 0; i < ndisp; ++i)
    if (s[i] < minsad) { minsad = s[i]; mind = i; }
  const int thresh = minsad + (minsad * UNIQUENESS_RATIO / 100);
  for (int i = 0; i < ndisp; ++i)
    if ((i < mind - 1 || i > mind + 1) && s[i] <= thresh) return -16;
  s[-1] = s[1];
  s[ndisp] = s[ndisp - 2];
  const int p = s[mind + 1], n = s[mind - 1];
  const int dd = p + n - 2 * s[mind] + abs(p - n);
  return (short)(((ndisp - mind - 1) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4);
}
}  // namespace

// Sparse StereoBM at one feature pixel (x, y) by one 256-thread workgroup; the disparity is returned in thread 0.
// Control flow is workgroup-uniform (the function contains barriers).
// BLOCK_C / NDISP_C: the block size and disparity range as compile-time constants (0: the run-time arguments).  The staging and
// the prefilter index their patches by  i / columns, i % columns : with run-time column counts every one of those is an integer
// division sequence (22 of them per thread in the staging alone); the reference's values (21, 48: src/image_processor.cpp:174)
// get their own instance, in which they are multiplications.
// NT: threads that work on the feature together — the 256 of a workgroup, or 64: ONE wavefront per feature (the grouped launch
// with the reference's sizes: a quarter of the wavefronts, no wavefront parked at a workgroup barrier while another stages).
template <int BLOCK_C, int NDISP_C, int NT>
__device__ __forceinline__ float stereo_at_block_t(const uint8_t* __restrict__ L, const uint8_t* __restrict__ R, int W, int H,
                                                   int stride, int ndisp_rt, int block_rt, int x, int y) {
  const int ndisp = NDISP_C ? NDISP_C : ndisp_rt, block = BLOCK_C ? BLOCK_C : block_rt;
  const int tid = threadIdx.x;
  const int half = block / 2;
  if (!(x >= ndisp - 1 + half && x < W - half && y >= half && y < H - half)) return -1.0f;  // workgroup-uniform
  constexpr int PR = MAX_BLOCK + 2;              // raw rows
  constexpr int LC = MAX_BLOCK + 2;              // raw left cols
  constexpr int RC = MAX_BLOCK + 2 + MAX_NDISP;  // raw right cols
  __shared__ uint8_t sLr[PR][LC + 1], sRr[PR][RC + 1];
  __shared__ __align__(16) uint8_t sLp[MAX_BLOCK][MAX_BLOCK + 3], sRp[MAX_BLOCK][MAX_BLOCK + MAX_NDISP + 3];  // rows are whole dwords (24 and 88 bytes): the SAD reads them as such
  static_assert((MAX_BLOCK + 3) % 4 == 0 && (MAX_BLOCK + MAX_NDISP + 3) % 4 == 0 && MAX_BLOCK % 4 == 1, "dword rows; the packed SAD below is written for 21 = 5 dwords + 1 byte");
  __shared__ int sSad[MAX_NDISP + 2];
  __shared__ int sT;
  const int rows = block + 2, lcols = block + 2, rcols = block + 1 + ndisp;
  const int ly0 = y - half - 1, lx0 = x - half - 1, rx0 = x - half - (ndisp - 1) - 1;
  // both raw patches in ONE memory round trip: every thread issues all of its loads (3 left + 8 right at the maximum block
  // and disparity range) before the first LDS store waits for any of them — a plain copy loop with a run-time trip count
  // is compiled as load, wait, store per iteration: ten dependent trips to L2 / HBM per feature
  {
    constexpr int RCC = NDISP_C ? MAX_BLOCK + 2 + NDISP_C : RC;  // right columns actually staged when the range is a constant
    constexpr int NL = (PR * LC + NT - 1) / NT, NR = (PR * RCC + NT - 1) / NT;
    uint8_t vl[NL], vr[NR];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = tid + NT * k;
      const int ic = min(i, rows * lcols - 1);  // unconditional load of a valid address: predication would put a wait behind each load
      const int r = ic / lcols, c = ic % lcols;
      vl[k] = L[(size_t)pf_row(ly0 + r, H) * stride + min(max(lx0 + c, 0), W - 1)];
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int i = tid + NT * k;
      const int ic = min(i, rows * rcols - 1);
      const int r = ic / rcols, c = ic % rcols;
      vr[k] = R[(size_t)pf_row(ly0 + r, H) * stride + min(max(rx0 + c, 0), W - 1)];
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = tid + NT * k;
      if (i < rows * lcols) sLr[i / lcols][i % lcols] = vl[k];
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int i = tid + NT * k;
      if (i < rows * rcols) sRr[i / rcols][i % rcols] = vr[k];
    }
  }
  for (int i = tid; i < MAX_NDISP + 2; i += NT) sSad[i] = 0;
  if (tid == 0) sT = 0;
  __syncthreads();
  // prefilter in LDS: the staged rows are already row-reflected, so the vertical taps are r-1, r, r+1
  const int pcols_r = block + ndisp - 1;
  for (int i = tid; i < block * block; i += NT) {
    const int r = i / block, c = i % block;
    const int gx = x - half + c, gy = y - half + r;
    int v = CAP;
    if (gx > 0 && gx < W - 1 && !((H & 1) && gy == H - 1)) {
      const int t = (sLr[r][c + 2] - sLr[r][c]) + 2 * (sLr[r + 1][c + 2] - sLr[r + 1][c]) + (sLr[r + 2][c + 2] - sLr[r + 2][c]);
      v = min(max(t, -CAP), CAP) + CAP;
    }
    sLp[r][c] = (uint8_t)v;
  }
  for (int i = tid; i < block * pcols_r; i += NT) {
    const int r = i / pcols_r, c = i % pcols_r;
    const int gx = x - half - (ndisp - 1) + c, gy = y - half + r;
    int v = CAP;
    if (gx > 0 && gx < W - 1 && !((H & 1) && gy == H - 1)) {
      const int t = (sRr[r][c + 2] - sRr[r][c]) + 2 * (sRr[r + 1][c + 2] - sRr[r + 1][c]) + (sRr[r + 2][c + 2] - sRr[r + 2][c]);
      v = min(max(t, -CAP), CAP) + CAP;
    }
    sRp[r][c] = (uint8_t)v;
  }
  __syncthreads();
  // lane = index i (disparity d = ndisp-1-i), wave = row group
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < ndisp) {
    int acc = 0;
    if (block == MAX_BLOCK) {
      // packed: a row of the left patch is 5 dwords + 1 byte, the right row starts `lane` bytes in — seven aligned dwords,
      // byte-aligned in registers — and v_sad_u8 sums four absolute differences per instruction: 13 LDS reads + 12 ALU
      // instructions per row instead of 42 + 63 (exact integers: the same sums)
      constexpr int LW = (MAX_BLOCK + 3) / 4, RW = (MAX_BLOCK + MAX_NDISP + 3) / 4;
      const uint32_t* Lw = reinterpret_cast<const uint32_t*>(&sLp[0][0]);
      const uint32_t* Rw = reinterpret_cast<const uint32_t*>(&sRp[0][0]);
      const int sh = lane & 3, w0 = lane >> 2;
      unsigned a4 = 0;
      for (int r = wave; r < MAX_BLOCK; r += NT / 64) {
        uint32_t l[LW], q[LW + 1];
#pragma unroll
        for (int k = 0; k < LW; ++k) l[k] = Lw[LW * r + k];
#pragma unroll
        for (int k = 0; k < LW + 1; ++k) q[k] = Rw[RW * r + w0 + k];
#pragma unroll
        for (int k = 0; k < LW - 1; ++k) a4 = __builtin_amdgcn_sad_u8(l[k], __builtin_amdgcn_alignbyte(q[k + 1], q[k], sh), a4);
        a4 = __builtin_amdgcn_sad_u8(l[LW - 1] & 0xFFu, __builtin_amdgcn_alignbyte(q[LW], q[LW - 1], sh) & 0xFFu, a4);
      }
      acc = (int)a4;
    } else {
      for (int r = wave; r < block; r += NT / 64)
        for (int c = 0; c < block; ++c) acc += abs((int)sLp[r][c] - (int)sRp[r][c + lane]);  // R col = x-half+c-d
    }
    atomicAdd(&sSad[lane + 1], acc);
  }
  if (wave == 0) {  // texture sum by one wave
    int t = 0;
    for (int i = lane; i < block * block; i += 64) t += abs((int)sLp[i / block][i % block] - CAP);
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if (lane == 0) sT = t;
  }
  __syncthreads();
  float d = 0.f;
  if (tid == 0) d = (float)bm_select(sSad + 1, ndisp, sT) * svo_ref::STEREO_DISPARITY_SCALE;
  return d;
}
template <int NT>
__device__ __forceinline__ float stereo_at_block(const uint8_t* __restrict__ L, const uint8_t* __restrict__ R, int W, int H,
                                                 int stride, int ndisp, int block, int x, int y) {
  if (NT == 64 || (block == svo_ref::STEREO_BLOCK_SIZE && ndisp == svo_ref::STEREO_NUM_DISPARITIES))  // workgroup-uniform; the one-wavefront form is launched for these sizes only
    return stereo_at_block_t<svo_ref::STEREO_BLOCK_SIZE, svo_ref::STEREO_NUM_DISPARITIES, NT>(L, R, W, H, stride, ndisp, block, x, y);
  if constexpr (NT != 64) return stereo_at_block_t<0, 0, NT>(L, R, W, H, stride, ndisp, block, x, y);
  return 0.f;
}

__global__ __launch_bounds__(256) void stereo_at_kernel(const uint8_t* __restrict__ L, const uint8_t* __restrict__ R,
                                                        int W, int H, int stride, int ndisp, int block,
                                                        const float* __restrict__ xy, const int* __restrict__ n_dev,
                                                        int n_host, float* __restrict__ disp) {
  svo_latency_critical();
  const int n = n_dev ? *n_dev : n_host;
  const int f = blockIdx.x;
  if (f >= n) return;
  // at<float>(it->y, it->x): truncation (SURVEY C-13)
  const float d = stereo_at_block<256>(L, R, W, H, stride, ndisp, block, (int)xy[2 * f], (int)xy[2 * f + 1]);
  if (threadIdx.x == 0) disp[f] = d;
}

// Sparse stereo AND the triangulation of src/image_processor.cpp:178-207 in one launch: one workgroup per feature, the
// last one to arrive triangulates / compacts all of them (tail_device.h) and publishes the completion word.
__device__ __forceinline__ void stereo_triangulate_body(const uint8_t* __restrict__ L, const uint8_t* __restrict__ R,
                                                        int W, int H, int stride, int ndisp, int block,
                                                        const float* __restrict__ xy, const int* __restrict__ n_dev,
                                                        int n_host, float* disp, const SvoMat4& M,
                                                        float* __restrict__ kept_xy, float* __restrict__ xyz,
                                                        int* __restrict__ kept_index, int* __restrict__ n_kept,
                                                        SvoPublish pub) {
  svo_latency_critical();
  __shared__ int sWaveT[4];
  __shared__ int sLast;
  const int n = n_dev ? *n_dev : n_host;
  const int f = blockIdx.x;
  if (f < n) {
    const float d = stereo_at_block<256>(L, R, W, H, stride, ndisp, block, (int)xy[2 * f], (int)xy[2 * f + 1]);
    if (threadIdx.x == 0) svo_wt_store(&disp[f], d);
  }
  if (!svo_last_arrival(pub.arrive, pub.target, &sLast)) return;
  svo_triangulate_block<256, true>(xy, disp, n, M, kept_xy, xyz, kept_index, n_kept, sWaveT);
  SvoPublish one = pub;
  one.arrive = nullptr;  // the arrivals have been counted: this workgroup publishes alone
  svo_publish_block(one);
}

__global__ __launch_bounds__(256) void stereo_triangulate_kernel(const uint8_t* __restrict__ L, const uint8_t* __restrict__ R,
                                                                 int W, int H, int stride, int ndisp, int block,
                                                                 const float* __restrict__ xy, const int* __restrict__ n_dev,
                                                                 int n_host, float* disp, SvoMat4 M,
                                                                 float* __restrict__ kept_xy, float* __restrict__ xyz,
                                                                 int* __restrict__ kept_index, int* __restrict__ n_kept,
                                                                 SvoPublish pub) {
  stereo_triangulate_body(L, R, W, H, stride, ndisp, block, xy, n_dev, n_host, disp, M, kept_xy, xyz, kept_index, n_kept, pub);
}

// stream-batched form (group_kernels.h): blockIdx.y = lane.  The dedup of src/image_processor.cpp:113-128 is folded in (one
// launch and one wait for wavefront slots less per keyframe; under the group load the separate dedup launch took 128 us for
// 15 us of work): wavefront 0 of a corner's workgroup tests it against the tracked inliers exactly as dedup_body does
// (sqrtf(dx^2 + dy^2) < min_d); a duplicate gets disparity 0, which the triangulation's validity test (d > 0, :194) drops —
// the surviving corners keep their detection order, as after the separate, order-preserving dedup compaction.
template <int NT>
__global__ __launch_bounds__(NT) void stereo_triangulate_group_kernel(SvoStereoTriLanes g) {
  int li, f;
  if (!svo_xcd_map_item(g.map, li, f)) return;
  const SvoStereoTriLane& a = g.lane[li];
  svo_latency_critical();
  __shared__ int sWaveT[NT / 64];
  __shared__ int sLast, sHit;
  // behind a PnP launch of the same line (round 5: no host turn in between) the matrix, the number of tracked inliers and "nothing to
  // do — the RANSAC bookkeeping wants more hypotheses first" come from the record that launch left (host/chain_math.h)
  const SvoChainRec* ch = a.chain;
  const bool idle = ch && ch->best == -2;
  const int n = idle ? 0 : (a.n_dev ? *a.n_dev : a.n_max);
  const int n_trk = ch ? ch->n_inl : a.n_trk;
  const float* trk = (ch && n_trk <= 0) ? nullptr : a.trk;
  if (f < n) {  // (one instance of the block matcher for the first keyframe — no tracked features to keep away from — and all later ones)
    const float x = a.xy[2 * f], y = a.xy[2 * f + 1];
    bool dup = false;
    if (trk) {  // uniform in the workgroup
      if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        bool hit = false;
        for (int j0 = 0; j0 < n_trk && !hit; j0 += 64) {
          const int j = j0 + lane;
          bool h = false;
          if (j < n_trk) {
            const float dx = x - trk[2 * j], dy = y - trk[2 * j + 1];
            h = sqrtf(dx * dx + dy * dy) < a.min_d;  // src/image_processor.cpp:118-123
          }
          hit = __any(h);
        }
        if (lane == 0) sHit = hit ? 1 : 0;
      }
      __syncthreads();
      dup = sHit != 0;  // workgroup-uniform
    }
    float d = 0.f;
    // at<float>(it->y, it->x): truncation (SURVEY C-13)
    if (!dup) d = stereo_at_block<NT>(a.left, a.right, g.w, g.h, g.stride, g.ndisp, g.block, (int)x, (int)y);
    if (threadIdx.x == 0) svo_wt_store(&a.disp[f], d);
  }
  if (!svo_last_arrival(a.pub.arrive, a.pub.target, &sLast)) return;
  SvoMat4 Mc = a.M;
  if (ch) for (int i = 0; i < 16; ++i) Mc.m[i] = ch->M[i];
  svo_triangulate_block<NT, true>(a.xy, a.disp, n, Mc, a.kept_xy, a.xyz, nullptr, a.n_kept, sWaveT);
  SvoPublish one = a.pub;
  one.arrive = nullptr;  // the arrivals have been counted: this workgroup publishes alone
  svo_publish_block(one);
}

int svo_kg_stereo_triangulate(svo_ctx* ctx, hipStream_t st, const SvoStereoTriLanes& lanes, int n_lanes, int grid_x) {
  SvoProfScope prof(ctx, SVO_PROF_STEREO_AT, st);
  // one wavefront per corner for the reference's block size and range (the instance with compile-time patch sizes), else a workgroup
  const dim3 grid = lanes.map.per_chunk > 0 ? dim3(lanes.map.grid()) : dim3(grid_x, n_lanes);
  if (lanes.block == svo_ref::STEREO_BLOCK_SIZE && lanes.ndisp == svo_ref::STEREO_NUM_DISPARITIES)
    hipLaunchKernelGGL(stereo_triangulate_group_kernel<64>, grid, dim3(64), 0, st, lanes);
  else
    hipLaunchKernelGGL(stereo_triangulate_group_kernel<256>, grid, dim3(256), 0, st, lanes);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

__global__ __launch_bounds__(256) void stereo_prefilter_kernel(const uint8_t* __restrict__ img, int W, int H, int stride,
                                                               uint8_t* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  auto I = [&](int xx, int yy) -> int { return img[(size_t)yy * stride + xx]; };
  out[(size_t)y * W + x] = (uint8_t)prefilter_at(I, x, y, W, H);
}

// Dense map.  Workgroup = 64 columns x 8 rows of output, prefiltered tiles in LDS.  The SAD over the block x block
// window is separable: per disparity slot, column threads form the vertical sums of |L - R| for the 8 output rows by a
// running window (add the entering row, subtract the leaving one), then the horizontal sums over `block` columns are
// formed for 4 adjacent outputs at a time (21 + 6 LDS reads instead of 4 x 21).  Three slots are processed per pass
// (3 x 84 column threads of the 256); slot `ndisp` is the texture sum |L - cap| (same window, same machinery).  All
// SADs of the tile (<= 441 * 62 < 2^16) are kept in LDS as uint16 [slot][pixel]; the winner / uniqueness / sub-pixel
// selection then reads them per pixel.  Integer arithmetic throughout: bit-identical to the direct double loop.
namespace {
constexpr int DT_W = 64, DT_H = 8, DT_PIX = DT_W * DT_H;
constexpr int DT_TH = DT_H + MAX_BLOCK - 1;         // 28 tile rows
constexpr int DT_TWL = DT_W + MAX_BLOCK - 1;        // 84 left tile columns
constexpr int DT_TWR = DT_TWL + MAX_NDISP;          // right tile columns
constexpr int DT_SLOTS = 3;                         // disparity slots per pass
}
__global__ __launch_bounds__(256) void stereo_dense_kernel(const uint8_t* __restrict__ Lp, const uint8_t* __restrict__ Rp,
                                                           int W, int H, int ndisp, int block,
                                                           int16_t* __restrict__ out) {
  __shared__ uint8_t sL[DT_TH][DT_TWL + 4], sR[DT_TH][DT_TWR + 4];
  __shared__ unsigned short sV[DT_SLOTS][DT_H][DT_TWL + 4];
  extern __shared__ unsigned short sSad[];  // [ndisp + 1][DT_PIX]
  const int half = block / 2;
  const int x0 = blockIdx.x * DT_W, y0 = blockIdx.y * DT_H;
  const int tid = threadIdx.x;
  const int th = DT_H + block - 1, twl = DT_W + block - 1, twr = twl + ndisp - 1;
  for (int i = tid; i < th * twl; i += 256) {
    const int r = i / twl, c = i % twl;
    const int gx = x0 - half + c, gy = y0 - half + r;
    sL[r][c] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? Lp[(size_t)gy * W + gx] : 0;
  }
  for (int i = tid; i < th * twr; i += 256) {
    const int r = i / twr, c = i % twr;
    const int gx = x0 - half - (ndisp - 1) + c, gy = y0 - half + r;
    sR[r][c] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? Rp[(size_t)gy * W + gx] : 0;
  }
  __syncthreads();
  const int nslots = ndisp + 1;
  const int vs = tid / DT_TWL, vc = tid % DT_TWL;   // vertical pass: slot-in-pass, tile column (tid < 252 active)
  for (int s0 = 0; s0 < nslots; s0 += DT_SLOTS) {
    // ---- vertical running sums
    if (vs < DT_SLOTS && vc < twl && s0 + vs < nslots) {
      const int slot = s0 + vs;
      const bool tex = slot == ndisp;
      auto AD = [&](int r) -> int {
        const int l = sL[r][vc];
        return tex ? abs(l - CAP) : abs(l - (int)sR[r][vc + slot]);
      };
      int sum = 0;
      for (int r = 0; r < block; ++r) sum += AD(r);
      sV[vs][0][vc] = (unsigned short)sum;
      for (int r = 1; r < DT_H; ++r) {
        sum += AD(r + block - 1) - AD(r - 1);
        sV[vs][r][vc] = (unsigned short)sum;
      }
    }
    __syncthreads();
    // ---- horizontal sums: work item = (slot-in-pass, row, group of 4 adjacent outputs)
    for (int item = tid; item < DT_SLOTS * DT_H * (DT_W / 4); item += 256) {
      const int hs = item / (DT_H * (DT_W / 4)), rem = item % (DT_H * (DT_W / 4));
      const int r = rem / (DT_W / 4), xg = (rem % (DT_W / 4)) * 4;
      if (s0 + hs >= nslots) continue;
      const unsigned short* v = &sV[hs][r][xg];
      int h0 = 0;
      for (int c = 0; c < block; ++c) h0 += v[c];
      const int h1 = h0 - v[0] + v[block], h2 = h1 - v[1] + v[block + 1], h3 = h2 - v[2] + v[block + 2];
      unsigned short* o = &sSad[(size_t)(s0 + hs) * DT_PIX + r * DT_W + xg];
      o[0] = (unsigned short)h0; o[1] = (unsigned short)h1; o[2] = (unsigned short)h2; o[3] = (unsigned short)h3;
    }
    __syncthreads();
  }
  // ---- selection (StereoBM winner, uniqueness, texture, sub-pixel): two pixels per thread
  for (int pix = tid; pix < DT_PIX; pix += 256) {
    const int x = x0 + (pix % DT_W), y = y0 + pix / DT_W;
    if (x >= W || y >= H) continue;
    int res = -16;
    if (x >= ndisp - 1 + half && x < W - half && y >= half && y < H - half) {
      auto S = [&](int i) -> int { return sSad[(size_t)i * DT_PIX + pix]; };
      const int tsum = S(ndisp);
      if (tsum >= TEXTURE_THRESHOLD) {
        int minsad = 0x7fffffff, mind = -1;
        for (int i = 0; i < ndisp; ++i) {
          const int v = S(i);
          if (v < minsad) { minsad = v; mind = i; }
        }
        const int thresh = minsad + (minsad * UNIQUENESS_RATIO / 100);
        bool unique = true;
        for (int i = 0; i < ndisp && unique; ++i)
          if ((i < mind - 1 || i > mind + 1) && S(i) <= thresh) unique = false;
        if (unique) {
          // borders as bm_select: s[-1] = s[1], s[ndisp] = s[ndisp - 2]
          const int p = mind + 1 < ndisp ? S(mind + 1) : S(ndisp - 2);
          const int n = mind - 1 >= 0 ? S(mind - 1) : S(1);
          const int dd = p + n - 2 * minsad + abs(p - n);
          res = (short)(((ndisp - mind - 1) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4);
        }
      }
    }
    out[(size_t)y * W + x] = (int16_t)res;
  }
}

// ----------------------------------------------------------------------------- host side
static int stereo_check(svo_ctx* ctx, const void* l, const void* r, int W, int H, int stride, int ndisp, int block) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, l && r, "stereo: null image");
  SVO_REQUIRE(ctx, W >= 3 && H >= 3 && W <= ctx->lim.max_width && H <= ctx->lim.max_height && stride >= W,
              "stereo: image size outside limits");
  SVO_REQUIRE(ctx, ndisp >= 16 && ndisp <= MAX_NDISP && ndisp % 16 == 0, "stereo: numDisparities must be 16..64, multiple of 16");
  SVO_REQUIRE(ctx, block >= 5 && block <= MAX_BLOCK && (block & 1), "stereo: blockSize must be odd, 5..21");
  return SVO_OK;
}

extern "C" int svo_stereo_disparity_at_dev(svo_ctx* ctx, const uint8_t* left, const uint8_t* right, int width,
                                           int height, int row_stride, int num_disparities, int block_size,
                                           const float* xy, const int* n_dev, int n_max, float* disp) {
  int rc = stereo_check(ctx, left, right, width, height, row_stride, num_disparities, block_size);
  if (rc) return rc;
  SVO_REQUIRE(ctx, n_max >= 0 && (n_max == 0 || (xy && disp)), "stereo_disparity_at: null buffer");
  if (n_max == 0) return SVO_OK;
  SvoProfScope prof(ctx, SVO_PROF_STEREO_AT);
  hipLaunchKernelGGL(stereo_at_kernel, dim3(n_max), dim3(256), 0, ctx->stream, left, right, width, height, row_stride,
                     num_disparities, block_size, xy, n_dev, n_max, disp);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

int svo_k_stereo_triangulate(svo_ctx* ctx, const uint8_t* left, const uint8_t* right, int width, int height, int row_stride,
                             int num_disparities, int block_size, const float* xy, const int* n_dev, int n_max, float* disp,
                             const SvoMat4& M, float* kept_xy, float* xyz, int* kept_index, int* n_kept, int word, SvoPublish* pub_out) {
  int rc = stereo_check(ctx, left, right, width, height, row_stride, num_disparities, block_size);
  if (rc) return rc;
  SVO_REQUIRE(ctx, n_max >= 1 && xy && disp && kept_xy && xyz && n_kept && pub_out, "stereo_triangulate: null buffer");
  SvoPublish pub = svo_publish_next(ctx, word, n_max);
  if (n_max == 1) { const SvoPublish a = svo_arrive_next(ctx, 1); pub.arrive = a.arrive; pub.target = a.target; }  // every workgroup counts
  *pub_out = pub;
  SvoProfScope prof(ctx, SVO_PROF_STEREO_AT);
  hipLaunchKernelGGL(stereo_triangulate_kernel, dim3(n_max), dim3(256), 0, ctx->stream, left, right, width, height, row_stride,
                     num_disparities, block_size, xy, n_dev, n_max, disp, M, kept_xy, xyz, kept_index, n_kept, pub);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  return SVO_OK;
}

extern "C" int svo_stereo_disparity_at(svo_ctx* ctx, const uint8_t* left, const uint8_t* right, int width, int height,
                                       int row_stride, int num_disparities, int block_size, const float* xy, int n,
                                       float* disp) {
  int rc = stereo_check(ctx, left, right, width, height, row_stride, num_disparities, block_size);
  if (rc) return rc;
  SVO_REQUIRE(ctx, n >= 0 && (n == 0 || (xy && disp)), "stereo_disparity_at: null buffer");
  if (n == 0) return SVO_OK;
  SvoScratch s(ctx);
  const size_t px = (size_t)width * height;
  uint8_t* dL = s.take<uint8_t>(px);
  uint8_t* dR = s.take<uint8_t>(px);
  float* dxy = s.take<float>(2 * (size_t)n);
  float* dd = s.take<float>(n);
  if (!dL || !dR || !dxy || !dd) { ctx->err = "stereo_disparity_at: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(dL, width, left, row_stride, width, height, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(dR, width, right, row_stride, width, height, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(dxy, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
  rc = svo_stereo_disparity_at_dev(ctx, dL, dR, width, height, width, num_disparities, block_size, dxy, nullptr, n, dd);
  if (rc) return rc;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(disp, dd, sizeof(float) * n, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return SVO_OK;
}

extern "C" int svo_stereo_bm(svo_ctx* ctx, const uint8_t* left, const uint8_t* right, int width, int height,
                             int row_stride, int num_disparities, int block_size, int16_t* disp16) {
  int rc = stereo_check(ctx, left, right, width, height, row_stride, num_disparities, block_size);
  if (rc) return rc;
  SVO_REQUIRE(ctx, disp16, "stereo_bm: null output");
  SvoScratch s(ctx);
  const size_t px = (size_t)width * height;
  uint8_t* dL = s.take<uint8_t>(px);
  uint8_t* dR = s.take<uint8_t>(px);
  uint8_t* dLp = s.take<uint8_t>(px);
  uint8_t* dRp = s.take<uint8_t>(px);
  int16_t* dD = s.take<int16_t>(px);
  if (!dL || !dR || !dLp || !dRp || !dD) { ctx->err = "stereo_bm: workspace too small"; return SVO_ERR_CAPACITY; }
  hipStream_t st = ctx->stream;
  SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(dL, width, left, row_stride, width, height, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemcpy2DAsync(dR, width, right, row_stride, width, height, hipMemcpyHostToDevice, st));
  const dim3 g1(svo_div_up(width, 64), svo_div_up(height, 4));
  hipLaunchKernelGGL(stereo_prefilter_kernel, g1, dim3(256), 0, st, dL, width, height, width, dLp);
  hipLaunchKernelGGL(stereo_prefilter_kernel, g1, dim3(256), 0, st, dR, width, height, width, dRp);
  const size_t sad_lds = sizeof(unsigned short) * (size_t)(num_disparities + 1) * DT_PIX;  // <= 66,560 B
  SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stereo_dense_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sad_lds));
  hipLaunchKernelGGL(stereo_dense_kernel, dim3(svo_div_up(width, DT_W), svo_div_up(height, DT_H)), dim3(256), sad_lds, st, dLp,
                     dRp, width, height, num_disparities, block_size, dD);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(disp16, dD, sizeof(int16_t) * px, hipMemcpyDeviceToHost, st));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return SVO_OK;
}
