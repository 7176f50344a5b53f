// "Predicate + stable compaction" tails that run in the LAST workgroup to arrive of a multi-workgroup kernel, so that a
// stage of the keyframe chain is one launch instead of two (flags + compaction, sparse stereo + triangulation).  With
// several stereo streams on the GPU each of these small dependent launches costs 25-55 us; the hand-over inside a launch
// uses the no-cache-maintenance protocol of DESIGN.md section 6: the producers' few words are written through (relaxed
// agent-scope stores), `s_waitcnt vmcnt(0)` = acknowledged, relaxed arrival counter, the last arrival reads the words
// with relaxed agent-scope loads (served from the coherence point, not from its own L2).
#ifndef SVO_TAIL_DEVICE_H_
#define SVO_TAIL_DEVICE_H_
#include "kernels.h"
#include "ref_constants.h"

#if defined(__HIPCC__)
// Every thread of every workgroup calls this after its write-through stores; true in all threads of the last workgroup.
__device__ __forceinline__ bool svo_last_arrival(unsigned* arrive, unsigned target, int* sFlag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) *sFlag = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == target;
  __syncthreads();
  return *sFlag != 0;
}

template <typename T>
__device__ __forceinline__ void svo_wt_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T svo_coherent_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Stable compaction step for one chunk of T items (T threads); returns this thread's output slot or -1.
template <int T>
__device__ __forceinline__ int svo_compact_slot(bool keep, int& base, int* sWave) {
  const unsigned long long mask = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int before = __popcll(mask & ((1ull << lane) - 1ull));
  if (lane == 0) sWave[wave] = __popcll(mask);
  __syncthreads();
  int off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < T / 64; ++w) {
    const int c = sWave[w];
    if (w < wave) off += c;
    total += c;
  }
  const int slot = keep ? base + off + before : -1;
  base += total;
  __syncthreads();
  return slot;
}

// src/image_processor.cpp:178-207 for n features by one workgroup of T threads.  COHERENT: `disp` was written by other
// workgroups of this very launch.
template <int T, bool COHERENT>
__device__ __forceinline__ void svo_triangulate_block(const float* __restrict__ xy, const float* disp, int n, const SvoMat4& M,
                                                      float* __restrict__ kept_xy, float* __restrict__ xyz,
                                                      int* __restrict__ kept_index, int* __restrict__ n_kept, int* sWave) {
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += T) {
    const int i = c0 + threadIdx.x;
    float x = 0.f, y = 0.f, d = 0.f;
    bool keep = false;
    if (i < n) {
      x = xy[2 * i]; y = xy[2 * i + 1];
      d = COHERENT ? svo_coherent_load(&disp[i]) : disp[i];
      keep = d > svo_ref::TRIANGULATE_MIN_DISPARITY;  // src/image_processor.cpp:194
    }
    const int slot = svo_compact_slot<T>(keep, base, sWave);
    if (slot >= 0) {
      const float v[4] = {x, y, d, 1.0f};
      float wv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += (double)M.m[4 * r + k] * (double)v[k];
        wv[r] = (float)s;
      }
      kept_xy[2 * slot] = x; kept_xy[2 * slot + 1] = y;
      xyz[3 * slot] = wv[0] / wv[3]; xyz[3 * slot + 1] = wv[1] / wv[3]; xyz[3 * slot + 2] = wv[2] / wv[3];
      if (kept_index) kept_index[slot] = i;
    }
  }
  if (threadIdx.x == 0) *n_kept = base;
}
#endif
#endif  // SVO_TAIL_DEVICE_H_
