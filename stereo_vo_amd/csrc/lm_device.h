// Device side of the LM step control: what host/lm.cpp + host/linalg.cpp do between two passes over the observations,
// executed inside the solve kernel so that a whole ceres::Solve (reference src/bundle_adjuster.cpp:140) is a single
// launch with no host round trip per iteration.  Every function applies the host's operations to every element in the
// host's order (declared arithmetic of host/linalg.cpp: each element receives  a_ij - l_i0 l_j0 - l_i1 l_j1 - ...  one
// product at a time in ascending k; forward substitution ascending, back substitution descending, rows scaled by the
// pivots' reciprocals r_i = 1 / l_ii), so the bits are
// the host's — checked by tests/test_ba.py::test_hip_device_cholesky_matches_host and by every pipeline parity test.
#ifndef SVO_LM_DEVICE_H_
#define SVO_LM_DEVICE_H_
#include <hip/hip_runtime.h>

#include "lm_math.h"

#if defined(__HIPCC__)
// value of `v` in lane `k` (k wave-uniform): two scalar reads instead of an LDS permute
__device__ __forceinline__ double svo_readlane_f64(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}

// Dense SPD solve in LDS by the calling workgroup (128 threads).  A: n x n row-major, lower triangle read, overwritten by
// L; b: right-hand side, overwritten by the solution; col: 6 n + 1 doubles of scratch (n for n > 128).  Right-looking: after column j is
// final, every trailing element (i, c), j < c <= i, receives  -= l_ij * l_cj  — the same subtraction, in the same
// ascending-k position of its sequence, as the left-looking loop.  Returns false (in every thread) when a pivot is not
// positive.  Ends with a barrier.
__device__ inline bool svo_dev_cholesky_solve(double* A, double* b, int n, double* col) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int tr = tid >> 3, tc = tid & 7, rstep = nt >> 3;  // trailing update: 8 columns x (threads / 8) rows per sweep
  if (n <= 64) {
    // Panels of PW columns (as host/linalg.cpp; 6 = one pose block of the reduced camera system: 7.8 us for n = 24 against 8.3 with
    // panels of 4 and 9.9 column by column): the first wavefront finishes a panel in registers — lane = row, a column's values
    // cross lanes through v_readlane — and only then the workgroup meets for the trailing update, which applies the panel's columns
    // to an element one after the other in ascending order.  Two barriers per PANEL instead of two per column; every element still
    // receives  a_ic - l_i0 l_c0 - l_i1 l_c1 - ...  in ascending k, so the bits are those of the column-by-column form below.
    // col: PW x n doubles (the panel's columns, col[q * n + i] = L[i][j0 + q]); col[PW * n] carries the verdict.
    constexpr int PW = 6;
    for (int j0 = 0; j0 < n; j0 += PW) {
      const int bw = n - j0 < PW ? n - j0 : PW;
      if (tid < 64) {
        const bool in = tid < n && tid >= j0;
        double pcol[PW];
#pragma unroll
        for (int q = 0; q < PW; ++q) pcol[q] = (in && q < bw) ? A[tid * n + j0 + q] : 0.0;
        bool ok = true;
#pragma unroll
        for (int q = 0; q < PW; ++q) {
          if (q < bw && ok) {  // wave-uniform
#pragma unroll
            for (int r = 0; r < q; ++r) pcol[q] -= pcol[r] * svo_readlane_f64(pcol[r], j0 + q);  // rows below the panel's r-th pivot hold l_i,j0+r; ascending r
            const double s = svo_readlane_f64(pcol[q], j0 + q);
            ok = s > 0;
            if (ok) {
              const double l = sqrt(s);
              const double v = pcol[q] / l;
              pcol[q] = tid == j0 + q ? l : (tid > j0 + q ? v : 0.0);  // (rows above the pivot take no part in later columns of the panel)
            }
          }
        }
        if (ok) {
#pragma unroll
          for (int q = 0; q < PW; ++q)
            if (q < bw && in && tid >= j0 + q) { A[tid * n + j0 + q] = pcol[q]; col[q * n + tid] = pcol[q]; }
        }
        if (tid == 0) col[PW * n] = ok ? 1.0 : -1.0;
      }
      __syncthreads();
      if (!(col[PW * n] > 0.0)) { __syncthreads(); return false; }  // uniform
      const int t0 = j0 + bw;  // first trailing row / column
      for (int i = t0 + tr; i < n; i += rstep) {
        double li[PW];
#pragma unroll
        for (int q = 0; q < PW; ++q) li[q] = q < bw ? col[q * n + i] : 0.0;
        for (int c = t0 + tc; c <= i; c += 8) {
          double v = A[i * n + c];
#pragma unroll
          for (int q = 0; q < PW; ++q) if (q < bw) v -= li[q] * col[q * n + c];
          A[i * n + c] = v;
        }
      }
      __syncthreads();
    }
  } else if (n <= 128) {
    // 64 < n <= 128 (the 20-keyframe reduced camera system of the bulk path, n = 114): the same panels, the first wavefront
    // holding TWO rows per lane — row `lane` in the low set, row `lane + 64` in the high set; a pivot row's values come from the set
    // it lives in (wave-uniform choice).  Every element still receives  a_ic - l_i0 l_c0 - l_i1 l_c1 - ...  in ascending k.
    constexpr int PW = 6;
    for (int j0 = 0; j0 < n; j0 += PW) {
      const int bw = n - j0 < PW ? n - j0 : PW;
      if (tid < 64) {
        const int r0 = tid, r1 = tid + 64;
        const bool in0 = r0 < n && r0 >= j0, in1 = r1 < n && r1 >= j0;
        double p0[PW], p1[PW];
#pragma unroll
        for (int q = 0; q < PW; ++q) { p0[q] = (in0 && q < bw) ? A[r0 * n + j0 + q] : 0.0; p1[q] = (in1 && q < bw) ? A[r1 * n + j0 + q] : 0.0; }
        bool ok = true;
#pragma unroll
        for (int q = 0; q < PW; ++q) {
          if (q < bw && ok) {  // wave-uniform
            const int pr = j0 + q;  // the pivot row
#pragma unroll
            for (int r = 0; r < q; ++r) {
              const double lp = pr < 64 ? svo_readlane_f64(p0[r], pr) : svo_readlane_f64(p1[r], pr - 64);  // l_{pr, j0 + r}
              p0[q] -= p0[r] * lp;
              p1[q] -= p1[r] * lp;
            }
            const double s = pr < 64 ? svo_readlane_f64(p0[q], pr) : svo_readlane_f64(p1[q], pr - 64);
            ok = s > 0;
            if (ok) {
              const double l = sqrt(s);
              const double v0 = p0[q] / l, v1 = p1[q] / l;
              p0[q] = r0 == pr ? l : (r0 > pr ? v0 : 0.0);
              p1[q] = r1 == pr ? l : (r1 > pr ? v1 : 0.0);
            }
          }
        }
        if (ok) {
#pragma unroll
          for (int q = 0; q < PW; ++q) {
            if (q < bw && in0 && r0 >= j0 + q) { A[r0 * n + j0 + q] = p0[q]; col[q * n + r0] = p0[q]; }
            if (q < bw && in1 && r1 >= j0 + q) { A[r1 * n + j0 + q] = p1[q]; col[q * n + r1] = p1[q]; }
          }
        }
        if (tid == 0) col[PW * n] = ok ? 1.0 : -1.0;
      }
      __syncthreads();
      if (!(col[PW * n] > 0.0)) { __syncthreads(); return false; }  // uniform
      const int t0 = j0 + bw;
      for (int i = t0 + tr; i < n; i += rstep) {
        double li[PW];
#pragma unroll
        for (int q = 0; q < PW; ++q) li[q] = q < bw ? col[q * n + i] : 0.0;
        for (int c = t0 + tc; c <= i; c += 8) {
          double v = A[i * n + c];
#pragma unroll
          for (int q = 0; q < PW; ++q) if (q < bw) v -= li[q] * col[q * n + c];
          A[i * n + c] = v;
        }
      }
      __syncthreads();
    }
    if (tid < 64) {  // substitutions by the first wavefront, two rows per lane, x in registers (as the n <= 64 form below)
      const int r0 = tid, r1 = tid + 64;
      const bool in0 = r0 < n, in1 = r1 < n;
      double b0 = in0 ? b[r0] : 0.0, b1 = in1 ? b[r1] : 0.0;
      const double i0 = 1.0 / (in0 ? A[r0 * n + r0] : 1.0), i1 = 1.0 / (in1 ? A[r1 * n + r1] : 1.0);
      for (int k = 0; k < n; ++k) {  // forward, ascending k
        const double l0 = (in0 && r0 > k) ? A[r0 * n + k] : 0.0, l1 = (in1 && r1 > k) ? A[r1 * n + k] : 0.0;
        const double xk = k < 64 ? svo_readlane_f64(b0, k) * svo_readlane_f64(i0, k) : svo_readlane_f64(b1, k - 64) * svo_readlane_f64(i1, k - 64);
        if (r0 == k) b0 = xk; else if (in0 && r0 > k) b0 -= l0 * xk;
        if (r1 == k) b1 = xk; else if (in1 && r1 > k) b1 -= l1 * xk;
      }
      for (int k = n - 1; k >= 0; --k) {  // backward, descending k
        const double l0 = r0 < k ? A[k * n + r0] : 0.0, l1 = (in1 && r1 < k) ? A[k * n + r1] : 0.0;
        const double xk = k < 64 ? svo_readlane_f64(b0, k) * svo_readlane_f64(i0, k) : svo_readlane_f64(b1, k - 64) * svo_readlane_f64(i1, k - 64);
        if (r0 == k) b0 = xk; else if (r0 < k) b0 -= l0 * xk;
        if (r1 == k) b1 = xk; else if (in1 && r1 < k) b1 -= l1 * xk;
      }
      if (in0) b[r0] = b0;
      if (in1) b[r1] = b1;
    }
    __syncthreads();
    return true;
  } else {
  for (int j = 0; j < n; ++j) {
    const double s = A[j * n + j];
    if (!(s > 0)) {  // uniform: every thread reads the same word
      __syncthreads();
      return false;
    }
    const double l = sqrt(s);
    for (int i = j + 1 + tid; i < n; i += nt) { const double v = A[i * n + j] / l; col[i] = v; A[i * n + j] = v; }
    __syncthreads();
    if (tid == 0) A[j * n + j] = l;  // behind the barrier: the other wavefront may still be reading the pivot
    for (int i = j + 1 + tr; i < n; i += rstep) {
      const double li = col[i];
      for (int c = j + 1 + tc; c <= i; c += 8) A[i * n + c] -= li * col[c];
    }
    __syncthreads();
  }
  }
  if (tid < 64 && n <= 64) {
    // substitutions by the first wavefront, lane = row, x in a register; L is final (barrier above), so its loads do not
    // wait for the running solution
    const bool in = tid < n;
    double bi = in ? b[tid] : 0.0;
    const double ri = 1.0 / (in ? A[tid * n + tid] : 1.0);  // the pivots' reciprocals, all lanes side by side: the 2 n divisions of the plain form were the chain
    for (int k = 0; k < n; ++k) {  // forward: b[i] -= L[i][k] x[k], ascending k
      const double lik = (in && tid > k) ? A[tid * n + k] : 0.0;
      const double xk = svo_readlane_f64(bi, k) * svo_readlane_f64(ri, k);
      if (tid == k) bi = xk;
      else if (in && tid > k) bi -= lik * xk;
    }
    for (int k = n - 1; k >= 0; --k) {  // backward: b[i] -= L[k][i] x[k], descending k
      const double lki = tid < k ? A[k * n + tid] : 0.0;
      const double xk = svo_readlane_f64(bi, k) * svo_readlane_f64(ri, k);
      if (tid == k) bi = xk;
      else if (tid < k) bi -= lki * xk;
    }
    if (in) b[tid] = bi;
  } else if (tid < 64) {
    for (int k = 0; k < n; ++k) {
      if (tid == 0) b[k] = b[k] * (1.0 / A[k * n + k]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double bk = b[k];
      for (int i = k + 1 + tid; i < n; i += 64) b[i] -= A[i * n + k] * bk;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int k = n - 1; k >= 0; --k) {
      if (tid == 0) b[k] = b[k] * (1.0 / A[k * n + k]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double bk = b[k];
      for (int i = tid; i < k; i += 64) b[i] -= A[k * n + i] * bk;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  __syncthreads();
  return true;
}

// ---- the throughput form of the dense SPD solve (bulk path's device-side step control, n <= 128: ba_bulk_control_kernel) -------
// svo_dev_cholesky_solve above reproduces host/linalg.cpp's bits: correctly rounded sqrt and divisions on a dependent chain —
// 140 us for n = 114 (measured, round 5), of which the in-panel chain of sqrt / divide sequences is 57, the LDS-bound trailing
// update 35, the two substitutions 14.  The bulk path sums in hardware order (tolerance-level parity), so its
// step control does not need the host's bits — it needs every rank to compute the SAME bits, which any deterministic sequence
// does.  This form therefore uses
//   * pivots by reciprocal square root: rs = rsq(s) + two Newton steps (relative error < 2e-16), l_jj = s rs, column scaled by rs;
//   * the right-hand side as one more row of the matrix: the forward substitution happens inside the factorisation (y = L^-1 b
//     is the row's panel solve), no separate chain;
//   * the backward substitution with the stored reciprocals.
// A: n x n row-major, lower triangle read, overwritten by L; b: right-hand side -> solution; col: 7 n + 8 doubles of scratch.
// Any workgroup size that is a multiple of 64 (written for 512).  Returns false (in every thread) on a non-positive pivot.
__device__ __forceinline__ double svo_rsqrt_newton(double s) {
  double r = __builtin_amdgcn_rsq(s);
  r = r * (1.5 - 0.5 * s * r * r);
  r = r * (1.5 - 0.5 * s * r * r);
  return r;
}
__device__ inline bool svo_dev_spd_solve_fast(double* A, double* b, int n, double* col, long long* prof = nullptr /* thread 0: ticks in [panel | trailing update + barriers | back substitution] */) {
  const int tid = threadIdx.x, nt = blockDim.x;
  constexpr int PW = 6;
  long long pt = prof && tid == 0 ? (long long)wall_clock64() : 0;
  auto lap = [&](int i) { if (prof && tid == 0) { const long long t = (long long)wall_clock64(); prof[i] += t - pt; pt = t; } };
  double* inv = col + PW * n;       // n: reciprocals of the diagonal of L
  double* yp = inv + n;             // PW: the panel's part of y = L^-1 b
  double* verdict = yp + PW;        // 1
  for (int j0 = 0; j0 < n; j0 += PW) {
    const int bw = n - j0 < PW ? n - j0 : PW;
    if (tid < 64) {
      const int r0 = tid, r1 = tid + 64;
      const bool in0 = r0 < n && r0 >= j0, in1 = r1 < n && r1 >= j0;
      double p0[PW], p1[PW], pb[PW];  // rows r0, r1 of the panel; the right-hand side's "row" (the same value in every lane)
#pragma unroll
      for (int q = 0; q < PW; ++q) {
        p0[q] = (in0 && q < bw) ? A[r0 * n + j0 + q] : 0.0;
        p1[q] = (in1 && q < bw) ? A[r1 * n + j0 + q] : 0.0;
        pb[q] = q < bw ? b[j0 + q] : 0.0;
      }
      bool ok = true;
#pragma unroll
      for (int q = 0; q < PW; ++q) {
        if (q < bw && ok) {  // wave-uniform
          const int pr = j0 + q;
#pragma unroll
          for (int r = 0; r < q; ++r) {
            const double lp = pr < 64 ? svo_readlane_f64(p0[r], pr) : svo_readlane_f64(p1[r], pr - 64);  // l_{pr, j0 + r}
            p0[q] -= p0[r] * lp;
            p1[q] -= p1[r] * lp;
            pb[q] -= pb[r] * lp;
          }
          const double s = pr < 64 ? svo_readlane_f64(p0[q], pr) : svo_readlane_f64(p1[q], pr - 64);
          ok = s > 0;
          if (ok) {
            const double rs = svo_rsqrt_newton(s), l = s * rs;
            p0[q] = r0 == pr ? l : (r0 > pr ? p0[q] * rs : 0.0);
            p1[q] = r1 == pr ? l : (r1 > pr ? p1[q] * rs : 0.0);
            pb[q] = pb[q] * rs;
            if (tid == 0) inv[pr] = rs;
          }
        }
      }
      if (ok) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
          if (q < bw && in0 && r0 >= j0 + q) { A[r0 * n + j0 + q] = p0[q]; col[q * n + r0] = p0[q]; }
          if (q < bw && in1 && r1 >= j0 + q) { A[r1 * n + j0 + q] = p1[q]; col[q * n + r1] = p1[q]; }
          if (q < bw && tid == 0) { b[j0 + q] = pb[q]; yp[q] = pb[q]; }
        }
      }
      if (tid == 0) *verdict = ok ? 1.0 : -1.0;
    }
    lap(0);
    __syncthreads();
    if (!(*verdict > 0.0)) { __syncthreads(); return false; }  // uniform
    const int t0 = j0 + bw;  // first trailing row / column
    // trailing update, rows paired (t0 + m, n - 1 - m) so that every group of eight threads walks the same number of columns
    const int rows = n - t0, pairs = (rows + 1) >> 1;
    for (int m = tid >> 3; m < pairs; m += nt >> 3) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int i = half == 0 ? t0 + m : n - 1 - m;
        if (half == 1 && i == t0 + m) continue;  // the middle row of an odd count
        double li[PW];
#pragma unroll
        for (int q = 0; q < PW; ++q) li[q] = q < bw ? col[q * n + i] : 0.0;
        for (int c = t0 + (tid & 7); c <= i; c += 8) {
          double v = A[i * n + c];
#pragma unroll
          for (int q = 0; q < PW; ++q) if (q < bw) v -= li[q] * col[q * n + c];
          A[i * n + c] = v;
        }
      }
    }
    // ... and the right-hand side's row: b_c -= sum_q y_q l_cq
    for (int c = t0 + tid; c < n; c += nt) {
      double v = b[c];
#pragma unroll
      for (int q = 0; q < PW; ++q) if (q < bw) v -= yp[q] * col[q * n + c];
      b[c] = v;
    }
    __syncthreads();
    lap(1);
  }
  // b holds y = L^-1 b; backward substitution by the first wavefront, two rows per lane, x in registers
  if (tid < 64) {
    const int r0 = tid, r1 = tid + 64;
    const bool in0 = r0 < n, in1 = r1 < n;
    double b0 = in0 ? b[r0] : 0.0, b1 = in1 ? b[r1] : 0.0;
    const double i0 = in0 ? inv[r0] : 1.0, i1 = in1 ? inv[r1] : 1.0;
    for (int k = n - 1; k >= 0; --k) {
      const double l0 = r0 < k ? A[k * n + r0] : 0.0, l1 = (in1 && r1 < k) ? A[k * n + r1] : 0.0;
      const double xk = k < 64 ? svo_readlane_f64(b0, k) * svo_readlane_f64(i0, k) : svo_readlane_f64(b1, k - 64) * svo_readlane_f64(i1, k - 64);
      if (r0 == k) b0 = xk; else if (r0 < k) b0 -= l0 * xk;
      if (r1 == k) b1 = xk; else if (in1 && r1 < k) b1 -= l1 * xk;
    }
    if (in0) b[r0] = b0;
    if (in1) b[r1] = b1;
  }
  lap(2);
  __syncthreads();
  return true;
}
#endif
#endif  // SVO_LM_DEVICE_H_
