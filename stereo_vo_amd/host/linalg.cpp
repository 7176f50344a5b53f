// Dense Cholesky solve of the reduced camera system (host side of the LM loop, n = 6 (K-1) <= 378).
//
// Declared arithmetic (shared with oracle/ora_ba.cpp's cholesky_solve, which is the plain left-looking
// loop): every element receives  a_ij - l_i0 l_j0 - l_i1 l_j1 - ...  with the products subtracted one at a
// time in ascending k, each multiply and subtract rounded separately (-ffp-contract=off, no FMA target); the two
// substitutions scale a row by the pivot's reciprocal r_i = 1 / l_ii, formed once (round 4: on the device the 2 n divisions
// of the plain form were a dependent chain).
// This file applies those same operations in right-looking order — after column j is final, it is
// subtracted from the trailing rows as contiguous axpy updates — so the compiler can vectorise without
// reassociating anything: the results are bit-identical to the left-looking loop, about 4x faster with AVX2 (n = 54 on an
// EPYC 9575F: 4.4 us per factor + solve with the AVX2 clone; tools/exp/chol_time.cpp).
#include <cmath>
#include <cstddef>
#include <vector>

#include "svo.h"

namespace {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)  // hipcc also parses host files in its device pass
#define SVO_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define SVO_CLONES
#endif

constexpr int PW = 4;  // panel width: the trailing matrix is read and written once per PW columns

SVO_CLONES bool chol_factor_solve(double* __restrict A, double* __restrict b, int n, double* __restrict col) {
  // col: PW contiguous copies of the panel's columns (col[p * n + c] = L[c][j0 + p])
  for (int j0 = 0; j0 < n; j0 += PW) {
    const int bw = n - j0 < PW ? n - j0 : PW;
    for (int p = 0; p < bw; ++p) {
      const int j = j0 + p;
      // panel-internal updates of column j (columns j0..j-1, ascending), then its square root and scaling
      for (int q = 0; q < p; ++q) {
        const double ljq = A[(size_t)j * n + j0 + q];
        for (int i = j; i < n; ++i) A[(size_t)i * n + j] -= A[(size_t)i * n + j0 + q] * ljq;
      }
      const double s = A[(size_t)j * n + j];
      if (!(s > 0)) return false;
      const double l = std::sqrt(s);
      A[(size_t)j * n + j] = l;
      double* __restrict cp = col + (size_t)p * n;
      for (int i = j + 1; i < n; ++i) {
        const double v = A[(size_t)i * n + j] / l;
        A[(size_t)i * n + j] = v;
        cp[i] = v;
      }
    }
    const int t0 = j0 + bw;  // first trailing row/column
    if (bw == PW) {
      const double* __restrict c0 = col; const double* __restrict c1 = col + n;
      const double* __restrict c2 = col + 2 * (size_t)n; const double* __restrict c3 = col + 3 * (size_t)n;
      for (int i = t0; i < n; ++i) {
        const double l0 = c0[i], l1 = c1[i], l2 = c2[i], l3 = c3[i];
        double* __restrict row = A + (size_t)i * n;
        for (int c = t0; c <= i; ++c) row[c] = (((row[c] - l0 * c0[c]) - l1 * c1[c]) - l2 * c2[c]) - l3 * c3[c];
      }
    } else {
      for (int p = 0; p < bw; ++p) {
        const double* __restrict cp = col + (size_t)p * n;
        for (int i = t0; i < n; ++i) {
          const double lij = cp[i];
          double* __restrict row = A + (size_t)i * n;
          for (int c = t0; c <= i; ++c) row[c] -= lij * cp[c];
        }
      }
    }
  }
  // substitutions scale by the pivots' reciprocals, formed once (declared arithmetic, round 4: see oracle/ora_ba.cpp)
  double* __restrict rd = col;  // the panel copies are idle now
  for (int i = 0; i < n; ++i) rd[i] = 1.0 / A[(size_t)i * n + i];
  for (int i = 0; i < n; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= A[(size_t)i * n + k] * b[k];
    b[i] = v * rd[i];
  }
  for (int i = n - 1; i >= 0; --i) {  // inner index DESCENDING: the order the device column sweep produces
    double v = b[i];
    for (int k = n - 1; k > i; --k) v -= A[(size_t)k * n + i] * b[k];
    b[i] = v * rd[i];
  }
  return true;
}
}  // namespace

// A: n x n row-major, lower triangle read and overwritten by L; b: right-hand side, overwritten by the solution.
bool svo_host_cholesky_solve(double* A, double* b, int n) {
  std::vector<double> col((size_t)PW * (size_t)(n > 0 ? n : 1));
  return chol_factor_solve(A, b, n, col.data());
}

extern "C" int svo_cholesky_solve(double* A, double* b, int n) {
  if (!A || !b || n < 0) return SVO_ERR_INVALID;
  return svo_host_cholesky_solve(A, b, n) ? SVO_OK : SVO_ERR_NUMERIC;
}
