// Developer experiment: wall clock of the host-side dense Cholesky solve of one LM iteration (n = 6 (K-1)).
//   g++ -O2 -o /tmp/chol_time tools/exp/chol_time.cpp -Lstereo_vo_amd -lsvo_hip -Wl,-rpath,$PWD/stereo_vo_amd && /tmp/chol_time
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
extern "C" int svo_cholesky_solve(double* A, double* b, int n);
int main() {
  for (int n : {30, 54, 114}) {
    std::mt19937_64 g(7);
    std::normal_distribution<double> N;
    std::vector<double> M(n * n), A0(n * n, 0.0), b0(n), A(n * n), b(n);
    for (auto& v : M) v = N(g);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += M[i * n + k] * M[j * n + k]; A0[i * n + j] = s + (i == j ? n : 0); }
    for (auto& v : b0) v = N(g);
    const int reps = 20000;
    double sink = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) { memcpy(A.data(), A0.data(), sizeof(double) * n * n); memcpy(b.data(), b0.data(), sizeof(double) * n); svo_cholesky_solve(A.data(), b.data(), n); sink += b[0]; }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("n = %3d: %.2f us per factor + solve (incl. %zu-byte copy)  [%g]\n", n, us, sizeof(double) * n * n, sink);
  }
  return 0;
}
