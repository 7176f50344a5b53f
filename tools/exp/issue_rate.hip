// What the instructions of the tracker's iteration cost on gfx950: SIMD cycles per wave64 instruction with 1, 2 and 4 wavefronts per
// SIMD issuing the SAME kind of instruction back to back on independent registers (round 5: the stream-batched tracker's throughput
// follows the number of CUs it may use and not its VALU count — which instructions are the expensive ones?).
// Every kernel: 1024 CUs-worth of one-wavefront workgroups (256 CUs x 4 SIMDs x W), each running ITER x 64 instructions.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL(NAME, ASM)                                                                                       \
  __global__ __launch_bounds__(64) void NAME(int iters, unsigned* out) {                                         \
    unsigned r0 = threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 + 11u, r5 = r0 + 13u, r6 = r0 + 17u, r7 = r0 + 19u; \
    unsigned a = blockIdx.x | 1u, b = 0x01020304u;                                                               \
    for (int i = 0; i < iters; ++i) {                                                                            \
      asm volatile(REP64(ASM) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)); \
    }                                                                                                            \
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345u) out[0] = r0;                                         \
  }

#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_MAD24(n) "v_mad_i32_i24 %" #n ", %8, %9, %" #n "\n"
#define A_MUL24(n) "v_mul_i32_i24 %" #n ", %" #n ", %8\n"
#define A_DOT2C(n) "v_dot2c_i32_i16 %" #n ", %8, %9\n"
#define A_DOT2(n) "v_dot2_i32_i16 %" #n ", %8, %9, %" #n "\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define A_ALIGN(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, 1\n"
#define A_DPP(n) "v_add_u32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_DPPB(n) "v_add_u32_dpp %" #n ", %" #n ", %" #n " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define A_SDWA(n) "v_sub_u32_sdwa %" #n ", %" #n ", sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define A_CVT(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define A_RNDNE(n) "v_rndne_f32 %" #n ", %" #n "\n"
#define A_MULF(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define A_ASHR(n) "v_ashrrev_i32 %" #n ", 9, %" #n "\n"
#define A_SNOP(n) "s_nop 0\n"
#define A_SALU(n) "s_add_u32 s20, s20, 1\n"

KERNEL(k_add, A_ADD)
KERNEL(k_mad24, A_MAD24)
KERNEL(k_mul24, A_MUL24)
KERNEL(k_dot2c, A_DOT2C)
KERNEL(k_dot2, A_DOT2)
KERNEL(k_perm, A_PERM)
KERNEL(k_align, A_ALIGN)
KERNEL(k_dpp, A_DPP)
KERNEL(k_dppb, A_DPPB)
KERNEL(k_sdwa, A_SDWA)
KERNEL(k_cvt, A_CVT)
KERNEL(k_rndne, A_RNDNE)
KERNEL(k_mulf, A_MULF)
KERNEL(k_ashr, A_ASHR)
KERNEL(k_snop, A_SNOP)

__global__ __launch_bounds__(64) void k_salu(int iters, unsigned* out) {
  for (int i = 0; i < iters; ++i) asm volatile(REP64(A_SALU) ::: "s20");
  if (iters == -1) out[0] = 1;
}
typedef void (*kern_t)(int, unsigned*);

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned* out = nullptr;
  if (hipMalloc((void**)&out, 64) != hipSuccess) return 1;
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const double ghz = khz * 1e-6;
  struct { const char* name; kern_t k; } ks[] = {
      {"v_add_u32", k_add}, {"v_mad_i32_i24", k_mad24}, {"v_mul_i32_i24", k_mul24}, {"v_dot2c_i32_i16", k_dot2c}, {"v_dot2_i32_i16", k_dot2},
      {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_align}, {"v_add_u32_dpp quad_perm", k_dpp}, {"v_add_u32_dpp row_bcast:15", k_dppb},
      {"v_sub_u32_sdwa", k_sdwa}, {"v_cvt_f32_i32", k_cvt}, {"v_rndne_f32", k_rndne}, {"v_mul_f32", k_mulf}, {"v_ashrrev_i32", k_ashr},
      {"s_nop 0", k_snop}};  // (k_salu: its asm clobbers the register the loop counter may live in — do not run)
  printf("%d CUs, %.2f GHz (reported clock); SIMD cycles per wave64 instruction = time x clock x (4 SIMDs x CUs) / (instructions x wavefronts)\n", cus, ghz);
  const int iters = 2000;
  for (auto& e : ks) {
    printf("%-28s", e.name);
    for (int w : {1, 2, 4, 8}) {
      const int grid = cus * 4 * w;
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(64), 0, 0, 10, out);
      hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(64), 0, 0, iters, out);
      const hipError_t err = hipDeviceSynchronize();
      if (err != hipSuccess) { printf("  %s\n", hipGetErrorString(err)); return 2; }
      const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const double per_wave_instr_cycles = s * ghz * 1e9 / ((double)iters * 64.0);  // cycles of wall clock per instruction of ONE wavefront
      printf("  %d/SIMD: %5.2f cyc per instr of a wave = %5.2f per SIMD", w, per_wave_instr_cycles, per_wave_instr_cycles / w);
    }
    printf("\n");
  }
  hipFree(out);
  return 0;
}
