// a9/a10/a12/a13 — sliding-window bundle adjustment: BundleAdjuster (reference src/bundle_adjuster.cpp:5-163)
// whose solve is ceres::Solve with DENSE_SCHUR (:9-12,140) over ReprojectionFactor residuals
// (src/reprojection_factor.cpp:10-88), quaternion (x) identity local parameterization (:19-20,123),
// oldest pose constant (:130).  LM semantics: SURVEY.md Appendix B; the step control itself is host/lm.cpp
// (svo_lm_solve), this file provides its two passes as HIP kernels plus the graph bookkeeping.
//
// Pass A "linearize" and pass B "backsub" are single sweeps over the observations (landmark-major CSR, wave chunks of
// <= 64 observations made of whole landmarks, lane = observation):
//   pass A: residual + analytic Jacobians (FP64 VALU, fused device function, nothing written back), per-landmark V / g_p
//           by in-wave segment gathers, 3x3 inverse, Y = W s Vd^-1 and the landmark's Schur contribution
//           -Y_k (W_k' s)^T  -> payload1 = [S | g_red | g_c | diag U | cost | sum g_p^2].
//   pass B: recomputes the landmark blocks (cheaper than 144 B/observation of W traffic), back-substitutes the camera
//           step, writes the candidate landmarks and evaluates the candidate cost
//           -> payload2 = [cost_new | model-change(points) | sum dp^2 | sum p^2].
// One exchange per LM iteration (host/lm.cpp): pass B of iteration i and pass A of iteration i+1 (at the candidate,
// with the radius an accepted step produces) run back to back and their payloads leave together:
//   deterministic mode (window-sized problems, everything the pipeline solves): every sum follows the DECLARED order of
//     oracle/ora_ba.cpp (round 4: "chunk order") — landmark sums in observation order; per wave chunk of <= 64
//     observations a partial of every payload element, its contributions added sequentially in (landmark, pair) order —
//     formed INSIDE the wavefront (per-observation Y / W s rows staged in LDS, one owner lane per destination row walks the
//     chunk's destination-ordered pair list); chunk partials added sequentially in chunk order (groups of G chunks first
//     when a problem has more than 128).  What crosses workgroups is E = 36 F(F+1)/2 + 33 F + 2 doubles per chunk (the
//     "wire format": Schur part of the upper pose-pair blocks | per pose g_c, -Y g_p, upper triangle of U | cost, sum g_p^2)
//     instead of rounds 1-3's per-pair 6x6 slots (7 MB written through and read back per LM iteration of a window).
//     [payload2 | decision | wire totals] goes straight into pinned host memory behind a completion word the host polls.
//     Two forms of the same arithmetic, chosen per solve:
//       host-driven           3 launches per LM iteration  ba_step_kernel -> ba_decide_linearize_kernel -> ba_reduce_kernel
//       device-resident       1 launch per SOLVE  ba_lm_kernel: the step control (host/lm.cpp's arithmetic) runs on the
//                                         device too, replicated in every workgroup; several solves (the lanes of a
//                                         pipeline group) share one launch, blockIdx.y = solve; workgroups hand over
//                                         TAGGED granules only (no counters, no cache maintenance)
//     The oracle performs the same sums in the same order, so the whole LM trajectory — and therefore every later PnP
//     inlier set — is bit-identical between CPU and GPU and independent of grid size.  (Needed because the reference's
//     problem has a scale gauge: with one fixed pose and only reprojection factors the iterates slide along a flat
//     direction and amplify any summation-order difference; measured 3e-2 pose drift otherwise.)
//   bulk modes (config 4; svo_ba_options.accumulation): hardware-order sums with a tolerance-level result.
//     ba_backsub_kernel, then ba_linearize_mfma_kernel (each landmark's Schur contribution as a rank-3 update of S on
//     the f64 matrix cores, <= 22 poses) or the LDS-atomic ba_linearize_kernel, both accumulating into ONE device
//     buffer [payload2 | payload1] that a single ncclAllReduce on the adjuster's stream sums over the ranks
//     (svo_ba_set_comm), followed by one D2H copy.
// The n x n (n = 6 (K-1) <= 114) Cholesky, step control and termination run on the host from the (all-reduced)
// payloads, so every rank of a sharded run takes identical decisions.  A rank holds all poses and its own landmarks.
#include <math.h>
#include <stdlib.h>

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <sched.h>
#include <chrono>
#include <deque>
#include <memory>
#include <vector>

#include "kernels.h"
#include "group_kernels.h"
#include "lm_decide.h"
#include "lm_device.h"
#include "ref_constants.h"
#include "reproj_device.h"

int svo_rccl_allreduce_f64(void* buf, size_t count, void* comm, hipStream_t stream, const char** err);  // csrc/rccl.hip
bool svo_throughput_mode();  // host/pipeline.cpp: more than two pipelines share the process

namespace {
std::atomic<int> g_group_lanes{0};  // lanes of the live pipeline groups of this process (svo_ba_note_group_lanes)
int g_ba_cu_share = 32;  // CUs of every 32 the adjusters' streams may use (SVO_BA_CU_SHARE; 32 = unmasked)
// Workgroups of admitted kernels whose workgroups wait for each other (see ba_fused_budget), per device: a process that drives
// several GPUs (one svo_ctx each) must not let one device's adjusters draw from another device's budget.
constexpr int SVO_MAX_DEVICES = 16;
std::atomic<int> g_fused_blocks[SVO_MAX_DEVICES];
int ba_fused_budget(int device);
// ... and across PROCESSES (round 5): the budget belongs to the GPU, not to a process.  /dev/shm/svo_admit_<PCI bus id> holds 64
// slots {pid, admitted workgroups}; a process claims a slot (or the slot of a process that no longer exists), mirrors its own
// admitted total there and counts the live slots of the others against the same budget.  Without /dev/shm (or with
// SVO_BA_XPROC=0) the admission is per process as before; a solve that still cannot become co-resident gives up within its
// bound and is re-run (ba_after_giveup), it is never lost.
struct XprocTable { int magic; int pad; int slot[64][2]; };
struct Xproc {
  XprocTable* t = nullptr;
  int mine = -1;
  bool tried = false;
};
Xproc g_xproc[SVO_MAX_DEVICES];
std::mutex g_xproc_mu;
Xproc& xproc_for(int dev) {
  Xproc& x = g_xproc[dev];
  std::lock_guard<std::mutex> g(g_xproc_mu);
  if (x.tried) return x;
  x.tried = true;
  if (const char* e = getenv("SVO_BA_XPROC")) if (*e && atoi(e) == 0) return x;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus) - 1, dev) != hipSuccess) return x;
  for (char* c = bus; *c; ++c) if (*c == ':' || *c == '.') *c = '_';
  char path[128];
  snprintf(path, sizeof(path), "/dev/shm/svo_admit_%s", bus);
  const int fd = open(path, O_RDWR | O_CREAT, 0666);
  if (fd < 0) return x;
  if (ftruncate(fd, sizeof(XprocTable)) != 0) { close(fd); return x; }
  void* m = mmap(nullptr, sizeof(XprocTable), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return x;
  XprocTable* t = static_cast<XprocTable*>(m);
  const int me = (int)getpid();
  for (int pass = 0; pass < 2 && x.mine < 0; ++pass)
    for (int i = 0; i < 64 && x.mine < 0; ++i) {
      int owner = __atomic_load_n(&t->slot[i][0], __ATOMIC_ACQUIRE);
      const bool dead = owner != 0 && owner != me && kill(owner, 0) != 0 && errno == ESRCH;
      if (owner == me || (pass == 1 && (owner == 0 || dead))) {
        if (owner == me || __atomic_compare_exchange_n(&t->slot[i][0], &owner, me, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
          __atomic_store_n(&t->slot[i][1], 0, __ATOMIC_RELEASE);
          x.mine = i;
        }
      }
    }
  if (x.mine >= 0) x.t = t; else munmap(m, sizeof(XprocTable));
  return x;
}
// workgroups the OTHER live processes have admitted on this device
int xproc_others(Xproc& x) {
  if (!x.t) return 0;
  int sum = 0;
  for (int i = 0; i < 64; ++i) {
    if (i == x.mine) continue;
    const int owner = __atomic_load_n(&x.t->slot[i][0], __ATOMIC_ACQUIRE);
    if (owner == 0) continue;
    const int b = __atomic_load_n(&x.t->slot[i][1], __ATOMIC_ACQUIRE);
    if (b > 0 && (kill(owner, 0) == 0 || errno != ESRCH)) sum += b;
  }
  return sum;
}
struct FusedAdmission {
  int blocks = 0, device = 0;
  bool admit(int n, int dev) {
    dev = dev >= 0 && dev < SVO_MAX_DEVICES ? dev : 0;
    const int budget = ba_fused_budget(dev);
    if (g_fused_blocks[dev].fetch_add(n, std::memory_order_acq_rel) + n > budget) { g_fused_blocks[dev].fetch_sub(n, std::memory_order_acq_rel); return false; }
    Xproc& x = xproc_for(dev);
    if (x.t) {  // publish first, then look at the others: two processes racing may both back off, never both pass
      const int mine_now = __atomic_add_fetch(&x.t->slot[x.mine][1], n, __ATOMIC_ACQ_REL);
      if (mine_now + xproc_others(x) > budget) {
        __atomic_sub_fetch(&x.t->slot[x.mine][1], n, __ATOMIC_ACQ_REL);
        g_fused_blocks[dev].fetch_sub(n, std::memory_order_acq_rel);
        return false;
      }
    }
    blocks = n; device = dev;
    return true;
  }
  void release() {
    if (!blocks) return;
    g_fused_blocks[device].fetch_sub(blocks, std::memory_order_acq_rel);
    Xproc& x = g_xproc[device];
    if (x.t) __atomic_sub_fetch(&x.t->slot[x.mine][1], blocks, __ATOMIC_ACQ_REL);
    blocks = 0;
  }
  ~FusedAdmission() { release(); }
};
constexpr double MIN_DIAG = 1e-6, MAX_DIAG = 1e32, LM_MIN_RADIUS_BULK = 1e-32;
constexpr int BULK_RING = 8;   // status records of the device-side step control (bulk path) the host may lag behind
constexpr int PAY2_SLOTS = 8;  // payload2 (4 doubles) is padded to 8 so that payload1 starts 64-byte aligned behind it

struct BaDev {
  int K = 0, n = 0, M = 0, L = 0, C = 0;
  const double* poses = nullptr;   // K x 7: the point pass A linearises at / pass B steps from
  double* cand_poses = nullptr;    // K x 7 device copy of the candidate poses (written by pass B's first workgroup)
  const double* points = nullptr;  // Npts x 3
  double* cand_points = nullptr;
  // per observation SLOT (slot = 64 * chunk + lane; chunks are padded to a full wave so that a lane finds everything it
  // needs with ONE dependent load round: no chunk table, no CSR walk):
  const int4* rec = nullptr;       // {pose k (-1: padding lane), landmark j, first lane of the landmark's segment, segment length}
  const double* obs_uv = nullptr;  // 2 per slot
  double* sp = nullptr;            // Npts x 3 point Jacobi scales
  double* pay1 = nullptr;          // device payload1 (bulk kernels accumulate here with atomics)
  double* pay2 = nullptr;          // device payload2
  double f = 0, cx = 0, cy = 0;
  // deterministic mode (chunk order): per-chunk destination tables + the partial store
  int det = 0;
  int G = 1, NG = 0, Epad = 0, E = 0;    // chunks per declared group, partials in the store, granules per partial (E rounded up), wire-format elements
  int CPW = 1, GS = 1;                   // chunks per workgroup = per stored partial (1, or G for very large problems: the workgroup then forms its group's sum itself), and stored partials per declared group in the level-2 sums (G / CPW)
  const uint16_t* tab = nullptr;   // chunk tables back to back (u16 words), see ba_build_tables
  const uint32_t* tab_off = nullptr;  // C + 1 offsets into tab
  int tab_lds_words = 0;              // host-driven kernels: u16 words of LDS behind the workgroup's staging area for its chunk's table (0: read from global memory)
  const unsigned* lm_key = nullptr;   // per landmark: low 32 bits of its feature id (null: the problem has none — bulk loads); the key of its landmark-store entry
  double* part1 = nullptr;         // granules {value, tag}: element e of group g at part1[2 * (g * Epad + e)] — a wavefront's partials are one contiguous run
  double* part2 = nullptr;         // granules: pass B's four sums of group g at part2[2 * ((parity * NG + g) * 4 + i)]
  double* pay1_out = nullptr;      // where ba_reduce_kernel writes the wire totals (pinned host memory when single-rank)
  double* pay2_out = nullptr;
  const double* step_in = nullptr;  // [dc (max(n,1)) | candidate poses (7K)]: pinned host memory (read in place, no H2D blit) or device
  // completion flag in pinned host memory (single-rank deterministic mode): the reduce kernel publishes `seq` after its
  // payload, the host polls the word instead of paying a stream wait
  int* flag = nullptr;
  unsigned* arrive = nullptr;   // device counter of finished reduce workgroups (monotone; target = total so far)
  unsigned arrive_target = 0;
  int seq = 0;
  double* ctl_dev = nullptr;    // device [accept (0/1) | next radius]: the chained decision, read by the next pass A
  unsigned long long pay_tag = 0;  // tag of the command in flight (unique per adjuster and command): every partial carries it, a reader takes a value only under the awaited tag
  int pay_parity = 0;           // which half of part2 the command in flight uses
};

// Scalars of the running LM iteration that the chained accept / radius decision needs (host/lm_decide.h).
struct LmCtl { double cost, mcc, radius, decrease_factor; int chain; };

// A chunk's partial sums (and pass B's four sums) cross workgroups — and XCDs, each with its own L2 — inside ONE launch
// when the whole solve is a single kernel.  Measured on MI355X (tools/exp/l2_invalidate.hip: a
// pointer chase through L2-resident data, 88 ns per load alone): with other streams executing agent-scope fences the same
// chase takes 295 / 830 / 1,480 ns per load (seq_cst = `buffer_wbl2` + `buffer_inv`, 1 / 4 / 8 aggressor streams), 190-390 ns
// with release fences only (`buffer_wbl2`), 170-480 ns with acquire fences only (`buffer_inv`); kernel boundaries of
// other streams cost nothing.  So the hand-over here uses NO cache maintenance instruction at all.
typedef double svo_d2 __attribute__((ext_vector_type(2)));
// Two adjacent slot words.  WT (the launch that also consumes them): one 16-byte write-through store (sc1: the data goes to
// the device's coherence point instead of staying dirty in this XCD's L2), so the hand-over needs no `buffer_wbl2` either —
// `s_waitcnt vmcnt(0)` says "arrived".  The s_nop covers the wide-store data hazard the compiler cannot see inside asm.
template <bool WT>
__device__ __forceinline__ void slot_store2(double* p, double x, double y) {
  if constexpr (WT) {
    const svo_d2 v = {x, y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  } else {
    p[0] = x; p[1] = y;
  }
}
__device__ __forceinline__ double slot_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A payload word on its way to the host (or to the all-reduce buffer): a relaxed system-scope store is written through every
// cache level, so that `s_waitcnt vmcnt(0)` means "it has arrived" without any cache maintenance (see reduce_publish).
__device__ __forceinline__ void pay_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// ... or on its way to the other workgroups of the same launch (P.pay_dev): the slots' protocol (sc1 write-through, read at the coherence point)
// Payload that stays on the device for the other workgroups of the SAME launch (ba_lm_kernel) travels as TAGGED GRANULES:
// 16 bytes {value, tag}, written by ONE 16-byte write-through store, the tag unique per solve and command.  A reader takes
// a value only when it finds the tag of the command it is waiting for, and reads again otherwise.  Why not "store,
// s_waitcnt / release fence, arrival counter, load" as everywhere else: measured on MI355X under load (8 stereo streams, 8
// solve launches in flight), a workgroup that had seen the arrival counter complete still read the PREVIOUS command's
// values of a few reduction slices — with sc1 and sc0 sc1 loads, with atomic read-modify-writes as loads, with fine-grained
// memory, with release fences in front of the arrivals — and a second read microseconds later returned the new ones: the
// counter (one memory channel) can be observed before a write-through store to another channel.  Replicated step
// control turns one such read into workgroups that take different branches; the tag makes the hand-over independent of
// any ordering between different addresses.  Since round 4 EVERYTHING that crosses workgroups of a launch — the chunks'
// partials included — travels this way (rounds 2-3 kept per-pair contribution slots under the counter protocol).
__device__ __forceinline__ void granule_store(double* g, double v, unsigned long long tag) { slot_store2<true>(g, v, __longlong_as_double((long long)tag)); }
// up to 8 granules per call, all loads in flight together; idx < 0: skipped.  Returns the mask of granules whose tag matched.
__device__ __forceinline__ unsigned granule_load8(const double* base, const int (&idx)[8], unsigned long long tag, double (&out)[8]) {
  svo_d2 v0, v1, v2, v3, v4, v5, v6, v7;
  const double* p0 = base + 2 * (idx[0] < 0 ? 0 : idx[0]); const double* p1 = base + 2 * (idx[1] < 0 ? 0 : idx[1]);
  const double* p2 = base + 2 * (idx[2] < 0 ? 0 : idx[2]); const double* p3 = base + 2 * (idx[3] < 0 ? 0 : idx[3]);
  const double* p4 = base + 2 * (idx[4] < 0 ? 0 : idx[4]); const double* p5 = base + 2 * (idx[5] < 0 ? 0 : idx[5]);
  const double* p6 = base + 2 * (idx[6] < 0 ? 0 : idx[6]); const double* p7 = base + 2 * (idx[7] < 0 ? 0 : idx[7]);
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
      : "memory");
  const svo_d2 v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
  unsigned ok = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    out[u] = v[u].x;
    if (idx[u] < 0 || (unsigned long long)__double_as_longlong(v[u].y) == tag) ok |= 1u << u;
  }
  return ok;
}
// one granule, waited for (bounded): false when its tag never showed up
__device__ __forceinline__ bool granule_wait(const double* base, int idx, unsigned long long tag, double& out) {
  const int only[8] = {idx, -1, -1, -1, -1, -1, -1, -1};
  double v[8];
  long long t0 = 0;
  for (unsigned spins = 0;; ++spins) {
    if (granule_load8(base, only, tag, v) & 1u) { out = v[0]; return true; }
    __builtin_amdgcn_s_sleep(2);
    if ((spins & 255u) == 255u) {  // the launch-wide deadline, see wait_until
      const long long tn = (long long)wall_clock64();
      if (!t0) t0 = tn;
      else if (tn - t0 > 300000000ll) return false;
    }
  }
}

// One "sentinel" granule per producer (granule first + i * stride, i < count), polled by the calling workgroup until it
// carries `tag`.  A producer writes its sentinel LAST, so a reader that has seen every sentinel finds (almost always)
// everything else in place on its first read — and reads again what is not: every granule carries the tag, the sentinel is
// only a hint that keeps the polling traffic at one load per producer and round instead of one per granule.  Ends with a
// barrier; false (block-uniform through *s_flag): a tag never showed up.
__device__ __forceinline__ bool wait_sentinels(const double* base, int first, int stride, int count, unsigned long long tag, int* s_flag) {
  if (threadIdx.x == 0) *s_flag = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < count; i += (int)blockDim.x) {
    const double* p = base + 2 * ((size_t)first + (size_t)i * stride);
    long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
      svo_d2 v;
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
      if ((unsigned long long)__double_as_longlong(v.y) == tag) break;
      __builtin_amdgcn_s_sleep(4);
      if ((spins & 255u) == 255u) {
        const long long tn = (long long)wall_clock64();
        if (!t0) t0 = tn;
        else if (tn - t0 > 300000000ll) { *s_flag = 0; break; }
      }
    }
  }
  __syncthreads();
  return *s_flag != 0;
}

// The decision for the summed payload2 -> ctl_dev (for the pass-A launch queued behind) and payload slots 4 / 5 (for the host).
__device__ __forceinline__ void decide_device(const LmCtl& c, double cost_new, double mc_points, double* ctl_dev, double* pay2buf) {
  const SvoLmDecision d = svo_lm_decide(c.cost, c.mcc, c.radius, c.decrease_factor, cost_new, mc_points);
  ctl_dev[0] = (double)d.accept; ctl_dev[1] = d.next_radius;
  pay_store(&pay2buf[4], (double)d.accept); pay_store(&pay2buf[5], d.next_radius);
}

// Pass A behind a chained decision: linearise at the candidate with the new radius (accepted) or at the current point
// with the reduced radius (rejected).
__device__ __forceinline__ void apply_ctl(BaDev& P, double& radius, const double* __restrict__ ctl) {
  if (!ctl) return;
  const double acc = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  radius = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (acc != 0.0) { P.points = P.cand_points; P.poses = P.cand_poses; }
}

// ---- bulk / sharded solves with the step control on the device (round 5; ba_bulk_control_kernel below) --------------------
// The LM state lives in device memory (`bctl`, BC_* doubles).  The host enqueues a FIXED sequence per LM iteration —
//   ba_backsub_kernel -> [all-reduce of payload2] -> pass A (MFMA / LDS-atomic kernel) -> [all-reduce of payload1] -> ba_bulk_control_kernel
// — without reading anything back: every kernel takes the point it works at (current / candidate buffers by BC_SEL), the radius
// and whether it has anything to do from that state, pass A takes Ceres' accept / radius decision itself from the (all-reduced)
// payload2 (svo_lm_decide: the same closed form in every workgroup and on every rank), and the control kernel does what
// host/lm.cpp does between two passes: Cholesky of the reduced camera system (n <= 128 in LDS), termination tests, pose update.
enum { BC_DONE = 0, BC_MODE, BC_SEL, BC_RADIUS, BC_DF, BC_COST, BC_INITIAL_COST, BC_MCC, BC_ITER, BC_SUCC, BC_TERM, BC_NEED_LIN,
       BC_LIN_CALLS, BC_STEP_CALLS, BC_NEXT_USED, BC_T0, BC_WORDS = 24 };
enum { BCM_FIRST = 0, BCM_STEP, BCM_RELIN };  // what the sequence slot in flight is: the first linearisation | pass B + chained pass A | pass A alone at the current point
struct BulkSel { const double* bctl; double* pts[2]; double* pos[2]; };  // bctl == null: the host-driven loop (arguments as passed)

__device__ __forceinline__ void bulk_select(BaDev& P, const BulkSel& bs) {
  const int sel = (int)bs.bctl[BC_SEL] & 1;
  P.points = bs.pts[sel]; P.cand_points = bs.pts[sel ^ 1]; P.poses = bs.pos[sel]; P.cand_poses = bs.pos[sel ^ 1];
}
// pass A of a sequence slot: false = nothing to do (the solve has terminated).  Block-uniform.
__device__ __forceinline__ bool bulk_apply(BaDev& P, double& radius, int& first_pass, const BulkSel& bs) {
  if (!bs.bctl) return true;
  const double* c = bs.bctl;
  if (c[BC_DONE] != 0.0) return false;
  const int mode = (int)c[BC_MODE];
  bulk_select(P, bs);
  first_pass = mode == BCM_FIRST;
  radius = c[BC_RADIUS];
  if (mode == BCM_STEP) {  // chained: at the candidate with the new radius, or — rejected / invalid step — at the current point with the reduced one
    const SvoLmDecision d = svo_lm_decide(c[BC_COST], c[BC_MCC], c[BC_RADIUS], c[BC_DF], P.pay2[0], P.pay2[1]);
    radius = d.next_radius;
    if (d.accept) { P.points = P.cand_points; P.poses = P.cand_poses; }
  }
  return true;
}

__device__ __forceinline__ bool inv3_sym(const double* V, double* Vi) {
  const double a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[8];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (!(fabs(det) > 0)) { for (int i = 0; i < 9; ++i) Vi[i] = 0.0; return false; }
  const double id = 1.0 / det;
  Vi[0] = c00 * id; Vi[1] = c01 * id; Vi[2] = c02 * id;
  Vi[3] = Vi[1]; Vi[4] = (a * f - c * c) * id; Vi[5] = (b * c - a * e) * id;
  Vi[6] = Vi[2]; Vi[7] = Vi[5]; Vi[8] = (a * d - b * b) * id;
  return true;
}

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src); }

// Landmark sums of the bulk (hardware-order) paths: NV per-observation terms are summed over each landmark's
// lane segment [first, last] by a segmented inclusive scan (log2(maxlen) shuffle steps) and the segment total
// is broadcast back — instead of every lane walking its whole segment.  Tree order: not for the declared-order
// (deterministic) mode.
template <int NV>
__device__ __forceinline__ void segment_totals(double (&v)[NV], int lane, int first, int last, int maxlen) {
  for (int off = 1; off < maxlen; off <<= 1) {
    const bool take = lane - off >= first;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const double u = __shfl_up(v[i], off);
      v[i] += take ? u : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = shfl_d(v[i], last);
}

// residual + tangent Jacobians of one observation
__device__ __forceinline__ void eval_obs(const double* __restrict__ pose, D3 p, double u, double v, double f, double cx,
                                         double cy, bool want_jc, double* r, double* Jc, double* Jp) {
  double Jq[14];
  reproj_full(pose, p, u, v, f, cx, cy, r, want_jc ? Jq : nullptr, Jp);
  if (want_jc) {
    const double w = pose[0], x = pose[1], y = pose[2], z = pose[3];
    const double T[4][3] = {{-x, -y, -z}, {w, z, -y}, {-z, w, x}, {y, -x, w}};
#pragma unroll
    for (int row = 0; row < 2; ++row) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += Jq[7 * row + k] * T[k][c];
        Jc[6 * row + c] = s;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) Jc[6 * row + 3 + c] = Jq[7 * row + 4 + c];
    }
  }
}
// the same with FMA contraction (bulk kernels only: hardware-ordered sums, tolerance-level parity)
#pragma clang fp contract(fast)
__device__ __forceinline__ void eval_obs_contract(const double* __restrict__ pose, D3 p, double u, double v, double f, double cx,
                                         double cy, bool want_jc, double* r, double* Jc, double* Jp) {
  double Jq[14];
  reproj_full_c(pose, p, u, v, f, cx, cy, r, want_jc ? Jq : nullptr, Jp);
  if (want_jc) {
    const double w = pose[0], x = pose[1], y = pose[2], z = pose[3];
    const double T[4][3] = {{-x, -y, -z}, {w, z, -y}, {-z, w, x}, {y, -x, w}};
#pragma unroll
    for (int row = 0; row < 2; ++row) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += Jq[7 * row + k] * T[k][c];
        Jc[6 * row + c] = s;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) Jc[6 * row + 3 + c] = Jq[7 * row + 4 + c];
    }
  }
}
#pragma clang fp contract(off)
}  // namespace

// One lane's observation: indices, its landmark's lane segment [first, first + len) inside the chunk, the landmark
// position and the measurement.
struct ObsRec { bool active; int o, k, j, first, len; D3 p; double u, v; };

__device__ __forceinline__ ObsRec load_obs(const BaDev& P, int chunk, int lane, const double* __restrict__ points) {
  ObsRec R{false, 0, 0, 0, lane, 0, D3{0, 0, 1}, 0.0, 0.0};
  R.o = chunk * 64 + lane;
  const int4 rc = P.rec[R.o];
  R.u = P.obs_uv[2 * R.o]; R.v = P.obs_uv[2 * R.o + 1];
  R.active = rc.x >= 0;
  if (R.active) {
    R.k = rc.x; R.j = rc.y; R.first = rc.z; R.len = rc.w;
    R.p = D3{points[3 * R.j], points[3 * R.j + 1], points[3 * R.j + 2]};
  }
  return R;
}

// What pass A computes before the trust-region radius enters (residual, Jacobians, the landmark's sums in observation order):
// inside ba_lm_kernel this part runs while the accept / radius decision of the step is still on its way.
struct LinPre { double r[2], Jc[12], Jp[6], V[9], gp[3], cost_l; int maxlen; };
// Per-lane constants of a solve that ba_lm_kernel keeps in registers from pass to pass: the landmark's Jacobi scales
// (fixed by the first pass A).
struct ChunkRegs { double s[3]; };

__device__ __forceinline__ void linearize_prefix(const BaDev& P, const ObsRec& R, const double* __restrict__ poses_, LinPre& q, double& lcost) {
  const bool active = R.active;
  const int k = R.k, first = R.first, len = R.len;
  q.r[0] = q.r[1] = 0.0;
#pragma unroll
  for (int i = 0; i < 12; ++i) q.Jc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) q.Jp[i] = 0.0;
  if (active) {
    eval_obs(poses_ + 7 * k, R.p, R.u, R.v, P.f, P.cx, P.cy, k > 0, q.r, q.Jc, q.Jp);
    lcost += 0.5 * (q.r[0] * q.r[0] + q.r[1] * q.r[1]);
  }
  const double my_cost = active ? 0.5 * (q.r[0] * q.r[0] + q.r[1] * q.r[1]) : 0.0;
  double cost_l = 0.0;  // landmark cost, summed in observation order (deterministic mode)
  int maxlen = len;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
  // landmark sums: every lane of a segment gathers the whole segment in observation order
#pragma unroll
  for (int i = 0; i < 9; ++i) q.V[i] = 0.0;
  q.gp[0] = q.gp[1] = q.gp[2] = 0.0;
  for (int t = 0; t < maxlen; ++t) {
    const int src = (first + t) & 63;
    double w[6], rr[2];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = shfl_d(q.Jp[i], src);
    rr[0] = shfl_d(q.r[0], src); rr[1] = shfl_d(q.r[1], src);
    const double ct = shfl_d(my_cost, src);
    if (t < len) {
      cost_l += ct;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        q.gp[a] += w[a] * rr[0] + w[3 + a] * rr[1];
#pragma unroll
        for (int b = 0; b < 3; ++b) q.V[3 * a + b] += w[a] * w[b] + w[3 + a] * w[3 + b];
      }
    }
  }
  q.cost_l = cost_l;
  q.maxlen = maxlen;
}

// Pass A behind the prefix, bulk modes (hardware-order sums): into the workgroup's LDS image of payload1 (ds_add_f64),
// lgp2 accumulates this lane's share of sum g_p^2.
__device__ __forceinline__ void linearize_suffix_bulk(const BaDev& P, const ObsRec& R, const LinPre& q, double radius, int first_pass, double* sS, double* sGred,
                                                      double* sGc, double* sDU, double& lgp2) {
  const int lane = threadIdx.x & 63, n = P.n;
  const bool active = R.active;
  const int k = R.k, j = R.j, first = R.first, len = R.len;
  const double (&r)[2] = q.r;
  const double (&Jc)[12] = q.Jc;
  const double (&Jp)[6] = q.Jp;
  const double (&V)[9] = q.V;
  const double (&gp)[3] = q.gp;
  const int maxlen = q.maxlen;
  double s[3] = {1, 1, 1};
  if (active) {
    if (first_pass) {
#pragma unroll
      for (int a = 0; a < 3; ++a) s[a] = 1.0 / (1.0 + sqrt(V[4 * a]));
      if (lane == first) { P.sp[3 * j] = s[0]; P.sp[3 * j + 1] = s[1]; P.sp[3 * j + 2] = s[2]; }
    } else {
      s[0] = P.sp[3 * j]; s[1] = P.sp[3 * j + 1]; s[2] = P.sp[3 * j + 2];
    }
    if (lane == first) lgp2 += gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2];
  }
  double Vd[9], Vi[9], gps[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    gps[a] = gp[a] * s[a];
#pragma unroll
    for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) Vd[4 * a] += fmin(fmax(Vd[4 * a], MIN_DIAG), MAX_DIAG) / radius;
  inv3_sym(Vd, Vi);
  double Ws[18], Y[18];
  const bool freep = active && k > 0;
  const int base = 6 * (k - 1);
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) Ws[3 * a + b] = freep ? (Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b]) * s[b] : 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) Y[3 * a + b] = Ws[3 * a] * Vi[b] + Ws[3 * a + 1] * Vi[3 + b] + Ws[3 * a + 2] * Vi[6 + b];
  if (freep) {
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      atomicAdd(&sGc[base + a], Jc[a] * r[0] + Jc[6 + a] * r[1]);
      atomicAdd(&sDU[base + a], Jc[a] * Jc[a] + Jc[6 + a] * Jc[6 + a]);
      atomicAdd(&sGred[base + a], -(Y[3 * a] * gps[0] + Y[3 * a + 1] * gps[1] + Y[3 * a + 2] * gps[2]));
#pragma unroll
      for (int b = 0; b < 6; ++b) atomicAdd(&sS[(base + a) * n + base + b], Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
    }
  }
  for (int t = 0; t < maxlen; ++t) {
    const int src = (first + t) & 63;
    const int kt = __shfl(k, src);
    double Wt[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) Wt[i] = shfl_d(Ws[i], src);
    if (freep && t < len && kt > 0 && src >= lane) {
      const int bt = 6 * (kt - 1);
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
          atomicAdd(&sS[(base + a) * n + bt + b],
                    -(Y[3 * a] * Wt[3 * b] + Y[3 * a + 1] * Wt[3 * b + 1] + Y[3 * a + 2] * Wt[3 * b + 2]));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Deterministic mode: the declared "chunk order" (oracle/ora_ba.cpp).  A wave chunk forms the partial of EVERY payload element
// it contributes to inside the wavefront; what leaves it is E doubles per chunk (group), not one 6x6 block per observation pair.
//   level 0  lanes stage their per-observation rows (W s, Y = W s Vd^-1: 36 doubles) in LDS; one OWNER lane per
//            destination row (upper pose-pair block q, row a) walks the chunk's destination-ordered pair list and adds
//            -(Y_i[a] . (W_t s)[b]) for b = 0..5 sequentially — (landmark, i, t) order, every element from +0.0.  Then the
//            lanes stage their 33 per-pose values (g_c | -Y g_p | upper triangle of J_c^T J_c) in the same LDS rows and the
//            owners of (pose, value) add them in lane order; the landmark scalars go through a small LDS list.
//   level 1  (only problems with more than 128 chunks) a wavefront walks the G chunks of its group and adds every new
//            partial to the group's running sum, kept in the partial store itself;
//   level 2  reduce_elements: thread = element, the NG group sums added sequentially.
// Partials travel as tagged granules {value, tag} (granule_store): a reader takes a value only under the awaited tag.
constexpr int REC_STRIDE = 38;        // doubles per lane of the staging rows (36 used; even: 16-byte alignment)
constexpr int LMS_STRIDE = 4;         // pass B: four scalars per landmark of a chunk, in the (idle) staging rows
constexpr int LMS_COL = 36;           // pass A: cost and g_p^2 of the landmark of rank r in the spare columns 36, 37 of row r
constexpr int PST_MAX = 2048;         // wire elements a wavefront stages in LDS before it posts them (larger E: posted one by one)
// LDS of an owner set (doubles): the staging rows (their two spare columns carry pass A's landmark scalars; pass B, which does
// not stage, keeps its four scalars per landmark in the rows themselves) | the E partials of the chunk on their way out
// (consecutive threads then post consecutive granules: whole lines) | 4 ints
__host__ __device__ static inline int wg_lds_doubles(int E) { return 64 * REC_STRIDE + (E <= PST_MAX ? ((E + 1) & ~1) : 0) + 2; }
struct WgLds { double* rec; double* pst; int* s_ne; uint16_t* tab; };  // tab: the host-driven kernels' LDS copy of the chunk table (behind wg_lds_doubles(E))
__device__ __forceinline__ WgLds wg_lds(double* base, int E) {
  WgLds L;
  const int pst_doubles = E <= PST_MAX ? ((E + 1) & ~1) : 0;
  L.rec = base;
  L.pst = pst_doubles ? base + 64 * REC_STRIDE : nullptr;
  L.s_ne = reinterpret_cast<int*>(base + 64 * REC_STRIDE + pst_doubles);
  L.tab = reinterpret_cast<uint16_t*>(base + 64 * REC_STRIDE + pst_doubles + 2);
  return L;
}
constexpr int TAB_LDS_WORDS = 1024;   // u16 words of a chunk table ba_lm_kernel keeps in LDS (larger tables: host-driven path)

// A chunk's destination table (host: ba_build_tables), u16 words:
//   bstart[nU + 1]  entry range of every upper pose-pair block (bstart[nU] = number of entries)
//   pstart[F + 1]   range of every free pose in the pose-lane list
//   ent[n_ent]      lane i | transposed << 7 | lane t << 8; destination-ordered, inside a destination in (landmark, i, t) order
//   plane[n_free]   bytes: lanes of the free observations, pose-ordered, ascending inside a pose
struct ChunkTab {
  const uint16_t* w; int nU, F;
  __device__ __forceinline__ int bstart(int q) const { return w[q]; }
  __device__ __forceinline__ int pstart(int k) const { return w[nU + 1 + k]; }
  __device__ __forceinline__ int ent(int e) const { return w[nU + F + 2 + e]; }
  __device__ __forceinline__ int plane(int x) const {
    const int at = 2 * (nU + F + 2 + (int)w[nU]) + x;  // byte offset
    return (w[at >> 1] >> (8 * (at & 1))) & 0xFF;
  }
};

// Where a chunk's partials go: group g of the partial store.  `first`: the group's first chunk (its partial starts the
// group's sum); otherwise the partial is added to what the group holds so far (same wavefront, acknowledged stores).
// part1 == null: the compact form (ba_lm_compact_kernel) — the chunk's partials stay in the workgroup's LDS (pst: the E elements of
// pass A; pst2: pass B's four sums) and are added to the running totals there, nothing crosses workgroups.
struct PartSink { double* part1; double* part2; double* pst; int Epad, E, NG, g, first, parity; unsigned long long tag; double* pst2; };
__device__ __forceinline__ void post1(const PartSink& k, int e, double v) {
  if (!k.part1) { k.pst[e] = v; return; }
  double* slot = k.part1 + 2 * ((size_t)k.g * k.Epad + e);
  if (!k.first) v = slot_load(slot) + v;
  granule_store(slot, v, k.tag);
}
__device__ __forceinline__ void emit1(const PartSink& k, int e, double v) {
  if (k.pst) k.pst[e] = v; else post1(k, e, v);
}
__device__ __forceinline__ void emit2(const PartSink& k, int i, double v) {
  if (!k.part1) { k.pst2[i] = v; return; }
  double* slot = k.part2 + 2 * (((size_t)k.parity * k.NG + k.g) * 4 + i);
  if (!k.first) v = slot_load(slot) + v;
  granule_store(slot, v, k.tag);
}

// "my stores have been acknowledged" (write-through stores: they are at the coherence point / in host memory)
__device__ __forceinline__ void stores_acknowledged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// acc += x[0], x[stride], ..., x[(count - 1) stride] in THAT order (count may differ per lane), eight LDS loads in flight:
// a plain loop pays one LDS round trip per term (a lone wavefront: ~100 cycles each)
__device__ __forceinline__ double lds_seq_sum(double acc, const double* x, int count, int stride) {
  int i = 0;
  for (; i + 8 <= count; i += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[(i + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  if (i < count) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[min(i + u, count - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (i + u < count) acc += v[u];
  }
  return acc;
}
// Levels 1 + 2 of the declared order over the C chunk partials x[0], x[stride], ... of one element: groups of G consecutive
// chunks, Q_g = P_c0 + P_c1 + ... from the group's first, total = Q_0 + Q_1 + ... from the first.  (Rounds 4's first version let
// ONE workgroup run a group's chunks one after the other and keep the running group sum in the store; every chunk has its own
// workgroup now — a 10-keyframe window has 200 chunks in groups of two, a 1280x720 window 1,500 in groups of twelve — and
// the grouping is applied here, where the partials are read anyway.  Same additions in the same order.)
__device__ __forceinline__ double grouped_seq_sum(const double* x, int C, int G, int stride) {
  if (G <= 1) return lds_seq_sum(x[0], x + stride, C - 1, stride);
  double total = 0.0;
  for (int c0 = 0; c0 < C; c0 += G) {
    const int m = min(G, C - c0);
    const double q = lds_seq_sum(x[c0 * stride], x + (c0 + 1) * stride, m - 1, stride);
    total = c0 == 0 ? q : total + q;
  }
  return total;
}
// LDS traffic of ONE wavefront is ordered; the fence only keeps the compiler from moving accesses across it
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// rank of the lane's landmark inside its chunk and the number of landmarks of the chunk
__device__ __forceinline__ void landmark_rank(const ObsRec& R, int& rank, int& nlm) {
  const int lane = threadIdx.x & 63;
  const unsigned long long firsts = __ballot(R.active && lane == R.first);
  nlm = __popcll(firsts);
  rank = __popcll(firsts & ((1ull << R.first) - 1ull));  // R.first <= lane < 64
}

// The Schur owners of a chunk: item = (block slot, row a, column group of WIDTH), dealt over the threads of the workgroup;
// `nonempty` (bit q: block q has entries; 0: every block, slot = q) lists the `nb` blocks that are walked.  Every element
// receives the same operations in the same order whatever WIDTH is: -(Y_i[a] . (W_t s)[b]) per pair, added in list order
// from +0.0.
template <int WIDTH>
__device__ __forceinline__ void schur_owners(const ChunkTab& T, const double* rec, const PartSink& sink, unsigned long long nonempty, int nb, int tid, int nt) {
  constexpr int SPLIT = 6 / WIDTH;
  for (int item = tid; item < nb * 6 * SPLIT; item += nt) {
    const int slot = item / (6 * SPLIT), rem = item - slot * (6 * SPLIT), a = rem / SPLIT, b0 = (rem - a * SPLIT) * WIDTH;
    int qb = slot;
    if (nonempty) {  // the slot-th set bit
      unsigned long long m = nonempty;
      for (int i = 0; i < slot; ++i) m &= m - 1ull;
      qb = __ffsll((long long)m) - 1;
    }
    double acc[WIDTH];
#pragma unroll
    for (int j = 0; j < WIDTH; ++j) acc[j] = 0.0;
    const int e0 = T.bstart(qb), e1 = T.bstart(qb + 1);
    if (WIDTH == 1) {
      // One element per owner: the chunks that get here touch one or two blocks with LONG lists (64 brand-new landmarks = one
      // block of 64 entries) and were the slowest chunks of every pass — everybody waits for them.  Four entries' rows are in
      // flight at once (entry -> rows is a dependent LDS chain), the additions stay in list order.  A transposed entry is the same
      // expression with the roles of (a, b) swapped: element (a, b) of B^T is  sum_c Y_i[b][c] (W_t s)[a][c]  — same products, same
      // order of the three-term sum, no branch.
      for (int e = e0; e < e1; e += 4) {
        int en[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) en[u] = T.ent(min(e + u, e1 - 1));
        double y[4][3], w[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double* ri = rec + (en[u] & 63) * REC_STRIDE;
          const double* rt = rec + ((en[u] >> 8) & 63) * REC_STRIDE;
          const bool tr = (en[u] & 0x80) != 0;
          const int ra = tr ? b0 : a, rb = tr ? a : b0;
#pragma unroll
          for (int c = 0; c < 3; ++c) { y[u][c] = ri[18 + 3 * ra + c]; w[u][c] = rt[3 * rb + c]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e + u < e1) acc[0] += -(y[u][0] * w[u][0] + y[u][1] * w[u][1] + y[u][2] * w[u][2]);
      }
      emit1(sink, 36 * qb + 6 * a + b0, acc[0]);
      continue;
    }
    int en_next = e0 < e1 ? T.ent(e0) : 0;  // the next entry is fetched one iteration ahead: entry -> rows is a dependent LDS chain
    for (int e = e0; e < e1; ++e) {
      const int en = en_next;
      if (e + 1 < e1) en_next = T.ent(e + 1);
      const double* ri = rec + (en & 63) * REC_STRIDE;
      const double* rt = rec + ((en >> 8) & 63) * REC_STRIDE;
      if (!(en & 0x80)) {
        const double y0 = ri[18 + 3 * a], y1 = ri[18 + 3 * a + 1], y2 = ri[18 + 3 * a + 2];
        double w[3 * WIDTH];
#pragma unroll
        for (int i = 0; i < 3 * WIDTH; ++i) w[i] = rt[3 * b0 + i];
#pragma unroll
        for (int j = 0; j < WIDTH; ++j) acc[j] += -(y0 * w[3 * j] + y1 * w[3 * j + 1] + y2 * w[3 * j + 2]);
      } else {  // the transpose of the pair's block: element (a, b) of the destination is B[b][a]
        const double w0 = rt[3 * a], w1 = rt[3 * a + 1], w2 = rt[3 * a + 2];
#pragma unroll
        for (int j = 0; j < WIDTH; ++j) acc[j] += -(ri[18 + 3 * (b0 + j)] * w0 + ri[18 + 3 * (b0 + j) + 1] * w1 + ri[18 + 3 * (b0 + j) + 2] * w2);
      }
    }
#pragma unroll
    for (int j = 0; j < WIDTH; ++j) emit1(sink, 36 * qb + 6 * a + b0 + j, acc[j]);
  }
}

// What a lane carries from its own arithmetic behind the prefix into the workgroup's owner phases.
struct SufRegs { double ov[33]; bool freep; };  // (W s and Y go straight into the lane's staging row: holding them too spilled 1.3 KB per lane in ba_lm_kernel)

// Pass A behind the prefix, one lane's arithmetic (registers; the landmark's cost and g_p^2 go to the spare columns of `rec`):
// W s, Y = (W s) Vd^-1 -> the lane's staging row; the 33 per-pose values (g_c | -Y g_p | upper triangle of J_c^T J_c) -> o.
__device__ __forceinline__ void suffix_math(const BaDev& P, const ObsRec& R, const LinPre& q, double radius, int first_pass, double* rec, ChunkRegs* cache, SufRegs& o) {
  const int lane = threadIdx.x & 63;
  const bool active = R.active;
  const int k = R.k, j = R.j, first = R.first;
  const double (&r)[2] = q.r;
  const double (&Jc)[12] = q.Jc;
  const double (&Jp)[6] = q.Jp;
  const double (&V)[9] = q.V;
  const double (&gp)[3] = q.gp;
  int rank, nlm;
  landmark_rank(R, rank, nlm);
  double s[3] = {1, 1, 1};
  if (active) {
    if (first_pass) {
#pragma unroll
      for (int a = 0; a < 3; ++a) s[a] = 1.0 / (1.0 + sqrt(V[4 * a]));
      if (lane == first) { P.sp[3 * j] = s[0]; P.sp[3 * j + 1] = s[1]; P.sp[3 * j + 2] = s[2]; }
      if (cache) { cache->s[0] = s[0]; cache->s[1] = s[1]; cache->s[2] = s[2]; }
    } else if (cache) {
      s[0] = cache->s[0]; s[1] = cache->s[1]; s[2] = cache->s[2];
    } else {
      s[0] = P.sp[3 * j]; s[1] = P.sp[3 * j + 1]; s[2] = P.sp[3 * j + 2];
    }
    if (lane == first) { rec[REC_STRIDE * rank + LMS_COL] = q.cost_l; rec[REC_STRIDE * rank + LMS_COL + 1] = gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2]; }
  }
  double Vd[9], Vi[9], gps[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    gps[a] = gp[a] * s[a];
#pragma unroll
    for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) Vd[4 * a] += fmin(fmax(Vd[4 * a], MIN_DIAG), MAX_DIAG) / radius;
  inv3_sym(Vd, Vi);
  const bool freep = active && k > 0;
  o.freep = freep;
  {
    double Ws[18], Y[18];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) Ws[3 * a + b] = freep ? (Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b]) * s[b] : 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) Y[3 * a + b] = Ws[3 * a] * Vi[b] + Ws[3 * a + 1] * Vi[3 + b] + Ws[3 * a + 2] * Vi[6 + b];
    double* row = rec + lane * REC_STRIDE;  // this lane's staging row [W s (18) | Y (18)]
#pragma unroll
    for (int i = 0; i < 18; ++i) { row[i] = Ws[i]; row[18 + i] = Y[i]; }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      o.ov[a] = Jc[a] * r[0] + Jc[6 + a] * r[1];
      o.ov[6 + a] = -(Y[3 * a] * gps[0] + Y[3 * a + 1] * gps[1] + Y[3 * a + 2] * gps[2]);
    }
  }
  int u = 12;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = a; b < 6; ++b) o.ov[u++] = Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
}

// The owner phases of ONE chunk: its lanes stage their rows, then every thread of the owner set takes destination rows.
//   WG = false: the owner set is the chunk's own wavefront (ba_lm_kernel: both wavefronts of a workgroup run their chunks'
//               phases side by side, each in its own LDS; hand-overs are wave-level fences);
//   WG = true : the owner set is the whole workgroup (the one-wavefront workgroups of the host-driven kernels).
//   (Tried in round 4: ONE staging area per 128-thread workgroup, its two wavefronts taking turns with all 128 lanes as
//   owners — 36 KB of LDS instead of 49, but the turns and their ten barriers took pass A from 10 to 22 us.)
//   rec: 64 x REC_STRIDE (spare columns: the chunk's landmark scalars), sink.pst: E or null, s_ne: 4 ints — LDS of the owner set.
template <bool WG>
__device__ __forceinline__ void chunk_owner_phases(const ObsRec& R, const ChunkTab& T, const SufRegs& o, double* rec, const PartSink& sink, int* s_ne, long long* t_schur = nullptr) {
  const int lane = threadIdx.x & 63;
  const int tid = WG ? (int)threadIdx.x : lane, nt = WG ? (int)blockDim.x : 64;
  const bool mine = WG ? (threadIdx.x >> 6) == 0 : true;
  auto sync = [&]() { if (WG) __syncthreads(); else wave_lds_fence(); };
  const int nU = T.nU, F = T.F;
  unsigned long long nonempty = 0ull;
  int nlm = 0;
  if (mine) {  // (its lanes' rows were staged by suffix_math)
    int rank;
    landmark_rank(R, rank, nlm);
    if (nU <= 64) nonempty = __ballot(lane < nU && T.bstart(lane + 1) > T.bstart(lane));
    if (WG && lane == 0) { s_ne[0] = (int)(nonempty & 0xFFFFFFFFull); s_ne[1] = (int)(nonempty >> 32); s_ne[2] = nlm; }
  }
  if (sink.pst) for (int e = tid; e < 36 * nU; e += nt) sink.pst[e] = 0.0;  // blocks the chunk does not touch: +0.0
  sync();
  if (WG) { nonempty = ((unsigned long long)(unsigned)s_ne[1] << 32) | (unsigned long long)(unsigned)s_ne[0]; nlm = s_ne[2]; }
  // ---- Schur part.  How many columns an owner takes depends on how many blocks the chunk touches at all: a chunk of new
  // landmarks (one observation each, all in the newest pose) has ONE block with 64 entries — six row owners would walk it
  // while the other lanes idle (measured: that chunk's pass took 19 us against a mean of 10 and everybody waited for it).
  if (nU <= 64) {
    const int nb_ne = __popcll(nonempty);
    if (!sink.pst) {  // posted straight from the owners: every element of every block must be written
      for (int e = tid; e < 36 * nU; e += nt) if (!((nonempty >> (e / 36)) & 1ull)) post1(sink, e, 0.0);
    }
    if (nb_ne * 36 <= nt) schur_owners<1>(T, rec, sink, nonempty, nb_ne, tid, nt);
    else if (nb_ne * 18 <= nt) schur_owners<2>(T, rec, sink, nonempty, nb_ne, tid, nt);
    else if (nb_ne * 12 <= nt) schur_owners<3>(T, rec, sink, nonempty, nb_ne, tid, nt);
    else schur_owners<6>(T, rec, sink, nonempty, nb_ne, tid, nt);
  } else {
    schur_owners<6>(T, rec, sink, 0ull, nU, tid, nt);
  }
  sync();
  if (t_schur && threadIdx.x == 0) *t_schur += (long long)wall_clock64();  // (diagnostics: the caller subtracts its own stamp)
  // ---- per-pose values: the rows now carry the 33 values of every free observation
  if (mine && o.freep) {
    double* row = rec + lane * REC_STRIDE;
#pragma unroll
    for (int i = 0; i < 33; ++i) row[i] = o.ov[i];
  }
  sync();
  for (int idx = tid; idx < 33 * F; idx += nt) {
    const int kp = idx / 33, el = idx - 33 * kp;
    double acc = 0.0;
    const int x0 = T.pstart(kp), x1 = T.pstart(kp + 1);
    for (int x = x0; x < x1; x += 8) {  // eight lanes' rows in flight (lane list -> row is a dependent LDS chain), added in list order
      int ln[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) ln[u] = T.plane(min(x + u, x1 - 1));
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = rec[ln[u] * REC_STRIDE + el];
#pragma unroll
      for (int u = 0; u < 8; ++u) if (x + u < x1) acc += v[u];
    }
    emit1(sink, 36 * nU + idx, acc);
  }
  // ---- cost, sum g_p^2: landmark order.  LAST: sum g_p^2 is the readers' sentinel.
  sync();
  if (sink.pst && sink.part1) {  // the chunk's partials leave as one contiguous run
    for (int e = tid; e < sink.E - 2; e += nt) post1(sink, e, sink.pst[e]);
  }
  if (tid < 2) post1(sink, 36 * nU + 33 * F + tid, lds_seq_sum(0.0, rec + LMS_COL + tid, nlm, REC_STRIDE));
  sync();  // rows, scalars and the outgoing partials are reused by the next chunk
}

// pass A of one chunk by a workgroup of the host-driven kernels (DET_THREADS threads): the first wavefront does the lanes'
// arithmetic (lane = observation), every thread of the workgroup then owns destination rows — the owner phases are half of a
// chunk's pass and run twice as wide (measured on the single-stream bench: the pass-A launch is its slowest chunk)
template <int NT>
__device__ __forceinline__ void linearize_chunk_wg1(const BaDev& P, const ObsRec& R, const double* __restrict__ poses_, double radius, int first_pass,
                                                    const ChunkTab& T, const WgLds& L, const PartSink& sink) {
  SufRegs o;
  o.freep = false;
  if (threadIdx.x < 64) {
    LinPre q;
    double unused = 0.0;
    linearize_prefix(P, R, poses_, q, unused);
    suffix_math(P, R, q, radius, first_pass, L.rec, nullptr, o);
  }
  if (NT > 64) __syncthreads(); else wave_lds_fence();  // the LDS copy of the chunk's table (stage_chunk_tab) is complete
  chunk_owner_phases<(NT > 64)>(R, T, o, L.rec, sink, L.s_ne);
}

// One-time reads of the problem image by ba_lm_kernel.  The image stays in PINNED HOST memory (no H2D copy launch in front of
// every solve: that copy was a 45 us blit kernel and a stream dependency per keyframe): `shift` leads from a device-arena
// address to the same word of the host image (0: the device arena is complete, a re-solve).  System-scope loads: the XCD's
// L2 may hold lines of the previous problem at these addresses.
template <typename T>
__device__ __forceinline__ T sys_load(const T* p, ptrdiff_t shift) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "word loads");
  return __hip_atomic_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + shift), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// 16 bytes per lane at system scope (sc0 sc1): consecutive lanes read consecutive records, so one instruction is a run of
// whole 64-byte PCIe reads — word-sized loads of the same records would fetch every line four times.
__device__ __forceinline__ uint4 sys_load16(const void* p, ptrdiff_t shift) {
  uint4 v;
  const char* q = reinterpret_cast<const char*>(p) + shift;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(q) : "memory");
  return v;
}
__device__ __forceinline__ ObsRec load_obs_image(const BaDev& P, int chunk, int lane, const double* __restrict__ points, ptrdiff_t shift) {
  ObsRec R{false, 0, 0, 0, lane, 0, D3{0, 0, 1}, 0.0, 0.0};
  R.o = chunk * 64 + lane;
  const uint4 rc = sys_load16(&P.rec[R.o], shift);
  const uint4 uv = sys_load16(&P.obs_uv[2 * R.o], shift);
  R.u = __hiloint2double((int)uv.y, (int)uv.x); R.v = __hiloint2double((int)uv.w, (int)uv.z);
  R.active = (int)rc.x >= 0;
  if (R.active) {
    R.k = (int)rc.x; R.j = (int)rc.y; R.first = (int)rc.z; R.len = (int)rc.w;
    R.p = D3{sys_load(&points[3 * R.j], shift), sys_load(&points[3 * R.j + 1], shift), sys_load(&points[3 * R.j + 2], shift)};
  }
  return R;
}

// this wavefront's chunk table -> LDS (ba_lm_kernel, once per solve): consecutive lanes read consecutive words of the image
__device__ __forceinline__ void load_chunk_table(const BaDev& P, int chunk, uint16_t* dst, ptrdiff_t shift) {
  const int lane = threadIdx.x & 63;
  const uint32_t o0 = sys_load(&P.tab_off[chunk], shift), o1 = sys_load(&P.tab_off[chunk + 1], shift);
  const int words32 = (int)((o1 - o0 + 1) >> 1);  // tables start on even u16 offsets
  const uint32_t* src = reinterpret_cast<const uint32_t*>(P.tab + o0);
  uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
  for (int i = lane; i < words32; i += 64) d32[i] = sys_load(src + i, shift);  // the host sized the LDS for the problem's largest table
  wave_lds_fence();
}

// Pass B for one wave chunk: back-substitution of the pose step dc_ at (poses_, R.p), candidate landmark (returned in
// `cand`, valid in every active lane of the landmark's segment; written to cand_points_ by the segment's first lane),
// candidate residual against cand_poses_.  Deterministic mode (lms != null): the chunk's four sums go to `sink`.  Otherwise
// a_* accumulate this lane's share of payload2.
__device__ __forceinline__ void backsub_chunk(const BaDev& P, const ObsRec& R, const double* __restrict__ poses_,
                                              const double* __restrict__ cand_poses_, const double* __restrict__ dc_,
                                              double* __restrict__ cand_points_, double radius, D3& cand, double& a_cost,
                                              double& a_mc, double& a_dp2, double& a_p2, const ChunkRegs* cache = nullptr,
                                              double* lms = nullptr, const PartSink* sink = nullptr) {
  const int lane = threadIdx.x & 63;
  const bool active = R.active;
  const bool det = lms != nullptr;
  const int k = R.k, j = R.j, first = R.first, len = R.len;
  double r[2] = {0, 0}, Jc[12], Jp[6], jd[2] = {0, 0};
  double det_c = 0.0, det_mc = 0.0, det_dp2 = 0.0, det_p2 = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) Jp[i] = 0.0;
  const D3 p = R.p;
  cand = p;
  if (active) {
    eval_obs(poses_ + 7 * k, p, R.u, R.v, P.f, P.cx, P.cy, k > 0, r, Jc, Jp);
    if (k > 0) {
      const double* d = dc_ + 6 * (k - 1);
#pragma unroll
      for (int a = 0; a < 6; ++a) { jd[0] += Jc[a] * d[a]; jd[1] += Jc[6 + a] * d[a]; }
    }
  }
  int maxlen = len;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
  double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0}, wd[3] = {0, 0, 0};
  if (!det) {
    double t[12] = {Jp[0] * Jp[0] + Jp[3] * Jp[3], Jp[0] * Jp[1] + Jp[3] * Jp[4], Jp[0] * Jp[2] + Jp[3] * Jp[5],
                    Jp[1] * Jp[1] + Jp[4] * Jp[4], Jp[1] * Jp[2] + Jp[4] * Jp[5], Jp[2] * Jp[2] + Jp[5] * Jp[5],
                    Jp[0] * r[0] + Jp[3] * r[1], Jp[1] * r[0] + Jp[4] * r[1], Jp[2] * r[0] + Jp[5] * r[1],
                    Jp[0] * jd[0] + Jp[3] * jd[1], Jp[1] * jd[0] + Jp[4] * jd[1], Jp[2] * jd[0] + Jp[5] * jd[1]};
    segment_totals<12>(t, lane, first, len > 0 ? first + len - 1 : lane, maxlen);
    V[0] = t[0]; V[1] = V[3] = t[1]; V[2] = V[6] = t[2]; V[4] = t[3]; V[5] = V[7] = t[4]; V[8] = t[5];
    gp[0] = t[6]; gp[1] = t[7]; gp[2] = t[8]; wd[0] = t[9]; wd[1] = t[10]; wd[2] = t[11];
  }
  for (int t = 0; t < (det ? maxlen : 0); ++t) {
    const int src = (first + t) & 63;
    double q[6], rr[2], dd[2];
#pragma unroll
    for (int i = 0; i < 6; ++i) q[i] = shfl_d(Jp[i], src);
    rr[0] = shfl_d(r[0], src); rr[1] = shfl_d(r[1], src);
    dd[0] = shfl_d(jd[0], src); dd[1] = shfl_d(jd[1], src);
    if (t < len) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        gp[a] += q[a] * rr[0] + q[3 + a] * rr[1];
        wd[a] += q[a] * dd[0] + q[3 + a] * dd[1];
#pragma unroll
        for (int b = 0; b < 3; ++b) V[3 * a + b] += q[a] * q[b] + q[3 + a] * q[3 + b];
      }
    }
  }
  if (active) {
    const double s[3] = {cache ? cache->s[0] : P.sp[3 * j], cache ? cache->s[1] : P.sp[3 * j + 1], cache ? cache->s[2] : P.sp[3 * j + 2]};
    double Vd[9], Vi[9], De[3], rh[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      rh[a] = -(gp[a] + wd[a]) * s[a];
#pragma unroll
      for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { De[a] = fmin(fmax(Vd[4 * a], MIN_DIAG), MAX_DIAG) / radius; Vd[4 * a] += De[a]; }
    inv3_sym(Vd, Vi);
    double np[3];
    const double pv[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double y = Vi[3 * a] * rh[0] + Vi[3 * a + 1] * rh[1] + Vi[3 * a + 2] * rh[2];
      const double d = y * s[a];
      np[a] = pv[a] + d;
      if (lane == first) {
        a_mc += 0.5 * y * (De[a] * y - gp[a] * s[a]);
        a_dp2 += d * d;
        a_p2 += pv[a] * pv[a];
      }
    }
    cand = D3{np[0], np[1], np[2]};
    if (lane == first) { cand_points_[3 * j] = np[0]; cand_points_[3 * j + 1] = np[1]; cand_points_[3 * j + 2] = np[2]; }
    double r0, r1;
    reproj_residual(cand_poses_ + 7 * k, cand, R.u, R.v, P.f, P.cx, P.cy, r0, r1);
    a_cost += 0.5 * (r0 * r0 + r1 * r1);
    det_c = 0.5 * (r0 * r0 + r1 * r1);
    det_mc = 0.0; det_dp2 = 0.0; det_p2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double y = Vi[3 * a] * rh[0] + Vi[3 * a + 1] * rh[1] + Vi[3 * a + 2] * rh[2];
      const double d = y * s[a];
      det_mc += 0.5 * y * (De[a] * y - gp[a] * s[a]);
      det_dp2 += d * d;
      det_p2 += pv[a] * pv[a];
    }
  }
  if (det) {  // candidate cost of the landmark in observation order, then the chunk's sums in landmark order
    double cn = 0.0;
    for (int t = 0; t < maxlen; ++t) {
      const double ct = shfl_d(det_c, (first + t) & 63);
      if (t < len) cn += ct;
    }
    int rank, nlm;
    landmark_rank(R, rank, nlm);
    if (active && lane == first) {
      double* lv = lms + LMS_STRIDE * rank;
      lv[0] = cn; lv[1] = det_mc; lv[2] = det_dp2; lv[3] = det_p2;
    }
    wave_lds_fence();
    if (lane < 4) emit2(*sink, lane, lds_seq_sum(0.0, lms + lane, nlm, LMS_STRIDE));
    wave_lds_fence();
  }
}

// [dc | candidate poses] -> LDS (the source may be pinned host memory: ONE PCIe round trip per workgroup instead of one
// per use); the first workgroup also leaves the device copy of the candidate poses that later launches linearise at.
__device__ __forceinline__ void stage_step(const BaDev& P, double* sStep) {
  const int nn = P.n > 0 ? P.n : 1, tot = nn + 7 * P.K;
  // the first two rounds as ONE round trip (window problems: 72-124 words on 64 or 128 threads): both loads are issued,
  // unconditionally and at clamped addresses, before either is stored — the plain loop waits for each PCIe read in turn
  const int i0 = threadIdx.x, i1 = threadIdx.x + blockDim.x;
  const double v0 = P.step_in[min(i0, tot - 1)], v1 = P.step_in[min(i1, tot - 1)];
  if (i0 < tot) { sStep[i0] = v0; if (blockIdx.x == 0 && i0 >= nn) P.cand_poses[i0 - nn] = v0; }
  if (i1 < tot) { sStep[i1] = v1; if (blockIdx.x == 0 && i1 >= nn) P.cand_poses[i1 - nn] = v1; }
  for (int i = threadIdx.x + 2 * blockDim.x; i < tot; i += blockDim.x) {
    const double v = P.step_in[i];
    sStep[i] = v;
    if (blockIdx.x == 0 && i >= nn) P.cand_poses[i - nn] = v;
  }
  __syncthreads();
}
constexpr int STEP_LDS_DOUBLES = 6 * 63 + 7 * 64;

// Every wait inside a launch is bounded by ONE wall-clock deadline (the 100 MHz constant clock, not a spin count whose
// duration depends on what is polled): a workgroup gives up LM_WAIT_TICKS after it started waiting — far beyond any
// solve, well inside the host's own 10 s limit on the completion word (ba_wait_flag) — and the host reports the solve.
constexpr long long LM_WAIT_TICKS = 300000000ll;  // 3 s

// `count` consecutive granules under `tag` -> dst (LDS), values only; up to 8 per thread in flight; a granule whose tag is
// not the awaited one is read again (bounded).  Any workgroup size; no barrier.  false: a tag never showed up.
__device__ __forceinline__ bool fetch_granules(double* dst, const double* src, int count, unsigned long long tag) {
  bool good = true;
  for (int base = 0; base < count; base += 8 * (int)blockDim.x) {
    int gi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * (int)blockDim.x + (int)threadIdx.x;
      gi[u] = i < count ? i : -1;
    }
    double v[8];
    unsigned ok = granule_load8(src, gi, tag, v);
    long long t0 = 0;
    for (unsigned spins = 0; ok != 0xFFu; ++spins) {
      __builtin_amdgcn_s_sleep(2);
      if ((spins & 255u) == 255u) {
        const long long tn = (long long)wall_clock64();
        if (!t0) t0 = tn;
        else if (tn - t0 > LM_WAIT_TICKS) break;
      }
      int again[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) again[u] = (ok >> u) & 1u ? -1 : gi[u];
      double w[8];
      const unsigned ok2 = granule_load8(src, again, tag, w);
#pragma unroll
      for (int u = 0; u < 8; ++u) if (!((ok >> u) & 1u) && ((ok2 >> u) & 1u)) { v[u] = w[u]; ok |= 1u << u; }
    }
    good = good && ok == 0xFFu;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (gi[u] >= 0) dst[gi[u]] = v[u];
  }
  return good;
}

// Level 2 for the wire elements [e0, e1): the NG group sums of every element (granules under `tag`) added sequentially,
// starting from the first.  sm: >= sm_doubles doubles of LDS; out(e, total).  Any workgroup size, block-uniform arguments;
// ends with a barrier.  false (block-uniform): a tag never showed up.
// GROUPED: the stored partials are summed in declared groups of P.GS (large windows of the host-driven path); ba_lm_kernel's
// problems have one chunk per group and instantiate the plain form only (its code is larger than the instruction cache as it is)
template <bool GROUPED, typename Out>
__device__ __forceinline__ bool reduce_elements(const BaDev& P, int e0, int e1, unsigned long long tag, double* sm, int sm_doubles, int* s_flag, Out out, long long* t_waited = nullptr) {
  const int NG = P.NG, Epad = P.Epad;
  const int per_round = max(1, sm_doubles / NG);
  // every group's last-posted element (sum g_p^2) first: one load per group and polling round
  const long long t_in = t_waited && threadIdx.x == 0 ? (long long)wall_clock64() : 0;
  if (!wait_sentinels(P.part1, P.E - 1, Epad, NG, tag, s_flag)) return false;
  if (t_waited && threadIdx.x == 0) *t_waited += (long long)wall_clock64() - t_in;
  for (int eb = e0; eb < e1; eb += per_round) {  // block-uniform trip count
    const int ne = min(per_round, e1 - eb);
    // item i = (group i / ne, element i % ne): consecutive threads read consecutive granules of a group's run
    bool good = true;
    for (int base = 0; base < ne * NG; base += 8 * (int)blockDim.x) {
      int gi[8], at[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * (int)blockDim.x + (int)threadIdx.x;
        const bool use = i < ne * NG;
        const int g = use ? i / ne : 0, el = use ? i - g * ne : 0;
        at[u] = use ? el * NG + g : -1;
        gi[u] = use ? g * Epad + eb + el : -1;
      }
      double v[8];
      unsigned ok = granule_load8(P.part1, gi, tag, v);
      long long t0 = 0;
      for (unsigned spins = 0; ok != 0xFFu; ++spins) {
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 255u) == 255u) {
          const long long tn = (long long)wall_clock64();
          if (!t0) t0 = tn;
          else if (tn - t0 > LM_WAIT_TICKS) break;
        }
        int again[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) again[u] = (ok >> u) & 1u ? -1 : gi[u];
        double w[8];
        const unsigned ok2 = granule_load8(P.part1, again, tag, w);
#pragma unroll
        for (int u = 0; u < 8; ++u) if (!((ok >> u) & 1u) && ((ok2 >> u) & 1u)) { v[u] = w[u]; ok |= 1u << u; }
      }
      good = good && ok == 0xFFu;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (at[u] >= 0) sm[at[u]] = v[u];
    }
    if (!good) *s_flag = 0;
    __syncthreads();
    for (int el = threadIdx.x; el < ne; el += (int)blockDim.x) {
      const double* row = sm + el * NG;
      out(eb + el, GROUPED ? grouped_seq_sum(row, NG, P.GS, 1) : lds_seq_sum(row[0], row + 1, NG - 1, 1));
    }
    __syncthreads();
  }
  return *s_flag != 0;
}

// payload2: the NG group sums of pass B's four scalars (granules under `tag`, half `parity` of the store) added
// sequentially from the first.  sm: >= 4 * NG doubles of LDS, sOut: 4.  Ends with a barrier; false: a tag never showed up.
template <bool GROUPED>
__device__ __forceinline__ bool sum_pay2(const BaDev& P, int parity, unsigned long long tag, double* sm, double* sOut, int* s_flag) {
  const int NG = P.NG;
  const double* base = P.part2 + 2 * ((size_t)parity * NG * 4);
  if (!wait_sentinels(base, 3, 4, NG, tag, s_flag)) return false;
  if (!GROUPED || P.GS <= 1) {
    if (!fetch_granules(sm, base, 4 * NG, tag)) *s_flag = 0;
    __syncthreads();
    if (threadIdx.x < 4) sOut[threadIdx.x] = lds_seq_sum(sm[threadIdx.x], sm + 4 + threadIdx.x, NG - 1, 4);
    __syncthreads();
    return *s_flag != 0;
  }
  // groups of G chunks (at most 128 of them): item = (group, scalar) forms the group's sum Q_g = P_c0 + P_c1 + ... itself,
  // eight granules in flight; then the four totals Q_0 + Q_1 + ... — sm: 4 x groups doubles
  const int G = P.GS, ngroups = (NG + G - 1) / G;
  bool good = true;
  for (int item = threadIdx.x; item < 4 * ngroups; item += (int)blockDim.x) {
    const int g = item >> 2, i = item & 3, c0 = g * G, m = min(G, NG - c0);
    double q = 0.0;
    for (int b0 = 0; b0 < m; b0 += 8) {
      int gi[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) gi[u] = b0 + u < m ? (c0 + b0 + u) * 4 + i : -1;
      double v[8];
      unsigned ok = granule_load8(base, gi, tag, v);
      long long t0 = 0;
      for (unsigned spins = 0; ok != 0xFFu; ++spins) {  // (the sentinels were seen: a granule under another tag is a rare late store)
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 255u) == 255u) {
          const long long tn = (long long)wall_clock64();
          if (!t0) t0 = tn;
          else if (tn - t0 > LM_WAIT_TICKS) break;
        }
        int again[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) again[u] = (ok >> u) & 1u ? -1 : gi[u];
        double w[8];
        const unsigned ok2 = granule_load8(base, again, tag, w);
#pragma unroll
        for (int u = 0; u < 8; ++u) if (!((ok >> u) & 1u) && ((ok2 >> u) & 1u)) { v[u] = w[u]; ok |= 1u << u; }
      }
      good = good && ok == 0xFFu;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (b0 + u < m) q = (b0 + u == 0) ? v[u] : q + v[u];
    }
    sm[4 * g + i] = q;
  }
  if (!good) *s_flag = 0;
  __syncthreads();
  if (threadIdx.x < 4) sOut[threadIdx.x] = lds_seq_sum(sm[threadIdx.x], sm + 4 + threadIdx.x, ngroups - 1, 4);
  __syncthreads();
  return *s_flag != 0;
}

// ---- pass A alone, bulk modes (first linearisation of a solve, re-linearisation after a rejected step or a missed speculation)
__global__ __launch_bounds__(256) void ba_linearize_kernel(BaDev P, double radius, int first_pass, const double* __restrict__ ctl, BulkSel bs) {
  svo_latency_critical();
  apply_ctl(P, radius, ctl);
  if (!bulk_apply(P, radius, first_pass, bs)) return;
  extern __shared__ double lds[];  // payload1 image: S (n*n) | gred (n) | gc (n) | dU (n) | cost | gp2
  const int n = P.n;
  const int pay1 = n * n + 3 * n + 2;
  double* sS = lds;
  double* sGred = sS + n * n;
  double* sGc = sGred + n;
  double* sDU = sGc + n;
  for (int i = threadIdx.x; i < pay1; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double lcost = 0.0, lgp2 = 0.0;
  const int wpb = blockDim.x >> 6;
  for (int chunk = blockIdx.x * wpb + wave; chunk < P.C; chunk += gridDim.x * wpb) {
    const ObsRec R = load_obs(P, chunk, lane, P.points);
    LinPre q;
    linearize_prefix(P, R, P.poses, q, lcost);
    linearize_suffix_bulk(P, R, q, radius, first_pass, sS, sGred, sGc, sDU, lgp2);
  }
  // block totals of cost / gp2
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { lcost += __shfl_xor(lcost, off); lgp2 += __shfl_xor(lgp2, off); }
  if (lane == 0) { atomicAdd(&lds[pay1 - 2], lcost); atomicAdd(&lds[pay1 - 1], lgp2); }
  __syncthreads();
  for (int i = threadIdx.x; i < pay1; i += blockDim.x) {
    const double v = lds[i];
    if (v != 0.0) atomicAdd(&P.pay1[i], v);
  }
}

// ---- deterministic mode, host-driven: one wavefront per workgroup, workgroup = group of chunks (window problems: one chunk)
__device__ __forceinline__ PartSink make_sink(const BaDev& P, int g, int first, double* pst) {
  return PartSink{P.part1, P.part2, pst, P.Epad, P.E, P.NG, g, first, P.pay_parity, P.pay_tag, nullptr};
}
__device__ __forceinline__ ChunkTab global_tab(const BaDev& P, int chunk) { return ChunkTab{P.tab + P.tab_off[chunk], (P.K - 1) * P.K / 2, P.K - 1}; }
// The chunk's table for the owner phases of a host-driven kernel: copied into the workgroup's LDS first when the host made room
// (P.tab_lds_words) — the owners' walk is a chain of dependent reads (entry -> rows), ~100 cycles per link from LDS against
// ~500+ from L2 (the single-stream pass A of round 4's first version: 20.7 us against 12.9 in round 3).  Called by the whole
// workgroup; the copy is visible after the next workgroup barrier.
__device__ __forceinline__ ChunkTab stage_chunk_tab(const BaDev& P, int chunk, const WgLds& L) {
  if (!P.tab_lds_words) return global_tab(P, chunk);
  const uint32_t o0 = P.tab_off[chunk], o1 = P.tab_off[chunk + 1];
  const int words32 = (int)((o1 - o0 + 1) >> 1);  // tables start on even u16 offsets
  const uint32_t* src = reinterpret_cast<const uint32_t*>(P.tab + o0);
  uint32_t* d32 = reinterpret_cast<uint32_t*>(L.tab);
  for (int i = threadIdx.x; i < words32; i += blockDim.x) d32[i] = src[i];
  return ChunkTab{L.tab, (P.K - 1) * P.K / 2, P.K - 1};
}

template <int NT>
__global__ __launch_bounds__(NT) void ba_linearize_det_kernel(BaDev P, double radius, int first_pass, const double* __restrict__ ctl) {
  svo_latency_critical();
  apply_ctl(P, radius, ctl);
  extern __shared__ double lds[];  // wg_lds_doubles(E)
  const WgLds L = wg_lds(lds, P.E);
  const int lane = threadIdx.x & 63, g = blockIdx.x;
  const int c0 = g * P.CPW, c1 = min(P.C, c0 + P.CPW);  // one chunk per workgroup (the declared groups of G chunks are then formed where the partials are summed) unless the partial store would get too large: then a workgroup runs its group's G chunks one after the other
  for (int chunk = c0; chunk < c1; ++chunk) {
    ObsRec R{false, 0, 0, 0, lane, 0, D3{0, 0, 1}, 0.0, 0.0};
    if (threadIdx.x < 64) R = load_obs(P, chunk, lane, P.points);
    if (chunk != c0) __syncthreads();  // the previous chunk's owners are done with the table
    const ChunkTab T = stage_chunk_tab(P, chunk, L);  // visible behind the barrier in front of the owner phases
    linearize_chunk_wg1<NT>(P, R, P.poses, radius, first_pass, T, L, make_sink(P, g, chunk == c0, L.pst));
    stores_acknowledged();  // the group's running sums are read back by the next chunk
  }
}

// ---- single rank, deterministic mode, chained iteration: pass A that FIRST forms payload2 from pass B's group sums (every
// workgroup redundantly, in the declared order: a few KB of L2 reads instead of a launch boundary), takes Ceres' accept /
// radius decision and linearises for it.  Workgroup 0 also delivers payload2 and the decision.
template <int NT>
__global__ __launch_bounds__(NT) void ba_decide_linearize_kernel(BaDev P, LmCtl ctl) {
  svo_latency_critical();
  extern __shared__ double lds[];  // wg_lds_doubles(E), the staging rows double as scratch of the sums
  __shared__ double sOut[4];
  __shared__ int sFlag;
  const WgLds L = wg_lds(lds, P.E);
  double* rec = L.rec;
  const int lane = threadIdx.x & 63, g = blockIdx.x;
  const int c0 = g * P.CPW, c1 = min(P.C, c0 + P.CPW);  // one chunk per workgroup (the declared groups of G chunks are then formed where the partials are summed) unless the partial store would get too large: then a workgroup runs its group's G chunks one after the other
  // the first chunk's records are requested before the sums (both candidate landing points: the decision is not known yet)
  ObsRec Rc = load_obs(P, c0 < P.C ? c0 : 0, lane, P.points);
  D3 pc = Rc.p;
  if (Rc.active) pc = D3{P.cand_points[3 * Rc.j], P.cand_points[3 * Rc.j + 1], P.cand_points[3 * Rc.j + 2]};
  ChunkTab T = stage_chunk_tab(P, c0 < P.C ? c0 : 0, L);  // on its way to LDS while the sums are collected
  (void)sum_pay2<true>(P, P.pay_parity, P.pay_tag, rec, sOut, &sFlag);  // pass B's launch is complete: the tags are there
  const SvoLmDecision dec = svo_lm_decide(ctl.cost, ctl.mcc, ctl.radius, ctl.decrease_factor, sOut[0], sOut[1]);
  if (blockIdx.x == 0 && threadIdx.x < 6)
    pay_store(&P.pay2_out[threadIdx.x], threadIdx.x < 4 ? sOut[threadIdx.x] : (threadIdx.x == 4 ? (double)dec.accept : dec.next_radius));
  const double* points_ = dec.accept ? P.cand_points : P.points;
  const double* poses_ = dec.accept ? P.cand_poses : P.poses;
  if (dec.accept) Rc.p = pc;
  for (int chunk = c0; chunk < c1; ++chunk) {
    if (chunk != c0) { Rc = load_obs(P, chunk, lane, points_); __syncthreads(); T = stage_chunk_tab(P, chunk, L); }
    linearize_chunk_wg1<NT>(P, Rc, poses_, dec.next_radius, 0, T, L, make_sink(P, g, chunk == c0, L.pst));
    stores_acknowledged();
  }
}

// ---- pass B alone (bulk modes; the speculative pass A follows as its own launch on the MFMA / LDS-atomic kernel)
__global__ __launch_bounds__(256) void ba_backsub_kernel(BaDev P, double radius, BulkSel bs) {
  if (bs.bctl) {  // device-side step control: the slot has a step to evaluate, or nothing to do
    if (bs.bctl[BC_DONE] != 0.0 || (int)bs.bctl[BC_MODE] != BCM_STEP) return;
    bulk_select(P, bs);
    radius = bs.bctl[BC_RADIUS];
  }
  __shared__ double sStep[STEP_LDS_DOUBLES];
  __shared__ double sAcc[4];
  if (threadIdx.x < 4) sAcc[threadIdx.x] = 0.0;
  stage_step(P, sStep);
  const double* dc_ = sStep;
  const double* cand_poses_ = sStep + (P.n > 0 ? P.n : 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a_cost = 0, a_mc = 0, a_dp2 = 0, a_p2 = 0;
  const int wpb = blockDim.x >> 6;
  for (int chunk = blockIdx.x * wpb + wave; chunk < P.C; chunk += gridDim.x * wpb) {
    const ObsRec R = load_obs(P, chunk, lane, P.points);
    D3 cand;
    backsub_chunk(P, R, P.poses, cand_poses_, dc_, P.cand_points, radius, cand, a_cost, a_mc, a_dp2, a_p2);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a_cost += __shfl_xor(a_cost, off); a_mc += __shfl_xor(a_mc, off);
    a_dp2 += __shfl_xor(a_dp2, off); a_p2 += __shfl_xor(a_p2, off);
  }
  if (lane == 0) { atomicAdd(&sAcc[0], a_cost); atomicAdd(&sAcc[1], a_mc); atomicAdd(&sAcc[2], a_dp2); atomicAdd(&sAcc[3], a_p2); }
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(&P.pay2[threadIdx.x], sAcc[threadIdx.x]);
}

// ---- deterministic mode, one LM iteration in one sweep: pass B at the current point, then (spec_radius > 0) pass A at
// the candidate it just formed, with the radius an accepted step will have.  One wave per workgroup (spreads the
// chunks over the CUs); the candidate landmark stays in registers between the passes.
template <int NT>
__global__ __launch_bounds__(NT) void ba_step_kernel(BaDev P, double radius, double spec_radius) {
  svo_latency_critical();
  __shared__ double sStep[STEP_LDS_DOUBLES];
  extern __shared__ double lds[];  // wg_lds_doubles(E)
  const WgLds L = wg_lds(lds, P.E);
  double* lms = L.rec;  // pass B's landmark scalars live in the idle staging rows
  const int lane = threadIdx.x & 63, g = blockIdx.x;
  const int c0 = g * P.CPW, c1 = min(P.C, c0 + P.CPW);  // one chunk per workgroup (the declared groups of G chunks are then formed where the partials are summed) unless the partial store would get too large: then a workgroup runs its group's G chunks one after the other
  // the observation records are requested BEFORE the step is staged: the HBM round trip and the PCIe round trip overlap
  ObsRec R = load_obs(P, c0 < P.C ? c0 : 0, lane, P.points);
  stage_step(P, sStep);
  const double* dc_ = sStep;
  const double* cand_poses_ = sStep + (P.n > 0 ? P.n : 1);
  double unused0 = 0, unused1 = 0, unused2 = 0, unused3 = 0;
  for (int chunk = c0; chunk < c1; ++chunk) {
    if (chunk != c0) R = load_obs(P, chunk, lane, P.points);
    const PartSink sink = make_sink(P, g, chunk == c0, L.pst);
    D3 cand = R.p;
    if (threadIdx.x < 64) backsub_chunk(P, R, P.poses, cand_poses_, dc_, P.cand_points, radius, cand, unused0, unused1, unused2, unused3, nullptr, lms, &sink);
    ChunkTab T = global_tab(P, chunk);
    if (spec_radius > 0) {
      __syncthreads();  // pass B kept its landmark scalars in the staging rows (and the previous chunk's owners are done with the table)
      T = stage_chunk_tab(P, chunk, L);
      R.p = cand;
      linearize_chunk_wg1<NT>(P, R, cand_poses_, spec_radius, 0, T, L, sink);
    }
    stores_acknowledged();
  }
}

// sharded runs: the decision is taken from the ALL-REDUCED payload2 (device buffer [payload2 | payload1])
__global__ __launch_bounds__(64) void ba_decide_kernel(LmCtl ctl, double* paybuf, double* ctl_dev) {
  if (threadIdx.x == 0 && blockIdx.x == 0) decide_device(ctl, paybuf[0], paybuf[1], ctl_dev, paybuf);
}

// ---- the step control of a bulk / sharded solve on the device: ONE workgroup behind the all-reduce of payload1 -------------
// What host/lm.cpp does between two passes, on the summed payloads in device memory: consume the step (Ceres' acceptance,
// radius update, tolerances: the SAME functions — host/lm_decide.h, host/lm_math.h), test the gradient, the iteration / time /
// radius limits, build the scaled damped reduced camera system, factor it (csrc/lm_device.h: the declared arithmetic of
// host/linalg.cpp, n <= 128 in LDS), form the pose step and the candidate poses, and leave state + step for the next slot's
// kernels.  Also clears the payload buffer for the next slot's accumulation and publishes a status record (pinned host memory)
// that the host reads only to decide how far ahead it may keep enqueuing — never inside an iteration's critical path.
// Every rank runs it on identical sums, so every rank takes identical decisions (the reason the wall-clock cap is tested on the
// all-reduced MEAN of the ranks' clocks: the payload's tail carries [elapsed seconds, 1.0] per rank).
struct LmDevOpt { int max_iterations; double function_tolerance, gradient_tolerance, parameter_tolerance, initial_radius, max_time_s; };
constexpr int BS_HEAD = 24;                      // status record: [seq | done | iterations | successful | termination | initial cost | cost | sel | stand-alone pass-A slots | steps | next linearisation used | mode | radius | elapsed s | - | -] then 7 K poses
constexpr int BS_DOUBLES = BS_HEAD + 7 * 64;
struct BulkCtlArgs {
  int n, K, slot, ring;
  double* pay;      // device [payload2 (PAY2_SLOTS) | payload1 (n n + 3 n + 2) | elapsed, ranks]
  double* bctl;     // device LM state (BC_*)
  double* sc;       // device: Jacobi scales of the pose columns, fixed by the first linearisation
  double* step;     // device [dc (max(n, 1)) | candidate poses (7 K)]: what ba_backsub_kernel stages
  double* pos[2];   // the two pose buffers (current = BC_SEL)
  double* status;   // pinned ring of BS_DOUBLES records
  LmDevOpt opt;
};
static inline size_t ba_bulk_ctl_lds_doubles(int n, int K) {
  const size_t nn = n > 0 ? n : 1;
  return (size_t)n * n + (3 * (size_t)n + 4) + 4 * nn + (nn + 14 * (size_t)K) + (7 * nn + 8) + 14 * (size_t)K;
}

__global__ __launch_bounds__(512) void ba_bulk_control_kernel(BulkCtlArgs a) {
  extern __shared__ double lds[];
  __shared__ double st[BC_WORDS];
  __shared__ int sAct;
  const int tid = threadIdx.x, nt = blockDim.x, n = a.n, K = a.K, nn = n > 0 ? n : 1;
  const LmDevOpt& opt = a.opt;
  for (int i = tid; i < BC_WORDS; i += nt) st[i] = a.bctl[i];
  __syncthreads();
  double* rec = a.status + (size_t)(a.slot % a.ring) * BS_DOUBLES;
  double* cS = lds;                         // n x n: S (lower triangle used) -> the scaled damped system -> L
  double* cV = cS + (size_t)n * n;          // g_red (n) | g_c (n) | diag U (n) | cost | sum g_p^2 | sum of the ranks' elapsed seconds | ranks
  double* cDf = cV + 3 * n + 4;
  double* cRhs = cDf + nn;
  double* cSc = cRhs + nn;
  double* cDc = cSc + nn;
  double* cTerm = cDc + nn;                 // nn + 14 K: per-element terms of the sequential sums
  double* cCol = cTerm + nn + 14 * K;       // 7 nn + 8: the solver's panel columns, reciprocal pivots, panel part of y
  double* cPose = cCol + 7 * nn + 8;        // 7 K current poses
  double* cCand = cPose + 7 * K;            // 7 K candidate poses
  enum { ACT_RETURN = 0, ACT_GRAD, ACT_LOOPTOP, ACT_SOLVE, ACT_FINISH };
  const bool was_done = st[BC_DONE] != 0.0;
  double elapsed_now = 0.0;
  // phase stamps of thread 0 (100 MHz ticks since kernel entry): loaded | step consumed | system built | factored + solved | step written
  long long tk0 = tid == 0 ? (long long)wall_clock64() : 0, tk[5] = {0, 0, 0, 0, 0};
  auto stamp = [&](int i) { if (tid == 0) tk[i] = (long long)wall_clock64() - tk0; };
  long long tsolve[3] = {0, 0, 0};
  if (!was_done) {
    const int mode = (int)st[BC_MODE], sel = (int)st[BC_SEL] & 1;
    const double* P1 = a.pay + PAY2_SLOTS;
    for (int i = tid; i < 3 * n + 4; i += nt) cV[i] = P1[(size_t)n * n + i];
    // every unordered pose pair was accumulated once (upper block), diagonal pose blocks in full: the lower triangle as
    // ba_payload1_out forms it (src[ij] + src[ji] off the diagonal blocks)
    // (The lower off-diagonal blocks of the payload are +0.0, so "src[ij] + src[ji]" is the upper element + 0.0.)  Read as the
    // payload lies in memory — wavefront = row, lanes = consecutive columns, eight rows in flight — and placed transposed.
    {
      const int wv = tid >> 6, ln = tid & 63, nwv = nt >> 6;
      for (int r0 = wv; r0 < n; r0 += 8 * nwv) {
        double v[8][2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + u * nwv, cb = 6 * (r / 6);  // first column of the row's own pose block
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = cb + ln + 64 * h;
            v[u][h] = (r < n && c < n) ? P1[(size_t)r * n + c] : 0.0;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = r0 + u * nwv, cb = 6 * (r / 6);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = cb + ln + 64 * h;
            if (r >= n || c >= n) continue;
            if (c < cb + 6) { if (c <= r) cS[r * n + c] = v[u][h]; }  // the row's diagonal pose block: its own lower element
            else cS[c * n + r] = v[u][h] + 0.0;                        // an upper block: the mirror's lower element
          }
        }
      }
    }
    for (int i = tid; i < 7 * K; i += nt) { cPose[i] = a.pos[sel][i]; cCand[i] = a.step[nn + i]; }
    if (mode != BCM_FIRST) for (int q = tid; q < n; q += nt) cSc[q] = a.sc[q];
    __shared__ double sPay2[4];
    if (tid < 4) sPay2[tid] = a.pay[tid];
    __syncthreads();
    // the payload buffer is consumed: clear it for the next slot's accumulation; its tail takes this rank's clock
    for (int i = tid; i < PAY2_SLOTS + n * n + 3 * n + 4; i += nt) a.pay[i] = 0.0;
    if (tid == 0) {
      const double now = (double)(long long)wall_clock64();
      if (mode == BCM_FIRST) st[BC_T0] = now;
      elapsed_now = 1e-8 * (now - st[BC_T0]);
    }
    __syncthreads();
    if (tid == 0) { a.pay[PAY2_SLOTS + (size_t)n * n + 3 * n + 2] = elapsed_now; a.pay[PAY2_SLOTS + (size_t)n * n + 3 * n + 3] = 1.0; }
    const double ranks = cV[3 * n + 3], mean_elapsed = ranks > 0 ? cV[3 * n + 2] / ranks : 0.0;
    stamp(0);
    int act = ACT_RETURN;
    if (mode == BCM_FIRST) {
      for (int q = tid; q < n; q += nt) { const double v = 1.0 / (1.0 + sqrt(cV[2 * n + q])); cSc[q] = v; a.sc[q] = v; }
      if (tid == 0) { st[BC_COST] = cV[3 * n]; st[BC_INITIAL_COST] = cV[3 * n]; st[BC_LIN_CALLS] = 1; }
      act = ACT_GRAD;
    } else if (mode == BCM_RELIN) {
      if (tid == 0) st[BC_NEED_LIN] = 0;
      act = ACT_SOLVE;
    } else {  // BCM_STEP: host/lm.cpp behind ops->step; the pass A that followed took svo_lm_decide's outcome (bulk_apply)
      for (int i = tid; i < 7 * K; i += nt) {
        const double dd = cCand[i] - cPose[i];
        cTerm[i] = dd * dd;
        cTerm[7 * K + i] = cPose[i] * cPose[i];
      }
      __syncthreads();
      if (tid == 0) {
        const double cost = st[BC_COST], mcc = st[BC_MCC], radius = st[BC_RADIUS], df = st[BC_DF];
        const double cost_new = sPay2[0], model_change = mcc + sPay2[1];
        const double step2 = lds_seq_sum(sPay2[2], cTerm + 7, 7 * K - 7, 1), x2 = lds_seq_sum(sPay2[3], cTerm + 7 * K + 7, 7 * K - 7, 1);  // host/lm.cpp's order, eight loads in flight
        const SvoLmDecision dec = svo_lm_decide(cost, mcc, radius, df, cost_new, sPay2[1]);  // what pass A linearised for
        int a_ = ACT_LOOPTOP, accepted = 0;
        if (!(model_change > 0)) {  // invalid step: pass A ran at the current point with radius / df — exactly what is needed
          st[BC_RADIUS] = radius / df; st[BC_DF] = df * 2; st[BC_NEED_LIN] = 0; st[BC_NEXT_USED] += 1;
        } else if (sqrt(step2) <= opt.parameter_tolerance * (sqrt(x2) + opt.parameter_tolerance)) {
          st[BC_TERM] = 0; a_ = ACT_FINISH;
        } else if (fabs(cost - cost_new) <= opt.function_tolerance * cost) {  // Ceres returns before the step is taken
          st[BC_TERM] = 0; a_ = ACT_FINISH;
        } else if (dec.accept) {
          accepted = 1;
          st[BC_SEL] = (double)(((int)st[BC_SEL] & 1) ^ 1);  // the candidate buffers become the current ones (op_accept)
          st[BC_COST] = cost_new; st[BC_SUCC] += 1; st[BC_RADIUS] = dec.next_radius; st[BC_DF] = 2.0; st[BC_NEXT_USED] += 1;
          a_ = ACT_GRAD;
        } else {
          st[BC_RADIUS] = dec.next_radius; st[BC_DF] = df * 2; st[BC_NEED_LIN] = 0; st[BC_NEXT_USED] += 1;
        }
        sAct = a_ | (accepted << 8);
      }
      __syncthreads();
      act = sAct & 0xFF;
      if (sAct >> 8) for (int i = tid; i < 7 * K; i += nt) cPose[i] = cCand[i];
      __syncthreads();
    }
    __syncthreads();
    stamp(1);
    for (;;) {  // block-uniform: every transition is decided by thread 0 and read between two barriers
      if (act == ACT_GRAD) {
        for (int q = tid; q < n; q += nt) cTerm[q] = cV[n + q] * cV[n + q];
        __syncthreads();
        if (tid == 0) {  // sqrt(sum g_p^2 + sum g_c^2), sequentially as host/lm.cpp's gradient_norm
          const double g2 = lds_seq_sum(cV[3 * n + 1], cTerm, n, 1);
          if (sqrt(g2) <= opt.gradient_tolerance) { st[BC_TERM] = 0; sAct = ACT_FINISH; } else sAct = ACT_LOOPTOP;
        }
        __syncthreads();
        act = sAct;
        __syncthreads();
      }
      if (act == ACT_LOOPTOP) {
        if (tid == 0) {
          int a_ = ACT_SOLVE;
          if (st[BC_ITER] >= (double)opt.max_iterations) { st[BC_TERM] = 1; a_ = ACT_FINISH; }
          else if (opt.max_time_s > 0 && mean_elapsed >= opt.max_time_s) { st[BC_TERM] = 1; a_ = ACT_FINISH; }  // src/bundle_adjuster.cpp:11
          else if (st[BC_RADIUS] <= LM_MIN_RADIUS_BULK) { st[BC_TERM] = 0; a_ = ACT_FINISH; }
          else {
            st[BC_ITER] += 1;
            if (st[BC_NEED_LIN] != 0.0) { st[BC_LIN_CALLS] += 1; st[BC_MODE] = (double)BCM_RELIN; a_ = ACT_RETURN; }
          }
          sAct = a_;
        }
        __syncthreads();
        act = sAct;
        __syncthreads();
      }
      if (act == ACT_FINISH) { if (tid == 0) st[BC_DONE] = 1.0; break; }
      if (act == ACT_RETURN) break;
      // ACT_SOLVE
      const double radius = st[BC_RADIUS];
      for (int q = tid; q < n; q += nt) {
        const double sq = cSc[q];
        cDf[q] = fmin(fmax(cV[2 * n + q] * sq * sq, MIN_DIAG), MAX_DIAG) / radius;
        cRhs[q] = -(cV[q] + cV[n + q]) * sq;
      }
      __syncthreads();
      for (int r = tid >> 6; r < n; r += nt >> 6) {  // wavefront = row, lanes = columns of the lower triangle
        const double sr = cSc[r];
        for (int c = tid & 63; c <= r; c += 64) {
          double w = cS[r * n + c] * sr * cSc[c];
          if (r == c) w += cDf[r];
          cS[r * n + c] = w;
        }
      }
      __syncthreads();
      stamp(2);
      const bool ok = n == 0 || svo_dev_spd_solve_fast(cS, cRhs, n, cCol, tid == 0 ? tsolve : nullptr);  // (deterministic, not the host's bits: see csrc/lm_device.h)
      stamp(3);
      if (ok) {
        for (int q = tid; q < n; q += nt) {
          const double rq = cRhs[q], sq = cSc[q];
          cDc[q] = rq * sq;
          cTerm[q] = 0.5 * rq * (cDf[q] * rq - cV[n + q] * sq);
        }
        __syncthreads();
        if (tid == 0) {
          st[BC_MCC] = lds_seq_sum(0.0, cTerm, n, 1); st[BC_STEP_CALLS] += 1; st[BC_MODE] = (double)BCM_STEP;
        }
        if (tid < K) {
          if (tid == 0) for (int q = 0; q < 7; ++q) cCand[q] = cPose[q];
          else svo_plus_pose(&cPose[7 * tid], &cDc[6 * (tid - 1)], &cCand[7 * tid]);
        }
        __syncthreads();
        const int nsel = (int)st[BC_SEL] & 1;
        for (int q = tid; q < nn; q += nt) a.step[q] = q < n ? cDc[q] : 0.0;
        for (int i = tid; i < 7 * K; i += nt) { a.step[nn + i] = cCand[i]; a.pos[nsel ^ 1][i] = cCand[i]; }
        break;
      }
      // not positive definite: invalid step, linearise again with the reduced radius
      if (tid == 0) { st[BC_RADIUS] = st[BC_RADIUS] / st[BC_DF]; st[BC_DF] = st[BC_DF] * 2; st[BC_NEED_LIN] = 1; }
      __syncthreads();
      act = ACT_LOOPTOP;
    }
    __syncthreads();
    for (int i = tid; i < BC_WORDS; i += nt) a.bctl[i] = st[i];
    // the accepted poses are the current buffer's content from now on
    { const int nsel = (int)st[BC_SEL] & 1; for (int i = tid; i < 7 * K; i += nt) a.pos[nsel][i] = cPose[i]; }
  } else {
    const int sel = (int)st[BC_SEL] & 1;
    for (int i = tid; i < 7 * K; i += nt) cPose[i] = a.pos[sel][i];
    __syncthreads();
  }
  // status record: poses and head (written through), acknowledged, then the sequence word
  if (st[BC_DONE] != 0.0) for (int i = tid; i < 7 * K; i += nt) pay_store(&rec[BS_HEAD + i], cPose[i]);
  if (tid == 0) {
    pay_store(&rec[1], st[BC_DONE]); pay_store(&rec[2], st[BC_ITER]); pay_store(&rec[3], st[BC_SUCC]); pay_store(&rec[4], st[BC_TERM]);
    pay_store(&rec[5], st[BC_INITIAL_COST]); pay_store(&rec[6], st[BC_COST]); pay_store(&rec[7], st[BC_SEL]); pay_store(&rec[8], st[BC_LIN_CALLS]);
    pay_store(&rec[9], st[BC_STEP_CALLS]); pay_store(&rec[10], st[BC_NEXT_USED]); pay_store(&rec[11], st[BC_MODE]); pay_store(&rec[12], st[BC_RADIUS]);
    pay_store(&rec[13], elapsed_now);
    stamp(4);
    for (int i = 0; i < 5; ++i) pay_store(&rec[14 + i], (double)tk[i]);
    for (int i = 0; i < 3; ++i) pay_store(&rec[19 + i], (double)tsolve[i]);
  }
  stores_acknowledged();
  __syncthreads();
  if (tid == 0) pay_store(&rec[0], (double)(a.slot + 1));
}

// Publishing without cache maintenance.  A compiler fence at agent or system scope is `buffer_wbl2` + `buffer_inv`: it
// writes back and INVALIDATES the whole L2 of the workgroup's XCD — harmless alone on the GPU, but with several stereo
// streams sharing the chip every workgroup of every adjuster kept emptying the L2 under the other streams' kernels.
// Nothing here needs it: the payload goes to fine-grained pinned host memory (hipHostMallocCoherent: uncached on the GPU
// side, every store is written through), so `s_waitcnt vmcnt(0)` alone says "my payload stores are acknowledged"; the
// arrival counter and the completion word are relaxed atomics (performed at the device's / system's coherence point).
// The last workgroup to arrive has, transitively, seen every payload store acknowledged before it writes the word.
__device__ __forceinline__ void reduce_publish(const BaDev& P) {
  if (!P.flag) return;
  stores_acknowledged();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(P.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == P.arrive_target) __hip_atomic_store(P.flag, P.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Level 2 of the declared order, host-driven form: workgroup b < nb1 sums RED_EPB wire elements -> pay1_out (pinned host
// memory, or the device buffer an all-reduce follows on); one more workgroup sums payload2 (and, single rank with a chained
// step, takes the decision for the pass-A launch queued behind).  The last workgroup to arrive publishes the completion word.
constexpr int RED_LDS_DOUBLES = 4096;
// wire elements per workgroup of the reduce launch: all partials of its elements in LDS at once (64 elements up to 64 chunks, fewer beyond)
__host__ __device__ inline int ba_reduce_epb(int NG) { const int e = RED_LDS_DOUBLES / (NG > 0 ? NG : 1); return e > 64 ? 64 : (e < 1 ? 1 : e); }
__host__ __device__ inline int ba_reduce_blocks(int E, int NG) { const int epb = ba_reduce_epb(NG); return (E + epb - 1) / epb; }

__global__ __launch_bounds__(256) void ba_reduce_kernel(BaDev P, int with_pay1, int with_pay2, LmCtl ctl) {
  svo_latency_critical();
  __shared__ double sm[RED_LDS_DOUBLES];
  __shared__ double sOut[4];
  __shared__ int sFlag;
  const int tid = threadIdx.x, b = blockIdx.x;
  const int nb1 = with_pay1 ? ba_reduce_blocks(P.E, P.NG) : 0, epb = ba_reduce_epb(P.NG);
  if (b < nb1) {
    double* out = P.pay1_out;
    (void)reduce_elements<true>(P, b * epb, min(P.E, (b + 1) * epb), P.pay_tag, sm, RED_LDS_DOUBLES, &sFlag, [out](int e, double v) { pay_store(&out[e], v); });
  } else if (with_pay2) {
    (void)sum_pay2<true>(P, P.pay_parity, P.pay_tag, sm, sOut, &sFlag);
    if (tid < 4) pay_store(&P.pay2_out[tid], sOut[tid]);
    if (ctl.chain) {  // single rank: these ARE the global sums; decide here, pass A is queued right behind this launch
      if (tid == 0) decide_device(ctl, sOut[0], sOut[1], P.ctl_dev, P.pay2_out);
    }
  }
  reduce_publish(P);
}

__device__ __forceinline__ bool wait_until(const unsigned* word, unsigned target, bool monotone) {
  long long t0 = 0;
  for (unsigned spins = 0;; ++spins) {
    const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (monotone ? (int)(v - target) >= 0 : v == target) return true;
    __builtin_amdgcn_s_sleep(2);
    if ((spins & 255u) == 255u) {
      const long long tn = (long long)wall_clock64();
      if (!t0) t0 = tn;
      else if (tn - t0 > LM_WAIT_TICKS) return false;  // something is badly wrong; never hang the GPU
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The WHOLE solve as one launch: ba_lm_kernel.  The host-driven path has the host in every LM iteration (Cholesky of the
// reduced camera system + Ceres' step control: 13 us of PCIe turnaround for 1.3-4.4 us of arithmetic, and one host thread
// per stereo stream).  Here the step control runs on the device, REPLICATED: every workgroup reads the summed payload and
// runs host/lm.cpp's step control itself — the SAME functions (host/lm_math.h, host/lm_decide.h; the Cholesky of
// csrc/lm_device.h applies host/linalg.cpp's operations in its order), hence the same bits and the same decision in every
// workgroup — and goes straight on to its share of the next pass.  Workgroups hand over nothing but TAGGED GRANULES:
//   pass B -> every wavefront posts its chunk's four sums -> EVERY workgroup collects all of them, adds them in the
//   declared order and takes Ceres' decision itself (the radius-free part of pass A runs before the collection) ->
//   pass A -> every wavefront posts its chunk's E partials -> every workgroup sums its slice of the elements over the
//   chunks (level 2) and posts the totals -> every workgroup collects the E totals and runs the step control.
// No arrival counter, no controller or reducer workgroup, no post / poll hop: a reader simply reads a granule again until
// it carries the tag of the command it is waiting for.  Three hand-overs per chained iteration, two per same-sweep
// iteration (after a saturated step pass A follows pass B directly with the predicted radius).
// The host launches, and later finds poses, landmarks, summary and ONE completion word in pinned memory.  Bit-identical to
// the host-driven loop (tests/test_ba.py, tests/test_pipeline.py): every deciding operation is the same IEEE operation
// wherever it runs.
// The wall-clock cap of src/bundle_adjuster.cpp:11: replicated controllers cannot each read a clock and agree, so
// workgroup 0 posts the time elapsed since its first pass with every reduction it takes part in (one more granule) and
// every controller tests THAT value at the top of the loop, where host/lm.cpp tests its clock.
struct LmDevArgs {
  unsigned* cnt;            // device counter block (LMC_*)
  const void* arena_src;    // pinned problem image to read in place (null: already on the device)
  void* arena_dst;
  size_t arena_bytes;
  double* points_a;         // the two landmark buffers (current / candidate, swapped by every accepted step)
  double* points_b;
  double* export_points;    // pinned: the solved landmarks, by landmark index (null: none)
  float4* store;            // device-resident landmark store (null: none): entry (key & store_mask) = {x, y, z, key} of every landmark of the solve, key = P.lm_key[j]
  unsigned store_mask;
  double* dev_res;          // device granules: the E wire totals of the running iteration | elapsed seconds
  double* host_result;      // pinned: [LMR_* summary | poses 7 K]
  int* host_flag;           // pinned completion word of the solve
  int host_seq;
  unsigned base_arrive;     // the delivery counter runs on from solve to solve (monotone, wrap-safe compares)
  int tab_words;            // u16 words of LDS per wavefront for its chunk table (a multiple of 4)
  LmDevOpt opt;
  unsigned* dbg;            // per workgroup 16 words: its last command (diagnostics of a solve that gave up; null: none)
  int test_giveup;          // test hook (SVO_BA_TEST_GIVEUP): the wide launch reports "gave up" at once, as if a bounded wait had run out
};
enum { LMC_ARRIVE = 0, LMC_WORDS = 16 };
enum { LMR_ITERATIONS = 0, LMR_SUCCESSFUL, LMR_TERMINATION, LMR_INITIAL_COST, LMR_FINAL_COST, LMR_LINEARIZE_CALLS, LMR_STEP_CALLS, LMR_SEL,
       LMR_T_WAIT, LMR_T_CTL, LMR_T_BODY, LMR_T_TOTAL, LMR_SAME_SWEEP, LMR_NEXT_USED, LMR_C_ARRIVE, LMR_TP0, LMR_DOUBLES = LMR_TP0 + 14 };
enum { LMS_START = 0, LMS_FIRST, LMS_RELIN, LMS_STEP, LMS_ACCEPT_RELIN, LMS_DELIVER };
enum { LMOP_EXIT = 0, LMOP_LINEARIZE, LMOP_ITERATE, LMOP_DELIVER, LMOP_ABORT };

struct LmDevState {
  double radius, df, cost, initial_cost, mcc, spec;  // spec: radius of the same-sweep pass A of the step in flight (0: none)
  double elapsed;                                    // seconds since workgroup 0's first pass, as posted with the last reduction
  double pay2[6];                                    // payload2 (+ accept, next radius) of a chained step, formed by lm_iterate
  int iterations, successful, termination, need_linearize, state, sel, chain, first, saturated;
  unsigned arrive_total;
  int lin_calls, step_calls, same_sweeps, next_used;
  int go, act, use_next, accepted, relin, bad, flag;
  unsigned long long tag;   // of the command in flight: bit 62 | (solve sequence << 20) | command number
  unsigned op_count;
  long long t0, t_wait, t_ctl, t_body, t_mark;  // 100 MHz ticks: collecting the totals, step control, passes
  long long tp[14];  // (slot 11: the controller's running mark; 10, 12: see lm_iterate) finer split (workgroup 0): pass B, radius-free part of pass A, collecting payload2, rest of pass A, own slice of level 2 (incl. waiting for the partials), -, collecting the totals, system build, Cholesky, step tail
};
constexpr double LM_MIN_RADIUS = 1e-32, LM_MAX_RADIUS = 1e16;

// controller workspace (doubles) — it lives in the LDS of the staging rows (a pass and a controller turn never overlap):
// payload image, 4 vectors of n, term scratch, U staging
__host__ __device__ static inline size_t ba_lm_ctl_doubles(int n, int K) { return (size_t)n * n + 3 * (size_t)n + 2 + 4 * (size_t)(n > 0 ? n : 1) + (size_t)(n > 0 ? n : 1) + 14 * (size_t)K + 21 * (size_t)(K > 1 ? K - 1 : 1) + 8; }
// dynamic LDS of ba_lm_kernel (doubles): [staging rows + landmark scalars of CPW wavefronts | controller workspace] (union) |
// chunk tables | Jacobi scales of the pose columns | step block [dc | candidate poses | current poses]
constexpr int LM_CPW = 2;
__host__ __device__ static inline int ba_wire_elements(int K) { const int F = K - 1; return 18 * F * (F + 1) + 33 * F + 2; }
__host__ __device__ static inline size_t ba_lm_union_doubles(int n, int K) { const size_t a = (size_t)LM_CPW * (size_t)wg_lds_doubles(ba_wire_elements(K)), b = ba_lm_ctl_doubles(n, K); return a > b ? a : b; }
__host__ __device__ static inline size_t ba_lm_lds_doubles(int n, int K, int tab_words /* per wavefront, a multiple of 4, <= TAB_LDS_WORDS */) {
  // behind the union and the tables: Jacobi scales (nn) | spare (nn) | step block [dc (nn) | candidate poses (7 K) | current poses (7 K)] — the kernel's
  // carve-up (cSc, sStep).  (Until the end of round 4 this said 2 nn: the current poses' tail lay nn doubles beyond the allocation, inside the
  // allocation granule for the 5-keyframe window and outside it from n = 36 on — NaN poses, found when larger windows first took this kernel.)
  return ba_lm_union_doubles(n, K) + (size_t)LM_CPW * (size_t)tab_words / 4 + 3 * (size_t)(n > 0 ? n : 1) + 14 * (size_t)K;
}

// wire total e -> its place in the payload image (cP) or the U triangles (cU)
__device__ __forceinline__ void lm_place_total(int e, double v, double* cP, double* cU, int F, int n) {
  const int nU = F * (F + 1) / 2;
  if (e < 36 * nU) {
    const int d = e / 36, el = e - 36 * d;
    int ka = 0, rest = d;  // d = ka F - ka (ka - 1) / 2 + (kb - ka), row-major over ka <= kb
    while (rest >= F - ka) { rest -= F - ka; ++ka; }
    const int kb = ka + rest;
    cP[(6 * ka + el / 6) * n + 6 * kb + el % 6] = v;
  } else if (e < 36 * nU + 33 * F) {
    const int k = (e - 36 * nU) / 33, el = (e - 36 * nU) - 33 * k;
    if (el < 6) cP[n * n + n + 6 * k + el] = v;            // g_c
    else if (el < 12) cP[n * n + 6 * k + (el - 6)] = v;    // g_red (the -Y g_p part)
    else cU[21 * k + (el - 12)] = v;
  } else {
    cP[n * n + 3 * n + (e - 36 * nU - 33 * F)] = v;
  }
}
// diagonal pose blocks take U; diag U.  Ends with a barrier.
__device__ __forceinline__ void lm_fold_u(double* cP, const double* cU, int F, int n) {
  for (int i = threadIdx.x; i < 36 * F; i += (int)blockDim.x) {
    const int k = i / 36, a = (i % 36) / 6, b = i % 6;
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    const double uu = cU[21 * k + lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
    double* s = &cP[(6 * k + a) * n + 6 * k + b];
    *s = uu + *s;
    if (a == b) cP[n * n + 2 * n + 6 * k + a] = uu;
  }
  __syncthreads();
}

// The E wire totals (granules under `tag`) -> the payload image [S | g_red | g_c | diag U | cost | sum g_p^2] in LDS, assembled as
// the oracle does: S[(k,a),(k,b)] = U_k[min][max] + Schur_(k,k)[a][b]; upper blocks as summed; the step control reads the
// lower triangle through the mirror (see the system build).  cU: 21 F doubles of scratch.  false: a tag never showed up.
__device__ __forceinline__ bool lm_fetch_totals(double* cP, double* cU, const double* res, int F, int n, unsigned long long tag, double* elapsed, int n_blocks, int* s_flag) {
  const int nU = F * (F + 1) / 2, E = 36 * nU + 33 * F + 2;
  {
    // the last element of every workgroup's slice, and the clock granule workgroup 0 writes behind its slice
    const int per = (E + n_blocks - 1) / n_blocks, slices = (E + per - 1) / per;
    if (!wait_sentinels(res, per - 1, per, slices - 1, tag, s_flag)) return false;
    if (!wait_sentinels(res, E - 1, 1, 2, tag, s_flag)) return false;
  }
  bool good = true;
  for (int base = 0; base < E + 1; base += 8 * (int)blockDim.x) {
    int gi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * (int)blockDim.x + (int)threadIdx.x;
      gi[u] = i < E + 1 ? i : -1;
    }
    double v[8];
    unsigned ok = granule_load8(res, gi, tag, v);
    long long t0 = 0;
    for (unsigned spins = 0; ok != 0xFFu; ++spins) {
      __builtin_amdgcn_s_sleep(2);
      if ((spins & 255u) == 255u) {
        const long long tn = (long long)wall_clock64();
        if (!t0) t0 = tn;
        else if (tn - t0 > LM_WAIT_TICKS) break;
      }
      int again[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) again[u] = (ok >> u) & 1u ? -1 : gi[u];
      double w[8];
      const unsigned ok2 = granule_load8(res, again, tag, w);
#pragma unroll
      for (int u = 0; u < 8; ++u) if (!((ok >> u) & 1u) && ((ok2 >> u) & 1u)) { v[u] = w[u]; ok |= 1u << u; }
    }
    good = good && ok == 0xFFu;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = gi[u];
      if (e < 0) continue;
      if (e < E) lm_place_total(e, v[u], cP, cU, F, n);
      else *elapsed = v[u];
    }
  }
  __syncthreads();
  lm_fold_u(cP, cU, F, n);
  return good;
}

// Controller turn (every thread of every workgroup; identical inputs -> identical state everywhere): consume the
// finished pass, run host/lm.cpp's step control up to the next pass.  Leaves the next pass' step block
// [dc | candidate poses | current poses] in sStep and its parameters in cs; returns an LMOP_* code.
// LOCAL (ba_lm_compact_kernel: ONE workgroup runs the whole solve): the summed payloads are not collected from granules, they sit in
// the workgroup's LDS — `tot` (E wire totals) and `tot2` (pass B's four sums); cs.elapsed is set by the caller from its own clock.
template <bool LOCAL, bool GROUPED = false>
__device__ __attribute__((noinline)) int lm_controller(const BaDev& P, const LmDevArgs& a, LmDevState& cs, double* cl, double* cSc, double* sStep, double* sOut4,
                                                        const double* tot = nullptr, const double* tot2 = nullptr) {
  const int tid = threadIdx.x, nt = blockDim.x, n = P.n, K = P.K, nn = n > 0 ? n : 1;
  const int grid = LOCAL ? 1 : (P.C + LM_CPW - 1) / LM_CPW;  // workgroups of THIS solve (the launch may hold several solves)
  const int pay1 = n * n + 3 * n + 2;
  double* cP = cl;             // payload image [S | g_red | g_c | diag U | cost | sum g_p^2]; S becomes the scaled system, then L
  double* cDf = cP + pay1;
  double* cRhs = cDf + nn;
  double* cCol = cRhs + nn;    // the Cholesky's panel columns: 6 n + 1 doubles, over cTerm and cU (both dead while it runs; 2 n + 35 K - 21 >= 6 n + 1 = 36 K - 35)
  double* cTerm = cCol + nn;   // nn + 14 K: per-element terms of the sequential sums
  double* cU = cTerm + nn + 14 * K;  // 21 F: the U triangles on their way into the diagonal blocks
  double* cDc = sStep;         // the step block is built in place
  double* cCand = sStep + nn;
  double* cPose = cCand + 7 * K;
  const LmDevOpt& opt = a.opt;
  enum { ACT_NONE = 0, ACT_LOOPTOP, ACT_FINISH, ACT_ACCEPT_TAIL, ACT_SOLVE };
  int act = ACT_NONE;
  auto issue = [&](int op) {
    if (tid == 0) {
      if (op == LMOP_DELIVER) cs.arrive_total += (unsigned)grid;
      cs.tag = (1ull << 62) | ((unsigned long long)(unsigned)a.host_seq << 20) | (unsigned long long)(++cs.op_count & 0xFFFFFu);
      const long long tn = (long long)wall_clock64();
      cs.t_ctl += tn - cs.t_mark; cs.t_mark = tn;
    }
    __syncthreads();
    return op;
  };

  auto cstamp = [&](int slot) { if (a.dbg && tid == 0) { const long long tn = (long long)wall_clock64(); cs.tp[slot] += tn - cs.tp[11]; cs.tp[11] = tn; } };
  const int st = cs.state;
  __syncthreads();  // thread 0 rewrites cs.state further down in this very turn: every wave must have read it first
  if (st == LMS_START) {
    if (tid == 0) {
      cs.go = 1;
      cs.radius = opt.initial_radius; cs.df = 2.0; cs.cost = 0.0; cs.initial_cost = 0.0; cs.mcc = 0.0; cs.elapsed = 0.0;
      cs.t0 = cs.t_mark = (long long)wall_clock64();
      cs.iterations = 0; cs.successful = 0; cs.termination = 1; cs.need_linearize = 0; cs.sel = 0; cs.chain = 0; cs.spec = 0.0; cs.saturated = 0;
      cs.arrive_total = a.base_arrive;
      cs.lin_calls = 1; cs.step_calls = 0; cs.same_sweeps = 0; cs.next_used = 0; cs.bad = 0; cs.op_count = 0; cs.tag = 0;
      cs.t_wait = cs.t_ctl = cs.t_body = 0;
      for (int i = 0; i < 14; ++i) cs.tp[i] = 0;
      cs.state = LMS_FIRST; cs.first = 1;
    }
    __syncthreads();
    {
      const ptrdiff_t shift = a.arena_src ? reinterpret_cast<const char*>(a.arena_src) - reinterpret_cast<const char*>(a.arena_dst) : 0;
      for (int i = tid; i < 7 * K; i += nt) cPose[i] = sys_load(&P.poses[i], shift);
    }
    return issue(LMOP_LINEARIZE);
  }
  if (st == LMS_DELIVER) return LMOP_EXIT;

  if (tid == 0) {
    const long long w0 = (long long)wall_clock64();
    cs.t_body += w0 - cs.t_mark;
    cs.t_mark = w0;
    cs.first = 0;
  }
  __syncthreads();
  if (a.dbg && tid == 0) cs.tp[11] = (long long)wall_clock64();

  bool fetch = true;  // (a stand-alone pass A has finished)
  if (st == LMS_STEP) {
    // payload2: a chained step formed it (and the decision) inside the pass; otherwise collect the group sums now
    if (!cs.chain) {
      if constexpr (LOCAL) {
        if (tid < 4) cs.pay2[tid] = tot2[tid];
      } else {
        if (!sum_pay2<GROUPED>(P, (int)(cs.op_count & 1u), cs.tag, cP, sOut4, &cs.flag)) { if (tid == 0) cs.bad = 1; }
        if (tid < 4) cs.pay2[tid] = sOut4[tid];
      }
      __syncthreads();
    }
    for (int i = tid; i < 7 * K; i += nt) {  // terms of the pose part of |step|^2 and |x|^2 (host/lm.cpp after ops->step)
      const double dd = cCand[i] - cPose[i];
      cTerm[i] = dd * dd;
      cTerm[7 * K + i] = cPose[i] * cPose[i];
    }
    __syncthreads();
    if (tid == 0) {
      const double* cPay2 = cs.pay2;
      const double cost_new = cPay2[0], model_change = cs.mcc + cPay2[1];
      double step2 = cPay2[2], x2 = cPay2[3];
      for (int i = 7; i < 7 * K; ++i) { step2 += cTerm[i]; x2 += cTerm[7 * K + i]; }
      // the linearisation that rode along: chained (decision taken inside the launch) or same sweep (predicted)
      const bool have_next = cs.chain != 0 || cs.spec > 0;
      const bool next_at_cand = cs.spec > 0 ? true : cPay2[4] != 0.0;
      const double next_radius = cs.spec > 0 ? cs.spec : cPay2[5];
      int a_ = ACT_LOOPTOP, use = 0, relin = 0;
      if (!(model_change > 0)) {  // invalid step: no model decrease
        cs.radius /= cs.df; cs.df *= 2; cs.saturated = 0;
        use = have_next && next_radius > 0 && next_radius == cs.radius && !next_at_cand;
        cs.need_linearize = !use;
      } else if (sqrt(step2) <= opt.parameter_tolerance * (sqrt(x2) + opt.parameter_tolerance)) {
        cs.termination = 0; a_ = ACT_FINISH;
      } else if (fabs(cs.cost - cost_new) <= opt.function_tolerance * cs.cost) {  // Ceres returns before the step is taken
        cs.termination = 0; a_ = ACT_FINISH;
      } else {
        const SvoLmDecision dec = svo_lm_decide(cs.cost, cs.mcc, cs.radius, cs.df, cost_new, cPay2[1]);
        if (dec.accept) {
          cs.sel ^= 1;  // the candidate becomes the current point (op_accept)
          cs.cost = cost_new;
          ++cs.successful;
          cs.saturated = dec.next_radius == fmin(LM_MAX_RADIUS, cs.radius / (1.0 / 3.0));  // rho >= 0.9368: predict the same for the next step
          cs.radius = dec.next_radius;
          cs.df = 2.0;
          use = have_next && next_radius > 0 && next_radius == cs.radius && next_at_cand;
          relin = !use;
          a_ = ACT_ACCEPT_TAIL;
        } else {
          cs.radius = dec.next_radius; cs.df *= 2; cs.saturated = 0;
          use = have_next && next_radius > 0 && next_radius == cs.radius && !next_at_cand;
          cs.need_linearize = !use;
        }
      }
      cs.accepted = a_ == ACT_ACCEPT_TAIL;
      cs.next_used += use;
      cs.act = a_; cs.use_next = use; cs.relin = relin;
    }
    __syncthreads();
    if (cs.accepted) for (int i = tid; i < 7 * K; i += nt) cPose[i] = cCand[i];
    // the totals of the pass A that rode along are collected even when they are not used: every slice owner has posted
    // them under this command's tag, and the elapsed time travels with them
    fetch = cs.chain != 0 || cs.spec > 0;
  }
  // the totals of a pass A: a stand-alone one's are the linearisation in use; one that rode along with a step is collected even
  // when it is not used (every slice owner has posted it under this command's tag, and the elapsed time travels with it).
  // ONE call site: the collection is 3 KB of code in a kernel that is larger than the instruction cache.
  if (fetch) {
    if (tid == 0) { const long long w0 = (long long)wall_clock64(); cs.t_mark = w0; }
    if constexpr (LOCAL) {
      for (int e = tid; e < P.E; e += nt) lm_place_total(e, tot[e], cP, cU, K - 1, n);
      __syncthreads();
      lm_fold_u(cP, cU, K - 1, n);
    } else {
      if (!lm_fetch_totals(cP, cU, a.dev_res, K - 1, n, cs.tag, &cs.elapsed, grid, &cs.flag)) { if (tid == 0) cs.bad = 1; }
    }
    if (tid == 0) { const long long w1 = (long long)wall_clock64(); cs.t_wait += w1 - cs.t_mark; cs.t_mark = w1; }
  }
  __syncthreads();
  if (st == LMS_STEP) {
    act = cs.act;
    const bool relin = act == ACT_ACCEPT_TAIL && cs.relin;
    __syncthreads();  // thread 0 writes cs.act again below
    if (relin) {  // the pass A that rode along did not run for this outcome: linearise now
      if (tid == 0) { ++cs.lin_calls; cs.state = LMS_ACCEPT_RELIN; }
      return issue(LMOP_LINEARIZE);
    }
  } else {
    if (st == LMS_FIRST) {
      if (tid == 0) { cs.cost = cP[pay1 - 2]; cs.initial_cost = cs.cost; }
      for (int q = tid; q < n; q += nt) cSc[q] = 1.0 / (1.0 + sqrt(cP[n * n + 2 * n + q]));
      act = ACT_ACCEPT_TAIL;  // the same gradient test
    } else if (st == LMS_RELIN) {
      if (tid == 0) cs.need_linearize = 0;
      act = ACT_SOLVE;
    } else {
      act = ACT_ACCEPT_TAIL;
    }
    __syncthreads();
  }

  __syncthreads();
  cstamp(6);
  if (cs.bad) return LMOP_ABORT;  // a granule's tag never showed up (bounded wait): give up, the host reports it
  for (;;) {  // uniform in the workgroup: every transition is decided by thread 0 and read between two barriers
    if (act == ACT_ACCEPT_TAIL) {
      for (int q = tid; q < n; q += nt) { const double g = cP[n * n + n + q]; cTerm[q] = g * g; }
      __syncthreads();
      if (tid == 0) {  // sqrt(sum g_p^2 + sum g_c^2): the 2-norm of the gradient (host/lm.cpp gradient_norm)
        double g2 = cP[pay1 - 1];
        for (int q = 0; q < n; ++q) g2 += cTerm[q];
        if (sqrt(g2) <= opt.gradient_tolerance) { cs.termination = 0; cs.act = ACT_FINISH; } else cs.act = ACT_LOOPTOP;
      }
      __syncthreads();
      act = cs.act;
      __syncthreads();
    }
    if (act == ACT_LOOPTOP) {
      if (tid == 0) {
        int a_ = ACT_SOLVE;
        if (cs.iterations >= opt.max_iterations) { cs.termination = 1; a_ = ACT_FINISH; }
        else if (opt.max_time_s > 0 && cs.elapsed >= opt.max_time_s) { cs.termination = 1; a_ = ACT_FINISH; }  // src/bundle_adjuster.cpp:11, on workgroup 0's posted clock
        else if (cs.radius <= LM_MIN_RADIUS) { cs.termination = 0; a_ = ACT_FINISH; }
        else {
          ++cs.iterations;
          if (cs.need_linearize) { ++cs.lin_calls; cs.state = LMS_RELIN; a_ = ACT_NONE; }
        }
        cs.act = a_;
      }
      __syncthreads();
      act = cs.act;
      __syncthreads();
      if (act == ACT_NONE) return issue(LMOP_LINEARIZE);
    }
    if (act == ACT_FINISH) {
      if (tid == 0) cs.state = LMS_DELIVER;
      if (blockIdx.x == 0) {
        if (tid == 0) {
          double* r = a.host_result;
          pay_store(&r[LMR_ITERATIONS], (double)cs.iterations); pay_store(&r[LMR_SUCCESSFUL], (double)cs.successful);
          pay_store(&r[LMR_TERMINATION], (double)cs.termination); pay_store(&r[LMR_INITIAL_COST], cs.initial_cost);
          pay_store(&r[LMR_FINAL_COST], cs.cost); pay_store(&r[LMR_LINEARIZE_CALLS], (double)cs.lin_calls);
          pay_store(&r[LMR_STEP_CALLS], (double)cs.step_calls); pay_store(&r[LMR_SEL], (double)cs.sel);
          const long long tn = (long long)wall_clock64();
          pay_store(&r[LMR_T_WAIT], (double)cs.t_wait); pay_store(&r[LMR_T_CTL], (double)(cs.t_ctl + (tn - cs.t_mark)));
          pay_store(&r[LMR_T_BODY], (double)cs.t_body); pay_store(&r[LMR_T_TOTAL], (double)(tn - cs.t0));
          pay_store(&r[LMR_SAME_SWEEP], (double)cs.same_sweeps); pay_store(&r[LMR_NEXT_USED], (double)cs.next_used);
          pay_store(&r[LMR_C_ARRIVE], (double)(cs.arrive_total + (unsigned)grid));  // + the delivery's arrivals
          for (int i = 0; i < 14; ++i) pay_store(&r[LMR_TP0 + i], (double)cs.tp[i]);
        }
        for (int i = tid; i < 7 * K; i += nt) pay_store(&a.host_result[LMR_DOUBLES + i], cPose[i]);
      }
      return issue(LMOP_DELIVER);
    }
    // ACT_SOLVE: scaled, damped reduced camera system (host/lm.cpp) -> Cholesky -> pose step
    const double radius = cs.radius;
    for (int q = tid; q < n; q += nt) {
      const double sq = cSc[q];
      cDf[q] = fmin(fmax(cP[n * n + 2 * n + q] * sq * sq, MIN_DIAG), MAX_DIAG) / radius;
      cRhs[q] = -(cP[n * n + q] + cP[n * n + n + q]) * sq;
    }
    __syncthreads();
    // lower triangle of S' = S sc_a sc_b (+ Df on the diagonal).  The totals hold the upper pose-pair blocks; a lower
    // block is the exact transpose of its mirror, diagonal blocks are complete.  A lower element of
    // an off-diagonal block never serves as the SOURCE of another element (sources are upper-block or diagonal-block
    // words), so the in-place write is race free.
    for (int r = tid >> 3; r < n; r += nt >> 3) {
      const double sr = cSc[r];
      for (int c = tid & 7; c <= r; c += 8) {
        const double v = (r / 6 > c / 6) ? cP[c * n + r] : cP[r * n + c];
        double w = v * sr * cSc[c];
        if (r == c) w += cDf[r];
        cP[r * n + c] = w;
      }
    }
    __syncthreads();
    cstamp(7);
    const bool ok = n == 0 || svo_dev_cholesky_solve(cP, cRhs, n, cCol);
    cstamp(8);
    if (ok) {
      for (int q = tid; q < n; q += nt) {
        const double rq = cRhs[q], sq = cSc[q];
        cDc[q] = rq * sq;
        cTerm[q] = 0.5 * rq * (cDf[q] * rq - cP[n * n + n + q] * sq);
      }
      __syncthreads();
      if (tid == 0) {
        double mcc = 0.0;
        for (int q = 0; q < n; ++q) mcc += cTerm[q];
        cs.mcc = mcc;
        cs.chain = 0; cs.spec = 0.0;
        if (cs.iterations < opt.max_iterations) {  // the last iteration cannot use a new linearisation
          // after a saturated step (Ceres' update is exactly radius / (1/3) for rho >= 0.9368) the next one is predicted
          // saturated too: pass A at the candidate runs with that radius in the SAME sweep as pass B — no hand-over
          // for the decision; a misprediction costs one stand-alone pass A (host/lm.cpp, "same sweep")
          if (cs.saturated) { cs.spec = fmin(LM_MAX_RADIUS, cs.radius / (1.0 / 3.0)); ++cs.same_sweeps; }
          else cs.chain = 1;
        }
        ++cs.step_calls;
        cs.state = LMS_STEP;
      }
      if (tid < K) {
        if (tid == 0) for (int q = 0; q < 7; ++q) cCand[q] = cPose[q];
        else svo_plus_pose(&cPose[7 * tid], &cDc[6 * (tid - 1)], &cCand[7 * tid]);
      }
      __syncthreads();
      cstamp(9);
      return issue(LMOP_ITERATE);
    }
    // not positive definite: invalid step, linearise again with the reduced radius
    if (tid == 0) { cs.radius /= cs.df; cs.df *= 2; cs.need_linearize = 1; }
    __syncthreads();
    act = ACT_LOOPTOP;
  }
}

// Landmarks of one wave chunk -> pinned memory (`out`, by landmark index).  A chunk's landmark indices ascend but need
// not be dense (landmarks without observations are skipped when chunks are formed): the export covers the SPAN
// [j0, j_last], staged in LDS and written by consecutive lanes to consecutive addresses (scattered 8-byte system-scope
// stores go out one PCIe write each); the slots of unobserved landmarks inside the span carry zeros, the host restores
// those from its own copy.  Spans beyond the staging fall back to per-landmark stores.
__device__ __forceinline__ void deliver_chunk_points(const ObsRec& R, double* out, double* stage, int stage_doubles) {
  const int lane = threadIdx.x & 63;
  const int j0 = __builtin_amdgcn_readfirstlane(R.j);  // lane 0 holds the chunk's first observation
  int jl = R.active ? R.j : j0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) jl = max(jl, __shfl_xor(jl, off));
  const int span = jl - j0 + 1;
  const bool first = R.active && lane == R.first;
  if (3 * span > stage_doubles) {
    if (first) { pay_store(&out[3 * (size_t)R.j], R.p.x); pay_store(&out[3 * (size_t)R.j + 1], R.p.y); pay_store(&out[3 * (size_t)R.j + 2], R.p.z); }
    return;
  }
  for (int i = lane; i < 3 * span; i += 64) stage[i] = 0.0;
  wave_lds_fence();
  if (first) { stage[3 * (R.j - j0)] = R.p.x; stage[3 * (R.j - j0) + 1] = R.p.y; stage[3 * (R.j - j0) + 2] = R.p.z; }
  wave_lds_fence();
  for (int i = lane; i < 3 * span; i += 64) pay_store(&out[3 * (size_t)j0 + i], stage[i]);
}

// One solve of a launch: everything the kernel needs, in pinned host memory owned by the adjuster (stable while its solve
// is in flight); the launch carries one pointer per solve, blockIdx.y selects it.
struct LmLane { BaDev P; LmDevArgs a; };
struct LmLanePtrs { const LmLane* p[SVO_MAX_LANES]; };

struct LmWave { ObsRec R; ChunkRegs c; D3 cand; };
struct LmShared { double sOut[4]; double sDec[2]; int sGo; };

// The owner phases of this wavefront's chunk, in its own LDS (chunk_owner_phases<false>).
__device__ __forceinline__ void lm_owner_phases(const BaDev& P, const LmWave& W, bool my_wave_works, const SufRegs& o, const uint16_t* tabs, const WgLds& L, long long* t_schur = nullptr) {
  if (!my_wave_works) return;
  const int wave = threadIdx.x >> 6;
  const ChunkTab T{tabs, (P.K - 1) * P.K / 2, P.K - 1};
  chunk_owner_phases<false>(W.R, T, o, L.rec, make_sink(P, (int)blockIdx.x * LM_CPW + wave, 1, L.pst), L.s_ne, t_schur);
}

// One command of ba_lm_kernel's passes: LMOP_ITERATE = pass B, [the decision,] pass A, this workgroup's slice of level 2;
// LMOP_LINEARIZE = pass A alone at the current point, then level 2.  ONE instance of every stage's code (prefix, suffix,
// owner phases, level 2) serves all of them: the kernel is larger than the instruction cache, and every inlined copy of a
// stage that this function used to hold (three of pass A's) cost every pass of every solve misses — measured: 5 KB more code,
// nowhere near pass A, made pass A 1.8 us slower.
template <bool GROUPED>
__device__ __forceinline__ bool lm_iterate(const BaDev& P, const LmDevArgs& a, LmWave& W, bool my_wave_works, const uint16_t* tabs, const WgLds& L,
                                           double* union_lds, int union_doubles, LmDevState& cs, double* sStep, LmShared& sh, int n_blocks, long long t_first, long long* tp,
                                           bool linearize_only) {
  const int tid = threadIdx.x, wave = tid >> 6;
  long long tmark = tp && tid == 0 ? (long long)wall_clock64() : 0;
  auto stamp = [&](int slot) { if (tp && tid == 0) { const long long tn = (long long)wall_clock64(); tp[slot] += tn - tmark; tmark = tn; } };
  const double* dc_ = sStep;
  const double* cand_poses_ = sStep + (P.n > 0 ? P.n : 1);
  const double* cur_poses_ = cand_poses_ + 7 * P.K;
  const double radius = cs.radius, spec_radius = linearize_only ? 0.0 : cs.spec;
  const int chain = linearize_only ? 0 : cs.chain;
  const int with_pay1 = linearize_only || chain || spec_radius > 0;
  const int my_chunk = (int)blockIdx.x * LM_CPW + wave;
  double* lms = L.rec;  // this wavefront's own LDS: pass B's landmark scalars live in the idle staging rows
  double unused0 = 0, unused1 = 0, unused2 = 0, unused3 = 0;
  const PartSink sink = make_sink(P, my_chunk, 1, L.pst);  // window problems: group = chunk
  LinPre pre;
  SufRegs o;
  o.freep = false;
  if (!linearize_only && my_wave_works)
    backsub_chunk(P, W.R, cur_poses_, cand_poses_, dc_, P.cand_points, radius, W.cand, unused0, unused1, unused2, unused3, &W.c, lms, &sink);
  stamp(0);
  if (with_pay1) {
    // what pass A linearises: the current point (a linearisation alone: first one, or after a rejected / mispredicted step), or
    // the candidate pass B just formed (same sweep: with the predicted radius; chained: with the radius of the decision below)
    ObsRec Ra = W.R;
    const double* poses_a = cur_poses_;
    double radius_a = radius;
    const int first_a = linearize_only ? cs.first : 0;
    if (!linearize_only) { Ra.p = W.cand; poses_a = cand_poses_; radius_a = spec_radius; }
#pragma nounroll
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (my_wave_works) linearize_prefix(P, Ra, poses_a, pre, unused0);  // radius-free; in a chained step the other wavefronts' sums are on their way meanwhile
      if (!chain || attempt == 1) break;
      stamp(1);
      // every workgroup collects pass B's sums itself and takes the decision: identical inputs, identical bits (the staging
      // rows are idle between the passes: they are the scratch of the collection)
      const bool ok = sum_pay2<GROUPED>(P, P.pay_parity, P.pay_tag, union_lds, sh.sOut, &sh.sGo);
      if (!ok) return false;
      if (tid == 0) {
        const SvoLmDecision dec = svo_lm_decide(cs.cost, cs.mcc, radius, cs.df, sh.sOut[0], sh.sOut[1]);
        sh.sDec[0] = (double)dec.accept; sh.sDec[1] = dec.next_radius;
        cs.pay2[0] = sh.sOut[0]; cs.pay2[1] = sh.sOut[1]; cs.pay2[2] = sh.sOut[2]; cs.pay2[3] = sh.sOut[3];
        cs.pay2[4] = (double)dec.accept; cs.pay2[5] = dec.next_radius;
      }
      __syncthreads();
      stamp(2);
      radius_a = sh.sDec[1];
      if (sh.sDec[0] != 0.0) break;  // accepted: the prefix at the candidate is the one to use
      Ra = W.R; poses_a = cur_poses_;  // rejected: the radius-free part again, at the current point
    }
    if (my_wave_works) suffix_math(P, Ra, pre, radius_a, first_a, L.rec, &W.c, o);
    stamp(10);  // the lanes' own arithmetic behind the decision
    if (tp && tid == 0) tp[12] -= tmark;
    lm_owner_phases(P, W, my_wave_works, o, tabs, L, tp ? tp + 12 : nullptr);  // slot 12: Schur owners alone; slot 3 (below): all owner phases
  }
  stamp(3);
  if (!with_pay1) return true;
  __syncthreads();  // both wavefronts are through their passes: their LDS becomes the scratch of level 2
  // level 2: this workgroup's slice of the wire elements.  A workgroup WITHOUT a slice must not wait for anybody's partials:
  // nobody waits for it in return, so the fast workgroups may already be posting the next command's (the hang of the first
  // version: E = 2 for a one-keyframe window, 494 elements over 47 workgroups leave two of them without a slice)
  const int per = (P.E + n_blocks - 1) / n_blocks;
  const int e0 = min(P.E, (int)blockIdx.x * per), e1 = min(P.E, e0 + per);
  double* res = a.dev_res;
  const unsigned long long tag = P.pay_tag;
  bool ok = true;
  if (e0 < e1) ok = reduce_elements<GROUPED>(P, e0, e1, tag, union_lds, union_doubles, &sh.sGo, [res, tag](int e, double v) { granule_store(&res[2 * e], v, tag); }, tp ? tp + 5 : nullptr);
  if (blockIdx.x == 0 && tid == 0)  // the clock every controller tests (see the kernel's header): seconds since this workgroup's first pass
    granule_store(&res[2 * P.E], 1e-8 * (double)((long long)wall_clock64() - t_first), tag);
  stamp(4);
  return ok;
}

#ifndef SVO_LM_WAVES_PER_EU  // developer experiments only (SVO_EXTRA_HIPFLAGS): the register budget of the solve kernel's wavefronts
#define SVO_LM_WAVES_PER_EU 2
#endif
// GROUPED: windows of 129..LM_MAX_CHUNKS_GROUPED chunks (the 10-keyframe windows of configs[2]): levels 1-2 of the declared order sum
// groups of G = ceil(C / 128) consecutive chunks first (reduce_elements<true>, sum_pay2<true>: the forms the host-driven kernels use).
// A kernel of its own (ba_lm_grouped_kernel) so that the plain one — every pipeline group's solve — keeps its code size.
template <bool GROUPED>
__device__ __forceinline__ void ba_lm_body(const LmLanePtrs& lanes) {
  extern __shared__ double lds[];  // ba_lm_lds_doubles(n, K)
  __shared__ LmShared sh;
  __shared__ LmDevState cs;
  __shared__ __align__(16) unsigned char sLaneRaw[sizeof(LmLane)];
  LmLane& sLane = *reinterpret_cast<LmLane*>(sLaneRaw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    // the solve's record: one PCIe read per workgroup, then LDS / registers
    const unsigned* src = reinterpret_cast<const unsigned*>(lanes.p[blockIdx.y]);
    unsigned* dst = reinterpret_cast<unsigned*>(&sLane);
    // system-scope loads: a plain load may be served from a line this XCD's L2 kept from the adjuster's PREVIOUS launch
    // (measured: workgroups that took the previous problem's chunk count and left at once)
    for (int i = tid; i < (int)(sizeof(LmLane) / 4); i += blockDim.x) dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  const int n_blocks = (sLane.P.C + LM_CPW - 1) / LM_CPW;  // workgroups of THIS solve; the launch is as wide as its largest solve
  if ((int)blockIdx.x >= n_blocks) return;
  BaDev P = sLane.P;
  const LmDevArgs& a = sLane.a;
  const int n = P.n, K = P.K, nn = n > 0 ? n : 1;
  const int union_doubles = (int)ba_lm_union_doubles(n, K);
  double* union_lds = lds;                                         // staging rows + landmark scalars | controller workspace
  const WgLds L = wg_lds(lds + wave * wg_lds_doubles(P.E), P.E);  // this wavefront's own staging rows and outgoing partials
  uint16_t* tabs = reinterpret_cast<uint16_t*>(lds + union_doubles) + wave * a.tab_words;
  double* cSc = lds + union_doubles + LM_CPW * a.tab_words / 4;  // persistent: Jacobi scales of the pose columns
  double* sStep = cSc + 2 * nn;                                    // [dc | candidate poses | current poses] (the second nn: spare)
  const int my_chunk = (int)blockIdx.x * LM_CPW + wave;
  const bool my_wave_works = my_chunk < P.C;
  if (a.test_giveup) {  // (test hook: the host must re-run this solve and lose nothing)
    if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.host_flag, -a.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  __builtin_amdgcn_s_setprio(3);
  P.step_in = nullptr;  // the step block is built in LDS by the step control
  P.flag = nullptr;
  if (tid == 0) { cs.state = LMS_START; cs.bad = 0; cs.accepted = 0; }
  // this wave's observations, landmarks and chunk table: loaded ONCE, read in place from the pinned problem image
  LmWave W;
  W.R = ObsRec{false, 0, 0, 0, lane, 0, D3{0, 0, 1}, 0.0, 0.0};
  W.cand = D3{0, 0, 1};
  W.c.s[0] = W.c.s[1] = W.c.s[2] = 1.0;
  const ptrdiff_t shift = a.arena_src ? reinterpret_cast<const char*>(a.arena_src) - reinterpret_cast<const char*>(a.arena_dst) : 0;
  if (my_wave_works) {
    W.R = load_obs_image(P, my_chunk, lane, a.points_a, shift);
    // the device copy of the current landmarks (DELIVER exports from the buffer the step control selected; pass B fills the other one)
    if (shift && W.R.active && lane == W.R.first) { a.points_a[3 * W.R.j] = W.R.p.x; a.points_a[3 * W.R.j + 1] = W.R.p.y; a.points_a[3 * W.R.j + 2] = W.R.p.z; }
    load_chunk_table(P, my_chunk, tabs, shift);
  }
  long long t_first = 0;
  __syncthreads();
  for (;;) {
    const int st_before = cs.state;
    const int op = lm_controller<false, GROUPED>(P, a, cs, union_lds, cSc, sStep, sh.sOut);
    if (a.dbg && tid == 0) {
      unsigned* g = a.dbg + 16 * blockIdx.x;
      g[0] = (unsigned)op; g[1] = (unsigned)cs.state; g[2] = (unsigned)cs.iterations; g[3] = (unsigned)cs.need_linearize;
      g[4] = (unsigned)cs.chain | ((unsigned)cs.bad << 8);  // bit 8: a tagged granule never arrived
      g[5] = cs.arrive_total; g[6] = cs.op_count; g[7] = (unsigned)cs.lin_calls;
    }
    if (op == LMOP_ABORT || cs.bad) {  // a bounded wait ran out: tell the host now (it re-runs the solve), the other workgroups follow within their own bound
      if (tid == 0) __hip_atomic_store(a.host_flag, -a.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    if (op == LMOP_EXIT) break;
    if (st_before == LMS_STEP && cs.accepted) W.R.p = W.cand;  // the step control took the step: the candidate is the current point
    if (!t_first) t_first = (long long)wall_clock64();
    const bool sel = cs.sel != 0;
    P.points = sel ? a.points_b : a.points_a;
    P.cand_points = sel ? a.points_a : a.points_b;
    P.pay_tag = cs.tag;
    P.pay_parity = (int)(cs.op_count & 1u);
    if (op == LMOP_DELIVER) {
      if (a.export_points && my_wave_works) {
        ObsRec R = W.R;  // the landmarks of the buffer the step control selected (a chained pass A may have run ahead of a step that was not taken)
        if (R.active) R.p = D3{P.points[3 * R.j], P.points[3 * R.j + 1], P.points[3 * R.j + 2]};
        deliver_chunk_points(R, a.export_points, union_lds + wave * ((64 * REC_STRIDE) / LM_CPW), (64 * REC_STRIDE) / LM_CPW);  // the staging rows are idle
      }
      if (a.store && my_wave_works && W.R.active && lane == W.R.first) {
        // get_world_points (src/bundle_adjuster.cpp:159-163: double -> float) for the next keyframe's PnP, served from the device:
        // one write-through 16-byte store per landmark, acknowledged before the completion word like everything else
        const int j = W.R.j;
        const unsigned key = sys_load(&P.lm_key[j], shift);
        const float x = (float)P.points[3 * j], y = (float)P.points[3 * j + 1], z = (float)P.points[3 * j + 2];
        const unsigned long long lo = ((unsigned long long)__float_as_uint(y) << 32) | __float_as_uint(x), hi = ((unsigned long long)key << 32) | __float_as_uint(z);
        slot_store2<true>(reinterpret_cast<double*>(a.store + (key & a.store_mask)), __longlong_as_double((long long)lo), __longlong_as_double((long long)hi));
      }
      if (a.dbg && tid == 0) {  // SVO_BA_TRACE: this workgroup's own split (ticks), for the spread over the workgroups of a solve
        unsigned* g = a.dbg + 16 * blockIdx.x;
        for (int i = 0; i < 6; ++i) g[8 + i] = (unsigned)cs.tp[i];
        g[8 + 3] = (unsigned)(cs.tp[3] + cs.tp[10]);  // rest of pass A = the lanes' arithmetic + the owner phases
        g[14] = (unsigned)cs.tp[6];
      }
      // everybody's results are out; the last workgroup to arrive publishes the host's completion word
      stores_acknowledged();
      __syncthreads();
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.cnt + LMC_ARRIVE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == cs.arrive_total) __hip_atomic_store(a.host_flag, a.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();
      continue;  // the next controller turn answers "delivered": everybody leaves
    }
    long long* tp = a.dbg ? cs.tp : nullptr;
    if (!lm_iterate<GROUPED>(P, a, W, my_wave_works, tabs, L, union_lds, union_doubles, cs, sStep, sh, n_blocks, t_first, op == LMOP_ITERATE ? tp : nullptr, op != LMOP_ITERATE)) {
      if (tid == 0) __hip_atomic_store(a.host_flag, -a.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(SVO_LM_WAVES_PER_EU, SVO_LM_WAVES_PER_EU))) void ba_lm_kernel(LmLanePtrs lanes) { ba_lm_body<false>(lanes); }
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(SVO_LM_WAVES_PER_EU, SVO_LM_WAVES_PER_EU))) void ba_lm_grouped_kernel(LmLanePtrs lanes) { ba_lm_body<true>(lanes); }

// ---------------------------------------------------------------------------------------------------
// The THROUGHPUT form of the device-resident solve: ba_lm_compact_kernel — ONE workgroup runs a whole solve (round 5).
// ba_lm_kernel spreads a window over ~47 workgroups of two 256-VGPR wavefronts that spend most of their life parked, waiting for
// each other's tagged granules: with 48 streams on the GPU ~10 resident solves held 46 % of all register files and the frame
// rate stopped growing with the number of streams.  Here a solve is NW <= 6 wavefronts in one workgroup:
//   * wavefront w takes chunks w, w + NW, w + 2 NW, ... IN TURN through its ONE staging area (rows + outgoing partials: the
//     same device functions as every other form — backsub_chunk, linearize_prefix, suffix_math, chunk_owner_phases);
//   * after every round of NW chunks the workgroup meets at a barrier and thread e adds the round's partials of wire element e to
//     the running total IN CHUNK ORDER — the declared order (groups of G chunks first when C > 128), so the bits are those of
//     every other form and of the oracle;
//   * the step control (lm_controller<true>: the same code as ba_lm_kernel's) runs ONCE per solve, on totals that never leave
//     the workgroup's LDS: no tagged granules, no sentinels, no replicated Cholesky, no hand-over latency;
//   * nothing waits for another workgroup, so there is no co-residency requirement, no admission budget and no way to give
//     up: the form is also the fallback of a wide solve that could not make progress.
// Per solve: 1/16 of the wide form's wavefronts and LDS for ~5x its latency — what a GPU shared by dozens of streams wants.
// The problem image is copied from pinned host memory into the device arena by the kernel itself (16-byte system-scope loads,
// eight in flight per thread) and read from there by plain loads (one workgroup = one CU = one L2: coherent by construction).
constexpr int LMC_MAX_WAVES = 6;
__host__ __device__ static inline int lmc_wave_doubles(int E) { return 64 * REC_STRIDE + ((E + 1) & ~1) + 2 + 4; }  // rows | partials | 4 ints | pass B's four sums
__host__ __device__ static inline size_t ba_lmc_union_doubles(int n, int K, int NW) {
  const size_t a = (size_t)NW * (size_t)lmc_wave_doubles(ba_wire_elements(K)), b = ba_lm_ctl_doubles(n, K);
  return a > b ? a : b;
}
// dynamic LDS (doubles): [NW wave areas | controller workspace] (union) | NW chunk tables | running totals (E) | running group sums (E) |
// pass B's totals and group sums (4 + 4) | Jacobi scales of the pose columns (nn) | spare (nn) | step block [dc (nn) | candidate poses (7 K) | current poses (7 K)]
__host__ __device__ static inline size_t ba_lmc_lds_doubles(int n, int K, int tab_words, int NW) {
  const size_t Ep = ((size_t)ba_wire_elements(K) + 1) & ~(size_t)1, nn = n > 0 ? n : 1;
  return ba_lmc_union_doubles(n, K, NW) + (size_t)NW * (size_t)tab_words / 4 + 2 * Ep + 8 + 3 * nn + 14 * (size_t)K;
}

// element `idx` of the running sums takes chunk c's partial p: groups of G consecutive chunks are summed from their first, the
// groups from the first (grouped_seq_sum's additions in its order)
__device__ __forceinline__ void lmc_acc(double* tot, double* q, int idx, int c, int C, int G, double p) {
  if (G <= 1) { tot[idx] = c == 0 ? p : tot[idx] + p; return; }
  const int r = c % G;
  const double qq = r == 0 ? p : q[idx] + p;
  if (r == G - 1 || c == C - 1) tot[idx] = c < G ? qq : tot[idx] + qq;
  else q[idx] = qq;
}

struct LmcLds { double* wave_area; uint16_t* tab; double *tot, *q, *tot2, *q2; int wave_doubles; };

// One sweep over all chunks of the solve.  do_b: pass B (step dc_ from the current point, candidate landmarks -> P.cand_points);
// do_a: pass A — at the candidate pass B just formed (a_from_b: same sweep), at the stored candidate (a_at_cand: behind an accepted
// chained decision) or at the current point.  Leaves the totals in L.tot2 / L.tot.  Block-uniform arguments.
__device__ __forceinline__ void lmc_sweep(const BaDev& P, const LmcLds& L, const double* sStep, bool do_b, bool do_a, bool a_from_b, bool a_at_cand,
                                          double radius_b, double radius_a, int first_a) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, NW = nt >> 6;
  const int C = P.C, G = P.G, E = P.E;
  const double* dc_ = sStep;
  const double* cand_poses_ = sStep + (P.n > 0 ? P.n : 1);
  const double* cur_poses_ = cand_poses_ + 7 * P.K;
  double* area = L.wave_area + (size_t)wave * L.wave_doubles;
  double* rec = area;
  double* pst = area + 64 * REC_STRIDE;
  int* s_ne = reinterpret_cast<int*>(pst + ((E + 1) & ~1));
  double* pst2 = pst + ((E + 1) & ~1) + 2;
  const PartSink sink{nullptr, nullptr, pst, P.Epad, E, P.NG, 0, 1, 0, 0ull, pst2};
  for (int c0 = 0; c0 < C; c0 += NW) {
    const int chunk = c0 + wave;
    if (chunk < C) {
      const ObsRec R = load_obs(P, chunk, lane, P.points);
      D3 cand = R.p;
      double u0 = 0, u1 = 0, u2 = 0, u3 = 0;
      if (do_a) {  // the chunk's table on its way to LDS while pass B / the prefix compute (plain loads: the arena was written by this workgroup)
        const uint32_t o0 = P.tab_off[chunk], o1 = P.tab_off[chunk + 1];
        const int words32 = (int)((o1 - o0 + 1) >> 1);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(P.tab + o0);
        uint32_t* d32 = reinterpret_cast<uint32_t*>(L.tab);
        for (int i = lane; i < words32; i += 64) d32[i] = src[i];
      }
      if (do_b) backsub_chunk(P, R, cur_poses_, cand_poses_, dc_, P.cand_points, radius_b, cand, u0, u1, u2, u3, nullptr, rec, &sink);
      if (do_a) {
        ObsRec Ra = R;
        const double* poses_a = cur_poses_;
        if (a_from_b) { Ra.p = cand; poses_a = cand_poses_; }
        else if (a_at_cand) {
          if (R.active) Ra.p = D3{P.cand_points[3 * R.j], P.cand_points[3 * R.j + 1], P.cand_points[3 * R.j + 2]};
          poses_a = cand_poses_;
        }
        LinPre pre;
        linearize_prefix(P, Ra, poses_a, pre, u0);
        ChunkRegs cr;
        cr.s[0] = cr.s[1] = cr.s[2] = 1.0;
        if (!first_a && R.active) { cr.s[0] = P.sp[3 * R.j]; cr.s[1] = P.sp[3 * R.j + 1]; cr.s[2] = P.sp[3 * R.j + 2]; }
        SufRegs o;
        o.freep = false;
        suffix_math(P, Ra, pre, radius_a, first_a, rec, &cr, o);
        wave_lds_fence();  // the table's LDS copy is complete
        const ChunkTab T{L.tab, (P.K - 1) * P.K / 2, P.K - 1};
        chunk_owner_phases<false>(Ra, T, o, rec, sink, s_ne);
      }
    }
    __syncthreads();
    // the round's partials join the running sums in chunk order
    const int nw = min(NW, C - c0);
    if (do_b && tid < 4)
      for (int w = 0; w < nw; ++w) lmc_acc(L.tot2, L.q2, tid, c0 + w, C, G, (L.wave_area + (size_t)w * L.wave_doubles)[64 * REC_STRIDE + ((E + 1) & ~1) + 2 + tid]);
    if (do_a)
      for (int e = tid; e < E; e += nt)
        for (int w = 0; w < nw; ++w) lmc_acc(L.tot, L.q, e, c0 + w, C, G, (L.wave_area + (size_t)w * L.wave_doubles)[64 * REC_STRIDE + e]);
    __syncthreads();
  }
}

__global__ __launch_bounds__(64 * LMC_MAX_WAVES) void ba_lm_compact_kernel(LmLanePtrs lanes) {
  extern __shared__ double lds[];  // ba_lmc_lds_doubles(n, K, tab_words, NW)
  __shared__ LmShared sh;
  __shared__ LmDevState cs;
  __shared__ __align__(16) unsigned char sLaneRaw[sizeof(LmLane)];
  LmLane& sLane = *reinterpret_cast<LmLane*>(sLaneRaw);
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, NW = nt >> 6;
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(lanes.p[blockIdx.y]);
    unsigned* dst = reinterpret_cast<unsigned*>(&sLane);
    for (int i = tid; i < (int)(sizeof(LmLane) / 4); i += nt) dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  BaDev P = sLane.P;
  const LmDevArgs& a = sLane.a;
  const int n = P.n, K = P.K, nn = n > 0 ? n : 1, E = P.E, Ep = (E + 1) & ~1;
  const int union_doubles = (int)ba_lmc_union_doubles(n, K, NW);
  LmcLds L;
  L.wave_area = lds;
  L.wave_doubles = lmc_wave_doubles(E);
  L.tab = reinterpret_cast<uint16_t*>(lds + union_doubles) + wave * a.tab_words;
  L.tot = lds + union_doubles + NW * a.tab_words / 4;
  L.q = L.tot + Ep;
  L.tot2 = L.q + Ep;
  L.q2 = L.tot2 + 4;
  double* cSc = L.q2 + 4;          // persistent: Jacobi scales of the pose columns
  double* sStep = cSc + 2 * nn;    // [dc | candidate poses | current poses] (the second nn: spare)
  // the problem image -> the device arena
  if (a.arena_src) {
    const char* src = reinterpret_cast<const char*>(a.arena_src);
    char* dst = reinterpret_cast<char*>(a.arena_dst);
    const size_t n16 = a.arena_bytes / 16;
    for (size_t base = 0; base < n16; base += 8 * (size_t)nt) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t i = base + (size_t)u * nt + tid;
        const char* q = src + 16 * (i < n16 ? i : n16 - 1);
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(q) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t i = base + (size_t)u * nt + tid;
        if (i < n16) *reinterpret_cast<uint4*>(dst + 16 * i) = v[u];
      }
    }
  }
  P.step_in = nullptr; P.flag = nullptr; P.part1 = nullptr; P.part2 = nullptr;
  __builtin_amdgcn_s_setprio(3);
  if (tid == 0) { cs.state = LMS_START; cs.bad = 0; cs.accepted = 0; cs.elapsed = 0.0; }
  __syncthreads();
  long long t_first = 0;
  LmDevArgs al = a;
  al.arena_src = nullptr;  // the controller reads the poses from the device arena now
  for (;;) {
    const int st_before = cs.state;
    if (tid == 0 && t_first) cs.elapsed = 1e-8 * (double)((long long)wall_clock64() - t_first);
    const int op = lm_controller<true>(P, al, cs, lds, cSc, sStep, sh.sOut, L.tot, L.tot2);
    if (op == LMOP_ABORT || op == LMOP_EXIT) break;
    (void)st_before;
    if (!t_first) t_first = (long long)wall_clock64();
    const bool sel = cs.sel != 0;
    P.points = sel ? a.points_b : a.points_a;
    P.cand_points = sel ? a.points_a : a.points_b;
    if (op == LMOP_DELIVER) {
      double* stage = lds + (size_t)wave * L.wave_doubles;  // the staging rows are idle
      for (int c0 = 0; c0 < P.C; c0 += NW) {
        const int chunk = c0 + wave;
        if (chunk >= P.C) break;  // (no barrier below: wave-level only)
        const ObsRec R = load_obs(P, chunk, lane, P.points);
        if (a.export_points) deliver_chunk_points(R, a.export_points, stage, 64 * REC_STRIDE);
        if (a.store && R.active && lane == R.first) {  // get_world_points (src/bundle_adjuster.cpp:159-163: double -> float) for the next keyframe's PnP
          const unsigned key = P.lm_key[R.j];
          const unsigned long long lo = ((unsigned long long)__float_as_uint((float)R.p.y) << 32) | __float_as_uint((float)R.p.x);
          const unsigned long long hi = ((unsigned long long)key << 32) | __float_as_uint((float)R.p.z);
          slot_store2<true>(reinterpret_cast<double*>(a.store + (key & a.store_mask)), __longlong_as_double((long long)lo), __longlong_as_double((long long)hi));
        }
        wave_lds_fence();
      }
      stores_acknowledged();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.host_flag, a.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __syncthreads();
      continue;  // the next controller turn answers "delivered"
    }
    const bool linearize_only = op != LMOP_ITERATE;
    const double radius = cs.radius, spec = linearize_only ? 0.0 : cs.spec;
    const int chain = linearize_only ? 0 : cs.chain;
    if (linearize_only) {
      lmc_sweep(P, L, sStep, false, true, false, false, 0.0, radius, cs.first);
    } else if (spec > 0) {
      lmc_sweep(P, L, sStep, true, true, true, false, radius, spec, 0);
    } else {
      lmc_sweep(P, L, sStep, true, false, false, false, radius, 0.0, 0);
      if (chain) {
        if (tid == 0) {  // Ceres' decision on the summed payload2 (the same function as everywhere)
          const SvoLmDecision dec = svo_lm_decide(cs.cost, cs.mcc, radius, cs.df, L.tot2[0], L.tot2[1]);
          sh.sDec[0] = (double)dec.accept; sh.sDec[1] = dec.next_radius;
          cs.pay2[0] = L.tot2[0]; cs.pay2[1] = L.tot2[1]; cs.pay2[2] = L.tot2[2]; cs.pay2[3] = L.tot2[3];
          cs.pay2[4] = (double)dec.accept; cs.pay2[5] = dec.next_radius;
        }
        __syncthreads();
        lmc_sweep(P, L, sStep, false, true, false, sh.sDec[0] != 0.0, 0.0, sh.sDec[1], 0);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// Bulk (non-deterministic) linearisation with the Schur products on the f64 matrix cores.
//
// For a landmark with V^-1 = Lc Lc^T (3x3 Cholesky) and Z_i = (W_i s) Lc (6x3 per observing pose), its whole
// contribution to the reduced camera matrix is  -Z Z^T  with Z the (6 x poses) x 3 stack: one rank-3 update
// of S.  S (n <= 128) is cut into 16x16 tiles; v_mfma_f64_16x16x4_f64 applies the update to one tile with
// K = 3 of its 4 k-slots used.  A workgroup is 8 waves; the accumulators are shared by the workgroup, not
// per wave: wave w keeps tiles (w, w..w+4 mod 8) — the circulant half of the symmetric matrix, 5 (4 for
// w >= 4) accumulator tiles = 40 VGPRs — in registers across ALL landmarks the workgroup sees and flushes
// them once at the end.
//   phase 1 (wave = one chunk of <= 64 observations, lane = observation; as ba_linearize_body): residual,
//            Jacobians, landmark sums by lane gathers, V^-1, Z -> LDS, plus a pose->lane byte table and the
//            tile-row mask of every landmark; U / g_c / g_red go to a small LDS image (ds_add_f64).
//   phase 2 (wave = tile row): for each of the 8 staged chunks, for each landmark whose mask touches the
//            row: gather the A operand (16 rows x 3) once, the B operands per touched tile, MFMA.
// Algorithmic work per landmark with L observations: 36 L^2 multiply-adds of Schur product — here
// 2048 flop per touched tile on the matrix pipe instead of 36 L(L+1)/2 LDS atomics.
constexpr int MF_WAVES = 8;
constexpr int MF_COPIES = 8;    // private copies of the U / g_c / g_red image (landmark index mod 8): same-pose lanes of a wave rarely share one
constexpr int MF_TBL_ROW = 32;   // bytes per landmark in the pose->lane table: free poses <= 21 (n <= 128)
typedef double mf_d4 __attribute__((ext_vector_type(4)));

static inline size_t ba_mfma_lds_bytes(int n, int F) {
  return sizeof(double) * ((size_t)MF_WAVES * 64 * 18 + MF_COPIES * ((size_t)F * 21 + 2 * (size_t)n) + 2) + (size_t)MF_WAVES * 64 * MF_TBL_ROW +
         sizeof(uint32_t) * MF_WAVES * 64 + sizeof(int) * MF_WAVES;
}

__global__ __launch_bounds__(512) void ba_linearize_mfma_kernel(BaDev P, double radius, int first_pass, const double* __restrict__ ctl, BulkSel bs) {
  apply_ctl(P, radius, ctl);
  if (!bulk_apply(P, radius, first_pass, bs)) return;
  extern __shared__ double lds[];
  const int n = P.n, F = P.K - 1;
  double* sZ = lds;                                  // [8][64][18]
  const int img = F * 21 + 2 * n;                    // one image: U upper triangles [F][21] | g_red [n] | g_c [n]
  double* sImg = sZ + MF_WAVES * 64 * 18;            // [MF_COPIES][img]
  double* sAcc = sImg + MF_COPIES * img;             // cost, sum g_p^2
  uint8_t* sTbl = reinterpret_cast<uint8_t*>(sAcc + 2);                         // [8][64][32]
  uint32_t* sMask = reinterpret_cast<uint32_t*>(sTbl + MF_WAVES * 64 * MF_TBL_ROW);  // [8][64]
  int* sNlm = reinterpret_cast<int*>(sMask + MF_WAVES * 64);                    // [8]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < MF_COPIES * img + 2; i += blockDim.x) sImg[i] = 0.0;
  __syncthreads();

  mf_d4 acc[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) acc[d] = mf_d4{0.0, 0.0, 0.0, 0.0};
  // operand coordinates of this lane for the row tile and the 5 column tiles: global row g = 16 t + (lane & 15)
  // -> (pose g / 6, component g % 6); k-slot kk = lane >> 4 (slot 3 is padding)
  const int kk = lane >> 4;
  // invalid lanes (padding rows / k-slot 3) read table byte MF_TBL_ROW-1, which no pose ever writes (-> 255 -> 0)
  int tix[5], go_[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const int g = 16 * ((wave + d) & 7) + (lane & 15);
    const int p = g / 6;
    const bool valid = g < n && kk < 3;
    tix[d] = valid ? p : MF_TBL_ROW - 1;
    go_[d] = valid ? 3 * (g - 6 * p) + kk : 0;
  }

  double lcost = 0.0, lgp2 = 0.0;
  const int groups = (P.C + MF_WAVES - 1) / MF_WAVES;
  // per-lane observation record of the NEXT group, fetched while the current group is in phase 2 (the
  // index -> landmark -> point chain is three dependent HBM/L2 round trips that two waves per SIMD cannot hide)
  struct Fetch { int c0, k, j, first, len; bool active; double u, v; D3 p; };
  auto fetch = [&](int grp) {
    Fetch f{0, 0, 0, lane, 0, false, 0.0, 0.0, D3{0, 0, 1}};
    const int chunk = grp * MF_WAVES + wave;
    if (grp < groups && chunk < P.C) {
      const int o = chunk * 64 + lane;
      const int4 rc = P.rec[o];
      f.active = rc.x >= 0;
      if (f.active) {
        f.k = rc.x; f.j = rc.y; f.first = rc.z; f.len = rc.w;
        f.p = D3{P.points[3 * f.j], P.points[3 * f.j + 1], P.points[3 * f.j + 2]};
        f.u = P.obs_uv[2 * o]; f.v = P.obs_uv[2 * o + 1];
      }
    }
    return f;
  };
  Fetch nxt = fetch(blockIdx.x);
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int chunk = grp * MF_WAVES + wave;
    const Fetch cur = nxt;
    // ---- phase 1 -------------------------------------------------------------------------------
    {
      uint64_t* t8 = reinterpret_cast<uint64_t*>(sTbl + wave * 64 * MF_TBL_ROW);
#pragma unroll
      for (int i = 0; i < MF_TBL_ROW / 8; ++i) t8[lane * (MF_TBL_ROW / 8) + i] = ~0ull;
      sMask[wave * 64 + lane] = 0u;
    }
    int nlm = 0;
    if (chunk < P.C) {
      const bool active = cur.active;
      const int k = cur.k, j = cur.j, first = cur.first, len = cur.len;
      double r[2] = {0, 0}, Jc[12], Jp[6];
#pragma unroll
      for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) Jp[i] = 0.0;
      if (active) {
        eval_obs_contract(P.poses + 7 * k, cur.p, cur.u, cur.v, P.f, P.cx, P.cy, k > 0, r, Jc, Jp);
        lcost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
      }
      int maxlen = len;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
      // landmark sums V = sum Jp^T Jp (symmetric: 6 terms), g_p = sum Jp^T r
      double V[9], gp[3];
      {
        double t[9] = {Jp[0] * Jp[0] + Jp[3] * Jp[3], Jp[0] * Jp[1] + Jp[3] * Jp[4], Jp[0] * Jp[2] + Jp[3] * Jp[5],
                       Jp[1] * Jp[1] + Jp[4] * Jp[4], Jp[1] * Jp[2] + Jp[4] * Jp[5], Jp[2] * Jp[2] + Jp[5] * Jp[5],
                       Jp[0] * r[0] + Jp[3] * r[1], Jp[1] * r[0] + Jp[4] * r[1], Jp[2] * r[0] + Jp[5] * r[1]};
        segment_totals<9>(t, lane, first, len > 0 ? first + len - 1 : lane, maxlen);
        V[0] = t[0]; V[1] = V[3] = t[1]; V[2] = V[6] = t[2]; V[4] = t[3]; V[5] = V[7] = t[4]; V[8] = t[5];
        gp[0] = t[6]; gp[1] = t[7]; gp[2] = t[8];
      }
      double s[3] = {1, 1, 1};
      if (active) {
        if (first_pass) {
#pragma unroll
          for (int a = 0; a < 3; ++a) s[a] = 1.0 / (1.0 + sqrt(V[4 * a]));
          if (lane == first) { P.sp[3 * j] = s[0]; P.sp[3 * j + 1] = s[1]; P.sp[3 * j + 2] = s[2]; }
        } else {
          s[0] = P.sp[3 * j]; s[1] = P.sp[3 * j + 1]; s[2] = P.sp[3 * j + 2];
        }
        if (lane == first) lgp2 += gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2];
      }
      double Vd[9], Vi[9], gps[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        gps[a] = gp[a] * s[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) Vd[3 * a + b] = V[3 * a + b] * s[a] * s[b];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) Vd[4 * a] += fmin(fmax(Vd[4 * a], MIN_DIAG), MAX_DIAG) / radius;
      inv3_sym(Vd, Vi);
      // V^-1 = Lc Lc^T
      double l00 = 0, l10 = 0, l20 = 0, l11 = 0, l21 = 0, l22 = 0;
      if (Vi[0] > 0) {
        l00 = sqrt(Vi[0]); l10 = Vi[3] / l00; l20 = Vi[6] / l00;
        const double d1 = Vi[4] - l10 * l10;
        if (d1 > 0) {
          l11 = sqrt(d1); l21 = (Vi[7] - l20 * l10) / l11;
          const double d2 = Vi[8] - l20 * l20 - l21 * l21;
          if (d2 > 0) l22 = sqrt(d2);
        }
      }
      const bool freep = active && k > 0;
      const int base = 6 * (k - 1);
      const unsigned long long flags = __ballot(active && lane == first);
      const int lm_local = __popcll(flags & ((2ull << lane) - 1ull)) - 1;
      double* sU = sImg + (lm_local & (MF_COPIES - 1)) * img;
      double* sGred = sU + F * 21;
      double* sGc = sGred + n;
      double Z[18];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double w[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) w[b] = freep ? (Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b]) * s[b] : 0.0;
        Z[3 * a] = w[0] * l00 + w[1] * l10 + w[2] * l20;
        Z[3 * a + 1] = w[1] * l11 + w[2] * l21;
        Z[3 * a + 2] = w[2] * l22;
        if (freep) {
          // g_red part: -(W s) V^-1 (g_p s)
          double y = 0;
#pragma unroll
          for (int b = 0; b < 3; ++b) y += (w[0] * Vi[b] + w[1] * Vi[3 + b] + w[2] * Vi[6 + b]) * gps[b];
          atomicAdd(&sGred[base + a], -y);
          atomicAdd(&sGc[base + a], Jc[a] * r[0] + Jc[6 + a] * r[1]);
        }
      }
      if (freep) {
        double* u = sU + (k - 1) * 21;
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) atomicAdd(&u[q++], Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b]);
      }
      double* z = sZ + (size_t)(wave * 64 + lane) * 18;
#pragma unroll
      for (int i = 0; i < 18; ++i) z[i] = Z[i];
      nlm = __popcll(flags);
      if (freep) {
        sTbl[(wave * 64 + lm_local) * MF_TBL_ROW + (k - 1)] = (uint8_t)lane;
        atomicOr(&sMask[wave * 64 + lm_local], (1u << (base >> 4)) | (1u << ((base + 5) >> 4)));
      }
    }
    if (lane == 0) sNlm[wave] = nlm;
    __syncthreads();
    nxt = fetch(grp + gridDim.x);
    // ---- phase 2: wave = tile row --------------------------------------------------------------
    for (int c = 0; c < MF_WAVES; ++c) {
      const int nl = sNlm[c];
      const uint8_t* tb = sTbl + c * 64 * MF_TBL_ROW;
      const double* zc = sZ + (size_t)c * 64 * 18;
      // landmarks of this chunk whose poses touch my tile row, as a wave-uniform bit set
      const uint32_t mvec = lane < nl ? sMask[c * 64 + lane] : 0u;
      unsigned long long todo = __ballot((mvec >> wave) & 1u);
      // two-stage pipeline: the pose->lane bytes of the next landmark are in flight while the current
      // landmark's operands are read and multiplied; no lane-divergent control flow in the loop
      int srcN[5];
      int lmi = -1;
      if (todo) {
        lmi = __builtin_ctzll(todo); todo &= todo - 1ull;
        const uint8_t* tl = tb + lmi * MF_TBL_ROW;
#pragma unroll
        for (int d = 0; d < 5; ++d) srcN[d] = tl[tix[d]];
      }
      while (lmi >= 0) {
        const uint32_t mask = __builtin_amdgcn_readlane(mvec, lmi);
        int src[5];
        double op[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) { src[d] = srcN[d]; op[d] = zc[(src[d] & 63) * 18 + go_[d]]; }
        lmi = -1;
        if (todo) {
          lmi = __builtin_ctzll(todo); todo &= todo - 1ull;
          const uint8_t* tl = tb + lmi * MF_TBL_ROW;
#pragma unroll
          for (int d = 0; d < 5; ++d) srcN[d] = tl[tix[d]];
        }
#pragma unroll
        for (int d = 0; d < 5; ++d) op[d] = src[d] != 255 ? op[d] : 0.0;
        const double a_op = -op[0];
#pragma unroll
        for (int d = 0; d < 5; ++d) {
          if (d == 4 && wave >= 4) continue;  // tile (w, w+4) is kept by w < 4 only
          if (!((mask >> ((wave + d) & 7)) & 1u)) continue;
          acc[d] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, op[d], acc[d], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // ---- flush: tiles -> S (each unordered pose pair once; diagonal pose blocks in full) -----------
  double* S = P.pay1;
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    if (d == 4 && wave >= 4) continue;
    const int cc = (wave + d) & 7;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int row = 16 * wave + (lane >> 4) + 4 * rg, col = 16 * cc + (lane & 15);
      const double v = acc[d][rg];
      if (v == 0.0 || row >= n || col >= n) continue;
      const int pr = row / 6, pc = col / 6;
      if (d == 0) {
        if (pr <= pc) atomicAdd(&S[(size_t)row * n + col], v);
      } else if (pr == pc) {
        atomicAdd(&S[(size_t)row * n + col], v);
        atomicAdd(&S[(size_t)col * n + row], v);
      } else if (pr < pc) {
        atomicAdd(&S[(size_t)row * n + col], v);
      } else {
        atomicAdd(&S[(size_t)col * n + row], v);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { lcost += __shfl_xor(lcost, off); lgp2 += __shfl_xor(lgp2, off); }
  if (lane == 0) { atomicAdd(&sAcc[0], lcost); atomicAdd(&sAcc[1], lgp2); }
  __syncthreads();
  double* gGred = S + (size_t)n * n;
  double* gGc = gGred + n;
  double* gDU = gGc + n;
  for (int i = threadIdx.x; i < F * 21; i += blockDim.x) {
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < MF_COPIES; ++c) v += sImg[c * img + i];
    if (v == 0.0) continue;
    const int p = i / 21;
    int q = i - 21 * p, a = 0;
    while (q >= 6 - a) { q -= 6 - a; ++a; }
    const int b = a + q, ra = 6 * p + a, rb = 6 * p + b;
    atomicAdd(&S[(size_t)ra * n + rb], v);
    if (a != b) atomicAdd(&S[(size_t)rb * n + ra], v);
    else atomicAdd(&gDU[ra], v);
  }
  for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) {  // g_red then g_c: adjacent in the image and in the payload
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < MF_COPIES; ++c) v += sImg[c * img + F * 21 + i];
    if (v != 0.0) atomicAdd(&gGred[i], v);
  }
  if (threadIdx.x < 2) atomicAdd(&gDU[n + threadIdx.x], sAcc[threadIdx.x]);
}

// ----------------------------------------------------------------------------- host side
struct svo_ba {
  svo_ctx* ctx = nullptr;
  svo_camera_info cam{};
  svo_ba_options opt{};
  int window_size = 5, max_landmarks = 0, max_obs = 0, max_poses = 0;
  svo_allreduce_fn allreduce = nullptr;   // in-process emulation of the collective (tests)
  void* allreduce_user = nullptr;
  void* comm = nullptr;                   // ncclComm_t of a sharded run (svo_ba_set_comm)
  // device problem
  BaDev d;
  size_t cap_points = 0, cap_obs = 0, cap_pay1 = 0;
  size_t cap_part1 = 0, cap_part2 = 0, cap_res = 0;   // granules of the partial store / of ba_lm_kernel's totals
  double* d_res = nullptr;          // ba_lm_kernel: the E wire totals of the running iteration (+ the posted clock), tagged granules
  unsigned long long tag_seq = 0;   // host-driven launches: one tag per op (ba_next_tag)
  int tab_max_words = 0;            // largest chunk table of the loaded problem (u16 words)
  double flops_iter = 0, bytes_iter = 0;                       // SURVEY 8(d) algorithmic f64 flops / bytes of ONE LM iteration of the loaded problem
  double acc_flops = 0, acc_bytes = 0, acc_solves = 0, acc_iters = 0;  // ... summed over the solves since the last svo_ba_work(reset)
  std::vector<double> h_poses;            // K x 7 current poses (updated in place by svo_lm_solve)
  int n_points = 0;
  uint8_t* d_arena = nullptr;     // all per-solve inputs in one allocation: one H2D per solve
  uint8_t* h_arena = nullptr;     // pinned staging image of the arena
  size_t arena_cap = 0;
  hipStream_t stream = nullptr;   // BA has its own stream so a solve can overlap the tracker's kernels
  bool stream_owned = true;       // false: adopted from the caller (svo_ba_use_stream: the lanes of a pipeline group share the group's lines)
  double* d_pay = nullptr;        // [payload2 (PAY2_SLOTS) | payload1]: one buffer, one all-reduce
  double* d_step = nullptr;       // device copy of [dc | candidate poses] (bulk / sharded runs)
  double* d_bctl = nullptr;       // bulk path, device-side step control: LM state (BC_*) | Jacobi scales of the pose columns
  double* h_bstat = nullptr;      // ... pinned: ring of status records (BS_DOUBLES each) | initial state image
  int bulk_ctl = -1;              // svo_ba_set_bulk_control: -1 automatic (on whenever eligible), 0 host-driven, 1 on
  // pinned block, fixed layout (never depends on the window size): [flag word (64 B) | step: dc + candidate poses |
  // payload2 (PAY2_SLOTS) | payload1]
  uint8_t* h_pin = nullptr;
  size_t pin_bytes = 0;
  int* h_flag = nullptr;
  double* h_step = nullptr;
  bool res_export = false;       // the device-resident solve delivers the landmarks into pinned memory
  bool host_points_valid = false;  // ... and did: h_arena + arena_pts_off holds the solved landmarks
  size_t arena_pts_off = 0, arena_bytes = 0;
  bool arena_dirty = false;      // h_arena holds a problem image that is not on the device yet
  double* h_out_points = nullptr;  // pinned, GPU-written only
  FusedAdmission res_admission;  // ba_lm_kernel's workgroups, admitted for the duration of a solve
  // device-resident solve (ba_lm_kernel): counter block, pinned result block, state of the launch in flight
  unsigned* d_lmc = nullptr;
  unsigned* d_lmdbg = nullptr;   // SVO_BA_TRACE: last command of every workgroup of ba_lm_kernel
  double* h_result = nullptr;    // pinned [LMR_DOUBLES | poses 7 Kmax], inside h_pin
  int device_lm = -1;            // svo_ba_set_device_lm: -1 automatic, 0 never, 1 whenever eligible
  int lm_form = -1;              // svo_ba_set_solve_form: which device-resident form — 0 wide (ba_lm_kernel: many workgroups, lowest latency), 1 compact (ba_lm_compact_kernel: one workgroup, smallest footprint), -1 automatic (SVO_BA_FORM, else wide)
  bool lm_compact_inflight = false;  // the launch in flight is the compact form (nothing admitted, no counter)
  int lm_penalty = 0;            // solves left that avoid the wide form after one of its launches gave up
  long wide_launches = 0;        // (test hook SVO_BA_TEST_GIVEUP counts them)
  long fallbacks = 0;            // device-resident solves that gave up and were re-run (svo_lm_stats.fallbacks of the last solve: 0 / 1)
  bool lm_inflight = false;      // a ba_lm_kernel has been launched and not yet joined
  hipStream_t lm_stream = nullptr;  // ... on this stream
  LmLane* h_lane = nullptr;      // pinned launch record of this adjuster's solve
  unsigned lm_base = 0;            // where the last solve left the delivery counter
  bool lm_counters_dirty = false;  // the last kernel did not leave through its clean exit: zero the counter before the next launch
  bool lm_have_base = false;       // lm_base describes the counter (false until the first clean solve)
  bool arena_partial = false;      // the last solve read the problem image from h_arena in place: d_arena lacks the tables, h_arena the solved state
  size_t arena_cpts_off = 0, arena_p0_off = 0, arena_p1_off = 0;
  std::chrono::steady_clock::time_point lm_t0;
  double* h_pay = nullptr;
  // current / candidate buffers of the running solve (swapped on every accepted step)
  double *cur_points = nullptr, *cand_points = nullptr, *cur_poses = nullptr, *cand_poses = nullptr;
  bool have_scale = false;
  std::vector<double> mirror;     // bulk modes: S with the pair blocks mirrored
  // sliding-window graph (BundleAdjuster state, host side; ids sequential — SURVEY C-3)
  struct Obs { float u, v; int64_t id; };
  struct PoseVar { double pose[7]; std::vector<Obs> obs; };
  std::deque<PoseVar> window;
  std::vector<double> feat_pos;  // 3 per feature id
  bool new_frame_added = false;
  std::vector<int64_t> solve_lm_ids;
  bool upload_has_ids = false;   // solve_lm_ids describes the landmarks of the problem being uploaded (svo_ba_solve_prepare)
  float4* store = nullptr; unsigned store_mask = 0;  // device-resident landmark store of the stream (svo_ba_attach_store), keyed by feature id
  float4* h_store_stage = nullptr; unsigned* h_store_slot = nullptr; size_t store_stage_cap = 0;  // pinned staging of the scatter behind a host-driven solve
  std::vector<double> s_poses, s_points, s_uv, s_out_pts;   // per-solve scratch of svo_ba_solve (kept: no allocation per keyframe)
  std::vector<int32_t> s_op, s_oj;
  std::vector<int32_t> u_lm_start, u_chunks, u_cnt;  // scratch of ba_upload
  std::vector<uint16_t> u_tab; std::vector<uint32_t> u_tab_off;  // the chunk tables of the loaded problem
  unsigned* d_arrive = nullptr; unsigned arrive_total = 0; int seq = 0;
  bool upload_pending = false;  // H2D of the problem image enqueued, not yet known complete
  bool mfma_ok = false;   // bulk problem eligible for ba_linearize_mfma_kernel (n <= 128, one observation per (landmark, pose))
  svo_lm_stats stats{};
  // SVO_TIMING accumulators
  double lm_tp[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double lm_wg_min[7] = {0, 0, 0, 0, 0, 0, 0}, lm_wg_mean[7] = {0, 0, 0, 0, 0, 0, 0}, lm_wg_max[7] = {0, 0, 0, 0, 0, 0, 0};  // SVO_BA_TRACE: spread of the per-workgroup split
  double lm_t_wait = 0, lm_t_ctl = 0, lm_t_body = 0, lm_t_total = 0; long lm_n = 0, lm_iters = 0, lm_same = 0, lm_used = 0, lm_steps = 0, lm_lins = 0;
  double t_lin = 0, t_step = 0, t_upload = 0, t_total = 0, t_prep = 0, t_read = 0; long n_lin = 0, n_step = 0, n_solves = 0, n_spec = 0, n_hit = 0;
};

static int ba_alloc(svo_ba* ba) {
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  const int Kmax = ba->max_poses, nmax = 6 * (Kmax - 1);
  ba->cap_points = ba->max_landmarks; ba->cap_obs = ba->max_obs;
  ba->cap_pay1 = std::max((size_t)nmax * nmax + 3 * (size_t)nmax + 2 + 2, (size_t)(18 * (Kmax - 1) * Kmax + 33 * (Kmax - 1) + 2));  // payload1 (+ the [elapsed, ranks] tail of the device-side step control), or the wire totals
  const size_t step_doubles = (size_t)(nmax > 0 ? nmax : 1) + 7 * (size_t)Kmax;
#define A(ptr, T, cnt) SVO_HIP_CHECK(ctx, hipMalloc((void**)&(ptr), sizeof(T) * (size_t)(cnt)))
  A(d.sp, double, 3 * ba->cap_points);
  A(ba->d_pay, double, PAY2_SLOTS + ba->cap_pay1);
  A(ba->d_step, double, step_doubles);
  A(ba->d_bctl, double, BC_WORDS + (nmax > 0 ? nmax : 1));
  SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&ba->h_bstat, sizeof(double) * ((size_t)BULK_RING * BS_DOUBLES + BC_WORDS), hipHostMallocCoherent));
  memset(ba->h_bstat, 0, sizeof(double) * ((size_t)BULK_RING * BS_DOUBLES + BC_WORDS));
  // The counter block of the device-resident solve: ordinary device memory (fine-grained memory LOST atomic increments
  // when it was tried in round 3: 38 workgroups had added to a counter that read 37).  The granule stores (partials, totals)
  // are sized by the problem: ba_ensure_partials.
  A(ba->d_lmc, unsigned, LMC_WORDS);
  SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&ba->h_lane, sizeof(LmLane), hipHostMallocCoherent));
  if (getenv("SVO_BA_TRACE")) A(ba->d_lmdbg, unsigned, 16 * 4096);
  SVO_HIP_CHECK(ctx, hipMemset(ba->d_lmc, 0, LMC_WORDS * sizeof(unsigned)));
  A(ba->d_arrive, unsigned, 16);  // [arrival counter | pad | chained decision: 2 doubles at +8 bytes | +32 bytes: pass-A done counter, pass-B arrival counter, decision post]
  SVO_HIP_CHECK(ctx, hipMemset(ba->d_arrive, 0, 16 * sizeof(unsigned)));
#undef A
  {
    // pre-size the per-solve input arena and the pair-block store for window-shaped problems (every landmark seen
    // at most once per pose) so that the hot path never allocates; bulk problems beyond this grow lazily
    const size_t M = ba->cap_obs, Kc = (size_t)Kmax;
    const size_t pairs = M * (Kc + 1) / 2 + 64;  // upper bound of the chunk tables' pair entries (2 bytes each)
    // per-slot arrays are padded to whole waves: a chunk closes when the next landmark would not fit, i.e. it holds more
    // than 64 - max(landmark length) observations; 2 M + 64 slots bound it for landmark lengths <= 32
    const size_t slots = 2 * M + 64;
    const size_t est = 16 * 3 * ba->cap_points + (16 + 16) * slots + 2 * pairs + (slots / 64 + 2) * (2 * (Kc * Kc + 2 * Kc + 4) + 64 + 4) + 2 * 56 * Kc + 16 * 256;
    if (est < ((size_t)512 << 20)) {
      ba->arena_cap = est;
      SVO_HIP_CHECK(ctx, hipMalloc((void**)&ba->d_arena, ba->arena_cap));
      SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&ba->h_arena, ba->arena_cap, hipHostMallocDefault));
    }
  }
  {
    // SVO_BA_CU_SHARE=n (n = 8 on MI355X: one shader engine of every XCD; see include/svo.h): window-sized adjusters
    // run on their own n CUs of every 32 and the context's stream on the others, so that the LM loop's small dependent
    // kernels never queue behind other stereo streams' wide LK launches.  Bulk-sized adjusters keep the whole GPU.
    const char* e = getenv("SVO_BA_CU_SHARE");
    const int nres = e ? atoi(e) : 0;
    if (nres > 0 && nres < 32 && ba->max_obs <= 100000) {
      uint32_t mask[8];
      for (int i = 0; i < 8; ++i) mask[i] = (1u << nres) - 1u;
      SVO_HIP_CHECK(ctx, hipExtStreamCreateWithCUMask(&ba->stream, 8, mask));
      g_ba_cu_share = nres;  // process-wide like the environment variable; ba_fused_budget() recomputes a device's budget when it changes
    } else {
      SVO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ba->stream, hipStreamNonBlocking));
    }
  }
  {
    // the granule stores for window-shaped problems up front (<= 128 chunks: one group per chunk), so that the hot path never
    // allocates; larger problems grow them in ba_ensure_partials
    const size_t Fm = (size_t)Kmax - 1, Em = ((18 * Fm * (Fm + 1) + 33 * Fm + 2) + 7) & ~(size_t)7;
    if (Em * 128 * 16 <= ((size_t)64 << 20)) {
      SVO_HIP_CHECK(ctx, hipMalloc((void**)&d.part1, 16 * Em * 128));
      SVO_HIP_CHECK(ctx, hipMemsetAsync(d.part1, 0, 16 * Em * 128, ba->stream));
      ba->cap_part1 = Em * 128;
      SVO_HIP_CHECK(ctx, hipMalloc((void**)&d.part2, 16 * 2 * 4 * 128));
      SVO_HIP_CHECK(ctx, hipMemsetAsync(d.part2, 0, 16 * 2 * 4 * 128, ba->stream));
      ba->cap_part2 = 2 * 4 * 128;
      SVO_HIP_CHECK(ctx, hipMalloc((void**)&ba->d_res, 16 * (Em + 1)));
      SVO_HIP_CHECK(ctx, hipMemsetAsync(ba->d_res, 0, 16 * (Em + 1), ba->stream));
      ba->cap_res = Em + 1;
    }
    // every fill of this function is complete before the adjuster is handed out (its streams are non-blocking: see ba_ensure_partials)
    SVO_HIP_CHECK(ctx, hipDeviceSynchronize());
  }
  // pinned block: [completion word 64 B | 128 B reserved | step: dc, candidate poses (, current poses) | payload]
  const size_t pin_step_doubles = step_doubles + 7 * (size_t)Kmax;
  // ... | landmarks delivered by ba_lm_kernel (written by the GPU only: the CPU never holds these lines dirty)]
  ba->pin_bytes = 64 + 128 + sizeof(double) * (pin_step_doubles + PAY2_SLOTS + ba->cap_pay1 + 8 + 3 * ba->cap_points + LMR_DOUBLES + 7 * (size_t)Kmax + 8);
  SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&ba->h_pin, ba->pin_bytes, hipHostMallocCoherent));  // fine-grained: see reduce_publish
  memset(ba->h_pin, 0, ba->pin_bytes);  // the flag word is compared by equality with a sequence number: never start from recycled bytes
  ba->h_flag = reinterpret_cast<int*>(ba->h_pin);
  ba->h_step = reinterpret_cast<double*>(ba->h_pin + 64 + 128);
  ba->h_pay = ba->h_step + pin_step_doubles;
  ba->h_out_points = ba->h_pay + ((PAY2_SLOTS + ba->cap_pay1 + 7) & ~(size_t)7);
  ba->h_result = ba->h_out_points + ((3 * ba->cap_points + 7) & ~(size_t)7);
  d.pay2 = ba->d_pay;
  d.pay1 = ba->d_pay + PAY2_SLOTS;
  return SVO_OK;
}

extern "C" void svo_ba_default_options(svo_ba_options* o) {
  if (!o) return;
  o->max_iterations = 50;                           // Ceres default
  o->max_time_s = svo_ref::BA_MAX_SOLVER_TIME_S;    // src/bundle_adjuster.cpp:11
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_radius = 1e4;
  o->max_features = svo_ref::MAX_FEATURES;          // src/bundle_adjuster.hpp:75
  o->accumulation = SVO_BA_ACC_AUTO;
}

extern "C" int svo_ba_create(svo_ctx* ctx, svo_ba** out, int window_size, const svo_camera_info* cam,
                             const svo_ba_options* opt, int max_landmarks, int max_observations) {
  if (!ctx || !out || !cam) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_REQUIRE(ctx, window_size >= 1 && window_size <= 64, "ba_create: window size must be 1..64");
  SVO_REQUIRE(ctx, max_landmarks >= 1 && max_observations >= 1, "ba_create: capacities must be positive");
  svo_ba* ba = new svo_ba();
  ba->ctx = ctx;
  ba->cam = *cam;
  if (opt) ba->opt = *opt; else svo_ba_default_options(&ba->opt);
  ba->window_size = window_size;
  ba->max_poses = window_size + 1 > 2 ? window_size + 1 : 2;
  if (ba->max_poses > 64) ba->max_poses = 64;
  ba->max_landmarks = max_landmarks;
  ba->max_obs = max_observations;
  int rc = ba_alloc(ba);
  if (rc) { svo_ba_destroy(ba); return rc; }
  *out = ba;
  return SVO_OK;
}

extern "C" void svo_ba_destroy(svo_ba* ba) {
  if (!ba) return;
  BaDev& d = ba->d;
  if (getenv("SVO_TIMING") && ba->n_lin)
    fprintf(stderr, "[svo ba] %ld solves: pass A alone %.1f us x %ld, step (pass B + speculative pass A) %.1f us x %ld, speculation %ld/%ld hit, "
                    "per solve: gather %.1f us, upload %.1f us, upload + LM %.1f us, read-back %.1f us\n", ba->n_solves, 1e3 * ba->t_lin / ba->n_lin, ba->n_lin,
            ba->n_step ? 1e3 * ba->t_step / ba->n_step : 0.0, ba->n_step, ba->n_hit, ba->n_spec, 1e3 * ba->t_prep / std::max(ba->n_solves, 1l),
            1e3 * ba->t_upload / std::max(ba->n_solves, 1l), 1e3 * ba->t_total / std::max(ba->n_solves, 1l), 1e3 * ba->t_read / std::max(ba->n_solves, 1l));
  if (getenv("SVO_TIMING") && ba->lm_n)
    fprintf(stderr, "[svo ba] device-resident solves: %ld, %.1f LM iterations each; per solve (workgroup 0, us): total %.1f = waiting for the passes %.1f + step control %.1f "
                    "+ own share of the passes %.1f; steps %ld (same sweep %ld, next linearisation used %ld), stand-alone linearisations %ld\n", ba->lm_n, (double)ba->lm_iters / ba->lm_n, 1e-2 * ba->lm_t_total / ba->lm_n, 1e-2 * ba->lm_t_wait / ba->lm_n,
            1e-2 * ba->lm_t_ctl / ba->lm_n, 1e-2 * ba->lm_t_body / ba->lm_n, ba->lm_steps, ba->lm_same, ba->lm_used, ba->lm_lins);
  if (getenv("SVO_TIMING") && ba->lm_n && ba->d_lmdbg)
    fprintf(stderr, "[svo ba]   per LM iteration (workgroup 0, us): pass B %.2f, radius-free part of pass A %.2f, collecting payload2 + decision %.2f, rest of pass A %.2f, "
                    "own slice of level 2 %.2f (of which waiting for everybody's partials %.2f) | collecting the totals + assembly %.2f, system build %.2f, Cholesky %.2f, step tail %.2f\n", 1e-2 * ba->lm_tp[0] / ba->lm_iters, 1e-2 * ba->lm_tp[1] / ba->lm_iters,
            1e-2 * ba->lm_tp[2] / ba->lm_iters, 1e-2 * (ba->lm_tp[3] + ba->lm_tp[10]) / ba->lm_iters, 1e-2 * ba->lm_tp[4] / ba->lm_iters, 1e-2 * ba->lm_tp[5] / ba->lm_iters,
            1e-2 * ba->lm_tp[6] / ba->lm_iters, 1e-2 * ba->lm_tp[7] / ba->lm_iters, 1e-2 * ba->lm_tp[8] / ba->lm_iters, 1e-2 * ba->lm_tp[9] / ba->lm_iters);
  if (getenv("SVO_TIMING") && ba->lm_n && ba->d_lmdbg)
    fprintf(stderr, "[svo ba]   rest of pass A, chained steps (workgroup 0's first wavefront, us per LM iteration): lanes' arithmetic behind the decision %.2f, Schur owners %.2f, "
                    "per-pose owners + posting the partials %.2f\n", 1e-2 * ba->lm_tp[10] / ba->lm_iters, 1e-2 * ba->lm_tp[12] / ba->lm_iters, 1e-2 * (ba->lm_tp[3] - ba->lm_tp[12]) / ba->lm_iters);
  if (getenv("SVO_TIMING") && ba->lm_n && ba->d_lmdbg) {
    static const char* nm[7] = {"pass B", "radius-free part of pass A", "collecting payload2", "rest of pass A", "own slice of level 2", "waiting for the partials", "collecting the totals"};
    for (int sl = 0; sl < 7; ++sl)
      fprintf(stderr, "[svo ba]   over the workgroups of a solve, per LM iteration (us): %-28s min %.2f mean %.2f max %.2f\n", nm[sl], 1e-2 * ba->lm_wg_min[sl] / ba->lm_iters,
              1e-2 * ba->lm_wg_mean[sl] / ba->lm_iters, 1e-2 * ba->lm_wg_max[sl] / ba->lm_iters);
  }
  if (ba->stream) (void)hipStreamSynchronize(ba->stream);
  void* ptrs[] = {ba->d_res, ba->d_lmdbg, ba->d_lmc, ba->d_arrive, ba->d_pay, ba->d_step, ba->d_bctl, d.sp, d.part1, d.part2, ba->d_arena};
  if (ba->h_bstat) (void)hipHostFree(ba->h_bstat);
  if (ba->h_arena) (void)hipHostFree(ba->h_arena);
  if (ba->h_store_stage) (void)hipHostFree(ba->h_store_stage);
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (ba->h_pin) (void)hipHostFree(ba->h_pin);
  if (ba->h_lane) (void)hipHostFree(ba->h_lane);
  if (ba->stream && ba->stream_owned) (void)hipStreamDestroy(ba->stream);
  delete ba;
}

extern "C" int svo_ba_set_allreduce(svo_ba* ba, svo_allreduce_fn fn, void* user) {
  if (!ba) return SVO_ERR_INVALID;
  ba->allreduce = fn;
  ba->allreduce_user = user;
  return SVO_OK;
}

extern "C" int svo_ba_set_comm(svo_ba* ba, void* nccl_comm) {
  if (!ba) return SVO_ERR_INVALID;
  ba->comm = nccl_comm;
  return SVO_OK;
}

extern "C" int svo_ba_set_device_lm(svo_ba* ba, int mode) {
  if (!ba || mode < -1 || mode > 1) return SVO_ERR_INVALID;
  ba->device_lm = mode;
  return SVO_OK;
}

// The adjuster's own HIP stream (uploads, host-driven solves, the landmark-store scatter) replaced by one of the caller's: a process
// with many adjusters then keeps few streams, and the runtime's stream -> hardware-queue binding (GPU_MAX_HW_QUEUES) stays one to one
// for the streams that carry the work (host/group.cpp).  Not while a solve is in flight.
int svo_ba_use_stream(svo_ba* ba, void* stream) {
  if (!ba || !stream || ba->lm_inflight) return SVO_ERR_INVALID;
  svo_use_device(ba->ctx);
  if (ba->stream) {
    (void)hipStreamSynchronize(ba->stream);
    if (ba->stream_owned) (void)hipStreamDestroy(ba->stream);
  }
  ba->stream = (hipStream_t)stream;
  ba->stream_owned = false;
  return SVO_OK;
}

// host/group.cpp: a pipeline group of `delta` lanes was created (> 0) or destroyed (< 0)
void svo_ba_note_group_lanes(int delta) { g_group_lanes.fetch_add(delta, std::memory_order_relaxed); }

extern "C" int svo_ba_set_solve_form(svo_ba* ba, int form) {
  if (!ba || form < -1 || form > 1) return SVO_ERR_INVALID;
  ba->lm_form = form;
  return SVO_OK;
}

extern "C" int svo_ba_set_bulk_control(svo_ba* ba, int mode) {
  if (!ba || mode < -1 || mode > 1) return SVO_ERR_INVALID;
  ba->bulk_ctl = mode;
  return SVO_OK;
}

extern "C" int svo_ba_last_stats(svo_ba* ba, svo_lm_stats* stats) {
  if (!ba || !stats) return SVO_ERR_INVALID;
  *stats = ba->stats;
  return SVO_OK;
}


// Upload a landmark-major problem (shared by the bulk API and the sliding-window solve).
// Destination index of pose-pair block (ka <= kb) among the F (F + 1) / 2 upper blocks, row-major.
static inline int ba_upper_index(int ka, int kb, int F) { return ka * F - ka * (ka - 1) / 2 + (kb - ka); }

// The granule stores of the loaded problem: partials of pass A (NG x Epad), of pass B (2 x NG x 4), ba_lm_kernel's totals
// (E + 1).  Window-sized problems reach their sizes within a few keyframes; nothing is allocated per solve afterwards.
// Fresh stores are zeroed: no command ever carries tag 0.
static int ba_ensure_partials(svo_ba* ba) {
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  auto grow = [&](double** p, size_t* cap, size_t granules) -> int {
    if (granules <= *cap) return SVO_OK;
    if (ba->upload_pending || ba->lm_inflight) SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
    if (*p) SVO_HIP_CHECK(ctx, hipFree(*p));
    *p = nullptr;
    const size_t want = granules + granules / 2 + 64;
    SVO_HIP_CHECK(ctx, hipMalloc((void**)p, 16 * want));
    // the fill must be COMPLETE before any solve posts into the store: the adjuster's streams are non-blocking streams, which
    // a hipMemset on the null stream is not ordered with (a fill that landed behind the first pass wiped its tags: the
    // reduction then waited three seconds for them and the solve returned zeros — seen once in a full test run)
    SVO_HIP_CHECK(ctx, hipMemsetAsync(*p, 0, 16 * want, ba->stream));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
    *cap = want;
    return SVO_OK;
  };
  int rc = grow(&d.part1, &ba->cap_part1, (size_t)d.Epad * (size_t)(d.NG > 0 ? d.NG : 1));
  if (!rc) rc = grow(&d.part2, &ba->cap_part2, 2 * 4 * (size_t)(d.NG > 0 ? d.NG : 1));
  if (!rc) rc = grow(&ba->d_res, &ba->cap_res, (size_t)d.E + 1);
  return rc;
}

static int ba_upload_checked(svo_ba* ba, int K, const double* poses7, int npts, const double* points3, int M,
                     const int32_t* op, const int32_t* oj, const double* uv) {
  svo_ctx* ctx = ba->ctx;
  SVO_REQUIRE(ctx, K >= 1 && K <= ba->max_poses, "ba: pose count outside the window capacity");
  SVO_REQUIRE(ctx, npts >= 0 && (size_t)npts <= ba->cap_points && M >= 0 && (size_t)M <= ba->cap_obs, "ba: problem exceeds capacity");
  BaDev& d = ba->d;
  if (ba->upload_pending) { SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream)); ba->upload_pending = false; }
  d.K = K; d.n = 6 * (K - 1); d.M = M; d.f = ba->cam.focal; d.cx = ba->cam.cx; d.cy = ba->cam.cy;
  ba->n_points = npts;
  // CSR over landmark index + wave chunks (<= 64 observations, whole landmarks).  All scratch vectors are members:
  // a keyframe's solve allocates nothing.
  std::vector<int32_t>&lm_start = ba->u_lm_start, &chunks = ba->u_chunks;
  lm_start.assign((size_t)npts + 1, 0);
  chunks.clear();
  for (int o = 0; o < M; ++o) {
    SVO_REQUIRE(ctx, oj[o] >= 0 && oj[o] < npts && op[o] >= 0 && op[o] < K, "ba: observation index out of range");
    SVO_REQUIRE(ctx, o == 0 || oj[o] >= oj[o - 1], "ba: observations must be sorted by landmark");
    lm_start[oj[o] + 1]++;
  }
  bool has_empty_landmark = false;
  for (int j = 0; j < npts; ++j) {
    SVO_REQUIRE(ctx, lm_start[j + 1] <= 64, "ba: a landmark has more than 64 observations");
    has_empty_landmark |= lm_start[j + 1] == 0;
    lm_start[j + 1] += lm_start[j];
  }
  // chunks (declared with the summation order, oracle/ora_ba.cpp): whole landmarks, greedily, at most 64 observations each
  chunks.push_back(0);
  int cur = 0;
  for (int j = 0; j < npts; ++j) {
    const int len = lm_start[j + 1] - lm_start[j];
    if (len == 0) continue;
    if (cur + len > 64) { chunks.push_back(lm_start[j]); cur = 0; }
    cur += len;
  }
  chunks.push_back(M);
  d.C = (int)chunks.size() - 1;
  d.L = npts;
  {
    // 466 flops per observation + per landmark 50 + 144 L + 216 L (L + 1) / 2; 24 B per observation, 48 B per landmark, 56 B per pose
    double fl = 466.0 * M;
    int n_seen = 0;
    for (int j = 0; j < npts; ++j) {
      const double Lj = lm_start[j + 1] - lm_start[j];
      if (Lj > 0) { fl += 50.0 + 144.0 * Lj + 108.0 * Lj * (Lj + 1.0); ++n_seen; }
    }
    ba->flops_iter = fl;
    ba->bytes_iter = 24.0 * M + 48.0 * n_seen + 56.0 * K;
  }
  hipStream_t st = ba->stream;
  std::vector<uint16_t>& tab = ba->u_tab;
  std::vector<uint32_t>& tab_off = ba->u_tab_off;
  tab.clear(); tab_off.clear();
  ba->tab_max_words = 0;
  {
    const int F = K - 1;
    bool dup = false;
    for (int j = 0; j < npts; ++j) {
      uint64_t seen = 0;
      for (int o = lm_start[j]; o < lm_start[j + 1]; ++o) {
        const uint64_t bit = 1ull << op[o];
        dup |= (seen & bit) != 0;
        seen |= bit;
      }
    }
    // deterministic mode ("chunk order"): G chunks per group, NG groups, E wire elements; the partial store holds E x NG granules
    const int nU = F * (F + 1) / 2;
    d.G = d.C <= 128 ? 1 : (d.C + 127) / 128;
    // partials in the store: one per chunk (the groups of G are applied by the level-2 sums) while that stays below 64 MB —
    // a 1280x720 window: 1,500 chunks x 1,920 elements = 46 MB; config 4 (6,900 chunks x 7,472 elements) keeps one partial per GROUP
    // (and at most 2,048 partials: the level-2 reducers hold all partials of an element in 4,096 doubles of LDS)
    d.CPW = ((size_t)((36 * nU + 33 * F + 2 + 7) & ~7) * (size_t)d.C * 16 <= ((size_t)64 << 20) && d.C <= 2048) ? 1 : d.G;
    d.GS = d.G / d.CPW;
    d.NG = d.C > 0 ? (d.C + d.CPW - 1) / d.CPW : 0;
    d.E = 36 * nU + 33 * F + 2;
    d.Epad = (d.E + 7) & ~7;
    d.det = (size_t)d.Epad * (size_t)(d.NG > 0 ? d.NG : 1) * 16 <= ((size_t)512 << 20) ? 1 : 0;
    if (ba->opt.accumulation == SVO_BA_ACC_ATOMICS || ba->opt.accumulation == SVO_BA_ACC_MFMA) d.det = 0;
    SVO_REQUIRE(ctx, !(ba->opt.accumulation == SVO_BA_ACC_DETERMINISTIC && !d.det), "ba: problem too large for deterministic accumulation");
    ba->mfma_ok = !dup && d.n <= 128 && d.n > 0 && ba->opt.accumulation != SVO_BA_ACC_ATOMICS;
    SVO_REQUIRE(ctx, !(ba->opt.accumulation == SVO_BA_ACC_MFMA && !ba->mfma_ok), "ba: MFMA accumulation needs <= 22 poses and one observation per (landmark, pose)");
    if (d.det) {
      // Chunk tables (ChunkTab): per chunk the pair entries of every UPPER pose-pair block in (landmark, i, t) order — pair
      // (i, t >= i) enters block (k_i, k_t) when k_i <= k_t and, transposed, block (k_t, k_i) when t != i and k_t <= k_i (two
      // observations of one landmark in one pose: both, the direct entry first) — and the lanes of every free pose.
      std::vector<int32_t>& cur = ba->u_cnt;  // fill cursors: nU block lists, then F pose lists
      tab_off.reserve((size_t)d.C + 1);
      tab.reserve((size_t)M * 3 + (size_t)d.C * (size_t)(nU + F + 4));
      for (int c = 0; c < d.C; ++c) {
        const int c0 = chunks[c], c1 = chunks[c + 1];
        cur.assign((size_t)nU + (size_t)F + 2, 0);
        // pass 1: list lengths
        for (int i = c0; i < c1; ++i) {
          const int ki = op[i] - 1;
          if (ki < 0) continue;
          cur[nU + ki]++;
          const int o1 = lm_start[oj[i] + 1];
          for (int t = i; t < o1; ++t) {
            const int kt = op[t] - 1;
            if (kt < 0) continue;
            if (ki <= kt) cur[ba_upper_index(ki, kt, F)]++;
            if (t != i && kt <= ki) cur[ba_upper_index(kt, ki, F)]++;
          }
        }
        tab_off.push_back((uint32_t)tab.size());
        const size_t t0 = tab.size();
        int n_ent = 0, n_free = 0;
        for (int q = 0; q < nU; ++q) { const int len = cur[q]; cur[q] = n_ent; tab.push_back((uint16_t)n_ent); n_ent += len; }
        tab.push_back((uint16_t)n_ent);
        for (int k = 0; k < F; ++k) { const int len = cur[nU + k]; cur[nU + k] = n_free; tab.push_back((uint16_t)n_free); n_free += len; }
        tab.push_back((uint16_t)n_free);
        const size_t e_at = tab.size();
        tab.resize(e_at + (size_t)n_ent + (size_t)((n_free + 1) / 2), 0);
        uint16_t* ent = tab.data() + e_at;
        uint8_t* plane = reinterpret_cast<uint8_t*>(ent + n_ent);
        // pass 2: fill, every list in (landmark, i, t) / lane order
        for (int i = c0; i < c1; ++i) {
          const int ki = op[i] - 1;
          if (ki < 0) continue;
          plane[cur[nU + ki]++] = (uint8_t)(i - c0);
          const int o1 = lm_start[oj[i] + 1];
          for (int t = i; t < o1; ++t) {
            const int kt = op[t] - 1;
            if (kt < 0) continue;
            if (ki <= kt) ent[cur[ba_upper_index(ki, kt, F)]++] = (uint16_t)((i - c0) | ((t - c0) << 8));
            if (t != i && kt <= ki) ent[cur[ba_upper_index(kt, ki, F)]++] = (uint16_t)((i - c0) | 0x80 | ((t - c0) << 8));
          }
        }
        if ((tab.size() - t0) & 1) tab.push_back(0);  // tables start on even u16 offsets (read as 32-bit words)
        ba->tab_max_words = std::max(ba->tab_max_words, (int)(tab.size() - t0));
      }
      tab_off.push_back((uint32_t)tab.size());
    }
  }
  // ---- one pinned staging image, one H2D: [points | points (candidate copy) | per-slot records | per-slot uv |
  //      chunk tables | table offsets | poses x 2]
  const size_t nslots = (size_t)d.C * 64;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = 0;
  const size_t o_pts = off; off = al(off + sizeof(double) * 3 * (size_t)npts);
  ba->arena_pts_off = o_pts; ba->host_points_valid = false;
  const size_t o_cpts = off; off = al(off + sizeof(double) * 3 * (size_t)npts);
  const size_t o_rec = off; off = al(off + sizeof(int4) * nslots);
  const size_t o_uv = off; off = al(off + sizeof(double) * 2 * nslots);
  const size_t o_tab = off; off = al(off + sizeof(uint16_t) * tab.size());
  const size_t o_toff = off; off = al(off + sizeof(uint32_t) * tab_off.size());
  const bool with_keys = ba->upload_has_ids && ba->store && (int)ba->solve_lm_ids.size() == npts;
  const size_t o_key = off; if (with_keys) off = al(off + sizeof(unsigned) * (size_t)npts);
  const size_t o_p0 = off; off = al(off + sizeof(double) * 7 * (size_t)K);
  const size_t o_p1 = off; off = al(off + sizeof(double) * 7 * (size_t)K);
  const size_t total = off;
  if (total > ba->arena_cap) {
    if (ba->d_arena) (void)hipFree(ba->d_arena);
    if (ba->h_arena) (void)hipHostFree(ba->h_arena);
    ba->d_arena = nullptr; ba->h_arena = nullptr;
    ba->arena_cap = total + total / 4 + 4096;
    SVO_HIP_CHECK(ctx, hipMalloc((void**)&ba->d_arena, ba->arena_cap));
    SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&ba->h_arena, ba->arena_cap, hipHostMallocDefault));
  }
  uint8_t* h = ba->h_arena;
  if (npts) { memcpy(h + o_pts, points3, sizeof(double) * 3 * (size_t)npts); memcpy(h + o_cpts, points3, sizeof(double) * 3 * (size_t)npts); }
  {
    int4* rec = reinterpret_cast<int4*>(h + o_rec);
    double* suv = reinterpret_cast<double*>(h + o_uv);
    for (int c = 0; c < d.C; ++c) {
      const int c0 = chunks[c], c1 = chunks[c + 1];
      for (int lane = 0; lane < 64; ++lane) {
        const size_t slot = (size_t)c * 64 + lane;
        const int o = c0 + lane;
        if (o < c1) {
          const int j = oj[o];
          rec[slot] = int4{op[o], j, lm_start[j] - c0, lm_start[j + 1] - lm_start[j]};
          suv[2 * slot] = uv[2 * (size_t)o]; suv[2 * slot + 1] = uv[2 * (size_t)o + 1];
        } else {
          rec[slot] = int4{-1, 0, lane, 0};
          suv[2 * slot] = 0.0; suv[2 * slot + 1] = 0.0;
        }
      }
    }
  }
  if (!tab.empty()) memcpy(h + o_tab, tab.data(), sizeof(uint16_t) * tab.size());
  if (!tab_off.empty()) memcpy(h + o_toff, tab_off.data(), sizeof(uint32_t) * tab_off.size());
  if (with_keys) { unsigned* k = reinterpret_cast<unsigned*>(h + o_key); for (int j = 0; j < npts; ++j) k[j] = (unsigned)ba->solve_lm_ids[(size_t)j]; }
  uint8_t* D = ba->d_arena;
  d.points = (double*)(D + o_pts); d.cand_points = (double*)(D + o_cpts);
  ba->cur_points = (double*)(D + o_pts); ba->cand_points = (double*)(D + o_cpts);
  ba->cur_poses = (double*)(D + o_p0); ba->cand_poses = (double*)(D + o_p1);
  d.rec = (const int4*)(D + o_rec); d.obs_uv = (const double*)(D + o_uv);
  d.tab = (const uint16_t*)(D + o_tab); d.tab_off = (const uint32_t*)(D + o_toff);
  d.tab_lds_words = (d.det && ba->tab_max_words > 0 && ba->tab_max_words <= 4096) ? ((ba->tab_max_words + 3) & ~3) : 0;  // host-driven kernels: the chunk's table in LDS (8 KB at most)
  d.lm_key = with_keys ? (const unsigned*)(D + o_key) : nullptr;
  memcpy(h + o_p0, poses7, sizeof(double) * 7 * (size_t)K);
  memcpy(h + o_p1, poses7, sizeof(double) * 7 * (size_t)K);
  ba->h_poses.assign(poses7, poses7 + 7 * (size_t)K);
  d.poses = (double*)(D + o_p0);
  d.cand_poses = (double*)(D + o_p1);
  // the image goes up when the solve knows how: read in place by ba_lm_kernel (no blit, no extra launch in front of it), or
  // by one H2D copy (ba_flush_arena)
  ba->arena_bytes = (total + 15) & ~(size_t)15;
  ba->arena_dirty = true;
  ba->arena_partial = false;
  ba->arena_cpts_off = o_cpts; ba->arena_p0_off = o_p0; ba->arena_p1_off = o_p1;
  (void)has_empty_landmark; (void)st;
  if (d.det) { const int rcp = ba_ensure_partials(ba); if (rcp) return rcp; }
  return SVO_OK;
}

// A rejected problem must not leave a half-committed one behind (dimensions of the new problem over the chunk layout and
// the host copies of the old one): after a failed load the adjuster holds NO problem — svo_ba_solve_problem refuses,
// svo_ba_read_problem copies nothing.
static int ba_upload(svo_ba* ba, int K, const double* poses7, int npts, const double* points3, int M,
                     const int32_t* op, const int32_t* oj, const double* uv) {
  const int rc = ba_upload_checked(ba, K, poses7, npts, points3, M, op, oj, uv);
  if (rc != SVO_OK) {
    ba->d.K = 0; ba->d.n = 0; ba->d.M = 0; ba->d.L = 0; ba->d.C = 0;
    ba->n_points = 0;
    ba->host_points_valid = false;
    ba->arena_dirty = false;
  }
  return rc;
}

// After a solve that read the problem image from pinned memory in place, the device arena holds only the landmark buffers and
// the host image still holds the INITIAL landmarks and poses.  Before anything else uses the loaded problem again (a second
// solve, a host-driven path), the host image takes the solved state and counts as not uploaded.
static int ba_refresh_arena_image(svo_ba* ba) {
  svo_ctx* ctx = ba->ctx;
  const size_t npts = (size_t)ba->n_points;
  if (npts) {
    double* img = reinterpret_cast<double*>(ba->h_arena + ba->arena_pts_off);
    if (ba->host_points_valid) {
      const std::vector<int32_t>& lm = ba->u_lm_start;
      for (size_t j = 0; j < npts; ++j) {
        if (j + 1 < lm.size() && lm[j + 1] == lm[j]) continue;  // never observed: keeps its uploaded value
        img[3 * j] = ba->h_out_points[3 * j]; img[3 * j + 1] = ba->h_out_points[3 * j + 1]; img[3 * j + 2] = ba->h_out_points[3 * j + 2];
      }
    } else {
      std::vector<double> tmp(3 * npts);
      SVO_HIP_CHECK(ctx, hipMemcpy(tmp.data(), ba->d.points, sizeof(double) * 3 * npts, hipMemcpyDeviceToHost));
      const std::vector<int32_t>& lm = ba->u_lm_start;
      for (size_t j = 0; j < npts; ++j) {
        if (j + 1 < lm.size() && lm[j + 1] == lm[j]) continue;
        img[3 * j] = tmp[3 * j]; img[3 * j + 1] = tmp[3 * j + 1]; img[3 * j + 2] = tmp[3 * j + 2];
      }
    }
    memcpy(ba->h_arena + ba->arena_cpts_off, img, sizeof(double) * 3 * npts);
  }
  memcpy(ba->h_arena + ba->arena_p0_off, ba->h_poses.data(), sizeof(double) * 7 * (size_t)ba->d.K);
  memcpy(ba->h_arena + ba->arena_p1_off, ba->h_poses.data(), sizeof(double) * 7 * (size_t)ba->d.K);
  uint8_t* D = ba->d_arena;
  ba->cur_points = (double*)(D + ba->arena_pts_off); ba->cand_points = (double*)(D + ba->arena_cpts_off);
  ba->cur_poses = (double*)(D + ba->arena_p0_off); ba->cand_poses = (double*)(D + ba->arena_p1_off);
  ba->d.points = ba->cur_points; ba->d.cand_points = ba->cand_points; ba->d.poses = ba->cur_poses; ba->d.cand_poses = ba->cand_poses;
  ba->arena_partial = false;
  ba->arena_dirty = true;
  return SVO_OK;
}

// The problem image -> device by one H2D copy on the adjuster's stream (every path but ba_lm_kernel's).
static int ba_flush_arena(svo_ba* ba) {
  if (ba->arena_partial) { const int rc = ba_refresh_arena_image(ba); if (rc) return rc; }
  if (!ba->arena_dirty) return SVO_OK;
  svo_ctx* ctx = ba->ctx;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_arena, ba->h_arena, ba->arena_bytes, hipMemcpyHostToDevice, ba->stream));
  ba->arena_dirty = false;
  // no wait here: the pinned staging image is next touched by the host after the solve that follows has
  // drained this stream (ba_lm)
  ba->upload_pending = true;
  return SVO_OK;
}

// ---- the two passes as svo_lm_ops (host/lm.cpp drives them) ---------------------------------------------------------
namespace {
inline std::chrono::steady_clock::time_point now() { return std::chrono::steady_clock::now(); }
inline double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
  return std::chrono::duration<double, std::milli>(b - a).count();
}

// single-rank deterministic solves: the reduce kernel writes the payload straight into pinned host memory and publishes
// a completion word (no copy kernel, no stream wait); SVO_BA_NO_POLL=1 restores device payload + D2H + stream wait
bool ba_zero_copy(const svo_ba* ba) {
  static const bool no_poll = getenv("SVO_BA_NO_POLL") != nullptr;
  return ba->d.det && !ba->comm && !ba->allreduce && !no_poll;
}

int ba_wait_flag(svo_ba* ba, int seq) {
  svo_ctx* ctx = ba->ctx;
  const auto t0 = now();
  unsigned spins = 0;
  for (;;) {
    const int v = __atomic_load_n(ba->h_flag, __ATOMIC_ACQUIRE);
    if (v == seq) break;
    if (v == -seq && seq != 0) { ctx->err = "ba: the device-resident solve gave up (a workgroup waited 3 s for another one's partial sums)"; return SVO_ERR_HIP; }
    __builtin_ia32_pause();
    if (++spins > 4096u && (spins & 63u) == 0) sched_yield();  // long wait: stay polite when threads outnumber cores
    if ((spins & 0xFFFFu) == 0 && ms_between(t0, now()) > 10000.0) {  // never expected: fall back to the stream wait
      SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
      if (__atomic_load_n(ba->h_flag, __ATOMIC_ACQUIRE) != seq) { ctx->err = "ba: completion word never arrived"; return SVO_ERR_HIP; }
    }
  }
  return SVO_OK;
}

// d_pay[off, off + cnt) summed in place over the ranks: ONE collective, asynchronous on the adjuster's stream with RCCL
int ba_allreduce(svo_ba* ba, size_t off, size_t cnt) {
  svo_ctx* ctx = ba->ctx;
  if (ba->comm) {
    const char* why = nullptr;
    if (svo_rccl_allreduce_f64(ba->d_pay + off, cnt, ba->comm, ba->stream, &why)) {
      ctx->err = std::string("ba: ncclAllReduce failed: ") + (why ? why : "?");
      return SVO_ERR_HIP;
    }
  } else if (ba->allreduce) {
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
    if (ba->allreduce(ba->d_pay + off, cnt, ba->allreduce_user)) { ctx->err = "ba: allreduce callback failed"; return SVO_ERR_INVALID; }
  }
  return SVO_OK;
}

// d_pay[off, off + cnt) -> h_pay, stream drained (the one host wait of an LM iteration outside the zero-copy mode)
int ba_fetch(svo_ba* ba, size_t off, size_t cnt) {
  svo_ctx* ctx = ba->ctx;
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(ba->h_pay + off, ba->d_pay + off, sizeof(double) * cnt, hipMemcpyDeviceToHost, ba->stream));
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
  return SVO_OK;
}

// payload1 as the step control wants it: [S full | g_red | g_c | diag U | cost | sum g_p^2].  Deterministic mode: `src` holds
// the E wire totals (Schur part of the upper pose-pair blocks | per pose g_c, -Y g_p, U triangle | cost, sum g_p^2), assembled
// exactly as the oracle does: S[(k,a),(k,b)] = U_k[min][max] + Schur_(k,k)[a][b]; the lower blocks are exact transposes.
// Bulk modes: `src` is payload1 with each unordered pose pair written once, mirrored here.
void ba_payload1_out(svo_ba* ba, const double* src, double* dst) {
  const BaDev& d = ba->d;
  const int n = d.n, K = d.K, F = K - 1;
  const size_t pay1 = (size_t)n * n + 3 * (size_t)n + 2;
  if (d.det) {
    const int nU = F * (F + 1) / 2;
    double* gred = dst + (size_t)n * n;
    double* gc = gred + n;
    double* dU = gc + n;
    for (int ka = 0; ka < F; ++ka)
      for (int kb = ka; kb < F; ++kb) {
        const double* blk = src + 36 * (size_t)ba_upper_index(ka, kb, F);
        const double* U = src + 36 * (size_t)nU + 33 * (size_t)ka + 12;
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 6; ++b) {
            double v = blk[6 * a + b];
            if (ka == kb) {
              const int lo = a < b ? a : b, hi = a < b ? b : a;
              v = U[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)] + v;
            }
            dst[(size_t)(6 * ka + a) * n + 6 * kb + b] = v;
            if (ka != kb) dst[(size_t)(6 * kb + b) * n + 6 * ka + a] = v;
          }
      }
    for (int k = 0; k < F; ++k) {
      const double* pv = src + 36 * (size_t)nU + 33 * (size_t)k;
      for (int a = 0; a < 6; ++a) {
        gc[6 * k + a] = pv[a];
        gred[6 * k + a] = pv[6 + a];
        dU[6 * k + a] = pv[12 + a * 6 - a * (a - 1) / 2];
      }
    }
    dst[pay1 - 2] = src[d.E - 2];
    dst[pay1 - 1] = src[d.E - 1];
    return;
  }
  memcpy(dst, src, sizeof(double) * pay1);
  for (int a = 0; a < K - 1; ++a)
    for (int b = 0; b < K - 1; ++b) {
      if (a == b) continue;
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          const size_t ij = (size_t)(6 * a + i) * n + 6 * b + j, ji = (size_t)(6 * b + j) * n + 6 * a + i;
          dst[ij] = src[ij] + src[ji];
        }
    }
}

// launch pass A in the bulk modes (accumulates into d.pay1, which the caller zeroed).  ctl != null: the point and the
// radius come from the chained decision on the device.
const BulkSel kNoBulk = {nullptr, {nullptr, nullptr}, {nullptr, nullptr}};
int ba_launch_bulk_linearize(svo_ba* ba, double radius, int first, const double* ctl, const BulkSel& bs = kNoBulk) {
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  if (d.C <= 0) return SVO_OK;
  const int n = d.n;
  const size_t pay1 = (size_t)n * n + 3 * (size_t)n + 2;
  SvoProfScope prof(ctx, SVO_PROF_BA_LINEARIZE, ba->stream);
  if (ba->mfma_ok) {
    const size_t mfma_lds = ba_mfma_lds_bytes(n, d.K - 1);
    const int mfma_grid = std::max(1, std::min(svo_div_up(d.C, MF_WAVES), 256));
    SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)ba_linearize_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mfma_lds));
    hipLaunchKernelGGL(ba_linearize_mfma_kernel, dim3(mfma_grid), dim3(64 * MF_WAVES), mfma_lds, ba->stream, d, radius, first, ctl, bs);
  } else {
    const size_t lds_bytes = pay1 * sizeof(double);
    const int grid = std::max(1, std::min(svo_div_up(d.C, 4), 512));
    if (lds_bytes > 64 * 1024)
      SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)ba_linearize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(ba_linearize_kernel, dim3(grid), dim3(256), lds_bytes, ba->stream, d, radius, first, ctl, bs);
  }
  return SVO_OK;
}

// deterministic mode: aim the reduce kernel at pinned host memory or at the device payload; `publish` = this launch is
// the last of the iteration: its last workgroup publishes the completion word the host polls
void ba_aim_reduce(svo_ba* ba, int n_blocks, bool publish) {
  BaDev& d = ba->d;
  const bool zc = ba_zero_copy(ba);
  d.pay2_out = zc ? ba->h_pay : ba->d_pay;
  d.pay1_out = d.pay2_out + PAY2_SLOTS;
  d.flag = zc && publish ? ba->h_flag : nullptr;
  d.arrive = ba->d_arrive;
  if (d.flag) { ba->arrive_total += (unsigned)n_blocks; d.arrive_target = ba->arrive_total; d.seq = ++ba->seq; }
}

const LmCtl kNoCtl = {0, 0, 0, 0, 0};

// one tag per host-driven op (pass B / pass A launches of one step-control call share it; every slot of the partial
// store is written at most once per op) — never 0, never in ba_lm_kernel's tag space (bit 62)
unsigned long long ba_next_tag(svo_ba* ba) { ba->d.pay_tag = ++ba->tag_seq; ba->d.pay_parity = (int)(ba->tag_seq & 1ull); return ba->d.pay_tag; }

// Workgroups of ba_lm_kernel WAIT for each other (they collect each other's granules), so all of one launch must be able
// to become resident while the waiting workgroups of every other adjuster's launch hold their wave slots: the process
// admits such launches only while their workgroups together fit in 7/8 of what the device can hold of this kernel
// (occupancy x CUs; kernels that never wait always drain and hand their slots over, so the sum of the WAITING workgroups
// is what has to fit).  A solve that is not admitted takes the host-driven path — same arithmetic, same results.  (Other
// PROCESSES on the GPU are not counted; the kernel's bounded waits turn that unlikely pile-up into a reported error,
// never a hang.)
int g_fused_per_cu = 0;  // workgroups per CU the budget is counted in (window-5 problems' occupancy)
int ba_fused_budget(int device) {
  static std::mutex mu;
  static int budget[SVO_MAX_DEVICES], share_of[SVO_MAX_DEVICES];  // 0: not computed yet; recomputed when SVO_BA_CU_SHARE changes
  device = device >= 0 && device < SVO_MAX_DEVICES ? device : 0;
  std::lock_guard<std::mutex> g(mu);
  if (budget[device] && share_of[device] == g_ba_cu_share) return budget[device];
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ba_lm_kernel, 128, sizeof(double) * ba_lm_lds_doubles(24, 5, 256)) != hipSuccess) return 0;
  g_fused_per_cu = per_cu;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  share_of[device] = g_ba_cu_share;
  // LDS is handed out in contiguous pieces: between the tracker's small workgroups a CU may not have room for the last
  // workgroup the occupancy query promises, so one per CU is left out of the count
  const int usable = per_cu >= 3 ? per_cu - 1 : (per_cu >= 1 ? 1 : 0);
  budget[device] = usable * cus / 8 * 7 * g_ba_cu_share / 32;  // a CU-masked stream holds proportionally fewer workgroups
  if (const char* e = getenv("SVO_BA_BUDGET_PERCENT")) {  // developer experiments (profiles/r04_exp_admission_budget.txt)
    const int pct = atoi(e);
    if (pct >= 10 && pct <= 400) budget[device] = (int)((long long)budget[device] * pct / 100);
  }
  return budget[device];
}

// What `grid` workgroups of ba_lm_kernel with `lds` bytes of dynamic LDS cost in the units of that budget: a
// larger reduced camera system (10-keyframe windows) lowers the kernel's occupancy, its workgroups then count for more.
int ba_lm_admission_cost(int grid, size_t lds, int device) {
  (void)ba_fused_budget(device);
  static std::mutex mu;
  static size_t cached_lds[8];
  static int cached_per_cu[8], n_cached = 0;
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> g(mu);
    for (int i = 0; i < n_cached; ++i) if (cached_lds[i] == lds) per_cu = cached_per_cu[i];
    if (!per_cu) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ba_lm_kernel, 128, lds) != hipSuccess || per_cu <= 0) return 1 << 30;
      if (n_cached < 8) { cached_lds[n_cached] = lds; cached_per_cu[n_cached++] = per_cu; }
    }
  }
  if (per_cu >= g_fused_per_cu) return grid;
  return (grid * g_fused_per_cu + per_cu - 1) / per_cu;
}

inline FusedAdmission* ba_resident_admission(svo_ba* ba) { return &ba->res_admission; }

// ---- device-resident solve (ba_lm_kernel): host side -------------------------------------------------------------
// SVO_BA_DEVICE_LM=0 / 1 forces; default since the end of round 4: ON for every window-sized single-rank deterministic solve that
// is eligible and admitted — one stream alone on the GPU runs 1,715-1,745 frames/s with it against 1,745-1,750 with the host-driven
// loop (41 against 40 us per LM iteration), many streams were always faster with it, and no host thread sits in the loop.
// (Rounds 3-4 kept the host loop for a lone stream: the device loop was 51 / 45 us per iteration then.)  A pipeline group always
// uses it (svo_ba_solve_launch); larger windows (more than 128 chunks) take the host-driven loop.
// Round 5: measured again with both loops in the tree — one stream alone: 1,803 frames/s host-driven against 1,700 device-resident
// (profiles/r05_exp_single_stream_paths.txt) — so a single pipeline takes the device-resident solve only while more than two
// pipelines are inside svo_pipeline_process_batch* (then the host thread per stream is what hurts); pipeline groups always do.
bool ba_device_lm_wanted() {
  static const char* e = getenv("SVO_BA_DEVICE_LM");
  if (e && *e) return atoi(e) != 0;
  return svo_throughput_mode();
}

int ba_lm_tab_words(const svo_ba* ba) { return std::max(64, (ba->tab_max_words + 63) & ~63); }
size_t ba_lm_lds_bytes(const svo_ba* ba) { return sizeof(double) * ba_lm_lds_doubles(ba->d.n, ba->d.K, ba_lm_tab_words(ba)); }

// Fills the adjuster's launch record for the loaded problem; false: not eligible (use the host-driven path).
// compact form: the waves per workgroup that fit 156 KB of dynamic LDS (the kernel keeps ~1.5 KB of static LDS), at most
// SVO_BA_COMPACT_WAVES (default 6) and not more than the problem has chunks; 0: not eligible
constexpr size_t LMC_LDS_BUDGET = 156 * 1024;
constexpr int LM_MAX_CHUNKS_GROUPED = 288;   // wide form: 129..288 chunks go to ba_lm_grouped_kernel (144 workgroups at one per CU fit the admission budget)
constexpr int LMC_MAX_CHUNKS = 256;   // beyond, a single workgroup's rounds take longer than the host-driven loop's launches
int ba_lmc_tab_words(const svo_ba* ba) { return std::max(64, (ba->tab_max_words + 63) & ~63); }
int ba_lmc_waves(const svo_ba* ba) {
  static const int cap = [] { const char* e = getenv("SVO_BA_COMPACT_WAVES"); const int v = e ? atoi(e) : LMC_MAX_WAVES; return v < 1 ? 1 : (v > LMC_MAX_WAVES ? LMC_MAX_WAVES : v); }();
  const BaDev& d = ba->d;
  int nw = std::min(cap, std::max(1, d.C));
  while (nw >= 1 && sizeof(double) * ba_lmc_lds_doubles(d.n, d.K, ba_lmc_tab_words(ba), nw) > LMC_LDS_BUDGET) --nw;
  return nw;
}
// which device-resident form a solve of this adjuster takes (see svo_ba_set_solve_form)
bool ba_wants_compact(const svo_ba* ba) {
  if (ba->lm_penalty > 0) return true;
  if (ba->lm_form >= 0) return ba->lm_form == 1;
  static const int env = [] { const char* e = getenv("SVO_BA_FORM"); return !e || !*e ? -1 : (e[0] == 'c' ? 1 : (e[0] == 'w' ? 0 : -1)); }();
  return env == 1;
}

bool ba_device_lm_fill(svo_ba* ba, int* cost, size_t* lds_out, bool forced, bool compact = false, int* waves_out = nullptr) {
  BaDev& d = ba->d;
  if (!d.det || d.C <= 0 || !ba_zero_copy(ba) || !(forced || ba->device_lm == 1 || (ba->device_lm < 0 && ba_device_lm_wanted())) || !ba->h_lane) return false;
  size_t lds = 0;
  int nw = 0;
  if (compact) {
    if (d.C > LMC_MAX_CHUNKS || ba->tab_max_words > 4096) return false;
    nw = ba_lmc_waves(ba);
    if (nw < 1) return false;
    lds = sizeof(double) * ba_lmc_lds_doubles(d.n, d.K, ba_lmc_tab_words(ba), nw);
  } else {
  // every chunk table in LDS, every chunk its own partial (beyond 128 chunks the grouped kernel sums groups of G chunks first), and
  // the launch must fit the admission budget at all (workgroups that wait for each other): window-sized problems
  if (d.C > LM_MAX_CHUNKS_GROUPED || d.CPW != 1 || ba->tab_max_words > TAB_LDS_WORDS) return false;
  // beyond 128 chunks (ba_lm_grouped_kernel) only on request: measured on configs[2]'s 10-keyframe windows (~210 chunks, E = 2,312 wire
  // elements) the host-driven loop is faster — 780 against 590 frames/s (profiles/r05_exp_single_stream_paths.txt)
  if (d.G > 1 && ba->device_lm != 1) return false;
  lds = ba_lm_lds_bytes(ba);
  if (lds > 120 * 1024) return false;  // n <= 100 or so; window problems are n <= 60
  }
  d.points = ba->cur_points; d.cand_points = ba->cand_points; d.poses = ba->cur_poses; d.cand_poses = ba->cand_poses;
  d.flag = nullptr;
  if (ba->arena_partial && ba_refresh_arena_image(ba)) return false;  // a re-solve after a zero-copy solve: the host image takes the solved state first
  LmLane& L = *ba->h_lane;
  L.P = d;
  LmDevArgs& a = L.a;
  a.cnt = ba->d_lmc;
  a.arena_src = ba->arena_dirty ? ba->h_arena : nullptr;  // read in place by the kernel (zero copy); null: the device arena is complete
  a.arena_dst = ba->d_arena; a.arena_bytes = ba->arena_bytes;
  a.points_a = ba->cur_points; a.points_b = ba->cand_points;
  a.export_points = ba->n_points ? ba->h_out_points : nullptr;
  a.store = d.lm_key ? ba->store : nullptr; a.store_mask = ba->store_mask;
  a.dev_res = ba->d_res;
  a.host_result = ba->h_result;
  a.host_flag = ba->h_flag; a.host_seq = ba->seq + 1;  // committed by the launch
  // the delivery counter runs on from solve to solve (monotone, wrap-safe compares): no clearing launch in front of a solve
  // unless the last one did not leave cleanly
  const bool fresh = ba->lm_counters_dirty || !ba->lm_have_base;
  a.base_arrive = fresh ? 0 : ba->lm_base;
  a.tab_words = compact ? ba_lmc_tab_words(ba) : ba_lm_tab_words(ba);
  a.opt.max_iterations = ba->opt.max_iterations;
  a.opt.function_tolerance = ba->opt.function_tolerance; a.opt.gradient_tolerance = ba->opt.gradient_tolerance;
  a.opt.parameter_tolerance = ba->opt.parameter_tolerance; a.opt.initial_radius = ba->opt.initial_radius;
  a.opt.max_time_s = ba->opt.max_time_s;  // src/bundle_adjuster.cpp:11, tested on workgroup 0's posted clock
  a.dbg = d.C <= 8192 ? ba->d_lmdbg : nullptr;
  {
    // SVO_BA_TEST_GIVEUP=n: every n-th wide launch of an adjuster reports "gave up" (tests/test_ba.py, tests/test_group.py)
    static const int every = [] { const char* e = getenv("SVO_BA_TEST_GIVEUP"); return e && *e ? atoi(e) : 0; }();
    a.test_giveup = (!compact && every > 0 && (++ba->wide_launches % every) == 0) ? 1 : 0;
  }
  *cost = compact ? 0 : ba_lm_admission_cost((d.C + LM_CPW - 1) / LM_CPW, lds, ba->ctx->device);
  *lds_out = lds;
  if (waves_out) *waves_out = nw;
  return true;
}

// ONE launch for the solves of `n` adjusters (the lanes of a pipeline group that reached a keyframe together; n = 1: a
// single pipeline) on stream `st`.  Returns how many were admitted and launched; `launched_mask` says which (an ineligible
// or not admitted adjuster is skipped, not a barrier for those behind it).  The others keep their loaded problem and can be
// offered again later, or solved by the host-driven path.
int ba_device_lm_launch(svo_ba** bas, int n, hipStream_t st, bool forced, unsigned long long* launched_mask) {
  if (launched_mask) *launched_mask = 0;
  // SVO_BA_OVERFLOW=1: a solve the admission budget refuses takes the compact form at once instead of waiting for the budget.  Off by
  // default: measured in round 5 (profiles/r05_exp_lanes_groups.txt) it helps only while streams share hardware queues (7 lines per group
  // on 16 queues: 30 k against 25 k frames/s at 128 lanes); with one hardware queue per stream the wide form alone is faster (37-51 k
  // against 27-36 k) — a 3 ms compact solve holds its line for 3 ms.
  static const bool overflow = [] { const char* e = getenv("SVO_BA_OVERFLOW"); return e && *e && atoi(e) != 0; }();
  (void)g_group_lanes;
  bool to_compact[SVO_MAX_LANES] = {};
  // ---- the wide form (ba_lm_kernel): admitted against the budget of co-resident waiting workgroups
  int launched_total = 0;
  for (int i = 0; i < n && i < SVO_MAX_LANES; ++i) bas[i]->lm_inflight = false;
  for (int grouped = 0; grouped < 2; ++grouped) {  // windows of up to 128 chunks: ba_lm_kernel; beyond: ba_lm_grouped_kernel
  LmLanePtrs ptrs;
  int launched = 0, max_c = 0;
  size_t max_lds = 0;
  svo_ba* took[SVO_MAX_LANES];
  int tidx[SVO_MAX_LANES];
  for (int i = 0; i < n && i < SVO_MAX_LANES; ++i) {
    svo_ba* ba = bas[i];
    int cost = 0;
    size_t lds = 0;
    if ((ba->d.G > 1) != (grouped != 0)) continue;
    if (ba_wants_compact(ba)) { to_compact[i] = true; continue; }
    if (!ba_device_lm_fill(ba, &cost, &lds, forced)) continue;
    if (lds > 32 * 1024 && hipFuncSetAttribute(grouped ? (const void*)ba_lm_grouped_kernel : (const void*)ba_lm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess) continue;
    if (!ba_resident_admission(ba)->admit(cost, ba->ctx->device)) { to_compact[i] = overflow; continue; }
    // the counter starts from zero: cleared in front of the launch (the adjuster's previous solve no longer touches it
    // once its completion word is out)
    if (ba->lm_counters_dirty || !ba->lm_have_base) {
      if (hipMemsetAsync(ba->d_lmc, 0, LMC_WORDS * sizeof(unsigned), st) != hipSuccess) { ba_resident_admission(ba)->release(); continue; }
    }
    ba->lm_counters_dirty = false;
    ptrs.p[launched] = ba->h_lane;
    took[launched] = ba;
    tidx[launched] = i;
    max_c = std::max(max_c, (ba->d.C + LM_CPW - 1) / LM_CPW);
    max_lds = std::max(max_lds, lds);
    ++launched;
  }
  if (launched) {
    const auto t0 = now();
    {
      SvoProfScope prof(took[0]->ctx, SVO_PROF_BA_STEP, st);
      if (grouped) hipLaunchKernelGGL(ba_lm_grouped_kernel, dim3(max_c, launched), dim3(128), max_lds, st, ptrs);
      else hipLaunchKernelGGL(ba_lm_kernel, dim3(max_c, launched), dim3(128), max_lds, st, ptrs);
    }
    if (hipGetLastError() != hipSuccess) {
      for (int i = 0; i < launched; ++i) ba_resident_admission(took[i])->release();
      launched = 0;
    }
    for (int i = 0; i < launched; ++i) {
      svo_ba* ba = took[i];
      ++ba->seq;  // = a.host_seq
      ba->lm_t0 = t0;
      if (ba->arena_dirty) ba->arena_partial = true;  // the kernel reads the host image in place: the device arena holds the landmark buffers only
      ba->arena_dirty = false;
      ba->lm_inflight = true; ba->lm_compact_inflight = false;
      ba->lm_stream = st;
      ba->res_export = ba->h_lane->a.export_points != nullptr;
      if (launched_mask) *launched_mask |= 1ull << tidx[i];
    }
  }
  launched_total += launched;
  }
  const int launched = launched_total;
  // ---- the compact form (ba_lm_compact_kernel: one workgroup per solve; nothing to admit, nothing to clear)
  int n_compact = 0;
  {
    LmLanePtrs cptrs;
    svo_ba* ctook[SVO_MAX_LANES];
    int cidx[SVO_MAX_LANES];
    int nw = LMC_MAX_WAVES;
    for (int i = 0; i < n && i < SVO_MAX_LANES; ++i) {
      svo_ba* ba = bas[i];
      if (!to_compact[i]) continue;
      int cost = 0, waves = 0;
      size_t lds = 0;
      if (!ba_device_lm_fill(ba, &cost, &lds, forced, true, &waves)) continue;
      cptrs.p[n_compact] = ba->h_lane; ctook[n_compact] = ba; cidx[n_compact] = i;
      nw = std::min(nw, waves);
      ++n_compact;
    }
    if (n_compact) {
      // one launch = one workgroup shape: the smallest wave count of its solves (fewer waves need less LDS), LDS of the largest
      size_t clds = 0;
      for (int k = 0; k < n_compact; ++k)
        clds = std::max(clds, sizeof(double) * ba_lmc_lds_doubles(ctook[k]->d.n, ctook[k]->d.K, ctook[k]->h_lane->a.tab_words, nw));
      static bool attr_set = false;
      if (!attr_set) attr_set = hipFuncSetAttribute((const void*)ba_lm_compact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LMC_LDS_BUDGET) == hipSuccess;
      const auto t0 = now();
      bool ok = attr_set;
      if (ok) {
        SvoProfScope prof(ctook[0]->ctx, SVO_PROF_BA_STEP, st);
        hipLaunchKernelGGL(ba_lm_compact_kernel, dim3(1, n_compact), dim3(64 * nw), clds, st, cptrs);
        ok = hipGetLastError() == hipSuccess;
      }
      if (!ok) n_compact = 0;
      for (int k = 0; k < n_compact; ++k) {
        svo_ba* ba = ctook[k];
        ++ba->seq;  // = a.host_seq
        ba->lm_t0 = t0;
        if (ba->arena_dirty) ba->arena_partial = true;  // the host image keeps the INITIAL state (see ba_refresh_arena_image)
        ba->arena_dirty = false;
        ba->lm_inflight = true; ba->lm_compact_inflight = true;
        ba->lm_stream = st;
        ba->res_export = ba->h_lane->a.export_points != nullptr;
        if (launched_mask) *launched_mask |= 1ull << cidx[k];
      }
    }
  }
  return launched + n_compact;
}

// Joins the launch: completion word, then poses / summary / counters out of the pinned result block.
int ba_device_lm_end(svo_ba* ba, svo_ba_summary* sum) {
  if (!ba->lm_inflight) return SVO_ERR_INVALID;
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  ba->lm_inflight = false;
  const bool was_compact = ba->lm_compact_inflight;
  ba->lm_compact_inflight = false;
  const int rc = ba_wait_flag(ba, ba->seq);
  ba_resident_admission(ba)->release();
  if (rc) {
    // a wait inside the kernel gave up (or the launch never ran): drain, and never trust the counter again
    (void)hipStreamSynchronize(ba->lm_stream);
    if (getenv("SVO_BA_TRACE")) {  // who stopped where
      unsigned c[LMC_WORDS] = {0};
      (void)hipMemcpy(c, ba->d_lmc, sizeof(c), hipMemcpyDeviceToHost);
      fprintf(stderr, "[svo ba] device solve gave up: chunks %d K %d delivered %u\n", d.C, d.K, c[LMC_ARRIVE]);
      if (ba->d_lmdbg && d.C <= 4096) {
        std::vector<unsigned> g(16 * (size_t)d.C);
        (void)hipMemcpy(g.data(), ba->d_lmdbg, sizeof(unsigned) * g.size(), hipMemcpyDeviceToHost);
        for (int b = 0; b < (d.C + LM_CPW - 1) / LM_CPW; ++b) {
          const unsigned* q = &g[16 * (size_t)b];
          if (b == 0 || memcmp(q, &g[0], 20) != 0)
            fprintf(stderr, "[svo ba]   workgroup %d: op %u state %u iterations %u need_linearize %u chain/bad %x arrive_total %u commands %u lin_calls %u\n", b, q[0], q[1],
                    q[2], q[3], q[4], q[5], q[6], q[7]);
        }
      }
    }
    if (!was_compact) ba->lm_counters_dirty = true;
    ba->host_points_valid = false;
    return rc;
  }
  const double* r = ba->h_result;
  if (!was_compact) { ba->lm_base = (unsigned)r[LMR_C_ARRIVE]; ba->lm_have_base = true; }
  memcpy(ba->h_poses.data(), r + LMR_DOUBLES, sizeof(double) * 7 * (size_t)d.K);
  if (r[LMR_SEL] != 0.0) { std::swap(ba->cur_points, ba->cand_points); std::swap(ba->cur_poses, ba->cand_poses); }
  ba->host_points_valid = ba->res_export;
  memset(&ba->stats, 0, sizeof(ba->stats));
  ba->stats.linearize_calls = (int)r[LMR_LINEARIZE_CALLS];
  ba->stats.step_calls = (int)r[LMR_STEP_CALLS];
  ba->stats.speculations = ba->stats.step_calls;
  ba->stats.speculation_hits = (int)r[LMR_NEXT_USED];
  ba->stats.single_exchange = (int)r[LMR_SAME_SWEEP];
  ba->lm_same += (long)r[LMR_SAME_SWEEP]; ba->lm_used += (long)r[LMR_NEXT_USED]; ba->lm_steps += (long)r[LMR_STEP_CALLS]; ba->lm_lins += (long)r[LMR_LINEARIZE_CALLS];
  for (int i = 0; i < 14; ++i) ba->lm_tp[i] += r[LMR_TP0 + i];
  if (ba->d_lmdbg && getenv("SVO_TIMING")) {
    const int nb = (d.C + LM_CPW - 1) / LM_CPW;
    std::vector<unsigned> g(16 * (size_t)nb);
    if (hipMemcpy(g.data(), ba->d_lmdbg, sizeof(unsigned) * g.size(), hipMemcpyDeviceToHost) == hipSuccess) {
      for (int sl = 0; sl < 7; ++sl) {
        double mn = 1e300, mx = 0, sum = 0;
        for (int b = 0; b < nb; ++b) { const double v = g[16 * (size_t)b + 8 + sl]; mn = std::min(mn, v); mx = std::max(mx, v); sum += v; }
        ba->lm_wg_min[sl] += mn; ba->lm_wg_max[sl] += mx; ba->lm_wg_mean[sl] += sum / nb;
      }
      static int traced = 0;
      if (getenv("SVO_BA_TRACE_WG") && traced < 2 && r[LMR_ITERATIONS] >= 4) {  // which chunks are the slow ones: rest of pass A of a workgroup's first wavefront against its chunk's shape
        ++traced;
        const int nU = (d.K - 1) * d.K / 2, F = d.K - 1;
        for (int b = 0; b < nb; ++b) {
          const int c = b * LM_CPW;
          const uint16_t* w = ba->u_tab.data() + ba->u_tab_off[(size_t)c];
          int ne = 0, longest = 0, plong = 0;
          for (int q = 0; q < nU; ++q) { const int len = w[q + 1] - w[q]; ne += len > 0; longest = std::max(longest, len); }
          for (int k = 0; k < F; ++k) plong = std::max(plong, (int)w[nU + 1 + k + 1] - (int)w[nU + 1 + k]);
          fprintf(stderr, "[svo ba wg] %2d: rest of pass A %.2f us/iteration, pass B %.2f | chunk %d: %d entries in %d blocks (longest list %d), longest pose list %d\n", b,
                  1e-2 * g[16 * (size_t)b + 8 + 3] / r[LMR_ITERATIONS], 1e-2 * g[16 * (size_t)b + 8 + 0] / r[LMR_ITERATIONS], c, (int)w[nU], ne, longest, plong);
        }
      }
    }
  }
  ba->lm_t_wait += r[LMR_T_WAIT]; ba->lm_t_ctl += r[LMR_T_CTL]; ba->lm_t_body += r[LMR_T_BODY]; ba->lm_t_total += r[LMR_T_TOTAL]; ba->lm_n++; ba->lm_iters += (long)r[LMR_ITERATIONS];
  if (sum) {
    sum->iterations = (int)r[LMR_ITERATIONS]; sum->successful_steps = (int)r[LMR_SUCCESSFUL]; sum->termination = (int)r[LMR_TERMINATION];
    sum->initial_cost = r[LMR_INITIAL_COST]; sum->final_cost = r[LMR_FINAL_COST];
    sum->solve_ms = ms_between(ba->lm_t0, now());
  }
  if (g_ba_cu_share != 32 && ba->lm_stream == ba->stream) SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));  // a CU-masked stream that never met a synchronisation point did not come back from hipStreamSynchronize at teardown (observed in round 2)
  d.points = ba->cur_points; d.cand_points = ba->cand_points; d.poses = ba->cur_poses; d.cand_poses = ba->cand_poses;
  return SVO_OK;
}

inline size_t wave_lds_bytes(const BaDev& d) { return sizeof(double) * (size_t)wg_lds_doubles(d.E) + sizeof(uint16_t) * (size_t)d.tab_lds_words; }
// workgroup width of the host-driven deterministic kernels: 128 (64, one wavefront doing everything: 1,553 against 1,620
// frames/s single stream; 256: the same as 128 — measured in round 4, the kernels stay templates on it)
#define SVO_DET_LAUNCH(kernel, grid, lds, st, ...) hipLaunchKernelGGL(kernel<128>, grid, dim3(128), lds, st, __VA_ARGS__)

int op_linearize(void* user, double radius, int first, double* pay1_out) {
  svo_ba* ba = static_cast<svo_ba*>(user);
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  hipStream_t st = ba->stream;
  const auto t0 = now();
  const int n = d.n;
  const size_t pay1 = (size_t)n * n + 3 * (size_t)n + 2;
  const size_t wire = d.det ? (size_t)d.E : pay1;  // what travels: the wire totals, or payload1 itself (bulk modes)
  d.points = ba->cur_points; d.cand_points = ba->cand_points; d.poses = ba->cur_poses; d.cand_poses = ba->cand_poses;
  if (d.det) {
    (void)ba_next_tag(ba);
    if (d.NG > 0) {
      SvoProfScope prof(ctx, SVO_PROF_BA_LINEARIZE, st);
      SVO_DET_LAUNCH(ba_linearize_det_kernel, dim3(d.NG), wave_lds_bytes(d), st, d, radius, first, (const double*)nullptr);  // one wave per workgroup: spreads the chunks over the CUs
    }
    const int nb = d.NG > 0 ? ba_reduce_blocks(d.E, d.NG) : 0;
    if (nb > 0) {
      ba_aim_reduce(ba, nb, true);
      hipLaunchKernelGGL(ba_reduce_kernel, dim3(nb), dim3(256), 0, st, d, 1, 0, kNoCtl);
      SVO_HIP_CHECK(ctx, hipGetLastError());
      if (d.flag) { const int rc = ba_wait_flag(ba, d.seq); if (rc) return rc; }
    } else {  // no observations at all: every total is zero
      SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
      memset(ba->h_pay, 0, sizeof(double) * (PAY2_SLOTS + wire));
      if (!ba_zero_copy(ba)) SVO_HIP_CHECK(ctx, hipMemsetAsync(ba->d_pay, 0, sizeof(double) * (PAY2_SLOTS + wire), st));
    }
  } else {
    SVO_HIP_CHECK(ctx, hipMemsetAsync(d.pay1, 0, sizeof(double) * pay1, st));
    const int rc = ba_launch_bulk_linearize(ba, radius, first, nullptr);
    if (rc) return rc;
    SVO_HIP_CHECK(ctx, hipGetLastError());
  }
  if (!ba_zero_copy(ba)) {
    int rc = ba_allreduce(ba, PAY2_SLOTS, wire);
    if (!rc) rc = ba_fetch(ba, PAY2_SLOTS, wire);
    if (rc) return rc;
  }
  ba_payload1_out(ba, ba->h_pay + PAY2_SLOTS, pay1_out);
  ba->t_lin += ms_between(t0, now()); ba->n_lin++;
  return SVO_OK;
}

// Pass B, and — in the same call, without the host in between — pass A for the next LM iteration:
//   ctl->spec_radius > 0  same sweep (deterministic: ONE kernel; bulk: back-to-back launches), ONE collective;
//   ctl->chain            pass B -> [all-reduce of payload2] -> decision on the device -> pass A -> [all-reduce of payload1].
int op_step(void* user, const double* dc, const double* cand_poses7, double radius, const svo_lm_step_ctl* ctl, double* pay2_out,
            double* pay1_next_out, double* next_radius, int* next_at_candidate) {
  svo_ba* ba = static_cast<svo_ba*>(user);
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  hipStream_t st = ba->stream;
  const auto t0 = now();
  const int n = d.n, K = d.K, nn = n > 0 ? n : 1;
  const size_t pay1 = (size_t)n * n + 3 * (size_t)n + 2;
  const size_t wire = d.det ? (size_t)d.E : pay1;
  const double spec_radius = ctl->spec_radius;
  const bool same_sweep = spec_radius > 0, chain = !same_sweep && ctl->chain != 0;
  const bool sharded = ba->comm || ba->allreduce;
  const LmCtl lc = {ctl->cost, ctl->mcc, radius, ctl->decrease_factor, chain && !sharded ? 1 : 0};
  *next_radius = 0.0; *next_at_candidate = 0;
  // [dc | candidate poses]: adjacent in the pinned block; the kernels stage them into LDS
  if (n > 0) memcpy(ba->h_step, dc, sizeof(double) * n);
  memcpy(ba->h_step + nn, cand_poses7, sizeof(double) * 7 * K);
  d.points = ba->cur_points; d.cand_points = ba->cand_points; d.poses = ba->cur_poses; d.cand_poses = ba->cand_poses;
  d.ctl_dev = reinterpret_cast<double*>(ba->d_arrive + 2);
  const bool zc = ba_zero_copy(ba);
  if (zc) {
    d.step_in = ba->h_step;  // read in place from pinned memory: no H2D blit per LM iteration
  } else {
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_step, ba->h_step, sizeof(double) * (nn + 7 * K), hipMemcpyHostToDevice, st));
    d.step_in = ba->d_step;
  }
  const bool next = same_sweep || chain;
  if (d.det && d.NG > 0) {
    (void)ba_next_tag(ba);
    const int nb = ba_reduce_blocks(d.E, d.NG);
    {
      SvoProfScope prof(ctx, SVO_PROF_BA_STEP, st);
      SVO_DET_LAUNCH(ba_step_kernel, dim3(d.NG), wave_lds_bytes(d), st, d, radius, same_sweep ? spec_radius : 0.0);
    }
    if (!chain) {
      const int blocks = (same_sweep ? nb : 0) + 1;
      ba_aim_reduce(ba, blocks, true);
      hipLaunchKernelGGL(ba_reduce_kernel, dim3(blocks), dim3(256), 0, st, d, same_sweep ? 1 : 0, 1, kNoCtl);
    } else if (!sharded) {
      // pass A forms payload2 and takes the decision itself: 3 launches per LM iteration
      ba_aim_reduce(ba, 1, false);
      SvoProfScope prof(ctx, SVO_PROF_BA_LINEARIZE, st);
      SVO_DET_LAUNCH(ba_decide_linearize_kernel, dim3(d.NG), wave_lds_bytes(d), st, d, lc);
    } else {
      ba_aim_reduce(ba, 1, false);
      hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, st, d, 0, 1, lc);
      int rc = ba_allreduce(ba, 0, PAY2_SLOTS);
      if (rc) return rc;
      const LmCtl lcs = {ctl->cost, ctl->mcc, radius, ctl->decrease_factor, 1};
      hipLaunchKernelGGL(ba_decide_kernel, dim3(1), dim3(64), 0, st, lcs, ba->d_pay, d.ctl_dev);
      SvoProfScope prof(ctx, SVO_PROF_BA_LINEARIZE, st);
      SVO_DET_LAUNCH(ba_linearize_det_kernel, dim3(d.NG), wave_lds_bytes(d), st, d, 0.0, 0, (const double*)d.ctl_dev);
    }
    if (chain) {
      ba_aim_reduce(ba, nb, true);
      hipLaunchKernelGGL(ba_reduce_kernel, dim3(nb), dim3(256), 0, st, d, 1, 0, kNoCtl);
    }
    SVO_HIP_CHECK(ctx, hipGetLastError());
    if (d.flag) { const int rc = ba_wait_flag(ba, d.seq); if (rc) return rc; }
  } else if (d.det) {  // no observations: zero sums, nothing to launch
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
    memset(ba->h_pay, 0, sizeof(double) * (PAY2_SLOTS + wire));
    if (!zc) SVO_HIP_CHECK(ctx, hipMemsetAsync(ba->d_pay, 0, sizeof(double) * (PAY2_SLOTS + wire), st));
    if (chain) {  // the decision of the chained step, as the kernels would take it
      const SvoLmDecision dec = svo_lm_decide(ctl->cost, ctl->mcc, radius, ctl->decrease_factor, 0.0, 0.0);
      ba->h_pay[4] = (double)dec.accept; ba->h_pay[5] = dec.next_radius;
      if (!zc) SVO_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_pay, ba->h_pay, sizeof(double) * PAY2_SLOTS, hipMemcpyHostToDevice, st));
    }
  } else {
    SVO_HIP_CHECK(ctx, hipMemsetAsync(ba->d_pay, 0, sizeof(double) * (next ? PAY2_SLOTS + pay1 : PAY2_SLOTS), st));
    if (d.C > 0) {
      SvoProfScope prof(ctx, SVO_PROF_BA_BACKSUB, st);
      const int grid = std::max(1, std::min(svo_div_up(d.C, 4), 512));
      hipLaunchKernelGGL(ba_backsub_kernel, dim3(grid), dim3(256), 0, st, d, radius, kNoBulk);
    }
    if (same_sweep) {  // pass A at the candidate the launch above just wrote
      d.points = ba->cand_points; d.poses = ba->cand_poses;
      const int rc = ba_launch_bulk_linearize(ba, spec_radius, 0, nullptr);
      d.points = ba->cur_points; d.poses = ba->cur_poses;
      if (rc) return rc;
    } else if (chain) {
      if (sharded) { const int rc = ba_allreduce(ba, 0, PAY2_SLOTS); if (rc) return rc; }
      const LmCtl lcs = {ctl->cost, ctl->mcc, radius, ctl->decrease_factor, 1};
      hipLaunchKernelGGL(ba_decide_kernel, dim3(1), dim3(64), 0, st, lcs, ba->d_pay, d.ctl_dev);
      const int rc = ba_launch_bulk_linearize(ba, 0.0, 0, d.ctl_dev);
      if (rc) return rc;
    }
    SVO_HIP_CHECK(ctx, hipGetLastError());
  }
  if (!zc) {
    // same sweep: ONE collective for both payloads; chained: payload2 was summed before the decision, payload1 now
    int rc = SVO_OK;
    if (same_sweep) rc = ba_allreduce(ba, 0, PAY2_SLOTS + wire);
    else if (chain) rc = ba_allreduce(ba, PAY2_SLOTS, wire);
    else rc = ba_allreduce(ba, 0, PAY2_SLOTS);
    if (!rc) rc = ba_fetch(ba, 0, next ? PAY2_SLOTS + wire : PAY2_SLOTS);
    if (rc) return rc;
  }
  memcpy(pay2_out, ba->h_pay, sizeof(double) * 4);
  if (same_sweep) {
    *next_radius = spec_radius; *next_at_candidate = 1;
  } else if (chain) {
    *next_at_candidate = ba->h_pay[4] != 0.0;
    *next_radius = ba->h_pay[5];
  }
  if (next) ba_payload1_out(ba, ba->h_pay + PAY2_SLOTS, pay1_next_out);
  ba->t_step += ms_between(t0, now()); ba->n_step++;
  return SVO_OK;
}

int op_accept(void* user) {
  svo_ba* ba = static_cast<svo_ba*>(user);
  std::swap(ba->cur_points, ba->cand_points);
  std::swap(ba->cur_poses, ba->cand_poses);  // the candidate poses are already on the device (pass B's first workgroup)
  return SVO_OK;
}
}  // namespace

// ---- bulk / sharded solve with the step control on the device ----------------------------------------------------------
// Eligible: hardware-order accumulation (the deterministic window path has its own device-resident form, ba_lm_kernel), a
// reduced camera system that fits one workgroup's LDS (n <= 128: up to 22 poses).  SVO_BA_BULK_CONTROL=0 / svo_ba_set_bulk_control
// restore the host-driven loop (host/lm.cpp).
static bool ba_bulk_control_wanted(const svo_ba* ba) {
  if (ba->d.det || ba->d.n > 128 || ba->d.K > 64) return false;
  if (ba->bulk_ctl >= 0) return ba->bulk_ctl == 1;
  static const char* e = getenv("SVO_BA_BULK_CONTROL");
  return e && *e && atoi(e) != 0;  // opt-in: one workgroup factors n = 114 in 125 us where a host core takes 35
}

// The host's whole part of a solve: enqueue the fixed sequence slot by slot, staying `ahead` slots in front of the last status
// record it has seen (SVO_BA_RUNAHEAD, default 2: the GPU never waits for the host; on termination ahead - 1 slots of no-op
// kernels and two small collectives are wasted once per solve).  Slot s is enqueued iff s < ahead or the record of slot s - ahead
// says "not done" — a function of the records only, never of timing: every rank of a sharded run enqueues the same
// number of collectives.
static int ba_lm_bulk_device(svo_ba* ba, svo_ba_summary* sum) {
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  hipStream_t st = ba->stream;
  const auto t_begin = now();
  const int n = d.n, K = d.K, nn = n > 0 ? n : 1;
  const size_t pay1 = (size_t)n * n + 3 * (size_t)n + 2, wire = pay1 + 2;  // + [elapsed seconds, ranks]
  static const int ahead_env = [] { const char* e = getenv("SVO_BA_RUNAHEAD"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > BULK_RING - 2 ? BULK_RING - 2 : v); }();
  const int ahead = ahead_env;
  {
    const int rcf = ba_flush_arena(ba);
    if (rcf) return rcf;
  }
  ba->host_points_valid = false;
  memset(&ba->stats, 0, sizeof(ba->stats));
  double host_ms = 0.0;
  // initial state -> device (from pinned memory), payload cleared, status ring forgotten
  double* init = ba->h_bstat + (size_t)BULK_RING * BS_DOUBLES;
  for (int i = 0; i < BC_WORDS; ++i) init[i] = 0.0;
  init[BC_MODE] = (double)BCM_FIRST; init[BC_RADIUS] = ba->opt.initial_radius; init[BC_DF] = 2.0; init[BC_TERM] = 1.0;
  for (int r = 0; r < BULK_RING; ++r) __atomic_store_n(reinterpret_cast<long long*>(&ba->h_bstat[(size_t)r * BS_DOUBLES]), 0ll, __ATOMIC_RELEASE);
  SVO_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_bctl, init, sizeof(double) * BC_WORDS, hipMemcpyHostToDevice, st));
  SVO_HIP_CHECK(ctx, hipMemsetAsync(ba->d_pay, 0, sizeof(double) * (PAY2_SLOTS + wire), st));
  d.points = ba->cur_points; d.cand_points = ba->cand_points; d.poses = ba->cur_poses; d.cand_poses = ba->cand_poses;
  d.step_in = ba->d_step;
  d.ctl_dev = nullptr;
  const BulkSel bs = {ba->d_bctl, {ba->cur_points, ba->cand_points}, {ba->cur_poses, ba->cand_poses}};
  BulkCtlArgs ca;
  ca.n = n; ca.K = K; ca.slot = 0; ca.ring = BULK_RING;
  ca.pay = ba->d_pay; ca.bctl = ba->d_bctl; ca.sc = ba->d_bctl + BC_WORDS; ca.step = ba->d_step;
  ca.pos[0] = ba->cur_poses; ca.pos[1] = ba->cand_poses;
  ca.status = ba->h_bstat;
  ca.opt.max_iterations = ba->opt.max_iterations; ca.opt.function_tolerance = ba->opt.function_tolerance;
  ca.opt.gradient_tolerance = ba->opt.gradient_tolerance; ca.opt.parameter_tolerance = ba->opt.parameter_tolerance;
  ca.opt.initial_radius = ba->opt.initial_radius; ca.opt.max_time_s = ba->opt.max_time_s;
  const size_t ctl_lds = sizeof(double) * ba_bulk_ctl_lds_doubles(n, K);
  SVO_REQUIRE(ctx, ctl_lds <= 158 * 1024, "ba: reduced camera system too large for the device-side step control");
  if (ctl_lds > 48 * 1024) SVO_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)ba_bulk_control_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctl_lds));
  int collectives = 0;
  auto control = [&](int slot) {
    ca.slot = slot;
    hipLaunchKernelGGL(ba_bulk_control_kernel, dim3(1), dim3(512), ctl_lds, st, ca);
  };
  const bool sharded = ba->comm || ba->allreduce;
  auto allreduce = [&](size_t off, size_t cnt) -> int {
    ++collectives;  // what N ranks issue; a single rank has nothing to sum
    return sharded ? ba_allreduce(ba, off, cnt) : SVO_OK;
  };
  auto record = [&](int slot) { return ba->h_bstat + (size_t)(slot % BULK_RING) * BS_DOUBLES; };
  auto wait_record = [&](int slot) -> int {  // blocks until the control kernel of `slot` has published
    const double* r = record(slot);
    const auto t0 = now();
    unsigned spins = 0;
    while (__atomic_load_n(reinterpret_cast<const long long*>(&r[0]), __ATOMIC_ACQUIRE) != (long long)__builtin_bit_cast(long long, (double)(slot + 1))) {
      __builtin_ia32_pause();
      if (++spins > 4096u && (spins & 63u) == 0) sched_yield();
      if ((spins & 0xFFFFu) == 0 && ms_between(t0, now()) > 20000.0) {
        (void)hipStreamSynchronize(st);
        if (r[0] != (double)(slot + 1)) { ctx->err = "ba: the device-side step control never published its status record"; return SVO_ERR_HIP; }
      }
    }
    return SVO_OK;
  };
  int rc = SVO_OK;
  // slot 0: the first linearisation
  {
    const auto h0 = now();
    rc = ba_launch_bulk_linearize(ba, ba->opt.initial_radius, 1, nullptr, bs);
    if (!rc) rc = allreduce(PAY2_SLOTS, wire);
    if (!rc) control(0);
    SVO_HIP_CHECK(ctx, hipGetLastError());
    host_ms += ms_between(h0, now());
  }
  if (rc) return rc;
  const int max_slots = 2 * ba->opt.max_iterations + 8;  // every slot is an LM iteration or follows a failed factorisation
  int enqueued = 1, last_seen = -1;
  bool done = false;
  double tk_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int tk_n = 0;
  for (int s = 1; s < max_slots && !done; ++s) {
    if (s >= ahead) {  // (waiting here overlaps the GPU's work on slots s - ahead + 1 .. s - 1)
      rc = wait_record(s - ahead);
      if (rc) return rc;
      last_seen = s - ahead;
      if (record(s - ahead)[1] != 0.0) { done = true; break; }
      for (int i = 0; i < 8; ++i) tk_sum[i] += record(s - ahead)[14 + i];
      ++tk_n;
    }
    const auto h0 = now();
    const int grid = std::max(1, std::min(svo_div_up(d.C, 4), 512));
    if (d.C > 0) {
      SvoProfScope prof(ctx, SVO_PROF_BA_BACKSUB, st);
      hipLaunchKernelGGL(ba_backsub_kernel, dim3(grid), dim3(256), 0, st, d, 0.0, bs);
    }
    rc = allreduce(0, PAY2_SLOTS);
    if (!rc) rc = ba_launch_bulk_linearize(ba, 0.0, 0, nullptr, bs);
    if (!rc) rc = allreduce(PAY2_SLOTS, wire);
    if (rc) return rc;
    control(s);
    SVO_HIP_CHECK(ctx, hipGetLastError());
    ++enqueued;
    host_ms += ms_between(h0, now());
  }
  // the last enqueued slot's record carries the final state (slots behind the terminating one re-publish it)
  rc = wait_record(enqueued - 1);
  if (rc) return rc;
  (void)last_seen;
  const double* r = record(enqueued - 1);
  if (r[1] == 0.0) { ctx->err = "ba: the device-side step control did not terminate within its slot bound"; return SVO_ERR_NUMERIC; }
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(st));
  memcpy(ba->h_poses.data(), r + BS_HEAD, sizeof(double) * 7 * (size_t)K);
  if (((int)r[7] & 1) != 0) { std::swap(ba->cur_points, ba->cand_points); std::swap(ba->cur_poses, ba->cand_poses); }
  ba->stats.linearize_calls = (int)r[8];
  ba->stats.step_calls = (int)r[9];
  ba->stats.speculations = (int)r[9];       // every step carried the next linearisation (chained decision)
  ba->stats.speculation_hits = (int)r[10];
  ba->stats.single_exchange = 0;
  ba->stats.collectives = collectives;
  ba->stats.device_control = 1;
  ba->stats.host_us = 1e3 * host_ms;
  if (getenv("SVO_TIMING") && tk_n)
    fprintf(stderr, "[svo ba] device-side step control, %d slots, thread 0 per slot (us since kernel entry): payloads loaded %.1f, step consumed %.1f, system built %.1f, "
                    "factored + solved %.1f, status written %.1f; inside the solve: panels %.1f, trailing updates + barriers %.1f, back substitution %.1f\n", tk_n, 1e-2 * tk_sum[0] / tk_n, 1e-2 * tk_sum[1] / tk_n, 1e-2 * tk_sum[2] / tk_n, 1e-2 * tk_sum[3] / tk_n, 1e-2 * tk_sum[4] / tk_n,
            1e-2 * tk_sum[5] / tk_n, 1e-2 * tk_sum[6] / tk_n, 1e-2 * tk_sum[7] / tk_n);
  if (sum) {
    sum->iterations = (int)r[2]; sum->successful_steps = (int)r[3]; sum->termination = (int)r[4];
    sum->initial_cost = r[5]; sum->final_cost = r[6];
    sum->solve_ms = ms_between(t_begin, now());
  }
  d.flag = nullptr;
  d.points = ba->cur_points; d.cand_points = ba->cand_points; d.poses = ba->cur_poses; d.cand_poses = ba->cand_poses;
  (void)nn;
  return SVO_OK;
}

// A device-resident solve gave up (ba_lm_kernel's workgroups wait for each other; on a GPU that another process fills, or with
// LDS handed out in pieces, a launch may not become co-resident within its 3 s bound).  Nothing is lost: the pinned problem
// image was only read, so the adjuster goes back to "loaded, not uploaded" and the caller solves it again — compact form or
// host-driven loop, the same bits.  The wide form is avoided for the next 64 solves of this adjuster.
static void ba_after_giveup(svo_ba* ba) {
  uint8_t* D = ba->d_arena;
  ba->cur_points = (double*)(D + ba->arena_pts_off); ba->cand_points = (double*)(D + ba->arena_cpts_off);
  ba->cur_poses = (double*)(D + ba->arena_p0_off); ba->cand_poses = (double*)(D + ba->arena_p1_off);
  ba->d.points = ba->cur_points; ba->d.cand_points = ba->cand_points; ba->d.poses = ba->cur_poses; ba->d.cand_poses = ba->cand_poses;
  ba->arena_partial = false;
  ba->arena_dirty = true;
  ba->host_points_valid = false;
  ba->upload_pending = false;
  ba->lm_penalty = 64;
  ++ba->fallbacks;
}

// ceres::Solve for the loaded problem: host/lm.cpp's step control over the HIP passes.
static int ba_lm(svo_ba* ba, svo_ba_summary* sum) {
  svo_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  svo_lm_ops ops;
  ops.user = ba;
  ops.linearize = op_linearize;
  ops.step = op_step;
  ops.accept = op_accept;
  memset(&ba->stats, 0, sizeof(ba->stats));
  int fell_back = 0;
  if (ba_device_lm_launch(&ba, 1, ba->stream, false, nullptr) == 1) {  // the whole solve is one launch: nothing for the host to do until the completion word
    const int rcd = ba_device_lm_end(ba, sum);
    d.flag = nullptr;
    if (rcd == SVO_OK) { if (ba->lm_penalty > 0) --ba->lm_penalty; return rcd; }
    ba_after_giveup(ba);  // the problem is still loaded (pinned image untouched): solve it again below, never lose the keyframe
    fell_back = 1;
    if (ba_device_lm_launch(&ba, 1, ba->stream, true, nullptr) == 1) {  // (the compact form: one workgroup, nothing it could wait for)
      const int rc2 = ba_device_lm_end(ba, sum);
      d.flag = nullptr;
      if (rc2 == SVO_OK) { ba->stats.fallbacks = 1; ctx->err.clear(); return rc2; }
      ba_after_giveup(ba);
    }
  }
  if (ba_bulk_control_wanted(ba)) return ba_lm_bulk_device(ba, sum);  // bulk / sharded: nothing on the host inside an LM iteration
  {
    const int rcf = ba_flush_arena(ba);
    if (rcf) return rcf;
  }
  ba->host_points_valid = false;
  const int rc = svo_lm_solve(d.K, ba->h_poses.data(), &ops, &ba->opt, sum, &ba->stats);
  ba->n_spec += ba->stats.speculations; ba->n_hit += ba->stats.speculation_hits;
  // nothing is pending on the zero-copy path either; the wait keeps later users of the stream ordered (the resident
  // kernel's exit has published a completion word instead: stream order alone protects the next upload)
  if (!ba->host_points_valid) SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
  d.flag = nullptr;
  // leave the result in d.points / d.poses
  d.points = ba->cur_points; d.cand_points = ba->cand_points; d.poses = ba->cur_poses; d.cand_poses = ba->cand_poses;
  if (fell_back && rc == SVO_OK) { ba->stats.fallbacks = 1; ctx->err.clear(); }
  return rc;
}

extern "C" int svo_ba_load_problem(svo_ba* ba, int n_poses, const double* poses7, int n_points, const double* points3,
                                   int n_obs, const int32_t* obs_pose, const int32_t* obs_point, const double* obs_uv) {
  if (!ba) return SVO_ERR_INVALID;
  svo_use_device(ba->ctx);
  SVO_REQUIRE(ba->ctx, poses7 && (n_points == 0 || points3) && (n_obs == 0 || (obs_pose && obs_point && obs_uv)),
              "ba_load_problem: null buffer");
  return ba_upload(ba, n_poses, poses7, n_points, points3, n_obs, obs_pose, obs_point, obs_uv);
}

extern "C" int svo_ba_solve_problem(svo_ba* ba, svo_ba_summary* summary) {
  if (!ba) return SVO_ERR_INVALID;
  svo_use_device(ba->ctx);
  SVO_REQUIRE(ba->ctx, ba->d.K >= 1, "ba_solve_problem: no problem loaded");
  const int rc = ba_lm(ba, summary);
  if (!rc) ba->upload_pending = false;  // every LM path ends with the stream drained
  return rc;
}

extern "C" int svo_ba_read_problem(svo_ba* ba, double* poses7, double* points3) {
  if (!ba) return SVO_ERR_INVALID;
  svo_ctx* ctx = ba->ctx;
  if (poses7) memcpy(poses7, ba->h_poses.data(), sizeof(double) * 7 * (size_t)ba->d.K);
  if (points3 && ba->n_points && ba->host_points_valid) {
    // delivered by ba_lm_kernel; a landmark without observations was never touched: it keeps its uploaded value
    memcpy(points3, ba->h_out_points, sizeof(double) * 3 * (size_t)ba->n_points);
    const double* in = reinterpret_cast<const double*>(ba->h_arena + ba->arena_pts_off);
    const std::vector<int32_t>& lm = ba->u_lm_start;
    for (int j = 0; j < ba->n_points && (size_t)j + 1 < lm.size(); ++j)
      if (lm[j + 1] == lm[j]) { points3[3 * j] = in[3 * j]; points3[3 * j + 1] = in[3 * j + 1]; points3[3 * j + 2] = in[3 * j + 2]; }
  } else if (points3 && ba->n_points && ba->arena_dirty) {
    // loaded, never solved: the image is still on the host
    memcpy(points3, ba->h_arena + ba->arena_pts_off, sizeof(double) * 3 * (size_t)ba->n_points);
  } else if (points3 && ba->n_points) {
    // through the pinned arena (idle once the solve has finished): the runtime's pageable path would stage and wait
    const size_t bytes = sizeof(double) * 3 * (size_t)ba->n_points;
    void* stage = bytes <= ba->arena_cap ? (void*)ba->h_arena : (void*)points3;
    SVO_HIP_CHECK(ctx, hipMemcpyAsync(stage, ba->d.points, bytes, hipMemcpyDeviceToHost, ba->stream));
    SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
    ba->upload_pending = false;
    if (stage != (void*)points3) memcpy(points3, stage, bytes);
  }
  return SVO_OK;
}

// ---- sliding-window graph (BundleAdjuster::add_keyframe / bundle_adjust / get_world_points)
extern "C" int svo_ba_add_keyframe(svo_ba* ba, const double* pose7, const int64_t* tracked_ids, const float* tracked_xy,
                                   int n_tracked, const float* new_xy, const float* new_xyz, int n_new, int64_t* new_ids,
                                   int* n_new_out) {
  if (!ba) return SVO_ERR_INVALID;
  svo_ctx* ctx = ba->ctx;
  SVO_REQUIRE(ctx, pose7 && n_tracked >= 0 && n_new >= 0 && n_new_out, "ba_add_keyframe: bad arguments");
  SVO_REQUIRE(ctx, (n_tracked == 0 || (tracked_ids && tracked_xy)) && (n_new == 0 || (new_xy && new_xyz && new_ids)),
              "ba_add_keyframe: null buffer");
  svo_ba::PoseVar pv;
  memcpy(pv.pose, pose7, sizeof(pv.pose));  // src/bundle_adjuster.cpp:63-70
  const int64_t nfeat = (int64_t)(ba->feat_pos.size() / 3);
  for (int i = 0; i < n_tracked; ++i) {   // :72-83
    SVO_REQUIRE(ctx, tracked_ids[i] >= 0 && tracked_ids[i] < nfeat, "ba_add_keyframe: unknown feature id");
    pv.obs.push_back({tracked_xy[2 * i], tracked_xy[2 * i + 1], tracked_ids[i]});
  }
  const int maxf = ba->opt.max_features;
  const int max_new = n_tracked > maxf ? 0 : maxf - n_tracked;  // :85-90 with the C-5 guard
  const int keep = n_new > max_new ? max_new : n_new;
  for (int i = 0; i < keep; ++i) {        // :92-122; ids sequential (C-3), new_ids = real ids only (C-4)
    const int64_t id = (int64_t)(ba->feat_pos.size() / 3);
    ba->feat_pos.push_back(new_xyz[3 * i]); ba->feat_pos.push_back(new_xyz[3 * i + 1]); ba->feat_pos.push_back(new_xyz[3 * i + 2]);
    new_ids[i] = id;
    pv.obs.push_back({new_xy[2 * i], new_xy[2 * i + 1], id});
  }
  *n_new_out = keep;
  ba->window.push_back(std::move(pv));
  if ((int)ba->window.size() > ba->window_size) ba->window.pop_front();  // :126-128 (remove_oldest_pose)
  ba->new_frame_added = true;                                            // :134
  return SVO_OK;
}

extern "C" int svo_ba_reset(svo_ba* ba) {
  if (!ba) return SVO_ERR_INVALID;
  ba->window.clear();
  ba->feat_pos.clear();
  ba->new_frame_added = false;
  ba->d.K = 0;
  return SVO_OK;
}

extern "C" int svo_ba_window_count(svo_ba* ba) { return ba ? (int)ba->window.size() : 0; }

extern "C" int svo_ba_get_pose(svo_ba* ba, int k, double* pose7) {
  if (!ba || !pose7) return SVO_ERR_INVALID;
  const int K = (int)ba->window.size();
  if (k < 0) k += K;
  SVO_REQUIRE(ba->ctx, k >= 0 && k < K, "ba_get_pose: slot out of range");
  memcpy(pose7, ba->window[k].pose, 7 * sizeof(double));
  return SVO_OK;
}

extern "C" int svo_ba_get_points(svo_ba* ba, const int64_t* ids, int n, float* xyz) {
  if (!ba) return SVO_ERR_INVALID;
  SVO_REQUIRE(ba->ctx, n >= 0 && (n == 0 || (ids && xyz)), "ba_get_points: null buffer");
  const int64_t nfeat = (int64_t)(ba->feat_pos.size() / 3);
  for (int i = 0; i < n; ++i) {  // src/bundle_adjuster.cpp:159-163 (double -> float)
    SVO_REQUIRE(ba->ctx, ids[i] >= 0 && ids[i] < nfeat, "ba_get_points: unknown feature id");
    for (int a = 0; a < 3; ++a) xyz[3 * i + a] = (float)ba->feat_pos[3 * ids[i] + a];
  }
  return SVO_OK;
}

// The landmark store behind a HOST-DRIVEN solve (a lane whose window was not eligible or not admitted for ba_lm_kernel): the
// solved landmarks as store entries in pinned memory, one scatter launch, joined before the caller goes on (the next PnP of
// the lane is launched on another stream).  Rare; the device-resident solve writes its entries itself.
__global__ __launch_bounds__(256) void ba_store_scatter_kernel(const float4* __restrict__ stage, int n, float4* __restrict__ store, unsigned mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 e = stage[i];
  store[__float_as_uint(e.w) & mask] = e;
}

static int ba_store_scatter(svo_ba* ba, const std::vector<int64_t>& ids, const std::vector<double>& pts) {
  svo_ctx* ctx = ba->ctx;
  const size_t n = ids.size();
  if (n > ba->store_stage_cap) {
    if (ba->h_store_stage) (void)hipHostFree(ba->h_store_stage);
    ba->h_store_stage = nullptr;
    ba->store_stage_cap = n + n / 2 + 256;
    SVO_HIP_CHECK(ctx, hipHostMalloc((void**)&ba->h_store_stage, sizeof(float4) * ba->store_stage_cap, hipHostMallocDefault));
  }
  for (size_t j = 0; j < n; ++j) {
    float4 e;
    e.x = (float)pts[3 * j]; e.y = (float)pts[3 * j + 1]; e.z = (float)pts[3 * j + 2];  // src/bundle_adjuster.cpp:159-163 (double -> float)
    const unsigned key = (unsigned)ids[j];
    memcpy(&e.w, &key, 4);
    ba->h_store_stage[j] = e;
  }
  hipLaunchKernelGGL(ba_store_scatter_kernel, dim3(svo_div_up((int)n, 256)), dim3(256), 0, ba->stream, (const float4*)ba->h_store_stage, (int)n, ba->store, ba->store_mask);
  SVO_HIP_CHECK(ctx, hipGetLastError());
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(ba->stream));
  return SVO_OK;
}

int svo_ba_attach_store(svo_ba* ba, float4* store, unsigned mask) {
  if (!ba || (store && (mask & (mask + 1)) != 0)) return SVO_ERR_INVALID;
  ba->store = store; ba->store_mask = mask;
  return SVO_OK;
}

// BundleAdjuster::bundle_adjust in three steps, so that a driver of several stereo streams (host/group.cpp) can assemble
// on worker threads, launch the solves of several adjusters as ONE kernel and join them later; svo_ba_solve = all three.
// prepare: host only (no launch) — the window's observations as a landmark-major problem image in pinned memory.
// Returns 1 when there is nothing to solve (src/bundle_adjuster.cpp:138), 0 when a problem is loaded.
int svo_ba_solve_prepare(svo_ba* ba) {
  if (!ba) return SVO_ERR_INVALID;
  svo_use_device(ba->ctx);
  if (!ba->new_frame_added) return 1;  // src/bundle_adjuster.cpp:138
  const auto tp0 = std::chrono::steady_clock::now();
  const int K = (int)ba->window.size();
  // Observations sorted by (landmark id, window slot) — the order ceres would visit nothing in particular, but the
  // declared summation order is defined on it.  Every keyframe's observation list is ascending in id when it comes
  // from the pipeline (tracked ids keep their order, new ids are larger), so a K-way merge replaces the sort; lists
  // from other callers are checked and fall back to a stable sort.
  std::vector<double>& poses = ba->s_poses; std::vector<double>& points = ba->s_points; std::vector<double>& uv = ba->s_uv;
  std::vector<int32_t>& op = ba->s_op; std::vector<int32_t>& oj = ba->s_oj;
  std::vector<int64_t>& lm_ids = ba->solve_lm_ids;
  poses.resize(7 * (size_t)K); points.clear(); uv.clear(); op.clear(); oj.clear(); lm_ids.clear();
  points.reserve(3 * 4096); uv.reserve(2 * 4096); op.reserve(4096); oj.reserve(4096); lm_ids.reserve(4096);
  for (int k = 0; k < K; ++k) memcpy(&poses[7 * k], ba->window[k].pose, 7 * sizeof(double));
  bool ascending = true;
  size_t total = 0;
  for (int k = 0; k < K; ++k) {
    const auto& o = ba->window[k].obs;
    total += o.size();
    for (size_t i = 1; i < o.size() && ascending; ++i) ascending = o[i - 1].id <= o[i].id;
  }
  auto emit = [&](int k, const svo_ba::Obs& o) {
    if (lm_ids.empty() || lm_ids.back() != o.id) {
      lm_ids.push_back(o.id);
      for (int a = 0; a < 3; ++a) points.push_back(ba->feat_pos[3 * o.id + a]);
    }
    op.push_back(k); oj.push_back((int32_t)lm_ids.size() - 1);
    uv.push_back(o.u); uv.push_back(o.v);
  };
  if (ascending && K <= 64) {
    const svo_ba::Obs* cur[64];
    const svo_ba::Obs* end[64];
    for (int k = 0; k < K; ++k) { const auto& o = ba->window[k].obs; cur[k] = o.data(); end[k] = o.data() + o.size(); }
    for (size_t done = 0; done < total; ++done) {
      int best = -1;
      int64_t best_id = 0;
      for (int k = 0; k < K; ++k)  // smallest id first; equal ids in window-slot order (what the stable sort gives)
        if (cur[k] != end[k] && (best < 0 || cur[k]->id < best_id)) { best = k; best_id = cur[k]->id; }
      emit(best, *cur[best]++);
    }
  } else {
    struct Flat { int k; const svo_ba::Obs* o; };
    std::vector<Flat> flat;
    flat.reserve(total);
    for (int k = 0; k < K; ++k)
      for (const auto& o : ba->window[k].obs) flat.push_back({k, &o});
    std::stable_sort(flat.begin(), flat.end(), [](const Flat& a, const Flat& b) { return a.o->id < b.o->id; });
    for (const Flat& f : flat) emit(f.k, *f.o);
  }
  const auto tu0 = std::chrono::steady_clock::now();
  ba->t_prep += std::chrono::duration<double, std::milli>(tu0 - tp0).count();
  ba->upload_has_ids = true;   // the landmark-store keys of this problem = lm_ids
  int rc = ba_upload(ba, K, poses.data(), (int)lm_ids.size(), points.data(), (int)op.size(), op.data(), oj.data(), uv.data());
  ba->upload_has_ids = false;
  if (rc) return rc;
  ba->t_upload += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count();
  return SVO_OK;
}

// launch: the prepared problems of `n` adjusters as ONE ba_lm_kernel launch on `stream`; returns how many (a prefix) were
// eligible and admitted.  The others: offer them again, or svo_ba_solve_finish runs the host-driven loop for them.
int svo_ba_solve_launch(svo_ba** bas, int n, void* stream, unsigned long long* launched_mask) {
  if (launched_mask) *launched_mask = 0;
  if (!bas || n < 1) return 0;
  svo_use_device(bas[0]->ctx);
  return ba_device_lm_launch(bas, n, stream ? (hipStream_t)stream : bas[0]->stream, true, launched_mask);
}

// 1: the launched solve has published its completion word (svo_ba_solve_finish will not block), 0: still running
int svo_ba_solve_poll(svo_ba* ba) {
  if (!ba || !ba->lm_inflight) return 1;
  const int v = __atomic_load_n(ba->h_flag, __ATOMIC_ACQUIRE);
  return (v == ba->seq || (v == -ba->seq && ba->seq != 0)) ? 1 : 0;  // published, or gave up (svo_ba_solve_finish re-runs it): either way the join will not wait
}

// finish: join the launched solve (or, if none was launched for this adjuster, run the host-driven loop now) and write
// poses and landmarks back into the graph (src/bundle_adjuster.cpp:146-155).
int svo_ba_solve_finish(svo_ba* ba, svo_ba_summary* summary) {
  if (!ba) return SVO_ERR_INVALID;
  svo_use_device(ba->ctx);
  if (summary) memset(summary, 0, sizeof(*summary));
  const auto tu0 = std::chrono::steady_clock::now();
  const int K = ba->d.K;
  std::vector<double>& poses = ba->s_poses; std::vector<double>& points = ba->s_points;
  std::vector<int64_t>& lm_ids = ba->solve_lm_ids;
  int rc;
  const bool on_device = ba->lm_inflight;
  bool on_device_ok = on_device;
  if (ba->lm_inflight) {
    rc = ba_device_lm_end(ba, summary); ba->d.flag = nullptr;
    if (rc == SVO_OK) { if (ba->lm_penalty > 0) --ba->lm_penalty; }
    else {  // gave up: never lose the keyframe — the problem is still loaded, solve it again (compact form, else host-driven)
      ba_after_giveup(ba);
      rc = ba_lm(ba, summary);
      if (rc == SVO_OK) { ba->stats.fallbacks = 1; ba->ctx->err.clear(); }
      on_device_ok = false;  // (the landmark store is written by the scatter below unless the re-run was device-resident — harmless twice)
    }
  } else rc = ba_lm(ba, summary);
  if (!rc) ba->upload_pending = false;
  ba->t_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tu0).count();
  ba->n_solves++;
  if (rc) return rc;
  if (summary && on_device) {  // device-resident solves only: the launches bench.py times with HIP events carry exactly these
    const double its = summary->iterations + 1;  // + the first linearisation
    ba->acc_flops += its * ba->flops_iter; ba->acc_bytes += its * ba->bytes_iter; ba->acc_solves += 1; ba->acc_iters += summary->iterations;
  }
  const auto tr0 = std::chrono::steady_clock::now();
  std::vector<double>& out_pts = ba->s_out_pts;
  out_pts.resize(points.size());
  rc = svo_ba_read_problem(ba, poses.data(), out_pts.data());
  if (rc) return rc;
  for (int k = 0; k < K; ++k) memcpy(ba->window[k].pose, &poses[7 * k], 7 * sizeof(double));
  for (size_t l = 0; l < lm_ids.size(); ++l)
    for (int a = 0; a < 3; ++a) ba->feat_pos[3 * lm_ids[l] + a] = out_pts[3 * l + a];
  ba->new_frame_added = false;  // :155
  if (ba->store && !on_device_ok && !lm_ids.empty()) {  // the device-resident solve wrote the landmark store itself
    rc = ba_store_scatter(ba, lm_ids, out_pts);
    if (rc) return rc;
  }
  ba->t_read += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count();
  return SVO_OK;
}

// algorithmic work (SURVEY 8d figures) of the solves finished since the last reset: [f64 flops, bytes, solves, LM iterations]
void svo_ba_work(svo_ba* ba, double* out4, int reset) {
  out4[0] = ba->acc_flops; out4[1] = ba->acc_bytes; out4[2] = ba->acc_solves; out4[3] = ba->acc_iters;
  if (reset) ba->acc_flops = ba->acc_bytes = ba->acc_solves = ba->acc_iters = 0;
}

extern "C" int svo_ba_solve(svo_ba* ba, svo_ba_summary* summary) {
  if (!ba) return SVO_ERR_INVALID;
  if (summary) memset(summary, 0, sizeof(*summary));
  const int rc = svo_ba_solve_prepare(ba);
  if (rc == 1) return SVO_OK;  // src/bundle_adjuster.cpp:138
  if (rc) return rc;
  return svo_ba_solve_finish(ba, summary);
}
