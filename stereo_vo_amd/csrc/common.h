// Shared internals of libsvo_hip.so (gfx950 only).
#ifndef SVO_COMMON_H_
#define SVO_COMMON_H_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "svo.h"

// Last-error text of a context.  The caller thread and the asynchronous bundle-adjustment worker may both report
// (and the caller clears / tests it between frames), so every access is serialised.
struct SvoErr {
  SvoErr& operator=(const std::string& v) { std::lock_guard<std::mutex> g(mu_); s_ = v; return *this; }
  SvoErr& operator=(const char* v) { std::lock_guard<std::mutex> g(mu_); s_ = v ? v : ""; return *this; }
  bool empty() const { std::lock_guard<std::mutex> g(mu_); return s_.empty(); }
  void clear() { std::lock_guard<std::mutex> g(mu_); s_.clear(); }
  const char* c_str() const {  // a per-thread copy: stays valid while another thread reports
    thread_local std::string copy;
    std::lock_guard<std::mutex> g(mu_);
    copy = s_;
    return copy.c_str();
  }
 private:
  mutable std::mutex mu_;
  std::string s_;
};

struct svo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  svo_limits lim{};
  SvoErr err;
  // persistent device workspace (allocated once in svo_create)
  uint8_t* d_ws = nullptr;   // generic scratch for host-pointer entry points
  size_t ws_bytes = 0;
  // front-end buffers, sized for max_batch frames
  float* d_eig = nullptr;              // f32 response map, eig_images * W*H: allocated on first use only (svo_ensure_eig: the
  size_t eig_images = 0;               // response tap svo_corner_response and the two-pass form SVO_CORNER_TWO_PASS=1)
  unsigned long long* d_raw = nullptr; // batch * raw_cap raw local maxima of the streaming detection pass (key<<32 | y<<16|x)
  size_t raw_cap = 0;
  unsigned* d_maxkey = nullptr;        // batch
  unsigned long long* d_cand = nullptr;  // batch * max_candidates  (key<<32 | raster idx)
  int* d_ncand = nullptr;              // batch
  int* d_cell_count = nullptr;         // batch * max_cells
  int* d_cell_start = nullptr;         // batch * (max_cells + 1)
  unsigned long long* d_sorted = nullptr;  // batch * max_candidates (cell-ordered keys)
  uint8_t* d_state = nullptr;          // batch * max_candidates
  int max_cells = 0;
  int* d_status = nullptr;             // device status word (capacity overflow etc.)
  // pinned host mirror for small readbacks
  void* h_pinned = nullptr;
  size_t pinned_bytes = 0;
  // per-kernel HIP-event timing (svo_profile_select / svo_profile_read)
  int prof_tag = 0;
  std::vector<hipEvent_t> prof_ev;  // start/stop pairs
  int prof_used = 0;                // pairs recorded
  // completion words (see SvoPublish)
  int word_seq = 0;
  unsigned arrive_total = 0;
};

// Host-visible completion words.  A kernel whose results the host needs writes them straight into pinned
// host memory and then publishes a sequence number (system-scope fence, then one release store); the host
// polls the word.  This replaces "D2H blit + hipStreamSynchronize" per readback: no copy kernel, no runtime
// call on the wait, and with many stereo streams per GPU no contention on the runtime's locks.
enum SvoWord { SVO_W_TRACK = 0, SVO_W_INIT, SVO_W_PNP_HYP, SVO_W_PNP_REF, SVO_W_TRI, SVO_W_N };
struct SvoPublish {
  int* word = nullptr;         // pinned host word
  int seq = 0;                 // value that means "this launch is complete"
  unsigned* arrive = nullptr;  // multi-workgroup kernels: device counter (monotone), last arrival publishes
  unsigned target = 0;
};
inline int* svo_word(svo_ctx* c, int which) { return reinterpret_cast<int*>(static_cast<char*>(c->h_pinned) + c->pinned_bytes - 4096) + 16 * which; }
inline SvoPublish svo_publish_next(svo_ctx* c, int which, int nblocks = 1) {
  SvoPublish p;
  p.word = svo_word(c, which);
  p.seq = ++c->word_seq;
  if (nblocks > 1) {
    c->arrive_total += (unsigned)nblocks;
    p.arrive = reinterpret_cast<unsigned*>(c->d_status + 8);
    p.target = c->arrive_total;
  }
  return p;
}
// arrival counter only (a hand-over between the workgroups of one launch, no host word)
inline SvoPublish svo_arrive_next(svo_ctx* c, int nblocks) {
  SvoPublish p;
  c->arrive_total += (unsigned)nblocks;
  p.arrive = reinterpret_cast<unsigned*>(c->d_status + 8);
  p.target = c->arrive_total;
  return p;
}
int svo_wait_word(svo_ctx* c, const SvoPublish& p);  // ctx.hip: bounded spin, falls back to a stream wait
int svo_ensure_eig(svo_ctx* c, size_t images);       // ctx.hip: the f32 response map, on first use

#if defined(__HIPCC__)
// First statement of every small kernel that sits on a stereo stream's serial path (the keyframe chain, the adjuster's
// passes): raised wave priority, so that on a SIMD shared with other streams' long tracker waves (priority 0, hundreds of
// microseconds of VALU work each) these few instructions issue first.
__device__ __forceinline__ void svo_latency_critical() { __builtin_amdgcn_s_setprio(3); }

// Called by EVERY thread of the workgroup after its last store to host memory.
// Cache maintenance is kept to the one operation the protocol needs.  `__threadfence_system()` and an acq_rel atomic are
// each `buffer_wbl2` + `buffer_inv`: the invalidate empties the whole L2 of the workgroup's XCD under every other kernel
// running there (with eight stereo streams per GPU that was a steady drizzle of L2 wipes).  Nobody here READS another
// workgroup's data, so no acquire is needed: a release fence per thread (write-back + wait for the acknowledgements),
// the barrier, a relaxed arrival, and a relaxed system-scope store of the word by the last arrival — which, through
// the counter, comes after every workgroup's acknowledged payload.
__device__ __forceinline__ void svo_publish_block(const SvoPublish& p) {
  if (!p.word) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (threadIdx.x == 0) {
    bool last = true;
    if (p.arrive) last = __hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == p.target;
    if (last) __hip_atomic_store(p.word, p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The same without ANY cache maintenance, for kernels whose host payload is a few words: write them with svo_host_store
// (relaxed system-scope stores are written through every cache level; `s_waitcnt vmcnt(0)` then means "acknowledged")
// and publish with svo_publish_block_wt.  Used where many workgroups publish (one `buffer_wbl2` per workgroup adds up:
// see the measurements at slot_store2 in ba.hip).
template <typename T>
__device__ __forceinline__ void svo_host_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void svo_publish_block_wt(const SvoPublish& p) {
  if (!p.word) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    bool last = true;
    if (p.arrive) last = __hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == p.target;
    if (last) __hip_atomic_store(p.word, p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#endif

enum SvoProfTag { SVO_PROF_NONE = 0, SVO_PROF_CORNER_RESPONSE, SVO_PROF_CORNER_NMS, SVO_PROF_CORNER_SELECT,
                  SVO_PROF_PYR_DOWN, SVO_PROF_LK_FB, SVO_PROF_STEREO_AT, SVO_PROF_TRIANGULATE, SVO_PROF_PNP_HYP,
                  SVO_PROF_PNP_REFINE, SVO_PROF_BA_LINEARIZE, SVO_PROF_BA_BACKSUB, SVO_PROF_BA_STEP };

// RAII event pair around one launch of the selected kernel (no-op for every other kernel).
struct SvoProfScope {
  svo_ctx* c;
  hipStream_t st;
  int slot = -1;
  SvoProfScope(svo_ctx* ctx, int tag, hipStream_t stream = nullptr) : c(ctx), st(stream ? stream : ctx->stream) {
    if (ctx->prof_tag == tag && 2 * (ctx->prof_used + 1) <= (int)ctx->prof_ev.size()) {
      slot = ctx->prof_used++;
      (void)hipEventRecord(ctx->prof_ev[2 * slot], st);
    }
  }
  ~SvoProfScope() {
    if (slot >= 0) (void)hipEventRecord(c->prof_ev[2 * slot + 1], st);
  }
};

#define SVO_HIP_CHECK(ctx, expr)                                                        \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                  \
      return SVO_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define SVO_REQUIRE(ctx, cond, msg)          \
  do {                                       \
    if (!(cond)) {                           \
      if (ctx) (ctx)->err = (msg);           \
      return SVO_ERR_INVALID;                \
    }                                        \
  } while (0)

inline int svo_div_up(int a, int b) { return (a + b - 1) / b; }

// Every C-ABI entry point may be called from a thread that never selected the context's GPU.
inline void svo_use_device(const svo_ctx* c) { if (c) (void)hipSetDevice(c->device); }

// Scratch carve-out from ctx->d_ws for host-pointer wrappers.
struct SvoScratch {
  svo_ctx* ctx;
  size_t off = 0;
  explicit SvoScratch(svo_ctx* c) : ctx(c) {}
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    if (off + n * sizeof(T) > ctx->ws_bytes) return nullptr;
    T* p = reinterpret_cast<T*>(ctx->d_ws + off);
    off += n * sizeof(T);
    return p;
  }
};

#ifdef __HIPCC__
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}
#endif

#endif  // SVO_COMMON_H_
