// Context lifetime and workspace (svo.h: svo_create / svo_destroy / svo_stream / svo_sync).
#include <algorithm>
#include <chrono>
#include <sched.h>

#include "common.h"

extern "C" const char* svo_version(void) { return "stereo_vo_amd 0.1 (gfx950, HIP, wave64)"; }

extern "C" const char* svo_last_error(const svo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" void* svo_stream(svo_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int svo_sync(svo_ctx* ctx) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return SVO_OK;
}

extern "C" int svo_profile_select(svo_ctx* ctx, const char* kernel) {
  if (!ctx) return SVO_ERR_INVALID;
  svo_use_device(ctx);
  static const char* names[] = {"", "corner_response", "corner_nms", "corner_select", "pyr_down", "lk_fb", "stereo_at",
                                "triangulate", "pnp_hypotheses", "pnp_refine", "ba_linearize", "ba_backsub", "ba_step"};
  int tag = 0;
  if (kernel && kernel[0]) {
    tag = -1;
    for (int i = 1; i < (int)(sizeof(names) / sizeof(names[0])); ++i)
      if (!strcmp(kernel, names[i])) tag = i;
    SVO_REQUIRE(ctx, tag > 0, "profile_select: unknown kernel name");
  }
  SVO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (tag && ctx->prof_ev.empty()) {
    ctx->prof_ev.resize(2 * 16384);
    for (auto& e : ctx->prof_ev) SVO_HIP_CHECK(ctx, hipEventCreate(&e));
  }
  ctx->prof_tag = tag;
  ctx->prof_used = 0;
  return SVO_OK;
}

extern "C" int svo_profile_read(svo_ctx* ctx, double* total_ms, int* launches) {
  if (!ctx || !total_ms || !launches) return SVO_ERR_INVALID;
  SVO_HIP_CHECK(ctx, hipDeviceSynchronize());  // profiled kernels may run on the BA stream
  double tot = 0.0;
  for (int i = 0; i < ctx->prof_used; ++i) {
    float ms = 0.f;
    SVO_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = ctx->prof_used;
  return SVO_OK;
}

extern "C" int svo_create(svo_ctx** out, int device, const svo_limits* lim) {
  if (!out || !lim) return SVO_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return SVO_ERR_NO_DEVICE;
  svo_ctx* c = new svo_ctx();
  c->device = device;
  c->lim = *lim;
  if (c->lim.max_batch < 1) c->lim.max_batch = 1;
  if (c->lim.max_corners < 1) c->lim.max_corners = 300;
  if (c->lim.max_candidates < 1024) c->lim.max_candidates = 1024;
  if (c->lim.max_features < 1) c->lim.max_features = 400;
  auto fail = [&](const char* what, hipError_t e) {
    fprintf(stderr, "svo_create: %s: %s\n", what, hipGetErrorString(e));
    svo_destroy(c);
    return SVO_ERR_HIP;
  };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
  {
    const char* ev = getenv("SVO_BA_CU_SHARE");  // see ba.hip / include/svo.h: the complement of the adjusters' CUs
    const int nres = ev ? atoi(ev) : 0;
    if (nres > 0 && nres < 32) {
      uint32_t mask[8];
      for (int i = 0; i < 8; ++i) mask[i] = ~((1u << nres) - 1u);
      if ((e = hipExtStreamCreateWithCUMask(&c->stream, 8, mask)) != hipSuccess) return fail("stream (CU mask)", e);
    } else if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return fail("stream", e);
  }
  const size_t px = (size_t)c->lim.max_width * c->lim.max_height;
  const size_t B = (size_t)c->lim.max_batch;
  // generic scratch: enough for two images + pyramids + dense outputs of the host-pointer wrappers
  c->ws_bytes = 24 * px + (size_t)(64 << 20);
  c->max_cells = (int)(px / 4 + 64);  // min_distance >= 2 px cells
  c->pinned_bytes = 1 << 20;
#define ALLOC(ptr, bytes) \
  if ((e = hipMalloc((void**)&(ptr), (bytes))) != hipSuccess) return fail(#ptr, e)
  ALLOC(c->d_ws, c->ws_bytes);
  // raw local maxima of the streaming detection pass: at most one 3x3 maximum per 2x2 pixels can exist (px / 4; ties on
  // plateaus aside).  How many are RECORDED depends on timing (a candidate is dropped early against the image's running
  // maximum), so the list is sized by that geometric bound, not by max_candidates (round 4 used 2 x max_candidates: contexts
  // with a small candidate bound could then overflow on textured frames although the final candidates fit — ADVICE r4),
  // under a budget of 1 GiB per context and never below twice the candidate bound; an overflow has its own status bit (8)
  // and message.  8 B per entry: 358 MB for the bench's 384-frame context at 1241x376.
  // (The f32 response map of rounds 1-2, 4 px bytes per frame, is not allocated here, see svo_ensure_eig.)
  {
    const size_t geometric = px / 4 + 4096, budget = ((size_t)1 << 27) / B;  // entries
    c->raw_cap = std::max(std::min(geometric, std::max(budget, (size_t)4096)), std::min(px / 2, (size_t)2 * (size_t)c->lim.max_candidates));
  }
  ALLOC(c->d_raw, B * c->raw_cap * sizeof(unsigned long long));
  ALLOC(c->d_maxkey, B * sizeof(unsigned));
  ALLOC(c->d_cand, B * c->lim.max_candidates * sizeof(unsigned long long));
  ALLOC(c->d_sorted, B * c->lim.max_candidates * sizeof(unsigned long long));
  ALLOC(c->d_state, B * c->lim.max_candidates);
  ALLOC(c->d_ncand, B * 32 * sizeof(int));  // one 128-B line per image (corner.hip NC_STRIDE)
  ALLOC(c->d_cell_count, B * (size_t)c->max_cells * sizeof(int));
  ALLOC(c->d_cell_start, B * ((size_t)c->max_cells + 1) * sizeof(int));
  ALLOC(c->d_status, 64);
#undef ALLOC
  if ((e = hipHostMalloc(&c->h_pinned, c->pinned_bytes, hipHostMallocDefault)) != hipSuccess) return fail("pinned", e);
  memset(c->h_pinned, 0, c->pinned_bytes);  // completion words are compared by equality: never start from recycled bytes
  if ((e = hipMemsetAsync(c->d_status, 0, 64, c->stream)) != hipSuccess) return fail("memset", e);
  if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return fail("sync", e);
  *out = c;
  return SVO_OK;
}

// the f32 response map for `images` frames (response tap / two-pass detection only)
int svo_ensure_eig(svo_ctx* c, size_t images) {
  if (c->d_eig && c->eig_images >= images) return SVO_OK;
  if (c->d_eig) { SVO_HIP_CHECK(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_eig); c->d_eig = nullptr; c->eig_images = 0; }
  const size_t px = (size_t)c->lim.max_width * c->lim.max_height;
  SVO_HIP_CHECK(c, hipMalloc((void**)&c->d_eig, images * px * sizeof(float)));
  c->eig_images = images;
  return SVO_OK;
}

int svo_wait_word(svo_ctx* c, const SvoPublish& p) {
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (__atomic_load_n(p.word, __ATOMIC_ACQUIRE) != p.seq) {
    __builtin_ia32_pause();
    if (++spins > 4096u && (spins & 63u) == 0) sched_yield();  // long wait: stay polite when threads outnumber cores
    if ((spins & 0xFFFFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
      // never expected: fall back to the stream wait so that a lost word cannot hang the caller
      SVO_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      if (__atomic_load_n(p.word, __ATOMIC_ACQUIRE) != p.seq) { c->err = "completion word never arrived"; return SVO_ERR_HIP; }
    }
  }
  return SVO_OK;
}

extern "C" void svo_destroy(svo_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void* ptrs[] = {c->d_ws, c->d_eig, c->d_raw, c->d_maxkey, c->d_cand, c->d_sorted, c->d_state, c->d_ncand,
                  c->d_cell_count, c->d_cell_start, c->d_status};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& e : c->prof_ev) (void)hipEventDestroy(e);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
