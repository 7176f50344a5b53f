#include "svo_adapter.hpp"

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace svo_adapter {
namespace {
std::mutex g_mu;
svo_ctx* g_ctx = nullptr;
bool g_tried = false;

int env_int(const char* name, int fallback) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : fallback;
}
}  // namespace

svo_ctx* context() {
  std::lock_guard<std::mutex> g(g_mu);
  if (g_ctx || g_tried) return g_ctx;
  g_tried = true;
  svo_limits lim;
  lim.max_width = env_int("SVO_ADAPTER_MAX_WIDTH", 1920);
  lim.max_height = env_int("SVO_ADAPTER_MAX_HEIGHT", 1200);
  lim.max_batch = 1;              // vo_node hands over one pair at a time (src/vo_node.cpp:141-144)
  lim.max_corners = 300;          // src/image_processor.cpp:22
  lim.max_candidates = 1 << 17;
  lim.max_features = 400;         // src/bundle_adjuster.hpp:75
  const int rc = svo_create(&g_ctx, env_int("SVO_ADAPTER_DEVICE", 0), &lim);
  if (rc != SVO_OK) {
    fprintf(stderr, "stereo_vo adapters: svo_create failed (status %d): no MI355X visible or allocation failed; "
                    "every call will return without effect\n", rc);
    g_ctx = nullptr;
  }
  return g_ctx;
}

void shutdown() {
  std::lock_guard<std::mutex> g(g_mu);
  if (g_ctx) svo_destroy(g_ctx);
  g_ctx = nullptr;
  g_tried = false;
}

double ba_max_time_s() {
  const char* v = getenv("SVO_ADAPTER_BA_MAX_TIME_S");
  return v && *v ? atof(v) : 0.1;  // src/bundle_adjuster.cpp:11
}

}  // namespace svo_adapter
