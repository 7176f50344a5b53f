// The XCD-aware item -> workgroup map of the stream-batched launches (stereo_vo_amd/csrc/xcd_map.h) walked on the host: the
// device function's own text is compiled here (blockIdx is a plain variable), every workgroup of the grid is asked for its
// (lane, item), and the invariants the arrival counters rely on are checked:
//   * every slot (lane j, item i < max(n_j, 1)) is served by exactly one workgroup — a lane's last-arrival target max(n_j, 1) is
//     reached exactly once, a lane without items still gets its one idle workgroup;
//   * the workgroups that serve nothing are exactly grid - total;
//   * the workgroups with the same (b mod 8) — one XCD — serve chunks/8 contiguous runs of slots.
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define __HIPCC__ 1
#define __device__
#define __forceinline__ inline
static struct { unsigned x, y, z; } blockIdx;
#include "xcd_map.h"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { ++fails; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 32); }

static void one_case(const std::vector<int>& counts, int chunks) {
  SvoXcdMap m;
  svo_xcd_map_fill(m, counts.data(), (int)counts.size(), chunks);
  const int L = (int)counts.size();
  int expect_total = 0;
  for (int c : counts) expect_total += c > 1 ? c : 1;
  CHECK(m.total == expect_total && m.prefix[L] == expect_total, "total %d, expected %d", m.total, expect_total);
  CHECK(m.grid() >= m.total && m.grid() < m.total + chunks, "grid %d for %d slots in %d chunks", m.grid(), m.total, chunks);
  CHECK(m.grid() % 8 == 0, "grid %d is not a multiple of 8", m.grid());
  std::vector<std::vector<int>> seen(L);
  for (int j = 0; j < L; ++j) seen[j].assign(counts[j] > 1 ? counts[j] : 1, 0);
  int idle = 0;
  std::vector<int> runs(8, 0), last(8, -2);
  for (int b = 0; b < m.grid(); ++b) {
    blockIdx.x = (unsigned)b; blockIdx.y = 0;
    int li = -1, it = -1;
    if (!svo_xcd_map_item(m, li, it)) { ++idle; continue; }
    CHECK(li >= 0 && li < L && it >= 0 && it < (int)seen[li].size(), "block %d -> lane %d item %d out of range", b, li, it);
    if (li < 0 || li >= L || it < 0 || it >= (int)seen[li].size()) continue;
    ++seen[li][it];
    const int slot = m.prefix[li] + it, x = b & 7;
    if (slot != last[x] + 1) ++runs[x];
    last[x] = slot;
  }
  for (int j = 0; j < L; ++j)
    for (size_t i = 0; i < seen[j].size(); ++i) CHECK(seen[j][i] == 1, "lane %d item %zu served %d times", j, i, seen[j][i]);
  CHECK(idle == m.grid() - m.total, "%d idle workgroups, expected %d", idle, m.grid() - m.total);
  for (int x = 0; x < 8; ++x) CHECK(runs[x] <= chunks / 8, "XCD label %d serves %d runs of slots, at most %d expected", x, runs[x], chunks / 8);
}

int main() {
  for (int chunks : {8, 16, 32, 64}) {
    one_case({0}, chunks);
    one_case({1}, chunks);
    one_case({5}, chunks);
    one_case({0, 0, 0}, chunks);
    one_case({731, 0, 1500, 1, 64, 63, 65}, chunks);
    one_case(std::vector<int>(SVO_MAX_LANES, 3300), chunks);
    for (int rep = 0; rep < 200; ++rep) {
      std::vector<int> c(1 + rnd() % SVO_MAX_LANES);
      for (int& v : c) v = (rnd() % 4 == 0) ? (int)(rnd() % 3) : (int)(rnd() % 3301);
      one_case(c, chunks);
    }
  }
  // the plain grid: blockIdx = (item, lane)
  SvoXcdMap m{};
  blockIdx.x = 17; blockIdx.y = 5;
  int li = -1, it = -1;
  CHECK(svo_xcd_map_item(m, li, it) && li == 5 && it == 17, "plain grid: lane %d item %d", li, it);
  if (fails) { fprintf(stderr, "%d checks failed\n", fails); return 1; }
  printf("xcd map ok\n");
  return 0;
}
