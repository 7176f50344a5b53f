// The XCD-aware item -> workgroup map of the stream-batched launches (group_kernels.h).  A header of its own, free of HIP
// includes, so that tests/sanitize/xcd_map_test.cpp can compile the device function's very text for the host and walk the grid.
#ifndef SVO_XCD_MAP_H_
#define SVO_XCD_MAP_H_

constexpr int SVO_MAX_LANES = 64;  // lanes of one pipeline group (round 5: 32 -> 64; the launches' lane records travel in the kernel-argument segment, masks are 64-bit)

// XCD-aware item -> workgroup map of a stream-batched launch (a speed choice only; per_chunk == 0: blockIdx = (item, lane)).  The
// launch's items (features, corners), lane after lane — a lane without items counts one idle workgroup: somebody has to arrive for
// it —, are cut into `chunks` contiguous chunks of per_chunk; workgroup b works on chunk (b % 8) + 8 * ((b / 8) / per_chunk):
// workgroups b and b + 8 share an XCD (observed dispatch, MI355X_MICROARCH.md), so an XCD's L2 sees the images of the two or three
// lanes its chunks lie in instead of every lane's.  prefix[j] = first slot of lane j, prefix[n_lanes] = total; a lane's arrival
// target then counts max(n, 1) workgroups instead of the grid's width.
constexpr int SVO_XCD_CHUNKS = 16;
struct SvoXcdMap {
  int per_chunk, total, chunks;  // chunks: a multiple of 8
  int prefix[SVO_MAX_LANES + 1];
  int grid() const { return chunks * per_chunk; }
};
// counts[j] = items of lane j (the map reserves max(counts[j], 1) slots)
inline void svo_xcd_map_fill(SvoXcdMap& m, const int* counts, int n_lanes, int chunks) {
  int tot = 0;
  for (int j = 0; j < n_lanes; ++j) { m.prefix[j] = tot; tot += counts[j] > 1 ? counts[j] : 1; }
  m.prefix[n_lanes] = tot;
  m.total = tot; m.chunks = chunks; m.per_chunk = (tot + chunks - 1) / chunks;
}
#if defined(__HIPCC__)
// false: this workgroup lies beyond the last chunk's end (nobody counts it).  Everything here is wave-uniform.
__device__ __forceinline__ bool svo_xcd_map_item(const SvoXcdMap& m, int& lane_index, int& item) {
  if (m.per_chunk <= 0) { lane_index = (int)blockIdx.y; item = (int)blockIdx.x; return true; }
  const int slot = (int)(blockIdx.x >> 3), sub = slot / m.per_chunk;
  const int idx = ((int)(blockIdx.x & 7) + 8 * sub) * m.per_chunk + (slot - sub * m.per_chunk);
  if (idx >= m.total) return false;
  int li = 0;
  while (idx >= m.prefix[li + 1]) ++li;
  lane_index = li; item = idx - m.prefix[li];
  return true;
}
#endif
#endif  // SVO_XCD_MAP_H_
