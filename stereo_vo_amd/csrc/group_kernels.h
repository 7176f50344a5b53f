// Stream-batched launches: ONE launch of a stage of the hot path covers several stereo streams ("lanes" of a pipeline
// group, host/group.cpp), blockIdx.y = lane or the XCD-aware map below.  The kernels run the SAME device bodies as the single-stream launches
// (kernels.h) on per-lane argument records that travel in the kernel-argument segment, so a lane's results are bit for
// bit those of its own svo_pipeline.  Why: with eight stereo streams as eight host threads and 16-24 hardware queues a
// launch -> completion round trip costs 30-90 us instead of 7-11 and concurrent small kernels run 3-5x their solo time
// (profiles/r02_exp_launch_rate.txt, r02_exp_placement.txt); one host thread and at most four queues avoid both.
// Reference stages: src/feature_tracker.cpp:18-67 (track), src/image_processor.cpp:76-80 (PnP), :113-128 (dedup),
// :173-207 (StereoBM at the features + triangulation).
#ifndef SVO_GROUP_KERNELS_H_
#define SVO_GROUP_KERNELS_H_
#include "kernels.h"

#include "chain_math.h"  // SvoChainRec
#include "xcd_map.h"  // SVO_MAX_LANES, SvoXcdMap

// a3 + the survivor filter of FeatureTracker::track_features, forward/backward LK and the stable compaction in ONE launch:
// the last wavefront of a lane to arrive compacts the lane (no second launch, no second round trip).
struct SvoLkLane {
  const uint8_t* pyr_prev; const uint8_t* pyr_next;
  const uint8_t* l0_prev; const uint8_t* l0_next;                  // level 0 of the two pyramids when it is read from the caller's images in place (null: inside the pyramid)
  const float* xy; const float* init_xy; const long long* ids;   // current features (device, or the pinned arrays a keyframe left)
  int n;
  float* fwd; uint8_t* keep; float* parallax;                      // per-feature scratch (device)
  float* kept_xy; float* init_dst; long long* ids_dst;             // compacted set (device)
  float* host_xy; long long* host_ids; int* host_n; float* host_av;  // pinned mirrors + (n_kept, av_parallax)
  unsigned* arrive; unsigned arrive_target;                        // device arrival counter of the lane (monotone)
  int* word; int seq;                                              // pinned completion word
};
struct SvoLkLanes { int w, h; SvoXcdMap map; SvoLkLane lane[SVO_MAX_LANES]; };
// map.per_chunk == 0: grid_x = the largest lane's feature count; every lane's arrive_target counts grid_x workgroups
int svo_kg_track(svo_ctx* ctx, hipStream_t st, const SvoLkLanes& lanes, int n_lanes, int grid_x);

// a5: the whole cv::solvePnPRansac of a lane (src/image_processor.cpp:72-80) as ONE launch: hypotheses (four per workgroup),
// RANSAC bookkeeping and refinement by the lane's last workgroup to arrive; the world points (get_world_points,
// src/bundle_adjuster.cpp:159-163) come from the lane's device-resident landmark store, keyed by feature id
// store entry (id mod capacity) = float4 {x, y, z, bits of the low 32 bits of the id}
struct SvoPnpLane {
  const long long* ids; const float4* store; unsigned store_mask;  // tracked features' ids (device) -> store entries
  const float* xy; int n; double f, cx, cy; double q0[4], t0[3]; double thr2, confidence; int iterations;
  int launched;  // hypotheses 0 .. launched-1 are computed by this launch (<= iterations); host_best = -2: the bookkeeping needs more of them
  double* hyp_pose; int* hyp_count; unsigned long long* hyp_mask; int mask_words;   // per-hypothesis scratch (device)
  double* out_pose; int* inliers; int* n_inliers; float* inlier_xy;                // device results (inlier_xy: the dedup stage's input)
  double* host_pose; int* host_inliers; int* host_nin; int* host_best; int* host_bad;  // pinned mirrors; host_best < 0: no model, host_bad != 0: a store entry under a foreign id
  unsigned* arrive; unsigned arrive_target;                                         // device arrival counter of the lane (monotone)
  int* word; int seq;                                                               // pinned completion word
  // round 5: the keyframe chain without a host turn — the launch's last workgroup also leaves the reprojection matrix of the refined
  // pose (or of the previous one: prev_rvec / prev_tvec, when no model was found) and the inlier count for the stereo +
  // triangulation launch queued right behind it (host/chain_math.h); chain == null: not asked for
  SvoChainRec* chain; float prev_rvec[3], prev_tvec[3]; float cam_f, cam_cx, cam_cy, cam_b;
};
struct SvoPnpLanes { SvoPnpLane lane[SVO_MAX_LANES]; };
int svo_kg_pnp(svo_ctx* ctx, hipStream_t st, const SvoPnpLanes& lanes, int n_lanes);  // grid from the lanes' `launched`
int svo_kg_pnp_workgroups(int launched);  // workgroups of a lane that computes `launched` hypotheses (what its arrive_target counts)
int svo_kg_pnp_first(int iterations);     // hypotheses of a lane's FIRST launch (the rest only if the bookkeeping asks for them)
int svo_pnp_update_num_iters(double p, double ep, int model_points, int max_iters);  // OpenCV's RANSACUpdateNumIters (host/pnp_iters.h)
int svo_pnp_model_points();

// a6 + a7 (sparse) + a8: every workgroup first applies the dedup predicate to its detected corner (a duplicate takes disparity 0
// and is dropped by the triangulation's own validity test, in the same index order as the separate dedup launch gave), then the
// disparity at the corner; triangulation / compaction by the lane's last workgroup
struct SvoStereoTriLane {
  const uint8_t* left; const uint8_t* right; const float* xy; const int* n_dev; int n_max; float* disp; SvoMat4 M;
  const float* trk; int n_trk; float min_d;  // a6 folded in: the tracked inliers a detected corner must keep min_d away from (null: no dedup, frame 0)
  float* kept_xy; float* xyz; int* n_kept; SvoPublish pub;
  const SvoChainRec* chain;  // non-null: M, the number of tracked inliers and "nothing to do" (best == -2) come from the PnP launch in front of this one
};
struct SvoStereoTriLanes { int w, h, stride, ndisp, block; SvoXcdMap map; SvoStereoTriLane lane[SVO_MAX_LANES]; };
// map.per_chunk == 0: grid_x = the largest lane's n_max; every lane's pub.target counts grid_x workgroups (pub.arrive must be set); with the map (lanes' n_dev must be null) max(n_max, 1)
int svo_kg_stereo_triangulate(svo_ctx* ctx, hipStream_t st, const SvoStereoTriLanes& lanes, int n_lanes, int grid_x);

// a12 in steps (csrc/ba.hip): assemble on any thread, launch the solves of several adjusters as ONE kernel, join later
int svo_ba_solve_prepare(svo_ba* ba);                          // 1: nothing to solve, 0: a problem is loaded, < 0: svo_status
int svo_ba_solve_launch(svo_ba** bas, int n, void* stream, unsigned long long* launched_mask);  // number launched; bit i of the mask: bas[i] was (an ineligible or not admitted adjuster is skipped)
int svo_ba_solve_poll(svo_ba* ba);                             // 1: finish will not block
int svo_ba_solve_finish(svo_ba* ba, svo_ba_summary* summary);  // join (or solve host-driven) + write back into the graph
void svo_ba_work(svo_ba* ba, double* out4, int reset);
// device-resident landmark store of the adjuster's stream (capacity = mask + 1 entries, a power of two): every solve writes the
// landmarks it optimised into it — ba_lm_kernel with its delivery, a host-driven solve through one scatter launch at its finish
void svo_ba_note_group_lanes(int delta);           // lanes of live pipeline groups (decides the admission overflow, csrc/ba.hip)
int svo_ba_use_stream(svo_ba* ba, void* stream);   // the adjuster's own stream replaced by one of the group's lines
int svo_ba_attach_store(svo_ba* ba, float4* store, unsigned mask);          // algorithmic [flops, bytes, solves, LM iterations] of the finished solves

#endif
