// KITTI-odometry ingestion, a non-ROS driver reproducing vo_node's loop, and ATE.
//   dataset layout  : src/kitti_node.cpp:37-68  (<data_path><SS>/image_0/%06d.png, image_1/..., poses file
//                     <data_path>data_odometry_poses/dataset/poses/SS.txt, 12 doubles per row = 3x4 [R|t])
//   driver semantics: src/vo_node.cpp:141-150   (process every frame, then bundle_adjust; published pose =
//                     camera in world: q_wc = conj(q), p_wc = q_wc * (-t)); one pose per processed frame (SURVEY C-12)
// The decoders, the poses reader and the ATE live in kitti_io.cpp (no GPU dependency).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <chrono>
#include <string>
#include <vector>

#include "common.h"

// The non-ROS driver: what kitti_node + vo_node do together, offline and deterministic.
extern "C" int svo_kitti_run(svo_ctx* ctx, const svo_pipeline_params* params, const char* data_path, int sequence,
                             int max_frames, double* traj_rt12, svo_run_stats* stats) {
  if (!ctx || !params || !data_path || !stats) return SVO_ERR_INVALID;
  memset(stats, 0, sizeof(*stats));
  char seq[8];
  snprintf(seq, sizeof(seq), "%02d", sequence);  // src/kitti_node.cpp:24-27
  const std::string root(data_path);
  std::vector<double> gt(12 * (size_t)(max_frames > 0 ? max_frames : 1));
  int n_gt = 0;
  const std::string pfile = root + "data_odometry_poses/dataset/poses/" + seq + ".txt";  // :37
  const bool have_gt = svo_kitti_read_poses(pfile.c_str(), gt.data(), max_frames, &n_gt) == SVO_OK && n_gt > 0;
  svo_pipeline* pipe = nullptr;
  int rc = svo_pipeline_create(ctx, &pipe, params);
  if (rc) return rc;
  const int W = params->width, H = params->height, B = ctx->lim.max_batch;
  const size_t px = (size_t)W * H;
  std::vector<uint8_t> L(px * B), R(px * B);
  std::vector<svo_frame_result> res(B);
  std::vector<double> est_xyz, gt_xyz;
  const auto t0 = std::chrono::steady_clock::now();
  int frame = 0;
  bool more = true;
  while (more && frame < max_frames) {
    int nb = 0;
    for (; nb < B && frame + nb < max_frames; ++nb) {
      char name[32];
      snprintf(name, sizeof(name), "%06d.png", frame + nb);  // :56-68
      int w = 0, h = 0, w2 = 0, h2 = 0;
      if (svo_image_read_gray((root + seq + "/image_0/" + name).c_str(), &L[px * nb], px, &w, &h) != SVO_OK ||
          svo_image_read_gray((root + seq + "/image_1/" + name).c_str(), &R[px * nb], px, &w2, &h2) != SVO_OK) { more = false; break; }
      if (w != W || h != H || w2 != W || h2 != H) { svo_pipeline_destroy(pipe); ctx->err = "kitti_run: image size differs from the pipeline parameters"; return SVO_ERR_INVALID; }
    }
    if (nb == 0) break;
    rc = svo_pipeline_process_batch(pipe, L.data(), R.data(), nb, res.data());  // process + bundle_adjust per frame (src/vo_node.cpp:141-148)
    if (rc) { svo_pipeline_destroy(pipe); return rc; }
    for (int i = 0; i < nb; ++i) {
      const svo_frame_result& r = res[i];
      stats->keyframes += r.is_keyframe;
      // camera in world (src/vo_node.cpp:149-150): q_wc = conj(q), p_wc = q_wc * (-t), float arithmetic as published
      const float w = (float)r.pose7[0], x = -(float)r.pose7[1], y = -(float)r.pose7[2], z = -(float)r.pose7[3];
      const float t[3] = {-(float)r.pose7[4], -(float)r.pose7[5], -(float)r.pose7[6]};
      const float Rm[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                           2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                           2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
      double p[3];
      for (int a = 0; a < 3; ++a) p[a] = Rm[3 * a] * t[0] + Rm[3 * a + 1] * t[1] + Rm[3 * a + 2] * t[2];
      if (r.pose7[0] == 0 && r.pose7[1] == 0 && r.pose7[2] == 0 && r.pose7[3] == 0) { p[0] = p[1] = p[2] = 0; }  // before the first keyframe
      const int f = frame + i;
      if (traj_rt12) {
        double* o = traj_rt12 + 12 * (size_t)f;
        for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) o[4 * a + b] = Rm[3 * a + b]; o[4 * a + 3] = p[a]; }
      }
      if (have_gt && f < n_gt && r.is_keyframe) {  // between keyframes the published pose is the last keyframe's (SURVEY C-12)
        for (int a = 0; a < 3; ++a) { est_xyz.push_back(p[a]); gt_xyz.push_back(gt[12 * (size_t)f + 4 * a + 3]); }
      }
    }
    frame += nb;
  }
  stats->frames = frame;
  stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  stats->ate_rmse = -1.0;
  if (est_xyz.size() >= 9) svo_ate_rmse(est_xyz.data(), gt_xyz.data(), (int)(est_xyz.size() / 3), 0, &stats->ate_rmse);
  svo_pipeline_destroy(pipe);
  return SVO_OK;
}
