// oracle: ImageProcessor::process (src/image_processor.cpp:18-163), FeatureTracker
// (src/feature_tracker.cpp:3-72), BundleAdjuster graph edits and solve (src/bundle_adjuster.cpp:41-163)
// and the driver rule "one process() then one bundle_adjust() per frame"
// (src/vo_node.cpp:141-148) — whole-pipeline CPU restatement built on the ora_* stages.
// TEST INFRASTRUCTURE ONLY.  Control flow, constants, id assignment and pose conventions follow
// first-party reference source; the stages it calls carry the pin status stated in svo_oracle.h.
// Stated differences from the reference as it would run under OpenCV 3.x (the HIP path makes the same choices):
//  (1) rvec / tvec are stored as float between frames, the type they are created with (:54-55).  OpenCV's
//      solvePnPRansac re-creates its outputs as CV_64F, after which `tmp.copyTo(hmat(cv::Rect(..)))`
//      (:131-134) has mismatched types and re-allocates the temporary ROI header instead of writing into the CV_32F
//      hmat, i.e. camera_pose would stay the identity from the second keyframe on (a latent bug).  The restatement
//      implements the evident intent: hmat = [R^T | -R^T t] from the float rvec / tvec.
//  (2) zero tracked features: av_parallax = 0 instead of the reference's 0/0 = NaN (:59,63); the keyframe gate then
//      fires through percent_lost = 1 either way.
//  (3) SURVEY C-1, C-3, C-4, C-5, C-7, C-9, C-12, C-13 as listed there (replicate / guard decisions).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <memory>
#include <unordered_map>
#include <vector>

#include "ora_constants.h"
#include "ora_trig.h"
#include "svo_oracle.h"

namespace {
struct Obs { float u, v; size_t id; };
struct PoseVar { double pose[7]; std::vector<Obs> obs; };
struct Feature { double pos[3]; size_t refcount; };

struct Keyframe {  // src/bundle_adjuster.hpp:22-46
  float position[3];
  float orientation[4];  // w x y z
  std::vector<float> tracked_2d;
  std::vector<size_t> tracked_ids;
  std::vector<float> new_2d, new_3d;
  std::vector<size_t> new_ids;
};

// cv::Rodrigues on a CV_32F rvec with declared arithmetic (ora_trig.h; round 5: the HIP path runs this conversion on the device)
void rodrigues_f(const float* rv, float* R9) { ora_trig::rodrigues_f(rv, R9); }

// Eigen::Quaternionf(Matrix3f) (src/image_processor.cpp:92), float arithmetic.
void quat_from_R(const float* m, float* q /*wxyz*/) {
  float t = m[0] + m[4] + m[8];
  if (t > 0.f) {
    t = std::sqrt(t + 1.0f);
    q[0] = 0.5f * t;
    t = 0.5f / t;
    q[1] = (m[7] - m[5]) * t; q[2] = (m[2] - m[6]) * t; q[3] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
    q[1 + i] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[1 + j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[1 + k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
}
}  // namespace

struct ora_pipeline {
  ora_pipeline_params prm;
  // BundleAdjuster state
  std::vector<Feature> features;
  std::deque<std::shared_ptr<PoseVar>> window;
  std::shared_ptr<Keyframe> last_keyframe;
  bool new_frame_added = false;
  int last_ba_iterations = 0;
  // FeatureTracker state
  std::vector<uint8_t> last_image;
  std::unordered_map<size_t, std::pair<float, float>> initial_features;
  std::vector<float> feature_set;
  std::vector<size_t> feature_ids;
  // ImageProcessor state
  float rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};

  void tracker_init(const uint8_t* img, const std::vector<float>& feats, const std::vector<size_t>& ids) {
    feature_ids = ids;  // src/feature_tracker.cpp:5-6
    feature_set = feats;
    initial_features.clear();
    for (size_t i = 0; i < ids.size(); ++i)
      initial_features.insert({ids[i], {feats[2 * i], feats[2 * i + 1]}});  // :10-12 (insert keeps the first)
    last_image.assign(img, img + (size_t)prm.width * prm.height);           // :14
  }

  void add_keyframe(std::shared_ptr<Keyframe> kf) {  // src/bundle_adjuster.cpp:60-135
    auto pv = std::make_shared<PoseVar>();
    for (int i = 0; i < 4; ++i) pv->pose[i] = kf->orientation[i];
    for (int i = 0; i < 3; ++i) pv->pose[4 + i] = kf->position[i];
    const size_t num_tracked = kf->tracked_ids.size();
    for (size_t i = 0; i < num_tracked; ++i) {
      const size_t id = kf->tracked_ids[i];
      features[id].refcount++;
      pv->obs.push_back({kf->tracked_2d[2 * i], kf->tracked_2d[2 * i + 1], id});
    }
    const size_t maxf = (size_t)prm.max_features;
    const size_t max_new = num_tracked > maxf ? 0 : maxf - num_tracked;  // C-5 guard
    if (kf->new_2d.size() / 2 > max_new) {  // :85-90 (C-4: new_ids holds real ids only)
      kf->new_2d.resize(2 * max_new);
      kf->new_3d.resize(3 * max_new);
    }
    const size_t num_new = kf->new_2d.size() / 2;
    for (size_t i = 0; i < num_new; ++i) {
      features.push_back(Feature());  // avail_ids is never refilled (C-3): ids are sequential
      const size_t id = features.size() - 1;
      for (int a = 0; a < 3; ++a) features[id].pos[a] = kf->new_3d[3 * i + a];
      features[id].refcount = 2;  // :113,116
      kf->new_ids.push_back(id);
      pv->obs.push_back({kf->new_2d[2 * i], kf->new_2d[2 * i + 1], id});
    }
    window.push_back(pv);
    if (window.size() > (size_t)prm.window_size) {  // :126-128, remove_oldest_pose :41-58
      for (const Obs& o : window.front()->obs) features[o.id].refcount--;
      window.pop_front();
    }
    last_keyframe = kf;
    new_frame_added = true;
  }

  void bundle_adjust() {  // src/bundle_adjuster.cpp:137-157
    last_ba_iterations = 0;
    if (!new_frame_added) return;
    const int K = (int)window.size();
    // gather landmarks referenced by the window, observations sorted by (landmark id, window slot)
    std::vector<std::pair<size_t, int>> order;  // (id, flat index)
    struct Flat { int k; float u, v; size_t id; };
    std::vector<Flat> flat;
    for (int k = 0; k < K; ++k)
      for (const Obs& o : window[k]->obs) flat.push_back({k, o.u, o.v, o.id});
    std::vector<int> perm(flat.size());
    for (size_t i = 0; i < perm.size(); ++i) perm[i] = (int)i;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return flat[a].id < flat[b].id; });
    std::vector<double> poses(7 * (size_t)K), points;
    std::vector<int32_t> op, oj;
    std::vector<double> uv;
    std::vector<size_t> lm_ids;
    for (int k = 0; k < K; ++k) std::memcpy(&poses[7 * k], window[k]->pose, 7 * sizeof(double));
    for (int idx : perm) {
      const Flat& f = flat[idx];
      if (lm_ids.empty() || lm_ids.back() != f.id) {
        lm_ids.push_back(f.id);
        for (int a = 0; a < 3; ++a) points.push_back(features[f.id].pos[a]);
      }
      op.push_back(f.k);
      oj.push_back((int32_t)lm_ids.size() - 1);
      uv.push_back(f.u);
      uv.push_back(f.v);
    }
    double summary[5];
    ora_ba_solve(K, poses.data(), (int)lm_ids.size(), points.data(), (int)op.size(), op.data(),
                 oj.data(), uv.data(), prm.focal, prm.cx, prm.cy, prm.ba_max_iterations, 1e-6, 1e-10,
                 1e-8, 1e4, prm.num_threads, nullptr, nullptr, summary);
    last_ba_iterations = (int)summary[0];
    for (int k = 0; k < K; ++k) std::memcpy(window[k]->pose, &poses[7 * k], 7 * sizeof(double));
    for (size_t l = 0; l < lm_ids.size(); ++l)
      for (int a = 0; a < 3; ++a) features[lm_ids[l]].pos[a] = points[3 * l + a];
    const double* pz = window.back()->pose;  // :146-153
    for (int i = 0; i < 4; ++i) last_keyframe->orientation[i] = (float)pz[i];
    for (int i = 0; i < 3; ++i) last_keyframe->position[i] = (float)pz[4 + i];
    new_frame_added = false;
  }

  void triangulate(const std::vector<float>& feats, const uint8_t* L, const uint8_t* R,
                   const float* pose16, std::vector<float>& out2d, std::vector<float>& out3d) {
    const int n = (int)feats.size() / 2;
    std::vector<float> disp(n), k2(2 * (size_t)n), k3(3 * (size_t)n);
    ora_stereo_disparity_at(L, R, prm.width, prm.height, prm.width, ora_k::kStereoNumDisparities, ora_k::kStereoBlockSize, feats.data(), n, disp.data());
    const int m = ora_triangulate(feats.data(), disp.data(), n, pose16, (float)prm.focal, (float)prm.cx,
                                  (float)prm.cy, (float)prm.baseline, k2.data(), k3.data(), nullptr);
    out2d.assign(k2.begin(), k2.begin() + 2 * m);
    out3d.assign(k3.begin(), k3.begin() + 3 * m);
  }

  void process(const uint8_t* L, const uint8_t* R, ora_frame_result* res) {
    std::memset(res, 0, sizeof(*res));
    const int W = prm.width, H = prm.height;
    std::vector<float> det(2 * (size_t)prm.max_corners);
    const int nd = ora_corner_detect(L, W, H, W, prm.max_corners, prm.quality,
                                     prm.min_feature_distance, det.data(), nullptr);  // :22
    det.resize(2 * (size_t)nd);
    res->n_detected = nd;
    if (nd < ora_k::kMinDetected) return;  // :23-25
    if (!last_keyframe) {  // :30-58
      const float eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      auto kf = std::make_shared<Keyframe>();
      kf->position[0] = kf->position[1] = kf->position[2] = 0.f;
      kf->orientation[0] = 1.f; kf->orientation[1] = kf->orientation[2] = kf->orientation[3] = 0.f;
      triangulate(det, L, R, eye, kf->new_2d, kf->new_3d);
      add_keyframe(kf);
      tracker_init(L, kf->new_2d, kf->new_ids);
      for (int i = 0; i < 3; ++i) rvec[i] = tvec[i] = 0.f;
      res->is_keyframe = 1;
      res->n_new = (int)kf->new_ids.size();
      return;
    }
    // FeatureTracker::track_features :62
    const int n = (int)feature_ids.size();
    std::vector<float> init(2 * (size_t)n), kept(2 * (size_t)n);
    std::vector<int> kidx(n);
    for (int i = 0; i < n; ++i) {
      const auto& p = initial_features.at(feature_ids[i]);
      init[2 * i] = p.first; init[2 * i + 1] = p.second;
    }
    float av = 0.f;
    const int m = ora_track_features(last_image.data(), L, W, H, W, feature_set.data(), init.data(), n,
                                     kept.data(), kidx.data(), &av);
    std::vector<size_t> new_ids_list(m);
    for (int i = 0; i < m; ++i) new_ids_list[i] = feature_ids[kidx[i]];  // C-1
    feature_set.assign(kept.begin(), kept.begin() + 2 * m);
    feature_ids = new_ids_list;
    const float percent_lost = (float)(1.0 - (double)((float)m / (float)initial_features.size()));
    last_image.assign(L, L + (size_t)W * H);
    res->n_tracked = m;
    res->av_parallax = av;
    res->percent_lost = percent_lost;
    if (av <= prm.parallax_thresh && (double)percent_lost < ora_k::kKeyframePercentLost) return;  // :63-65
    // PnP :67-92
    std::vector<float> wp(3 * (size_t)m);
    for (int i = 0; i < m; ++i)
      for (int a = 0; a < 3; ++a) wp[3 * i + a] = (float)features[feature_ids[i]].pos[a];  // get_world_points
    std::vector<int> inl(m > 0 ? m : 1);
    double rv[3] = {rvec[0], rvec[1], rvec[2]}, tv[3] = {tvec[0], tvec[1], tvec[2]};
    const int ni = ora_pnp_ransac(wp.data(), feature_set.data(), m, (float)prm.focal, (float)prm.cx,
                                  (float)prm.cy, rv, tv, ora_k::kPnpIterations, ora_k::kPnpReprojError, ora_k::kPnpConfidence, inl.data());
    for (int i = 0; i < 3; ++i) { rvec[i] = (float)rv[i]; tvec[i] = (float)tv[i]; }  // CV_32F in/out
    res->n_inliers = ni;
    float Rm[9], q[4];
    rodrigues_f(rvec, Rm);
    quat_from_R(Rm, q);
    auto kf = std::make_shared<Keyframe>();
    for (int i = 0; i < 3; ++i) kf->position[i] = tvec[i];
    for (int i = 0; i < 4; ++i) kf->orientation[i] = q[i];
    for (int i = 0; i < ni; ++i) {  // :104-108
      kf->tracked_ids.push_back(feature_ids[inl[i]]);
      kf->tracked_2d.push_back(feature_set[2 * inl[i]]);
      kf->tracked_2d.push_back(feature_set[2 * inl[i] + 1]);
    }
    // dedup :113-128
    std::vector<float> fresh(2 * (size_t)nd);
    const int nf = ora_dedup(det.data(), nd, kf->tracked_2d.data(), ni, prm.min_feature_distance, fresh.data());
    fresh.resize(2 * (size_t)nf);
    // hmat = [R^T | -R^T t] :130-134 (float Mats, gemm accumulates in double)
    float hmat[16] = {0};
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) hmat[4 * i + j] = Rm[3 * j + i];
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += (double)(-Rm[3 * k + i]) * (double)tvec[k];
      hmat[4 * i + 3] = (float)s;
    }
    hmat[15] = 1.f;
    triangulate(fresh, L, R, hmat, kf->new_2d, kf->new_3d);  // :137-142
    add_keyframe(kf);                                        // :144
    std::vector<float> f2d(kf->tracked_2d);                  // :148-162
    f2d.insert(f2d.end(), kf->new_2d.begin(), kf->new_2d.end());
    std::vector<size_t> ids(kf->tracked_ids);
    ids.insert(ids.end(), kf->new_ids.begin(), kf->new_ids.end());
    tracker_init(L, f2d, ids);
    res->is_keyframe = 1;
    res->n_new = (int)kf->new_ids.size();
  }
};

extern "C" ora_pipeline* ora_pipeline_create(const ora_pipeline_params* p) {
  auto* o = new ora_pipeline();
  o->prm = *p;
  return o;
}
extern "C" void ora_pipeline_destroy(ora_pipeline* p) { delete p; }

extern "C" void ora_pipeline_process(ora_pipeline* p, const uint8_t* left, const uint8_t* right,
                                     ora_frame_result* res) {
  p->process(left, right, res);
  if (p->last_keyframe) {  // src/vo_node.cpp:146-148
    p->bundle_adjust();
    res->ba_iterations = p->last_ba_iterations;
    for (int i = 0; i < 4; ++i) res->pose7[i] = p->last_keyframe->orientation[i];
    for (int i = 0; i < 3; ++i) res->pose7[4 + i] = p->last_keyframe->position[i];
  }
}

extern "C" int ora_pipeline_get_tracked(ora_pipeline* p, int64_t* ids, float* xy, int capacity) {
  const int n = (int)p->feature_ids.size();
  for (int i = 0; i < n && i < capacity; ++i) {
    ids[i] = (int64_t)p->feature_ids[i];
    xy[2 * i] = p->feature_set[2 * i];
    xy[2 * i + 1] = p->feature_set[2 * i + 1];
  }
  return n;
}

// The literals above as one table (order = tests/test_constants.py ORACLE_NAMES): lets the tests compare this
// restatement's constants with the values extracted from the reference text.
extern "C" int ora_reference_constants(double* out, int capacity) {
  const double v[] = {(double)ora_k::kMinDetected, ora_k::kKeyframePercentLost, (double)ora_k::kPnpIterations, (double)ora_k::kPnpReprojError,
                      ora_k::kPnpConfidence, (double)ora_k::kStereoNumDisparities, (double)ora_k::kStereoBlockSize,
                      (double)ora_k::kStereoDisparityScale, (double)ora_k::kTriangulateMinDisparity, (double)ora_k::kLkWin,
                      (double)ora_k::kLkMaxLevel, (double)ora_k::kLkMaxIterations, ora_k::kLkEpsilon, (double)ora_k::kLkMinEigThreshold,
                      ora_k::kFbMaxDistance, (double)ora_k::kMaxParallax};
  const int n = (int)(sizeof(v) / sizeof(v[0]));
  for (int i = 0; i < n && i < capacity; ++i) out[i] = v[i];
  return n;
}
