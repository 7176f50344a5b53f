// Test translation unit: makes exactly the calls the reference's src/vo_node.cpp makes on the three class surfaces
// (constructors :104-119, StereoPair queue :66-73,141-144, get_last_keyframe / bundle_adjust :146-148, pose inversion
// :149-150, get_drawing :188), against the adapters in adapters/ and — in this image — the syntax-only stub headers in
// tests/stubs/.  ROS is replaced by the deterministic synthetic stereo stream of libsvo_hip.so (svo_synth_*).
// It also exercises the rest of the public surfaces directly (FeatureTracker::init / track_features /
// get_tracked_features / draw_track, BundleAdjuster::add_keyframe / get_world_points, ReprojectionFactor::Evaluate).
// One line per frame on stdout; tests/test_adapters.py compares them with the C-ABI pipeline on the same frames.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <opencv2/opencv.hpp>
#include <Eigen/Dense>
#include "bundle_adjuster.hpp"
#include "feature_tracker.hpp"
#include "image_processor.hpp"
#include "reprojection_factor.hpp"
#include <string>
#include <queue>

using namespace Eigen;
using namespace std;

static const float parallax_thresh = 20;         // src/vo_node.cpp:33-36
static const float min_feature_distance = 30;
static const size_t sliding_window_size = 5;

static unsigned fbits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

int main(int argc, char **argv) {
  const int n_frames = argc > 1 ? atoi(argv[1]) : 10;
  const int width = argc > 2 ? atoi(argv[2]) : 496, height = argc > 3 ? atoi(argv[3]) : 160;
  svo_synth_params sp;
  svo_synth_default_params(&sp, width, height);
  if (argc > 4) sp.focal = atof(argv[4]);
  if (argc > 5) sp.seed = strtoull(argv[5], nullptr, 0);

  float focal_length = (float)sp.focal, cx = (float)sp.cx, cy = (float)sp.cy, baseline = (float)sp.baseline;

  // Populate camera matrix with intrinsics (src/vo_node.cpp:104-108)
  cv::Mat camera_matrix = cv::Mat::eye(3, 3, CV_32F);
  camera_matrix.at<float>(0, 0) = focal_length;
  camera_matrix.at<float>(0, 2) = cx;
  camera_matrix.at<float>(1, 1) = focal_length;
  camera_matrix.at<float>(1, 2) = cy;

  CameraInfo info = {focal_length, cx, cy, 0, 0, 0, 0, baseline};  // :110

  shared_ptr<BundleAdjuster> bundle_adjuster = make_shared<BundleAdjuster>(sliding_window_size, info);  // :112
  shared_ptr<FeatureTracker> feature_tracker = make_shared<FeatureTracker>();                          // :113
  ImageProcessor image_processor(camera_matrix, feature_tracker, bundle_adjuster, baseline, min_feature_distance,
                                 parallax_thresh);                                                    // :114-119
  if (!svo_adapter::context()) {
    // no MI355X: the adapters must degrade to the reference's "void + return" convention, not crash
    cv::Mat blank(height, width, CV_8UC1);
    image_processor.process(StereoPair(blank, blank, 0.0));
    bundle_adjuster->bundle_adjust();
    const bool none = bundle_adjuster->get_last_keyframe() == nullptr && feature_tracker->get_drawing().empty();
    printf("no-device ok=%d\n", none ? 1 : 0);
    return none ? 3 : 1;
  }

  shared_ptr<queue<StereoPair>> image_queue = make_shared<queue<StereoPair>>();  // :123
  cv::Mat tracking_image;

  for (int f = 0; f < n_frames; ++f) {
    // handle_images::operator() (:60-73): push one synchronised mono8 pair
    cv::Mat left(height, width, CV_8UC1), right(height, width, CV_8UC1);
    if (svo_synth_render(&sp, f, left.data, right.data) != SVO_OK) return 2;
    double time = 0.1 * f;
    if (!image_queue->empty() && time - image_queue->back().t < 0.05) continue;
    image_queue->push(StereoPair(left, right, time));

    while (!image_queue->empty()) {  // :141-144
      image_processor.process(image_queue->front());
      image_queue->pop();
    }
    const svo::ImageProcessor::Stats *st = image_processor.last_stats();
    printf("frame %d det %d trk %d inl %d new %d kf %d par %08x lost %08x", f, st->n_detected, st->n_tracked, st->n_inliers,
           st->n_new, st->is_keyframe, fbits(st->av_parallax), fbits(st->percent_lost));

    if (bundle_adjuster->get_last_keyframe() != nullptr) {  // :146
      bundle_adjuster->bundle_adjust();
      shared_ptr<Keyframe> keyframe = bundle_adjuster->get_last_keyframe();
      Quaternionf orientation = keyframe->orientation.conjugate();  // :149
      Vector3f position = orientation * (-keyframe->position);      // :150
      printf(" pose %08x %08x %08x %08x %08x %08x %08x", fbits(keyframe->orientation.w()), fbits(keyframe->orientation.x()),
             fbits(keyframe->orientation.y()), fbits(keyframe->orientation.z()), fbits(keyframe->position(0)),
             fbits(keyframe->position(1)), fbits(keyframe->position(2)));
      printf(" cam %.6f %.6f %.6f %.6f %.6f %.6f %.6f", orientation.w(), orientation.x(), orientation.y(), orientation.z(),
             position(0), position(1), position(2));
      printf(" kfsizes %zu %zu %zu %zu img %dx%d", keyframe->tracked_ids.size(), keyframe->new_ids.size(),
             keyframe->new_features_2d.size(), keyframe->new_features_3d.size(), keyframe->image.cols, keyframe->image.rows);
      tracking_image = feature_tracker->get_drawing();  // :188
    }
    vector<cv::Point2f> feats;
    vector<size_t> ids;
    feature_tracker->get_tracked_features(feats, ids);
    printf(" tracked %zu", ids.size());
    for (size_t i = 0; i < ids.size(); ++i) printf(" %zu:%08x:%08x", ids[i], fbits(feats[i].x), fbits(feats[i].y));
    printf("\n");
  }
  // the drawing: an rgb8 image of the keyframe size with green arrow pixels on it once something was tracked
  size_t green = 0;
  if (!tracking_image.empty() && tracking_image.type() == CV_8UC3)
    for (int y = 0; y < tracking_image.rows; ++y)
      for (int x = 0; x < tracking_image.cols; ++x) {
        const uchar *p = tracking_image.ptr<uchar>(y) + 3 * x;
        green += p[0] == 0 && p[1] == 255 && p[2] == 0;
      }
  printf("drawing %dx%d type %d green %zu\n", tracking_image.cols, tracking_image.rows, tracking_image.type(), green);

  // ---- the rest of the public surfaces, used directly -------------------------------------------------------------
  {
    FeatureTracker tracker;  // src/feature_tracker.hpp:20-54
    cv::Mat l0(height, width, CV_8UC1), r0(height, width, CV_8UC1), l1(height, width, CV_8UC1), r1(height, width, CV_8UC1);
    svo_synth_render(&sp, 0, l0.data, r0.data);
    svo_synth_render(&sp, 1, l1.data, r1.data);
    // corners of frame 0 through the C-ABI (the adapter ImageProcessor detects inside the library)
    vector<float> xy(2 * 300);
    int n = 0;
    svo_corner_detect(svo_adapter::context(), l0.data, width, height, (int)(size_t)l0.step, 300, 0.1, min_feature_distance, xy.data(), &n);
    vector<cv::Point2f> pts;
    vector<size_t> ids;
    for (int i = 0; i < n; ++i) { pts.push_back(cv::Point2f(xy[2 * i], xy[2 * i + 1])); ids.push_back(100 + i); }
    tracker.init(l0, pts, ids);
    float av = 0, lost = 0;
    tracker.track_features(av, lost, l1, true);
    vector<cv::Point2f> out;
    vector<size_t> oid;
    tracker.get_tracked_features(out, oid);
    tracker.draw_track();
    cv::Mat d = tracker.get_drawing();
    printf("direct-tracker init %d kept %zu par %08x lost %08x drawing %dx%d", n, oid.size(), fbits(av), fbits(lost), d.cols, d.rows);
    for (size_t i = 0; i < oid.size(); ++i) printf(" %zu:%08x:%08x", oid[i], fbits(out[i].x), fbits(out[i].y));
    printf("\n");

    // duplicate ids (src/feature_tracker.cpp:10-12: map::insert keeps the FIRST feature of an id; :47,64 then read that entry
    // for the parallax of every feature carrying the id and count the map's entries): features 2m and 2m+1 share id 500+m
    {
      FeatureTracker dup;
      vector<size_t> ids2;
      for (int i = 0; i < n; ++i) ids2.push_back(500 + i / 2);
      dup.init(l0, pts, ids2);
      float av2 = 0, lost2 = 0;
      dup.track_features(av2, lost2, l1, true);
      vector<cv::Point2f> out2;
      vector<size_t> oid2;
      dup.get_tracked_features(out2, oid2);
      printf("dup-tracker init %d kept %zu par %08x lost %08x", n, oid2.size(), fbits(av2), fbits(lost2));
      for (size_t i = 0; i < oid2.size(); ++i) printf(" %zu:%08x:%08x", oid2[i], fbits(out2[i].x), fbits(out2[i].y));
      printf("\n");
    }

    BundleAdjuster adjuster(3, info);  // src/bundle_adjuster.hpp:86-126
    vector<cv::Point2f> n2;
    vector<cv::Point3f> n3;
    for (int i = 0; i < 450; ++i) {  // more than max_features: add_keyframe must truncate to 400 (:85-90)
      n2.push_back(cv::Point2f(10.f + i, 20.f));
      n3.push_back(cv::Point3f(0.01f * i, 0.f, 5.f));
    }
    shared_ptr<Keyframe> kf = make_shared<Keyframe>(Vector3f::Zero(), Quaternionf::Identity(), l0, vector<cv::Point2f>(),
                                                    vector<size_t>(), n2, n3);
    adjuster.add_keyframe(kf);
    vector<cv::Point3f> wp;
    adjuster.get_world_points(wp, kf->new_ids);
    printf("direct-adjuster new2d %zu new3d %zu ids %zu first %zu last %zu same_kf %d wp %zu wp7 %.4f\n", kf->new_features_2d.size(),
           kf->new_features_3d.size(), kf->new_ids.size(), kf->new_ids.empty() ? 0 : kf->new_ids.front(),
           kf->new_ids.empty() ? 0 : kf->new_ids.back(), adjuster.get_last_keyframe() == kf ? 1 : 0, wp.size(), wp.size() > 7 ? wp[7].x : -1.f);

    ReprojectionFactor factor(320.5, 110.25, info);  // src/reprojection_factor.hpp:9-13
    double pose[7] = {0.999, 0.01, -0.02, 0.03, 0.1, -0.2, 0.3}, point[3] = {0.5, -0.25, 8.0};
    double const *params[2] = {pose, point};
    double r[2], jq[14], jp[6];
    double *jac[2] = {jq, jp};
    const bool ok1 = factor.Evaluate(params, r, jac);
    double r2[2];
    double *jac_none[2] = {nullptr, nullptr};
    const bool ok2 = factor.Evaluate(params, r2, jac_none) && factor.Evaluate(params, r2, nullptr);  // Ceres' null conventions
    printf("direct-factor ok %d %d r %.17g %.17g same %d j5 %.1f j11 %.1f jq0 %.17g jp0 %.17g\n", ok1, ok2, r[0], r[1],
           r[0] == r2[0] && r[1] == r2[1], jq[5], jq[11], jq[0], jp[0]);
  }
  svo_adapter::shutdown();
  return 0;
}
