// FeatureTracker over svo::FeatureTracker.  Replaces src/feature_tracker.cpp of the reference.
#include "feature_tracker.hpp"

FeatureTracker::FeatureTracker() {
  svo_ctx *ctx = svo_adapter::context();
  if (!ctx) return;
  int max_w = 1920, max_h = 1200;
  if (const char *v = getenv("SVO_ADAPTER_MAX_WIDTH")) if (*v) max_w = atoi(v);
  if (const char *v = getenv("SVO_ADAPTER_MAX_HEIGHT")) if (*v) max_h = atoi(v);
  impl_ = make_shared<svo::FeatureTracker>(ctx, (int)max_features, max_w, max_h);
  if (!impl_->ok()) { impl_.reset(); return; }
  impl_->enable_drawing(true);  // ImageProcessor::process calls draw_track() on every keyframe (src/image_processor.cpp:146)
}

void FeatureTracker::init(const cv::Mat &image, const vector<cv::Point2f> &features, const vector<size_t> &ids) {
  const uint8_t *data; int w, h, stride;
  if (!impl_ || !svo_adapter::mono8(image, &data, &w, &h, &stride)) return;
  vector<svo::Point2f> f(features.size());
  for (size_t i = 0; i < features.size(); ++i) f[i] = svo::Point2f{features[i].x, features[i].y};
  if (impl_->init_host(data, w, h, stride, f, ids) != SVO_OK) return;
  initial_image = image.clone();  // :14-15
}

void FeatureTracker::track_features(float &av_parallax, float &percent_lost, const cv::Mat &image, bool flow_back) {
  const uint8_t *data; int w, h, stride;
  if (!impl_ || !svo_adapter::mono8(image, &data, &w, &h, &stride)) return;
  (void)impl_->track_features_host(av_parallax, percent_lost, data, w, h, stride, flow_back);
}

void FeatureTracker::get_tracked_features(vector<cv::Point2f> &features, vector<size_t> &ids) {
  features.clear(); ids.clear();
  if (!impl_) return;
  vector<svo::Point2f> f;
  impl_->get_tracked_features(f, ids);
  features.resize(f.size());
  for (size_t i = 0; i < f.size(); ++i) features[i] = cv::Point2f(f[i].x, f[i].y);
}

void FeatureTracker::rasterise() {
  drawn_serial = impl_->drawing_serial();
  const uint8_t *data; int w, h, stride;
  if (!svo_adapter::mono8(initial_image, &data, &w, &h, &stride)) return;
  track_drawing.create(h, w, CV_8UC3);  // :75-76 (clone + GRAY2RGB)
  const vector<svo::Point2f> &from = impl_->drawn_initial(), &to = impl_->drawn_current();
  // green arrows, thickness 4, keyframe position -> current position (:77-82), this repository's rasteriser
  (void)svo_draw_track(data, w, h, stride, reinterpret_cast<const float *>(from.data()), reinterpret_cast<const float *>(to.data()),
                       (int)to.size(), track_drawing.data);
}

void FeatureTracker::draw_track() {
  if (!impl_) return;
  impl_->draw_track();  // snapshot of (initial_features.at(id), feature_set[i]) for every tracked feature
  rasterise();
}

void FeatureTracker::finish_keyframe(const cv::Mat &keyframe_image) {
  if (!impl_) return;
  if (impl_->drawing_serial() != drawn_serial) rasterise();  // arrows over the PREVIOUS keyframe image (:146 runs before :162)
  initial_image = keyframe_image.clone();
}

cv::Mat FeatureTracker::get_drawing() {
  if (impl_ && impl_->drawing_serial() != drawn_serial) rasterise();
  if (track_drawing.empty()) {  // :86-89: nothing drawn yet -> the keyframe image as RGB
    const uint8_t *data; int w, h, stride;
    if (!svo_adapter::mono8(initial_image, &data, &w, &h, &stride)) return track_drawing;
    track_drawing.create(h, w, CV_8UC3);
    (void)svo_draw_track(data, w, h, stride, nullptr, nullptr, 0, track_drawing.data);
  }
  return track_drawing;
}
