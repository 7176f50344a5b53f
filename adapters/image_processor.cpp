// ImageProcessor over svo::ImageProcessor.  Replaces src/image_processor.cpp of the reference.
#include "image_processor.hpp"

ImageProcessor::ImageProcessor(cv::Mat cam_mat, shared_ptr<FeatureTracker> tracker, shared_ptr<BundleAdjuster> adjuster,
                               float bline, float min_feature_distance, float parallax_thresh)
    : bundle_adjuster(adjuster), feature_tracker(tracker), camera_matrix(cam_mat), baseline(bline),
      min_feature_distance(min_feature_distance), parallax_thresh(parallax_thresh) {
  svo_ctx *ctx = svo_adapter::context();
  if (!ctx || !tracker || !adjuster || !tracker->impl() || !adjuster->impl()) return;
  if (cam_mat.rows != 3 || cam_mat.cols != 3 || cam_mat.type() != CV_32F) return;  // the reference reads it with at<float>
  float K[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K[3 * i + j] = cam_mat.at<float>(i, j);
  // 300 corners at quality 0.1: the literals of src/image_processor.cpp:22
  impl_.reset(new svo::ImageProcessor(ctx, K, tracker->impl(), adjuster->impl(), bline, min_feature_distance,
                                      parallax_thresh, 300, 0.1, 1));
  if (!impl_->ok()) impl_.reset();
}

ImageProcessor::~ImageProcessor() {}

void ImageProcessor::process(const StereoPair &stereo_pair) {
  const uint8_t *l, *r; int w, h, sl, w2, h2, sr;
  if (!impl_ || !svo_adapter::mono8(stereo_pair.left, &l, &w, &h, &sl) || !svo_adapter::mono8(stereo_pair.right, &r, &w2, &h2, &sr) ||
      w != w2 || h != h2)
    return;
  // a non-zero library status maps to the reference's convention: the frame is skipped silently
  if (impl_->process_host(l, sl, r, sr, w, h, stereo_pair.t) != SVO_OK) return;
  if (impl_->stats().is_keyframe) {
    feature_tracker->finish_keyframe(stereo_pair.left);        // draw_track over the previous keyframe, then the new image
    bundle_adjuster->note_keyframe_image(stereo_pair.left);    // Keyframe::image of get_last_keyframe()
  }
}
