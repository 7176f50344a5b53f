// Every first-party literal of the reference that the hot path depends on, by name (citations: reference file:line).
// The kernels and the host chain use THESE names; svo_reference_constants() (include/svo.h) reports them, and
// tests/test_constants.py checks them against tests/golden/constants_golden.json, which is extracted from the
// reference's source text.
#ifndef SVO_REF_CONSTANTS_H_
#define SVO_REF_CONSTANTS_H_

namespace svo_ref {
constexpr int GFTT_MAX_CORNERS = 300;          // src/image_processor.cpp:22
constexpr double GFTT_QUALITY = 0.1;           // src/image_processor.cpp:22
constexpr int MIN_DETECTED = 4;                // src/image_processor.cpp:23
constexpr double KEYFRAME_PERCENT_LOST = 0.4;  // src/image_processor.cpp:63
constexpr int PNP_ITERATIONS = 100;            // src/image_processor.cpp:80
constexpr float PNP_REPROJ_ERROR = 8.0f;       // src/image_processor.cpp:80
constexpr double PNP_CONFIDENCE = 0.99;        // src/image_processor.cpp:80
constexpr int STEREO_NUM_DISPARITIES = 16 * 3; // src/image_processor.cpp:174
constexpr int STEREO_BLOCK_SIZE = 21;          // src/image_processor.cpp:174
constexpr float STEREO_DISPARITY_SCALE = 1.0f / 16.0f;  // src/image_processor.cpp:176
constexpr float TRIANGULATE_MIN_DISPARITY = 0.0f;       // src/image_processor.cpp:194 (exclusive)
constexpr int LK_WIN = 21;                     // src/feature_tracker.cpp:24
constexpr int LK_MAX_LEVEL = 3;                // src/feature_tracker.cpp:24
constexpr int LK_MAX_ITERATIONS = 30;          // src/feature_tracker.cpp:25
constexpr double LK_EPSILON = 0.01;            // src/feature_tracker.cpp:25
constexpr float LK_MIN_EIG_THRESHOLD = 1e-2f;  // src/feature_tracker.cpp:26
constexpr double FB_MAX_DISTANCE = 2.0;        // src/feature_tracker.cpp:47
constexpr float MAX_PARALLAX = 200.f;          // src/feature_tracker.cpp:53
constexpr int DRAW_THICKNESS = 4;              // src/feature_tracker.cpp:81
constexpr float PARALLAX_THRESH = 20.f;        // src/vo_node.cpp:33
constexpr float MIN_FEATURE_DISTANCE = 30.f;   // src/vo_node.cpp:34
constexpr int SLIDING_WINDOW_SIZE = 5;         // src/vo_node.cpp:36
constexpr int MAX_FEATURES = 400;              // src/bundle_adjuster.hpp:75
constexpr double BA_MAX_SOLVER_TIME_S = 0.1;   // src/bundle_adjuster.cpp:11
constexpr int BA_NUM_THREADS = 4;              // src/bundle_adjuster.cpp:12 (Ceres' host threads; reported, unused on the GPU)
}  // namespace svo_ref

#endif
