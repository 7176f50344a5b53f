// oracle: the reference's first-party literals as THIS restatement uses them (citations: reference file:line).
// TEST INFRASTRUCTURE ONLY.  Kept separate from the product's csrc/ref_constants.h on purpose: tests/test_constants.py
// checks golden (extracted from the reference text) == product == oracle, three independent statements.
#ifndef ORA_CONSTANTS_H_
#define ORA_CONSTANTS_H_
namespace ora_k {
const int kMinDetected = 4;                 // src/image_processor.cpp:23
const double kKeyframePercentLost = 0.4;    // src/image_processor.cpp:63
const int kPnpIterations = 100;             // src/image_processor.cpp:80
const float kPnpReprojError = 8.0f;         // src/image_processor.cpp:80
const double kPnpConfidence = 0.99;         // src/image_processor.cpp:80
const int kStereoNumDisparities = 16 * 3;   // src/image_processor.cpp:174
const int kStereoBlockSize = 21;            // src/image_processor.cpp:174
const float kStereoDisparityScale = 1.0f / 16.0f;  // src/image_processor.cpp:176
const float kTriangulateMinDisparity = 0.0f;       // src/image_processor.cpp:194 (exclusive)
const int kLkWin = 21;                      // src/feature_tracker.cpp:24
const int kLkMaxLevel = 3;                  // src/feature_tracker.cpp:24
const int kLkMaxIterations = 30;            // src/feature_tracker.cpp:25
const double kLkEpsilon = 0.01;             // src/feature_tracker.cpp:25
const float kLkMinEigThreshold = 1e-2f;     // src/feature_tracker.cpp:26
const double kFbMaxDistance = 2.0;          // src/feature_tracker.cpp:47
const float kMaxParallax = 200.f;           // src/feature_tracker.cpp:53
}  // namespace ora_k
#endif
