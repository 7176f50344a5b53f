// f4 — FeatureTracker::draw_track / get_drawing (reference src/feature_tracker.cpp:74-91): the keyframe image as RGB
// with one green arrow (thickness 4) per tracked feature from its keyframe position to its current position.
// Visualisation only, host side.  The reference draws with cv::arrowedLine; OpenCV is not available, so
// the rasteriser here is this repository's own (integer Bresenham centre line stamped with a 4-px disc brush, tip
// strokes of 0.1 x length at +-45 degrees as cv::arrowedLine places them): same picture, not pixel-identical.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "ref_constants.h"
#include "svo.h"

namespace {
void stamp(uint8_t* rgb, int w, int h, int cx, int cy, int thickness) {
  // pixels whose centre lies within thickness/2 of (cx, cy)
  const int r = thickness / 2;
  const int r2 = (thickness * thickness) / 4;
  for (int dy = -r; dy <= r; ++dy)
    for (int dx = -r; dx <= r; ++dx) {
      if (dx * dx + dy * dy > r2) continue;
      const int x = cx + dx, y = cy + dy;
      if (x < 0 || y < 0 || x >= w || y >= h) continue;
      uint8_t* p = rgb + 3 * ((size_t)y * w + x);
      p[0] = 0; p[1] = 255; p[2] = 0;  // CV_RGB(0, 255, 0) on an RGB image
    }
}

void line(uint8_t* rgb, int w, int h, int x0, int y0, int x1, int y1, int thickness) {
  const int dx = std::abs(x1 - x0), sx = x0 < x1 ? 1 : -1;
  const int dy = -std::abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
  int err = dx + dy;
  // bound the walk: an end point far outside the image (a lost track) must not cost millions of steps
  for (int guard = 0; guard < 4 * (w + h) + 16; ++guard) {
    stamp(rgb, w, h, x0, y0, thickness);
    if (x0 == x1 && y0 == y1) break;
    const int e2 = 2 * err;
    if (e2 >= dy) { err += dy; x0 += sx; }
    if (e2 <= dx) { err += dx; y0 += sy; }
  }
}
}  // namespace

extern "C" int svo_draw_track(const uint8_t* gray, int width, int height, int row_stride, const float* from_xy,
                              const float* to_xy, int n, uint8_t* rgb) {
  if (!gray || !rgb || width < 1 || height < 1 || row_stride < width || n < 0 || (n > 0 && (!from_xy || !to_xy)))
    return SVO_ERR_INVALID;
  for (int y = 0; y < height; ++y)  // cv::cvtColor(GRAY2RGB), src/feature_tracker.cpp:76
    for (int x = 0; x < width; ++x) {
      const uint8_t v = gray[(size_t)y * row_stride + x];
      uint8_t* p = rgb + 3 * ((size_t)y * width + x);
      p[0] = v; p[1] = v; p[2] = v;
    }
  const int thickness = svo_ref::DRAW_THICKNESS;  // :81
  const double tip_length = 0.1;  // cv::arrowedLine default
  for (int i = 0; i < n; ++i) {
    const double ax = from_xy[2 * i], ay = from_xy[2 * i + 1], bx = to_xy[2 * i], by = to_xy[2 * i + 1];
    if (!(std::isfinite(ax) && std::isfinite(ay) && std::isfinite(bx) && std::isfinite(by))) continue;
    const int x0 = (int)std::lround(ax), y0 = (int)std::lround(ay), x1 = (int)std::lround(bx), y1 = (int)std::lround(by);
    line(rgb, width, height, x0, y0, x1, y1, thickness);
    const double tip = std::sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)) * tip_length;
    const double ang = std::atan2(ay - by, ax - bx);
    const double kPi4 = 0.78539816339744830962;
    for (int s = -1; s <= 1; s += 2) {
      const int px = (int)std::lround(bx + tip * std::cos(ang + s * kPi4)), py = (int)std::lround(by + tip * std::sin(ang + s * kPi4));
      line(rgb, width, height, px, py, x1, y1, thickness);
    }
  }
  return SVO_OK;
}
