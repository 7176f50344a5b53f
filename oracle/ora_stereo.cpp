// oracle: cv::StereoBM::create(ndisp, block)->compute(L, R, disp) followed by
// convertTo(CV_32F, 1/16) — src/image_processor.cpp:173-176.
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED vs OpenCV; restates SURVEY.md Appendix A.2.
// All integer arithmetic, so any evaluation order gives the same result.
//   prefilter XSOBEL cap 31; minDisparity 0; textureThreshold 10; uniquenessRatio 15; no speckle
//   filter; no L-R check.  FILTERED = -16 (=> -1.0f).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ora_constants.h"
#include "svo_oracle.h"

namespace {
const int kCap = 31, kTextureThreshold = 10, kUniquenessRatio = 15;

// Disparity (CV_16S, 4 fractional bits) at one pixel of the valid rectangle, from prefiltered images.
int16_t bm_pixel(const uint8_t* lp, const uint8_t* rp, int w, int x, int y, int ndisp, int half) {
  const int16_t FILTERED = (int16_t)(-16);
  std::vector<int> sad(ndisp + 2, 0);  // sad[i+1], index i = ndisp-1-d
  int tsum = 0;
  for (int dy = -half; dy <= half; ++dy) {
    const uint8_t* lrow = lp + (size_t)(y + dy) * w;
    const uint8_t* rrow = rp + (size_t)(y + dy) * w;
    for (int dx = -half; dx <= half; ++dx) {
      const int lv = lrow[x + dx];
      tsum += std::abs(lv - kCap);
      for (int d = 0; d < ndisp; ++d) sad[ndisp - 1 - d + 1] += std::abs(lv - (int)rrow[x + dx - d]);
    }
  }
  if (tsum < kTextureThreshold) return FILTERED;
  int* s = sad.data() + 1;
  int minsad = 0x7fffffff, mind = -1;
  for (int i = 0; i < ndisp; ++i)
    if (s[i] < minsad) { minsad = s[i]; mind = i; }
  const int thresh = minsad + (minsad * kUniquenessRatio / 100);
  for (int i = 0; i < ndisp; ++i)
    if ((i < mind - 1 || i > mind + 1) && s[i] <= thresh) return FILTERED;
  s[-1] = s[1];
  s[ndisp] = s[ndisp - 2];
  const int p = s[mind + 1], n = s[mind - 1];
  const int dd = p + n - 2 * s[mind] + std::abs(p - n);
  return (int16_t)((((ndisp - mind - 1) * 256 + (dd != 0 ? (p - n) * 256 / dd : 0) + 15) >> 4));
}
}  // namespace

extern "C" void ora_stereo_prefilter(const uint8_t* img, int w, int h, int stride, int cap,
                                     uint8_t* out) {
  auto row = [&](int y) -> const uint8_t* {
    if (y < 0) y = h > 1 ? 1 : 0;          // row -1 == row 1
    if (y >= h) y = h > 1 ? h - 2 : 0;     // row H  == row H-2
    return img + (size_t)y * stride;
  };
  int y = 0;
  for (; y < h - 1; y += 2) {
    for (int k = 0; k < 2; ++k) {
      const int yy = y + k;
      const uint8_t *r0 = row(yy - 1), *r1 = row(yy), *r2 = row(yy + 1);
      uint8_t* d = out + (size_t)yy * w;
      d[0] = d[w - 1] = (uint8_t)cap;
      for (int x = 1; x < w - 1; ++x) {
        const int v = (r0[x + 1] - r0[x - 1]) + 2 * (r1[x + 1] - r1[x - 1]) + (r2[x + 1] - r2[x - 1]);
        d[x] = (uint8_t)(std::min(std::max(v, -cap), cap) + cap);
      }
    }
  }
  for (; y < h; ++y)  // odd leftover row
    for (int x = 0; x < w; ++x) out[(size_t)y * w + x] = (uint8_t)cap;
}

extern "C" void ora_stereo_bm(const uint8_t* left, const uint8_t* right, int w, int h, int stride,
                              int ndisp, int block, int16_t* disp16) {
  std::vector<uint8_t> lp((size_t)w * h), rp((size_t)w * h);
  ora_stereo_prefilter(left, w, h, stride, kCap, lp.data());
  ora_stereo_prefilter(right, w, h, stride, kCap, rp.data());
  const int half = block / 2;
  const int x0 = ndisp - 1 + half, x1 = w - half, y0 = half, y1 = h - half;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int16_t v = -16;
      if (x >= x0 && x < x1 && y >= y0 && y < y1) v = bm_pixel(lp.data(), rp.data(), w, x, y, ndisp, half);
      disp16[(size_t)y * w + x] = v;
    }
}

extern "C" void ora_stereo_disparity_at(const uint8_t* left, const uint8_t* right, int w, int h,
                                        int stride, int ndisp, int block, const float* xy, int n,
                                        float* disp) {
  std::vector<uint8_t> lp((size_t)w * h), rp((size_t)w * h);
  ora_stereo_prefilter(left, w, h, stride, kCap, lp.data());
  ora_stereo_prefilter(right, w, h, stride, kCap, rp.data());
  const int half = block / 2;
  const int x0 = ndisp - 1 + half, x1 = w - half, y0 = half, y1 = h - half;
  for (int i = 0; i < n; ++i) {
    const int x = (int)xy[2 * i], y = (int)xy[2 * i + 1];  // at<float>(it->y, it->x) truncation
    int16_t v = -16;
    if (x >= x0 && x < x1 && y >= y0 && y < y1) v = bm_pixel(lp.data(), rp.data(), w, x, y, ndisp, half);
    disp[i] = (float)v * ora_k::kStereoDisparityScale;
  }
}
