// oracle: cv::goodFeaturesToTrack(img, out, maxCorners, quality, minDistance) with blockSize=3,
// useHarris=false, no mask — the call at src/image_processor.cpp:22.
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED vs OpenCV (no OpenCV here, no reference vectors);
// restates SURVEY.md Appendix A.1 with this declared operation order:
//   1. scale = 1/(4*3*255).  Dx: row pass d[x]=float(I[x+1]-I[x-1]) (exact), column pass
//      (top+bot)*k1 + mid*k0 with k1=float(scale), k0=float(2*scale); Dy: row pass
//      float(I[x])*k0 + float(I[x-1]+I[x+1])*k1, column pass bot-top.  BORDER_REFLECT_101.
//   2. xx=dx*dx, xy=dx*dy, yy=dy*dy in f32.
//   3. 3x3 unnormalised box sum in double: per row (l+m+r), then (top+mid+bot); REFLECT_101 applied
//      to the *covariance* coordinates; rounded to f32.
//   4. a=xx*0.5f, c=yy*0.5f, eig=(a+c)-sqrtf((a-c)*(a-c)+b*b)   (no FMA contraction).
//   5. maxVal over image; thr=float(double(maxVal)*quality); keep eig>thr; candidate iff interior
//      pixel (1<=x<=W-2, 1<=y<=H-2), value != 0 and value == 3x3 max of thresholded map
//      (neighbours outside the image ignored).
//   6. total order: value descending, then raster index (y*W+x) descending.
//   7. greedy min-distance on a grid of cell=round(minDistance): reject if an accepted corner in
//      the 3x3 neighbouring cells has dx*dx+dy*dy < minDistance^2; stop at maxCorners.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "svo_oracle.h"

namespace {
inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}
}  // namespace

extern "C" void ora_corner_response(const uint8_t* img, int w, int h, int stride, float* eig) {
  const double scale = 1.0 / (4.0 * 3.0 * 255.0);
  const float k1 = (float)(1.0 * scale);
  const float k0 = (float)(2.0 * scale);
  std::vector<float> xx((size_t)w * h), xy((size_t)w * h), yy((size_t)w * h);
  auto I = [&](int x, int y) -> int { return img[(size_t)reflect101(y, h) * stride + reflect101(x, w)]; };
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      // row pass results for rows y-1, y, y+1
      float rdx[3], rdy[3];
      for (int j = -1; j <= 1; ++j) {
        const int l = I(x - 1, y + j), m = I(x, y + j), r = I(x + 1, y + j);
        rdx[j + 1] = (float)(r - l);
        rdy[j + 1] = (float)m * k0 + (float)(l + r) * k1;
      }
      const float dx = (rdx[0] + rdx[2]) * k1 + rdx[1] * k0;
      const float dy = rdy[2] - rdy[0];
      const size_t o = (size_t)y * w + x;
      xx[o] = dx * dx;
      xy[o] = dx * dy;
      yy[o] = dy * dy;
    }
  }
  auto box = [&](const std::vector<float>& c, int x, int y) -> float {
    double s[3];
    for (int j = -1; j <= 1; ++j) {
      const size_t row = (size_t)reflect101(y + j, h) * w;
      s[j + 1] = ((double)c[row + reflect101(x - 1, w)] + (double)c[row + x]) +
                 (double)c[row + reflect101(x + 1, w)];
    }
    return (float)((s[0] + s[1]) + s[2]);
  };
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float a = box(xx, x, y) * 0.5f;
      const float b = box(xy, x, y);
      const float c = box(yy, x, y) * 0.5f;
      const float d = a - c;
      eig[(size_t)y * w + x] = (a + c) - std::sqrt(d * d + b * b);
    }
}

extern "C" int ora_corner_detect(const uint8_t* img, int w, int h, int stride, int max_corners,
                                 double quality, double min_distance, float* out_xy,
                                 int* n_candidates) {
  std::vector<float> eig((size_t)w * h);
  ora_corner_response(img, w, h, stride, eig.data());
  float maxv = eig[0];
  for (size_t i = 1; i < eig.size(); ++i) maxv = std::max(maxv, eig[i]);
  const float thr = (float)((double)maxv * quality);
  auto T = [&](int x, int y) -> float {
    const float v = eig[(size_t)y * w + x];
    return v > thr ? v : 0.0f;
  };
  struct Cand { float v; int idx; };
  std::vector<Cand> cand;
  for (int y = 1; y < h - 1; ++y)
    for (int x = 1; x < w - 1; ++x) {
      const float v = T(x, y);
      if (v == 0.0f) continue;
      float m = v;
      for (int j = -1; j <= 1; ++j)
        for (int i = -1; i <= 1; ++i) m = std::max(m, T(x + i, y + j));
      if (v == m) cand.push_back({v, y * w + x});
    }
  if (n_candidates) *n_candidates = (int)cand.size();
  std::sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) {
    return a.v > b.v || (a.v == b.v && a.idx > b.idx);
  });
  int n = 0;
  if (min_distance >= 1.0) {
    const int cell = (int)std::lrint(min_distance);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    std::vector<std::vector<int>> grid((size_t)gw * gh);
    const float md2 = (float)min_distance * (float)min_distance;
    for (const Cand& c : cand) {
      const int y = c.idx / w, x = c.idx % w;
      const int xc = x / cell, yc = y / cell;
      const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
      const int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
      bool good = true;
      for (int yy = y1; yy <= y2 && good; ++yy)
        for (int xx = x1; xx <= x2 && good; ++xx)
          for (int k : grid[(size_t)yy * gw + xx]) {
            const float dx = (float)(x - k % w), dy = (float)(y - k / w);
            if (dx * dx + dy * dy < md2) { good = false; break; }
          }
      if (!good) continue;
      grid[(size_t)yc * gw + xc].push_back(c.idx);
      out_xy[2 * n] = (float)x;
      out_xy[2 * n + 1] = (float)y;
      if (++n == max_corners) break;
    }
  } else {
    for (const Cand& c : cand) {
      out_xy[2 * n] = (float)(c.idx % w);
      out_xy[2 * n + 1] = (float)(c.idx / w);
      if (++n == max_corners) break;
    }
  }
  return n;
}
