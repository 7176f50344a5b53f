// KITTI-odometry ingestion and ATE — the part without any GPU dependency (decoders read files from
// disk: every length, dimension and chunk order is checked; tests/test_host_sanitizers.py runs this file under
// AddressSanitizer + UBSan against malformed inputs).
//   dataset layout  : src/kitti_node.cpp:37-68  (<data_path><SS>/image_0/%06d.png, image_1/..., poses file
//                     <data_path>data_odometry_poses/dataset/poses/SS.txt, 12 doubles per row = 3x4 [R|t])
// PNG decoding uses zlib only (no libpng/OpenCV in this image): 8/16-bit, gray / gray+alpha / RGB / RGBA,
// non-interlaced; colour is reduced with the usual 0.299/0.587/0.114 weights.  PGM (P5) is accepted too.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#include "svo.h"

namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool read_file(const char* path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok;
}

int decode_pgm(const std::vector<uint8_t>& d, uint8_t* buf, size_t cap, int* w, int* h) {
  size_t pos = 2;
  int vals[3], got = 0;
  while (got < 3 && pos < d.size()) {
    if (d[pos] == '#') { while (pos < d.size() && d[pos] != '\n') ++pos; continue; }
    if (d[pos] <= ' ') { ++pos; continue; }
    if (d[pos] < '0' || d[pos] > '9') return SVO_ERR_INVALID;  // anything but a digit here is not a PGM header
    long long v = 0;
    while (pos < d.size() && d[pos] >= '0' && d[pos] <= '9') {
      v = v * 10 + (d[pos++] - '0');
      if (v > 65535) return SVO_ERR_INVALID;  // image sides are bounded by the corner key packing (y << 16 | x)
    }
    vals[got++] = (int)v;
  }
  if (got < 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0) return SVO_ERR_INVALID;
  ++pos;  // single whitespace after maxval
  const size_t n = (size_t)vals[0] * (size_t)vals[1];
  if (n > cap || pos + n > d.size()) return SVO_ERR_CAPACITY;
  memcpy(buf, d.data() + pos, n);
  *w = vals[0]; *h = vals[1];
  return SVO_OK;
}

int decode_png(const std::vector<uint8_t>& d, uint8_t* buf, size_t cap, int* w, int* h) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (d.size() < 33 || memcmp(d.data(), sig, 8)) return SVO_ERR_INVALID;
  size_t pos = 8;
  int W = 0, H = 0, depth = 0, ctype = 0;
  bool have_ihdr = false;
  std::vector<uint8_t> idat;
  while (pos + 12 <= d.size()) {
    const uint32_t len = be32(&d[pos]);
    const uint8_t* type = &d[pos + 4];
    const uint8_t* data = &d[pos + 8];
    if ((size_t)len > d.size() || pos + 12 + (size_t)len > d.size()) return SVO_ERR_INVALID;
    if (!memcmp(type, "IHDR", 4)) {
      // exactly one IHDR, first chunk, 13 bytes (a short one would be read past its end)
      if (have_ihdr || pos != 8 || len != 13) return SVO_ERR_INVALID;
      have_ihdr = true;
      const uint32_t w32 = be32(data), h32 = be32(data + 4);
      if (w32 == 0 || h32 == 0 || w32 > 65535u || h32 > 65535u) return SVO_ERR_INVALID;
      W = (int)w32; H = (int)h32; depth = data[8]; ctype = data[9];
      if (data[10] != 0 || data[11] != 0 || data[12] != 0) return SVO_ERR_INVALID;  // interlaced PNGs unsupported
    } else if (!memcmp(type, "IDAT", 4)) {
      if (!have_ihdr) return SVO_ERR_INVALID;
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + len;
  }
  int channels = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
  if (!channels || (depth != 8 && depth != 16) || W <= 0 || H <= 0) return SVO_ERR_INVALID;
  if ((size_t)W * H > cap) return SVO_ERR_CAPACITY;
  const int bpp = channels * depth / 8;
  const size_t rowbytes = (size_t)W * bpp;
  std::vector<uint8_t> raw((rowbytes + 1) * H);
  uLongf rawlen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return SVO_ERR_INVALID;
  std::vector<uint8_t> prev(rowbytes, 0), cur(rowbytes);
  for (int y = 0; y < H; ++y) {
    const uint8_t ft = raw[(rowbytes + 1) * y];
    const uint8_t* in = &raw[(rowbytes + 1) * y + 1];
    for (size_t i = 0; i < rowbytes; ++i) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int v = in[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: {
          const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return SVO_ERR_INVALID;
      }
      cur[i] = (uint8_t)v;
    }
    uint8_t* out = buf + (size_t)y * W;
    const int step = depth / 8;  // 16-bit samples: most significant byte
    for (int x = 0; x < W; ++x) {
      const uint8_t* px = &cur[(size_t)x * bpp];
      if (channels <= 2) out[x] = px[0];
      else out[x] = (uint8_t)((299 * px[0] + 587 * px[step] + 114 * px[2 * step] + 500) / 1000);
    }
    prev.swap(cur);
  }
  *w = W; *h = H;
  return SVO_OK;
}

// Horn's closed-form absolute orientation: rotation (unit quaternion) maximising sum (R a_i) . b_i.
void horn_rotation(const double M[9], double R[9]) {
  const double Sxx = M[0], Sxy = M[1], Sxz = M[2], Syx = M[3], Syy = M[4], Syz = M[5], Szx = M[6], Szy = M[7], Szz = M[8];
  double N[16] = {Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx,
                  Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz,
                  Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy,
                  Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz};
  double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 64; ++sweep) {  // cyclic Jacobi on the symmetric 4x4
    double off = 0;
    for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += N[4 * p + q] * N[4 * p + q];
    if (off < 1e-30) break;
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (fabs(N[4 * p + q]) < 1e-300) continue;
        const double th = (N[4 * q + q] - N[4 * p + p]) / (2 * N[4 * p + q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
        const double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 4; ++k) { const double a = N[4 * k + p], b = N[4 * k + q]; N[4 * k + p] = c * a - s * b; N[4 * k + q] = s * a + c * b; }
        for (int k = 0; k < 4; ++k) { const double a = N[4 * p + k], b = N[4 * q + k]; N[4 * p + k] = c * a - s * b; N[4 * q + k] = s * a + c * b; }
        for (int k = 0; k < 4; ++k) { const double a = V[4 * k + p], b = V[4 * k + q]; V[4 * k + p] = c * a - s * b; V[4 * k + q] = s * a + c * b; }
      }
  }
  int best = 0;
  for (int k = 1; k < 4; ++k) if (N[5 * k] > N[5 * best]) best = k;
  const double w = V[best], x = V[4 + best], y = V[8 + best], z = V[12 + best];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
}  // namespace

extern "C" int svo_image_read_gray(const char* path, uint8_t* buf, size_t cap, int* w, int* h) {
  if (!path || !buf || !w || !h) return SVO_ERR_INVALID;
  std::vector<uint8_t> d;
  if (!read_file(path, d) || d.size() < 8) return SVO_ERR_INVALID;
  if (d[0] == 'P' && d[1] == '5') return decode_pgm(d, buf, cap, w, h);
  return decode_png(d, buf, cap, w, h);
}

extern "C" int svo_kitti_read_poses(const char* poses_file, double* rt12, int cap_frames, int* n) {
  if (!poses_file || !rt12 || !n) return SVO_ERR_INVALID;
  FILE* f = fopen(poses_file, "r");
  if (!f) return SVO_ERR_INVALID;
  int k = 0;
  double v[12];
  while (k < cap_frames) {  // src/kitti_node.cpp:47-50: rows of 12 whitespace-separated doubles
    int got = 0;
    for (; got < 12; ++got)
      if (fscanf(f, "%lf", &v[got]) != 1) break;
    if (got < 12) break;
    memcpy(rt12 + 12 * (size_t)k, v, sizeof(v));
    ++k;
  }
  fclose(f);
  *n = k;
  return SVO_OK;
}

// Absolute trajectory error: RMSE of positions after the best rigid (optionally similarity) alignment est -> gt.
extern "C" int svo_ate_rmse(const double* est_xyz, const double* gt_xyz, int n, int with_scale, double* rmse) {
  if (!est_xyz || !gt_xyz || !rmse || n < 3) return SVO_ERR_INVALID;
  double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { ca[k] += est_xyz[3 * i + k]; cb[k] += gt_xyz[3 * i + k]; }
  for (int k = 0; k < 3; ++k) { ca[k] /= n; cb[k] /= n; }
  double M[9] = {0}, sa = 0;
  for (int i = 0; i < n; ++i) {
    double a[3], b[3];
    for (int k = 0; k < 3; ++k) { a[k] = est_xyz[3 * i + k] - ca[k]; b[k] = gt_xyz[3 * i + k] - cb[k]; sa += a[k] * a[k]; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[3 * r + c] += a[r] * b[c];
  }
  double R[9];
  horn_rotation(M, R);
  double scale = 1.0;
  if (with_scale && sa > 0) {
    double num = 0;
    for (int i = 0; i < n; ++i) {
      double a[3], b[3];
      for (int k = 0; k < 3; ++k) { a[k] = est_xyz[3 * i + k] - ca[k]; b[k] = gt_xyz[3 * i + k] - cb[k]; }
      for (int r = 0; r < 3; ++r) num += b[r] * (R[3 * r] * a[0] + R[3 * r + 1] * a[1] + R[3 * r + 2] * a[2]);
    }
    scale = num / sa;
  }
  double sse = 0;
  for (int i = 0; i < n; ++i) {
    double a[3];
    for (int k = 0; k < 3; ++k) a[k] = est_xyz[3 * i + k] - ca[k];
    for (int r = 0; r < 3; ++r) {
      const double e = scale * (R[3 * r] * a[0] + R[3 * r + 1] * a[1] + R[3 * r + 2] * a[2]) + cb[r] - gt_xyz[3 * i + r];
      sse += e * e;
    }
  }
  *rmse = sqrt(sse / n);
  return SVO_OK;
}

