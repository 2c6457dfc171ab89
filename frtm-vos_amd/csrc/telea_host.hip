// HOST-side hole fill of the reference's first-frame augmentation (round 6; VERDICT r5 "Next" #9):
//   image = cv2.inpaint(image, mask1, inpaintRadius = d, flags = cv2.INPAINT_TELEA)            reference model/augmenter.py:317-324 (d = 1 at :497)
// The reference runs this step on the CPU through OpenCV, once per object on its first frame.  OpenCV is absent here; this is a restatement from the
// published algorithm (A. Telea, "An image inpainting technique based on the fast marching method", J. Graphics Tools 9(1), 2004) in the structure
// the oracle's restatement (oracle/aug_ref.py: telea_fill_ref) takes from OpenCV's modules/photo/src/inpaint.cpp -- arrays padded by one pixel, a
// stable priority queue on the arrival time T, a pixel inpainted when it is first REACHED, weights dir * dst * lev, the image-gradient term
// normalised by its own length -- with the SAME arithmetic (which operation is float, which double) so that the two agree bit for bit
// (tests/test_cpu_host.py).  PARITY WITH OpenCV ITSELF IS UNPINNED, like the oracle's.
//
// It is an OPTION of the product (ImageAugmenter(fill='telea')): the default first-frame fill is the device-side pull-push pyramid
// (csrc/image_ops.hip), whose effect on J&F against this fill is measured in profiles/r06_fill_evidence.txt.  Fast marching is sequential by
// nature (a heap ordered by arrival time), which is why it runs on the host -- as it does in the reference.  No GPU involved: every pointer is host memory.
#include <cmath>
#include <cstdint>
#include <queue>
#include <vector>
#include "frtm_common.h"
#include "../../include/frtm_hip.h"

#pragma clang fp contract(off)        // numpy does not fuse multiply-add: neither may this file

namespace {
constexpr unsigned char KNOWN = 0, BAND = 1, INSIDE = 2;
struct Item { float t; unsigned seq; int i, j; };
struct Later { bool operator()(const Item& a, const Item& b) const { return a.t > b.t || (a.t == b.t && a.seq > b.seq); } };
}  // namespace

extern "C" int frtm_telea_inpaint_u8(const unsigned char* image_chw, const unsigned char* hole_hw, int C, int H, int W, int radius,
                                     unsigned char* out_chw) {
  FRTM_CHECK_ARG(image_chw && hole_hw && out_chw && C >= 1 && C <= 4 && H >= 2 && W >= 2 && radius >= 1 && radius <= 32, "frtm_telea_inpaint_u8: bad argument");
  const int rows = H + 2, cols = W + 2;
  const size_t hw = (size_t)H * W;
  std::vector<unsigned char> f((size_t)rows * cols, KNOWN);
  std::vector<float> t((size_t)rows * cols, 1.0e6f);
  std::vector<float> out((size_t)C * hw);
  for (size_t q = 0; q < (size_t)C * hw; ++q) out[q] = (float)image_chw[q];
  auto F = [&](int i, int j) -> unsigned char& { return f[(size_t)i * cols + j]; };
  auto T = [&](int i, int j) -> float& { return t[(size_t)i * cols + j]; };
  auto mk = [&](int i, int j) { return i >= 1 && i <= H && j >= 1 && j <= W && hole_hw[(size_t)(i - 1) * W + (j - 1)] != 0; };
  std::priority_queue<Item, std::vector<Item>, Later> heap;
  unsigned seq = 0;
  // narrow band = cross-dilated hole minus the hole, its border row / column cleared; raster order, all at T = 0
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) {
      if (mk(i, j)) { F(i, j) = INSIDE; continue; }
      const bool band = (mk(i - 1, j) || mk(i + 1, j) || mk(i, j - 1) || mk(i, j + 1)) && i > 0 && i < rows - 1 && j > 0 && j < cols - 1;
      if (band) { F(i, j) = BAND; T(i, j) = 0.f; heap.push(Item{0.f, seq++, i, j}); }
    }
  auto solve = [&](int i1, int j1, int i2, int j2) -> double {
    const double a11 = (double)T(i1, j1), a22 = (double)T(i2, j2);
    const double m12 = a11 < a22 ? a11 : a22;
    if (F(i1, j1) != INSIDE) {
      if (F(i2, j2) != INSIDE) {
        if (std::fabs(a11 - a22) >= 1.0) return 1 + m12;
        return (a11 + a22 + std::sqrt(2 - (a11 - a22) * (a11 - a22))) * 0.5;
      }
      return 1 + a11;
    }
    if (F(i2, j2) != INSIDE) return 1 + a22;
    return 1 + m12;
  };
  auto px = [&](int c, int y, int x) -> float { return out[(size_t)c * hw + (size_t)y * W + x]; };
  while (!heap.empty()) {
    const Item it = heap.top();
    heap.pop();
    const int ii = it.i, jj = it.j;
    F(ii, jj) = KNOWN;
    const int nb[4][2] = {{ii - 1, jj}, {ii, jj - 1}, {ii + 1, jj}, {ii, jj + 1}};
    for (int q = 0; q < 4; ++q) {
      const int i = nb[q][0], j = nb[q][1];
      if (i <= 1 || j <= 1 || i > rows - 1 || j > cols - 1) continue;          // (image row 0 / column 0 are never filled: OpenCV's bounds test)
      if (i >= rows || j >= cols || F(i, j) != INSIDE) continue;
      double d0 = solve(i - 1, j, i, j - 1), d1 = solve(i + 1, j, i, j - 1), d2 = solve(i - 1, j, i, j + 1), d3 = solve(i + 1, j, i, j + 1);
      double dm = d0 < d1 ? d0 : d1; dm = dm < d2 ? dm : d2; dm = dm < d3 ? dm : d3;
      const float dist = (float)dm;
      T(i, j) = dist;
      float gtx, gty;                                                            // gradient of T, one-sided next to pixels that are still inside
      if (F(i, j + 1) != INSIDE) gtx = (F(i, j - 1) != INSIDE) ? (T(i, j + 1) - T(i, j - 1)) * 0.5f : (T(i, j + 1) - T(i, j));
      else gtx = (F(i, j - 1) != INSIDE) ? (T(i, j) - T(i, j - 1)) : 0.f;
      if (F(i + 1, j) != INSIDE) gty = (F(i - 1, j) != INSIDE) ? (T(i + 1, j) - T(i - 1, j)) * 0.5f : (T(i + 1, j) - T(i, j));
      else gty = (F(i - 1, j) != INSIDE) ? (T(i, j) - T(i - 1, j)) : 0.f;
      float Ia[4] = {0.f, 0.f, 0.f, 0.f}, Jx[4] = {0.f, 0.f, 0.f, 0.f}, Jy[4] = {0.f, 0.f, 0.f, 0.f};
      float ssum = 1.0e-20f;
      for (int k = i - radius; k <= i + radius; ++k) {
        const int km = k - 1 + (k == 1), kp = k - 1 - (k == rows - 2);
        for (int l = j - radius; l <= j + radius; ++l) {
          const int lm = l - 1 + (l == 1), lp = l - 1 - (l == cols - 2);
          if (!(k > 0 && l > 0 && k < rows - 1 && l < cols - 1)) continue;
          if (F(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > radius * radius) continue;
          const float ry = (float)(i - k), rx = (float)(j - l);
          const float len2 = rx * rx + ry * ry;
          const float dst = (float)(1.0 / ((double)len2 * std::sqrt((double)len2)));
          const float lev = (float)(1.0 / (1 + std::fabs((double)T(k, l) - (double)T(i, j))));
          float dr = rx * gtx + ry * gty;
          if (std::fabs((double)dr) <= 0.01) dr = 0.000001f;
          const float w = std::fabs(dst * lev * dr);
          for (int c = 0; c < C; ++c) {
            // image gradient at the known pixel (k, l), from known neighbours only (image coordinates = padded - 1, clamped)
            float gix, giy;
            if (F(k, l + 1) != INSIDE) gix = (F(k, l - 1) != INSIDE) ? (px(c, km, lp + 1) - px(c, km, lm - 1)) * 2.0f : (px(c, km, lp + 1) - px(c, km, lm));
            else gix = (F(k, l - 1) != INSIDE) ? (px(c, km, lp) - px(c, km, lm - 1)) : 0.f;
            if (F(k + 1, l) != INSIDE) giy = (F(k - 1, l) != INSIDE) ? (px(c, kp + 1, lm) - px(c, km - 1, lm)) * 2.0f : (px(c, kp + 1, lm) - px(c, km, lm));
            else giy = (F(k - 1, l) != INSIDE) ? (px(c, kp, lm) - px(c, km - 1, lm)) : 0.f;
            Ia[c] += w * px(c, km, lm);
            Jx[c] -= w * (gix * rx);
            Jy[c] -= w * (giy * ry);
          }
          ssum += w;
        }
      }
      for (int c = 0; c < C; ++c) {
        const float sat = Ia[c] / ssum + (Jx[c] + Jy[c]) / (std::sqrt(Jx[c] * Jx[c] + Jy[c] * Jy[c]) + 1.0e-20f) + 0.5f;
        float r = std::nearbyintf(sat);                                          // cv::saturate_cast<uchar>(float): round to nearest (even), saturate
        r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
        out[(size_t)c * hw + (size_t)(i - 1) * W + (j - 1)] = r;
      }
      F(i, j) = BAND;
      heap.push(Item{dist, seq++, i, j});
    }
  }
  for (size_t q = 0; q < (size_t)C * hw; ++q) out_chw[q] = (unsigned char)out[q];
  return FRTM_OK;
}
