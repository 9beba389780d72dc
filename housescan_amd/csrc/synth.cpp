// synth.cpp -- deterministic synthetic depth stream: a five-sided box room open toward the
// camera, a sphere and a block, rendered analytically as z-depth in uint16 millimetres in the frame layout
// HouseScan receives from OpenNI2 (housescan/HoniHelper.hs:34-36; index i = y*w + x, Main.hs:1298-1300).
// Host-only; all arithmetic in binary64.
#include <cmath>
#include <cstdint>
#include <limits>

#include "../../include/hskinfu.h"

namespace {
const double kPi = 3.14159265358979323846;
const double WX0 = 0.2, WX1 = 2.8, WY0 = 0.3, WY1 = 2.7, WZ1 = 2.8, WZ0 = -5.0;  // room (open at low z)
const double SC[3] = {1.0, 1.9, 1.8}, SR = 0.35;                                  // sphere
const double B0[3] = {1.8, 1.7, 1.2}, B1[3] = {2.4, 2.7, 1.7};                    // block

inline bool in_room(const double p[3], int skip) {
  const double e = 1e-9;
  if (skip != 0 && (p[0] < WX0 - e || p[0] > WX1 + e)) return false;
  if (skip != 1 && (p[1] < WY0 - e || p[1] > WY1 + e)) return false;
  if (skip != 2 && (p[2] < WZ0 - e || p[2] > WZ1 + e)) return false;
  return true;
}

// smallest positive ray parameter s with o + s d on the scene, or +inf
double trace(const double o[3], const double d[3]) {
  double best = std::numeric_limits<double>::infinity();
  // room walls (seen from inside)
  const double planes[5][2] = {{0, WX0}, {0, WX1}, {1, WY0}, {1, WY1}, {2, WZ1}};
  for (int i = 0; i < 5; ++i) {
    const int ax = (int)planes[i][0];
    const double c = planes[i][1];
    if (d[ax] == 0.0) continue;
    const double s = (c - o[ax]) / d[ax];
    if (!(s > 1e-9) || s >= best) continue;
    // only the face looking into the room
    const bool low = (i == 0 || i == 2);
    if (low ? !(d[ax] < 0.0) : !(d[ax] > 0.0)) continue;
    const double p[3] = {o[0] + s * d[0], o[1] + s * d[1], o[2] + s * d[2]};
    if (in_room(p, ax)) best = s;
  }
  // sphere
  {
    const double oc[3] = {o[0] - SC[0], o[1] - SC[1], o[2] - SC[2]};
    const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const double b = 2.0 * (oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2]);
    const double c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - SR * SR;
    const double disc = b * b - 4.0 * a * c;
    if (disc >= 0.0) {
      const double s = (-b - std::sqrt(disc)) / (2.0 * a);
      if (s > 1e-9 && s < best) best = s;
    }
  }
  // block (slab method)
  {
    double t0 = 0.0, t1 = std::numeric_limits<double>::infinity();
    bool ok = true;
    for (int ax = 0; ax < 3 && ok; ++ax) {
      if (d[ax] == 0.0) {
        if (o[ax] < B0[ax] || o[ax] > B1[ax]) ok = false;
        continue;
      }
      double a = (B0[ax] - o[ax]) / d[ax], b = (B1[ax] - o[ax]) / d[ax];
      if (a > b) {
        const double t = a;
        a = b;
        b = t;
      }
      if (a > t0) t0 = a;
      if (b < t1) t1 = b;
      if (t0 > t1) ok = false;
    }
    if (ok && t0 > 1e-9 && t0 < best) best = t0;
  }
  return best;
}
}  // namespace

// Pose of frame k: yaw 12 deg * sin(2 pi k / 150), pitch 4 deg * sin(2 pi k / 100), position on a circle of
// radius 0.15 m in the x-z plane through the KinFu start pose (1.5, 1.5, -0.3).
extern "C" int hsk_synth_pose(int frame, float pose[16]) {
  if (!pose) return HSK_ERR_ARG;
  const double th = 2.0 * kPi * (double)frame / 150.0;
  const double yaw = 12.0 * kPi / 180.0 * std::sin(th);
  const double pitch = 4.0 * kPi / 180.0 * std::sin(2.0 * kPi * (double)frame / 100.0);
  const double cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch);
  // R = Ry(yaw) * Rx(pitch)
  const double R[9] = {cy, sy * sp, sy * cp, 0.0, cp, -sp, -sy, cy * sp, cy * cp};
  const double t[3] = {1.5 + 0.15 * std::sin(th), 1.5, -0.3 + 0.15 * (1.0 - std::cos(th))};
  for (int i = 0; i < 3; ++i) {
    pose[i * 4] = (float)R[i * 3];
    pose[i * 4 + 1] = (float)R[i * 3 + 1];
    pose[i * 4 + 2] = (float)R[i * 3 + 2];
    pose[i * 4 + 3] = (float)t[i];
  }
  pose[12] = pose[13] = pose[14] = 0.0f;
  pose[15] = 1.0f;
  return HSK_OK;
}

extern "C" int hsk_synth_render(const float pose[16], int w, int h, float fx, float fy, float cx, float cy,
                                uint16_t* depth) {
  if (!pose || !depth || w <= 0 || h <= 0) return HSK_ERR_ARG;
  double R[9], o[3];
  for (int i = 0; i < 3; ++i) {
    R[i * 3] = pose[i * 4];
    R[i * 3 + 1] = pose[i * 4 + 1];
    R[i * 3 + 2] = pose[i * 4 + 2];
    o[i] = pose[i * 4 + 3];
  }
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      // camera-space direction with unit z: the ray parameter IS the z-depth
      const double dc[3] = {((double)u - (double)cx) / (double)fx, ((double)v - (double)cy) / (double)fy, 1.0};
      const double d[3] = {R[0] * dc[0] + R[1] * dc[1] + R[2] * dc[2], R[3] * dc[0] + R[4] * dc[1] + R[5] * dc[2],
                           R[6] * dc[0] + R[7] * dc[1] + R[8] * dc[2]};
      const double s = trace(o, d);
      uint16_t mm = 0;
      if (s < 10.0) {
        const double r = std::nearbyint(s * 1000.0);
        if (r >= 1.0 && r <= 65535.0) mm = (uint16_t)r;
      }
      depth[(size_t)v * w + u] = mm;
    }
  return HSK_OK;
}
