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

// ---------------------------------------------------------------------------------------------------------------
// Closed rooms for the room-stitching configurations (BASELINE configs[0] and [4]): a box room seen from inside,
// furniture standing on the floor (y grows DOWNWARD in the KinFu frame, so the floor is the high-y wall), and a
// turntable trajectory that looks all the way round from near the room's centre.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct Box {
  double lo[3], hi[3];
};
struct RoomScene {
  Box room;
  Box blocks[23];
  double sphere_c[3], sphere_r;
};

RoomScene make_room(int variant) {
  static const double ext[4][6] = {{0.25, 2.75, 0.30, 2.70, 0.20, 2.80},
                                   {0.40, 2.60, 0.30, 2.70, 0.25, 2.75},
                                   {0.20, 2.80, 0.35, 2.65, 0.50, 2.50},
                                   {0.50, 2.50, 0.30, 2.70, 0.30, 2.70}};
  const double* e = ext[((variant % 4) + 4) % 4];
  RoomScene s;
  s.room = {{e[0], e[2], e[4]}, {e[1], e[3], e[5]}};
  const double x0 = e[0], x1 = e[1], y1 = e[3], z0 = e[4], z1 = e[5];
  // every wall carries pieces at camera height (y ~ 1.5): their sides and tops pin the translation that a bare wall
  // leaves free for point-to-plane ICP
  s.blocks[0] = {{x0, y1 - 0.80, z0}, {x0 + 0.50, y1, z0 + 0.80}};                  // low cupboard, corner (x0, z0)
  s.blocks[1] = {{x1 - 0.70, y1 - 1.10, z1 - 0.50}, {x1, y1, z1}};                  // chest, corner (x1, z1)
  s.blocks[2] = {{1.20, y1 - 0.60, z1 - 0.40}, {1.90, y1, z1}};                     // bench on the z1 wall
  s.blocks[3] = {{x0, y1 - 1.50, 1.30}, {x0 + 0.40, y1, 1.90}};                     // tall cabinet on the x0 wall
  s.blocks[4] = {{x1 - 0.30, 1.10, 0.90}, {x1, 1.50, 1.60}};                        // shelf on the x1 wall
  s.blocks[5] = {{x0, 0.80, 2.00}, {x0 + 0.25, 1.15, z1 - 0.15}};                   // shelf on the x0 wall
  s.blocks[6] = {{0.60, 1.30, z1 - 0.35}, {1.05, y1, z1}};                          // bookcase on the z1 wall
  s.blocks[7] = {{1.60, 1.00, z0}, {2.20, y1, z0 + 0.45}};                          // wardrobe on the z0 wall
  s.blocks[8] = {{0.95, 1.25, z0}, {1.35, 1.75, z0 + 0.20}};                        // picture box on the z0 wall
  s.blocks[9] = {{x1 - 0.55, 0.95, 1.85}, {x1, y1, 2.10}};                          // pillar on the x1 wall
  s.blocks[10] = {{1.25, 1.15, z1 - 0.25}, {1.85, 1.40, z1}};                       // shelf on the z1 wall
  s.blocks[11] = {{x0, 1.35, 0.95}, {x0 + 0.20, 1.60, 1.25}};                       // small box on the x0 wall
  // two rails running round the room (a picture rail and a dado rail): horizontal edges in EVERY view, which is
  // what keeps the vertical translation observable while the camera faces a flat wall
  const double rail[2][3] = {{1.15, 1.27, 0.10}, {1.80, 1.90, 0.06}};  // y from, y to, depth
  for (int r = 0; r < 2; ++r) {
    const double ya = rail[r][0], yb = rail[r][1], dp = rail[r][2];
    s.blocks[12 + 4 * r + 0] = {{x0, ya, z0}, {x0 + dp, yb, z1}};
    s.blocks[12 + 4 * r + 1] = {{x1 - dp, ya, z0}, {x1, yb, z1}};
    s.blocks[12 + 4 * r + 2] = {{x0, ya, z0}, {x1, yb, z0 + dp}};
    s.blocks[12 + 4 * r + 3] = {{x0, ya, z1 - dp}, {x1, yb, z1}};
  }
  // the ceiling (low y) carries two crossing beams and a lamp box: the upward-looking turn would otherwise see one
  // flat ceiling and one flat wall, free to slide along their common edge
  const double y0 = e[2];
  s.blocks[20] = {{x0, y0, 1.00}, {x1, y0 + 0.15, 1.15}};
  s.blocks[21] = {{1.70, y0, z0}, {1.85, y0 + 0.12, z1}};
  s.blocks[22] = {{1.25, y0, 1.75}, {1.55, y0 + 0.35, 2.05}};
  s.sphere_c[0] = x1 - 0.45;
  s.sphere_c[1] = 1.55;
  s.sphere_c[2] = z0 + 0.50;
  s.sphere_r = 0.25;
  return s;
}

double trace_box_outside(const Box& b, const double o[3], const double d[3]) {  // slab method, entry point
  double t0 = 0.0, t1 = std::numeric_limits<double>::infinity();
  for (int ax = 0; ax < 3; ++ax) {
    if (d[ax] == 0.0) {
      if (o[ax] < b.lo[ax] || o[ax] > b.hi[ax]) return std::numeric_limits<double>::infinity();
      continue;
    }
    double a = (b.lo[ax] - o[ax]) / d[ax], c = (b.hi[ax] - o[ax]) / d[ax];
    if (a > c) {
      const double t = a;
      a = c;
      c = t;
    }
    if (a > t0) t0 = a;
    if (c < t1) t1 = c;
    if (t0 > t1) return std::numeric_limits<double>::infinity();
  }
  return t0 > 1e-9 ? t0 : std::numeric_limits<double>::infinity();
}

double trace_room(const RoomScene& s, const double o[3], const double d[3]) {
  double best = std::numeric_limits<double>::infinity();
  // the six walls from inside: the exit point of the ray from the room box
  {
    double t1 = std::numeric_limits<double>::infinity();
    bool inside = true;
    for (int ax = 0; ax < 3; ++ax) {
      if (o[ax] < s.room.lo[ax] || o[ax] > s.room.hi[ax]) inside = false;
      if (d[ax] == 0.0) continue;
      const double c = ((d[ax] > 0.0 ? s.room.hi[ax] : s.room.lo[ax]) - o[ax]) / d[ax];
      if (c < t1) t1 = c;
    }
    if (inside && t1 > 1e-9) best = t1;
  }
  for (const Box& b : s.blocks) {
    const double t = trace_box_outside(b, o, d);
    if (t < best) best = t;
  }
  {
    const double oc[3] = {o[0] - s.sphere_c[0], o[1] - s.sphere_c[1], o[2] - s.sphere_c[2]};
    const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const double b = 2.0 * (oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2]);
    const double c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - s.sphere_r * s.sphere_r;
    const double disc = b * b - 4.0 * a * c;
    if (disc >= 0.0) {
      const double t = (-b - std::sqrt(disc)) / (2.0 * a);
      if (t > 1e-9 && t < best) best = t;
    }
  }
  return best;
}
}  // namespace

// Room `variant` in its own scan frame: extents (x0, x1, y0, y1, z0, z1) in metres.
extern "C" int hsk_synth_room_extents(int variant, float extents[6]) {
  if (!extents) return HSK_ERR_ARG;
  const RoomScene s = make_room(variant);
  for (int ax = 0; ax < 3; ++ax) {
    extents[2 * ax] = (float)s.room.lo[ax];
    extents[2 * ax + 1] = (float)s.room.hi[ax];
  }
  return HSK_OK;
}

// Turntable pose `frame` of `n_frames`: THREE full turns of yaw from near the room's centre -- the first level
// (pitch 12 deg * sin(3 yaw)), the second swinging up to 38 deg one way and the third 38 deg the other way (half
// sines), so that ceiling and floor are scanned once the walls are in the model.  Position: a circle of radius
// 0.10 m about the centre.  Frame 0 looks along +z from (cx, cy, cz - 0.10).
extern "C" int hsk_synth_room_pose(int variant, int frame, int n_frames, float pose[16]) {
  if (!pose || n_frames < 1) return HSK_ERR_ARG;
  const RoomScene s = make_room(variant);
  const double c[3] = {0.5 * (s.room.lo[0] + s.room.hi[0]), 0.5 * (s.room.lo[1] + s.room.hi[1]), 0.5 * (s.room.lo[2] + s.room.hi[2])};
  const double u = (double)frame / (double)n_frames;
  const double yaw = 6.0 * kPi * u;
  double pitch;
  if (u < 1.0 / 3.0)
    pitch = 12.0 * kPi / 180.0 * std::sin(3.0 * yaw);
  else if (u < 2.0 / 3.0)
    pitch = 38.0 * kPi / 180.0 * std::sin(3.0 * kPi * (u - 1.0 / 3.0));
  else
    pitch = -38.0 * kPi / 180.0 * std::sin(3.0 * kPi * (u - 2.0 / 3.0));
  const double cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch);
  const double R[9] = {cy, sy * sp, sy * cp, 0.0, cp, -sp, -sy, cy * sp, cy * cp};  // Ry(yaw) Rx(pitch)
  const double t[3] = {c[0] - 0.10 * std::sin(yaw), c[1], c[2] - 0.10 * std::cos(yaw)};
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

extern "C" int hsk_synth_room_render(int variant, const float pose[16], int w, int h, float fx, float fy, float cx, float cy,
                                     uint16_t* depth) {
  if (!pose || !depth || w <= 0 || h <= 0) return HSK_ERR_ARG;
  const RoomScene s = make_room(variant);
  double R[9], o[3];
  for (int i = 0; i < 3; ++i) {
    R[i * 3] = pose[i * 4];
    R[i * 3 + 1] = pose[i * 4 + 1];
    R[i * 3 + 2] = pose[i * 4 + 2];
    o[i] = pose[i * 4 + 3];
  }
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      const double dc[3] = {((double)u - (double)cx) / (double)fx, ((double)v - (double)cy) / (double)fy, 1.0};
      const double d[3] = {R[0] * dc[0] + R[1] * dc[1] + R[2] * dc[2], R[3] * dc[0] + R[4] * dc[1] + R[5] * dc[2],
                           R[6] * dc[0] + R[7] * dc[1] + R[8] * dc[2]};
      const double t = trace_room(s, o, d);
      uint16_t mm = 0;
      if (t < 10.0) {
        const double r = std::nearbyint(t * 1000.0);
        if (r >= 1.0 && r <= 65535.0) mm = (uint16_t)r;
      }
      depth[(size_t)v * w + u] = mm;
    }
  return HSK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Holes as a sensor makes them.  A takeDepthSnapshot frame (housescan/HoniHelper.hs:20-36) of a
// structured-light camera has 10-30 % invalid pixels in CONTIGUOUS regions, not SURVEY.md 8(d)'s independent 2 %:
//   * grazing rays: no return where the ray meets the surface at |n . d| < 0.15 (81 deg from the normal);
//   * shadows: the projector sits 75 mm beside the camera, so behind every depth discontinuity along x the far surface
//     is unlit over a band of  b f (1 / z_near - 1 / z_far)  pixels -- kept within 3..5 px; 3 px under horizontal edges;
//   * the range cut: nothing beyond range_cut_m (<= 0: 3.5 m);
//   * absorbing != 0: dark furniture returns nothing (the block of the open scene; the chest and the wardrobe of a room);
//   * and 8(d)'s noise, sigma_mm x (z / 1 m)^2, on what is left -- a counter-based generator keyed by (seed, pixel): the
//     frame is a pure function of its arguments.
// scene < 0: the open scene of 8(d) (hsk_synth_render); 0..3: closed room `scene` (hsk_synth_room_render).
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct Hit {
  double s;
  double n[3];
  int object;   // 0 = a wall of the room, 1 + k = piece of furniture k, 100 = the sphere
};
inline void set_axis_normal(Hit& h, int ax) {
  h.n[0] = h.n[1] = h.n[2] = 0.0;
  h.n[ax] = 1.0;
}
// the entry face of a box (slab method) with its axis
double trace_box_axis(const double lo[3], const double hi[3], const double o[3], const double d[3], int* axis) {
  double t0 = 0.0, t1 = std::numeric_limits<double>::infinity();
  int a0 = 0;
  for (int ax = 0; ax < 3; ++ax) {
    if (d[ax] == 0.0) {
      if (o[ax] < lo[ax] || o[ax] > hi[ax]) return std::numeric_limits<double>::infinity();
      continue;
    }
    double a = (lo[ax] - o[ax]) / d[ax], c = (hi[ax] - o[ax]) / d[ax];
    if (a > c) {
      const double t = a;
      a = c;
      c = t;
    }
    if (a > t0) {
      t0 = a;
      a0 = ax;
    }
    if (c < t1) t1 = c;
    if (t0 > t1) return std::numeric_limits<double>::infinity();
  }
  *axis = a0;
  return t0 > 1e-9 ? t0 : std::numeric_limits<double>::infinity();
}
bool trace_sphere(const double c[3], double r, const double o[3], const double d[3], Hit& best) {
  const double oc[3] = {o[0] - c[0], o[1] - c[1], o[2] - c[2]};
  const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const double b = 2.0 * (oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2]);
  const double cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - r * r;
  const double disc = b * b - 4.0 * a * cc;
  if (disc < 0.0) return false;
  const double t = (-b - std::sqrt(disc)) / (2.0 * a);
  if (!(t > 1e-9 && t < best.s)) return false;
  best.s = t;
  best.object = 100;
  for (int i = 0; i < 3; ++i) best.n[i] = (o[i] + t * d[i] - c[i]) / r;
  return true;
}
// the same surfaces as trace() / trace_room(), with the normal at the hit (the depths agree with those renders bit for bit:
// the same expressions decide the nearest hit)
Hit trace_open_n(const double o[3], const double d[3]) {
  Hit best;
  best.object = 0;
  best.s = std::numeric_limits<double>::infinity();
  set_axis_normal(best, 2);
  const double planes[5][2] = {{0, WX0}, {0, WX1}, {1, WY0}, {1, WY1}, {2, WZ1}};
  for (int i = 0; i < 5; ++i) {
    const int ax = (int)planes[i][0];
    const double c = planes[i][1];
    if (d[ax] == 0.0) continue;
    const double s = (c - o[ax]) / d[ax];
    if (!(s > 1e-9) || s >= best.s) continue;
    const bool low = (i == 0 || i == 2);
    if (low ? !(d[ax] < 0.0) : !(d[ax] > 0.0)) continue;
    const double p[3] = {o[0] + s * d[0], o[1] + s * d[1], o[2] + s * d[2]};
    if (in_room(p, ax)) {
      best.s = s;
      set_axis_normal(best, ax);
    }
  }
  trace_sphere(SC, SR, o, d, best);
  int ax = 0;
  const double t = trace_box_axis(B0, B1, o, d, &ax);
  if (t < best.s) {
    best.s = t;
    best.object = 1;
    set_axis_normal(best, ax);
  }
  return best;
}
Hit trace_room_n(const RoomScene& sc, const double o[3], const double d[3]) {
  Hit best;
  best.object = 0;
  best.s = std::numeric_limits<double>::infinity();
  set_axis_normal(best, 2);
  {
    double t1 = std::numeric_limits<double>::infinity();
    int a1 = 0;
    bool inside = true;
    for (int ax = 0; ax < 3; ++ax) {
      if (o[ax] < sc.room.lo[ax] || o[ax] > sc.room.hi[ax]) inside = false;
      if (d[ax] == 0.0) continue;
      const double c = ((d[ax] > 0.0 ? sc.room.hi[ax] : sc.room.lo[ax]) - o[ax]) / d[ax];
      if (c < t1) {
        t1 = c;
        a1 = ax;
      }
    }
    if (inside && t1 > 1e-9) {
      best.s = t1;
      set_axis_normal(best, a1);
    }
  }
  int piece = 0;
  for (const Box& b : sc.blocks) {
    int ax = 0;
    const double t = trace_box_axis(b.lo, b.hi, o, d, &ax);
    ++piece;
    if (t < best.s) {
      best.s = t;
      best.object = piece;
      set_axis_normal(best, ax);
    }
  }
  trace_sphere(sc.sphere_c, sc.sphere_r, o, d, best);
  return best;
}
inline uint64_t mix64(uint64_t x) {  // splitmix64's finaliser
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
inline double gauss(uint64_t seed, uint64_t index) {  // Box-Muller on two counter-based uniforms
  const uint64_t a = mix64(seed ^ mix64(2 * index)), b = mix64(seed ^ mix64(2 * index + 1));
  const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740993.0);  // (0, 1)
  const double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);            // [0, 1)
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * kPi * u2);
}
}  // namespace

extern "C" int hsk_synth_render_sensor(int scene, const float pose[16], int w, int h, float fx, float fy, float cx, float cy,
                                       uint64_t seed, float sigma_mm, float range_cut_m, int absorbing, uint16_t* depth,
                                       double* hole_fraction) {
  if (!pose || !depth || w <= 0 || h <= 0) return HSK_ERR_ARG;
  const double kGraze = 0.15, kRange = range_cut_m > 0.0f ? (double)range_cut_m : 3.5, kEdge = 0.05, kBaseline = 0.075;
  double R[9], o[3];
  for (int i = 0; i < 3; ++i) {
    R[i * 3] = pose[i * 4];
    R[i * 3 + 1] = pose[i * 4 + 1];
    R[i * 3 + 2] = pose[i * 4 + 2];
    o[i] = pose[i * 4 + 3];
  }
  RoomScene room;
  if (scene >= 0) room = make_room(scene);
  const size_t P = (size_t)w * h;
  double* z = new double[P];          // z-depth in metres, 0 = no return
  unsigned char* bad = new unsigned char[P]();
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      const double dc[3] = {((double)u - (double)cx) / (double)fx, ((double)v - (double)cy) / (double)fy, 1.0};
      const double d[3] = {R[0] * dc[0] + R[1] * dc[1] + R[2] * dc[2], R[3] * dc[0] + R[4] * dc[1] + R[5] * dc[2],
                           R[6] * dc[0] + R[7] * dc[1] + R[8] * dc[2]};
      const Hit hit = scene >= 0 ? trace_room_n(room, o, d) : trace_open_n(o, d);
      const size_t i = (size_t)v * w + u;
      z[i] = 0.0;
      if (!(hit.s < 10.0)) continue;
      z[i] = hit.s;  // (unit camera-z direction: the ray parameter is the z-depth)
      const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      const double cosang = std::fabs(hit.n[0] * d[0] + hit.n[1] * d[1] + hit.n[2] * d[2]) / len;
      if (cosang < kGraze || hit.s > kRange) bad[i] = 1;
      // absorbing surfaces (dark furniture returns nothing): the block of the open scene; the chest and the wardrobe of a room
      if (absorbing && (scene >= 0 ? (hit.object == 2 || hit.object == 8) : hit.object == 1)) bad[i] = 1;
    }
  // shadow bands on the far side of the discontinuities of the TRUE depth (before any pixel is dropped)
  for (int v = 0; v < h; ++v)
    for (int u = 0; u + 1 < w; ++u) {
      const double a = z[(size_t)v * w + u], b = z[(size_t)v * w + u + 1];
      if (a == 0.0 || b == 0.0 || std::fabs(a - b) <= kEdge) continue;
      const double zn = a < b ? a : b, zf = a < b ? b : a;
      int band = (int)std::nearbyint(kBaseline * (double)fx * (1.0 / zn - 1.0 / zf));
      band = band < 3 ? 3 : (band > 5 ? 5 : band);
      const int dir = a < b ? 1 : -1, u0 = a < b ? u + 1 : u;   // the far pixel and outwards from the edge
      for (int k = 0; k < band; ++k) {
        const int uu = u0 + dir * k;
        if (uu < 0 || uu >= w) break;
        if (std::fabs(z[(size_t)v * w + uu] - zf) > kEdge && k > 0) break;   // (the band ends where the far surface does)
        bad[(size_t)v * w + uu] = 1;
      }
    }
  for (int v = 0; v + 1 < h; ++v)
    for (int u = 0; u < w; ++u) {
      const double a = z[(size_t)v * w + u], b = z[(size_t)(v + 1) * w + u];
      if (a == 0.0 || b == 0.0 || std::fabs(a - b) <= kEdge) continue;
      const double zf = a < b ? b : a;
      const int dir = a < b ? 1 : -1, v0 = a < b ? v + 1 : v;
      for (int k = 0; k < 3; ++k) {
        const int vv = v0 + dir * k;
        if (vv < 0 || vv >= h) break;
        if (std::fabs(z[(size_t)vv * w + u] - zf) > kEdge && k > 0) break;
        bad[(size_t)vv * w + u] = 1;
      }
    }
  size_t holes = 0;
  for (size_t i = 0; i < P; ++i) {
    uint16_t mm = 0;
    if (z[i] != 0.0 && !bad[i]) {
      const double noisy = z[i] * 1000.0 + (sigma_mm > 0.0f ? gauss(seed, (uint64_t)i) * (double)sigma_mm * z[i] * z[i] : 0.0);
      const double r = std::nearbyint(noisy);
      if (r >= 1.0 && r <= 65535.0) mm = (uint16_t)r;
    }
    holes += mm == 0;
    depth[i] = mm;
  }
  if (hole_fraction) *hole_fraction = (double)holes / (double)P;
  delete[] z;
  delete[] bad;
  return HSK_OK;
}
