// products.cpp -- products on the file seam between KinFu and HouseScan (host-only):
//   <room>/cloud_bin.pcd, <room>/cloud_downsampled.pcd  -- read by loadRoom / cloudFromFile
//   (housescan/Main.hs:1738-1762, :1334-1345 via PCD.loadXyz :1320-1323; printed at :2437).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/hskinfu.h"

// binary PCD v0.7 with float32 x y z, the layout pcd-loader's loadXyz expects
extern "C" int hsk_write_pcd_xyz(const char* path, const float* xyz, size_t n) {
  if (!path || (!xyz && n)) return HSK_ERR_ARG;
  FILE* f = fopen(path, "wb");
  if (!f) return HSK_ERR_STATE;
  fprintf(f,
          "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
          "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n",
          n, n);
  const size_t wrote = n ? fwrite(xyz, 12, n, f) : 0;
  const int rc = fclose(f);
  return (wrote == n && rc == 0) ? HSK_OK : HSK_ERR_STATE;
}

// voxel-grid centroid downsample (what produces cloud_downsampled.pcd); output ordered by leaf index
extern "C" int hsk_voxel_downsample(const float* xyz, size_t n, float leaf, float* out, size_t cap, size_t* n_out) {
  if (!n_out || (!xyz && n) || !(leaf > 0.0f)) return HSK_ERR_ARG;
  struct Item {
    uint64_t key;
    uint32_t idx;
  };
  std::vector<Item> items;
  items.reserve(n);
  const float inv = 1.0f / leaf;
  for (size_t i = 0; i < n; ++i) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!(x == x) || !(y == y) || !(z == z)) continue;
    const int64_t ix = (int64_t)std::floor(x * inv) + (1 << 20);
    const int64_t iy = (int64_t)std::floor(y * inv) + (1 << 20);
    const int64_t iz = (int64_t)std::floor(z * inv) + (1 << 20);
    if (ix < 0 || iy < 0 || iz < 0 || ix >= (1 << 21) || iy >= (1 << 21) || iz >= (1 << 21)) continue;
    items.push_back(Item{((uint64_t)iz << 42) | ((uint64_t)iy << 21) | (uint64_t)ix, (uint32_t)i});
  }
  std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.key != b.key ? a.key < b.key : a.idx < b.idx; });
  size_t m = 0;
  for (size_t i = 0; i < items.size();) {
    size_t j = i;
    double sx = 0, sy = 0, sz = 0;
    while (j < items.size() && items[j].key == items[i].key) {
      sx += xyz[3 * (size_t)items[j].idx];
      sy += xyz[3 * (size_t)items[j].idx + 1];
      sz += xyz[3 * (size_t)items[j].idx + 2];
      ++j;
    }
    const double c = (double)(j - i);
    if (out && m < cap) {
      out[3 * m] = (float)(sx / c);
      out[3 * m + 1] = (float)(sy / c);
      out[3 * m + 2] = (float)(sz / c);
    }
    ++m;
    i = j;
  }
  *n_out = m;
  return HSK_OK;
}

// ------------------------------------------------------------------------------------------------------
// Plane detection on the fused cloud: what HouseScan's loadRoom consumes beside the cloud --
//   <room>/planes.txt               one plane per line "a b c d" in PCL form ax + by + cz + d = 0
//                                   (parsed by planeEqsFromFile, housescan/Main.hs:1379-1389, which negates d)
//   <room>/cloud_plane_hull<k>.pcd  polygon vertices of plane k in drawing order (Main.hs:1395-1400, :758-763)
// Deterministic sequential RANSAC (fixed-seed LCG) + PCA refit + 2-D convex hull of the inliers.
// ------------------------------------------------------------------------------------------------------
namespace {
struct Lcg {
  uint64_t s;
  uint32_t next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33);
  }
};

// smallest-eigenvalue eigenvector of a symmetric 3x3 matrix (cyclic Jacobi)
void smallest_eigvec(double A[3][3], double n[3]) {
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 32; ++sweep) {
    double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    if (off < 1e-30) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (std::fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int m = 0;
  if (A[1][1] < A[m][m]) m = 1;
  if (A[2][2] < A[m][m]) m = 2;
  for (int k = 0; k < 3; ++k) n[k] = V[k][m];
}
}  // namespace

// planes_abcd: up to max_planes x 4 floats (unit normal a,b,c and d of ax+by+cz+d=0); labels (optional, n ints):
// index of the plane a point was assigned to, or -1.  Planes are reported in detection order (largest first).
extern "C" int hsk_detect_planes(const float* xyz, size_t n, float dist_thresh, float min_fraction, int max_planes,
                                 int iterations, float* planes_abcd, int* labels, int* n_planes) {
  if (!xyz || !planes_abcd || !n_planes || max_planes <= 0 || !(dist_thresh > 0.0f) || iterations <= 0) return HSK_ERR_ARG;
  std::vector<int> lab(n, -1);
  std::vector<uint32_t> rest(n);
  for (size_t i = 0; i < n; ++i) rest[i] = (uint32_t)i;
  Lcg rng{0x9E3779B97F4A7C15ull};
  const size_t min_inl = (size_t)std::max(3.0, (double)min_fraction * (double)n);
  int found = 0;
  while (found < max_planes && rest.size() >= std::max<size_t>(min_inl, 3)) {
    // score candidates on a bounded subsample of the remaining points
    const size_t m = rest.size();
    const size_t stride = std::max<size_t>(1, m / 20000);
    double best_n[3] = {0, 0, 1}, best_d = 0;
    size_t best_cnt = 0;
    for (int it = 0; it < iterations; ++it) {
      const float* p0 = xyz + 3 * (size_t)rest[rng.next() % m];
      const float* p1 = xyz + 3 * (size_t)rest[rng.next() % m];
      const float* p2 = xyz + 3 * (size_t)rest[rng.next() % m];
      const double u[3] = {(double)p1[0] - p0[0], (double)p1[1] - p0[1], (double)p1[2] - p0[2]};
      const double v[3] = {(double)p2[0] - p0[0], (double)p2[1] - p0[1], (double)p2[2] - p0[2]};
      double nn[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
      const double len = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
      if (len < 1e-9) continue;
      for (double& c : nn) c /= len;
      const double d = -(nn[0] * p0[0] + nn[1] * p0[1] + nn[2] * p0[2]);
      size_t cnt = 0;
      for (size_t i = 0; i < m; i += stride) {
        const float* p = xyz + 3 * (size_t)rest[i];
        if (std::fabs(nn[0] * p[0] + nn[1] * p[1] + nn[2] * p[2] + d) <= dist_thresh) ++cnt;
      }
      if (cnt > best_cnt) {
        best_cnt = cnt;
        best_d = d;
        best_n[0] = nn[0];
        best_n[1] = nn[1];
        best_n[2] = nn[2];
      }
    }
    if (best_cnt * stride < min_inl) break;
    // refit twice by PCA on the inliers of the current estimate
    for (int pass = 0; pass < 2; ++pass) {
      double mean[3] = {0, 0, 0};
      size_t cnt = 0;
      for (uint32_t idx : rest) {
        const float* p = xyz + 3 * (size_t)idx;
        if (std::fabs(best_n[0] * p[0] + best_n[1] * p[1] + best_n[2] * p[2] + best_d) <= dist_thresh) {
          mean[0] += p[0];
          mean[1] += p[1];
          mean[2] += p[2];
          ++cnt;
        }
      }
      if (cnt < 3) break;
      for (double& c : mean) c /= (double)cnt;
      double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (uint32_t idx : rest) {
        const float* p = xyz + 3 * (size_t)idx;
        if (std::fabs(best_n[0] * p[0] + best_n[1] * p[1] + best_n[2] * p[2] + best_d) <= dist_thresh) {
          const double q[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) C[a][b] += q[a] * q[b];
        }
      }
      double nn[3];
      smallest_eigvec(C, nn);
      const double len = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
      if (len < 1e-12) break;
      for (int a = 0; a < 3; ++a) best_n[a] = nn[a] / len;
      best_d = -(best_n[0] * mean[0] + best_n[1] * mean[1] + best_n[2] * mean[2]);
    }
    // assign and remove the inliers
    std::vector<uint32_t> keep;
    keep.reserve(rest.size());
    size_t cnt = 0;
    for (uint32_t idx : rest) {
      const float* p = xyz + 3 * (size_t)idx;
      if (std::fabs(best_n[0] * p[0] + best_n[1] * p[1] + best_n[2] * p[2] + best_d) <= dist_thresh) {
        lab[idx] = found;
        ++cnt;
      } else {
        keep.push_back(idx);
      }
    }
    if (cnt < min_inl) {  // the refit lost the support: undo and stop
      for (int& l : lab)
        if (l == found) l = -1;
      break;
    }
    planes_abcd[4 * found + 0] = (float)best_n[0];
    planes_abcd[4 * found + 1] = (float)best_n[1];
    planes_abcd[4 * found + 2] = (float)best_n[2];
    planes_abcd[4 * found + 3] = (float)best_d;
    ++found;
    rest.swap(keep);
  }
  if (labels)
    for (size_t i = 0; i < n; ++i) labels[i] = lab[i];
  *n_planes = found;
  return HSK_OK;
}

// Convex hull (in the plane, counter-clockwise about the normal) of the points labelled `plane`; hull_xyz gets up
// to cap vertices lying exactly on the plane.
extern "C" int hsk_plane_hull(const float* xyz, size_t n, const int* labels, int plane, const float abcd[4], float* hull_xyz,
                              size_t cap, size_t* n_hull) {
  if (!xyz || !labels || !abcd || !n_hull) return HSK_ERR_ARG;
  const double nn[3] = {abcd[0], abcd[1], abcd[2]};
  // orthonormal basis (e1, e2) of the plane
  double a[3] = {1, 0, 0};
  if (std::fabs(nn[0]) > 0.9) {
    a[0] = 0;
    a[1] = 1;
  }
  double e1[3] = {a[1] * nn[2] - a[2] * nn[1], a[2] * nn[0] - a[0] * nn[2], a[0] * nn[1] - a[1] * nn[0]};
  const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
  for (double& c : e1) c /= l1;
  const double e2[3] = {nn[1] * e1[2] - nn[2] * e1[1], nn[2] * e1[0] - nn[0] * e1[2], nn[0] * e1[1] - nn[1] * e1[0]};
  struct P2 {
    double x, y;
  };
  std::vector<P2> pts;
  for (size_t i = 0; i < n; ++i)
    if (labels[i] == plane) {
      const float* p = xyz + 3 * i;
      pts.push_back({e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2], e2[0] * p[0] + e2[1] * p[1] + e2[2] * p[2]});
    }
  *n_hull = 0;
  if (pts.size() < 3) return HSK_OK;
  std::sort(pts.begin(), pts.end(), [](const P2& u, const P2& v) { return u.x != v.x ? u.x < v.x : u.y < v.y; });
  auto cross = [](const P2& o, const P2& u, const P2& v) { return (u.x - o.x) * (v.y - o.y) - (u.y - o.y) * (v.x - o.x); };
  std::vector<P2> h(2 * pts.size());
  size_t k = 0;
  for (size_t i = 0; i < pts.size(); ++i) {  // Andrew's monotone chain
    while (k >= 2 && cross(h[k - 2], h[k - 1], pts[i]) <= 0) --k;
    h[k++] = pts[i];
  }
  for (size_t i = pts.size() - 1, t = k + 1; i > 0; --i) {
    while (k >= t && cross(h[k - 2], h[k - 1], pts[i - 1]) <= 0) --k;
    h[k++] = pts[i - 1];
  }
  h.resize(k > 1 ? k - 1 : k);
  *n_hull = h.size();
  const double off = -(double)abcd[3];  // points on the plane: x = u e1 + v e2 + off * n
  for (size_t i = 0; i < h.size() && i < cap && hull_xyz; ++i)
    for (int c = 0; c < 3; ++c) hull_xyz[3 * i + c] = (float)(h[i].x * e1[c] + h[i].y * e2[c] + off * nn[c]);
  return HSK_OK;
}

// planes.txt in the format planeEqsFromFile parses (Main.hs:1379-1389): "a b c d" per line, lines separated by \n
extern "C" int hsk_write_planes_txt(const char* path, const float* planes_abcd, int n_planes) {
  if (!path || (!planes_abcd && n_planes)) return HSK_ERR_ARG;
  FILE* f = fopen(path, "w");
  if (!f) return HSK_ERR_STATE;
  for (int i = 0; i < n_planes; ++i)
    fprintf(f, "%.9g %.9g %.9g %.9g%s", planes_abcd[4 * i], planes_abcd[4 * i + 1], planes_abcd[4 * i + 2],
            planes_abcd[4 * i + 3], i + 1 < n_planes ? "\n" : "");
  return fclose(f) == 0 ? HSK_OK : HSK_ERR_STATE;
}

// ------------------------------------------------------------------------------------------------------
// Transforms on the seam back from HouseScan: HouseScan exports each room's placement as a
// row-major, LEFT-multiplicative 4x4 -- as one CSV line for `pcl_transform_point_cloud -matrix`
// (roomProjectionToString, housescan/Main.hs:2271-2284) and as a 4-line .xf file for `plyxform`
// (roomProjectionToXfFormat, Main.hs:2289-2302).  These read/write both forms and apply them to a cloud.
// ------------------------------------------------------------------------------------------------------
extern "C" int hsk_write_xf(const char* path, const float m[16]) {
  if (!path || !m) return HSK_ERR_ARG;
  FILE* f = fopen(path, "w");
  if (!f) return HSK_ERR_STATE;
  for (int r = 0; r < 4; ++r) fprintf(f, "%.9g %.9g %.9g %.9g\n", m[4 * r], m[4 * r + 1], m[4 * r + 2], m[4 * r + 3]);
  return fclose(f) == 0 ? HSK_OK : HSK_ERR_STATE;
}

// accepts the .xf layout (whitespace separated) and the CSV layout (comma separated): 16 numbers, row-major
extern "C" int hsk_read_xf(const char* path, float m[16]) {
  if (!path || !m) return HSK_ERR_ARG;
  FILE* f = fopen(path, "r");
  if (!f) return HSK_ERR_STATE;
  int n = 0;
  double v;
  while (n < 16) {
    int c = fgetc(f);
    if (c == EOF) break;
    if (c == ',' || c == ' ' || c == '\n' || c == '\t' || c == '\r') continue;
    ungetc(c, f);
    if (fscanf(f, "%lf", &v) != 1) break;
    m[n++] = (float)v;
  }
  fclose(f);
  return n == 16 ? HSK_OK : HSK_ERR_STATE;
}

// p' = M p for packed xyz points (in place allowed); w is assumed 1 and the last row (0 0 0 1)
extern "C" int hsk_transform_cloud(const float* xyz, size_t n, const float m[16], float* out) {
  if ((!xyz && n) || !m || (!out && n)) return HSK_ERR_ARG;
  for (size_t i = 0; i < n; ++i) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    out[3 * i] = m[0] * x + m[1] * y + m[2] * z + m[3];
    out[3 * i + 1] = m[4] * x + m[5] * y + m[6] * z + m[7];
    out[3 * i + 2] = m[8] * x + m[9] * y + m[10] * z + m[11];
  }
  return HSK_OK;
}

// ------------------------------------------------------------------------------------------------------
// Recorded depth streams: no OpenNI / .oni here, so the core has its own raw container.
//   bytes 0..3 "HSKD", u32 version = 1, u32 width, u32 height, u32 n_frames, f32 fx, fy, cx, cy, then
//   n_frames x (width*height) little-endian uint16 millimetres, row-major -- the frame layout of
//   takeDepthSnapshot (housescan/HoniHelper.hs:20-36).
// ------------------------------------------------------------------------------------------------------
struct hsk_depth_stream {
  FILE* f;
  uint32_t w, h, n;
  float intr[4];
  bool writing;
};

extern "C" hsk_depth_stream* hsk_stream_create(const char* path, int w, int h, float fx, float fy, float cx, float cy) {
  if (!path || w <= 0 || h <= 0) return nullptr;
  FILE* f = fopen(path, "wb");
  if (!f) return nullptr;
  hsk_depth_stream* s = new hsk_depth_stream{f, (uint32_t)w, (uint32_t)h, 0, {fx, fy, cx, cy}, true};
  const uint32_t hdr[5] = {0x444B5348u /* "HSKD" */, 1u, s->w, s->h, 0u};
  fwrite(hdr, 4, 5, f);
  fwrite(s->intr, 4, 4, f);
  return s;
}

extern "C" hsk_depth_stream* hsk_stream_open(const char* path, int* w, int* h, int* n_frames, float intr[4]) {
  if (!path) return nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return nullptr;
  uint32_t hdr[5];
  float in4[4];
  if (fread(hdr, 4, 5, f) != 5 || fread(in4, 4, 4, f) != 4 || hdr[0] != 0x444B5348u || hdr[1] != 1u) {
    fclose(f);
    return nullptr;
  }
  hsk_depth_stream* s = new hsk_depth_stream{f, hdr[2], hdr[3], hdr[4], {in4[0], in4[1], in4[2], in4[3]}, false};
  if (w) *w = (int)s->w;
  if (h) *h = (int)s->h;
  if (n_frames) *n_frames = (int)s->n;
  if (intr)
    for (int i = 0; i < 4; ++i) intr[i] = in4[i];
  return s;
}

extern "C" int hsk_stream_write(hsk_depth_stream* s, const uint16_t* depth) {
  if (!s || !s->writing || !depth) return HSK_ERR_ARG;
  const size_t px = (size_t)s->w * s->h;
  if (fwrite(depth, 2, px, s->f) != px) return HSK_ERR_STATE;
  s->n += 1;
  return HSK_OK;
}

// frame `index` (0-based) into depth; HSK_ERR_ARG when out of range
extern "C" int hsk_stream_read(hsk_depth_stream* s, int index, uint16_t* depth) {
  if (!s || s->writing || !depth || index < 0 || (uint32_t)index >= s->n) return HSK_ERR_ARG;
  const size_t px = (size_t)s->w * s->h;
  if (fseek(s->f, (long)(36 + (size_t)index * px * 2), SEEK_SET) != 0) return HSK_ERR_STATE;
  return fread(depth, 2, px, s->f) == px ? HSK_OK : HSK_ERR_STATE;
}

extern "C" int hsk_stream_info(const hsk_depth_stream* s, int* w, int* h, int* n_frames, float intr[4]) {
  if (!s) return HSK_ERR_ARG;
  if (w) *w = (int)s->w;
  if (h) *h = (int)s->h;
  if (n_frames) *n_frames = (int)s->n;
  if (intr)
    for (int i = 0; i < 4; ++i) intr[i] = s->intr[i];
  return HSK_OK;
}

extern "C" int hsk_stream_close(hsk_depth_stream* s) {
  if (!s) return HSK_ERR_ARG;
  int rc = HSK_OK;
  if (s->writing) {  // patch the frame count
    if (fseek(s->f, 16, SEEK_SET) != 0 || fwrite(&s->n, 4, 1, s->f) != 1) rc = HSK_ERR_STATE;
  }
  if (fclose(s->f) != 0) rc = HSK_ERR_STATE;
  delete s;
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// mesh products: weld a triangle soup by exact vertex coordinates, write a binary .ply mesh
// ---------------------------------------------------------------------------------------------------------------
#include <cstring>
#include <unordered_map>

namespace {
struct VKey {
  uint32_t a, b, c;
  bool operator==(const VKey& o) const { return a == o.a && b == o.b && c == o.c; }
};
struct VKeyHash {
  size_t operator()(const VKey& k) const {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ k.a;
    h = (h ^ (h >> 29)) * 0xbf58476d1ce4e5b9ull ^ k.b;
    h = (h ^ (h >> 32)) * 0x94d049bb133111ebull ^ k.c;
    return (size_t)(h ^ (h >> 31));
  }
};
// vertices in order of first appearance; -0.0 is folded onto +0.0 so that equal points weld
void weld(const float* tri, size_t n_tri, std::vector<float>& verts, std::vector<int32_t>& idx) {
  std::unordered_map<VKey, int32_t, VKeyHash> seen;
  seen.reserve(n_tri * 2);
  idx.resize(n_tri * 3);
  for (size_t i = 0; i < n_tri * 3; ++i) {
    float p[3] = {tri[3 * i] + 0.0f, tri[3 * i + 1] + 0.0f, tri[3 * i + 2] + 0.0f};
    VKey k;
    std::memcpy(&k, p, 12);
    auto it = seen.find(k);
    if (it == seen.end()) {
      const int32_t id = (int32_t)(verts.size() / 3);
      seen.emplace(k, id);
      verts.insert(verts.end(), p, p + 3);
      idx[i] = id;
    } else {
      idx[i] = it->second;
    }
  }
}
}  // namespace

extern "C" int hsk_weld_triangles(const float* tri_xyz, size_t n_triangles, float* vertices, size_t cap_vertices, size_t* n_vertices,
                                  int32_t* indices) {
  if ((!tri_xyz && n_triangles) || !n_vertices) return HSK_ERR_ARG;
  std::vector<float> verts;
  std::vector<int32_t> idx;
  weld(tri_xyz, n_triangles, verts, idx);
  *n_vertices = verts.size() / 3;
  if (!vertices && !indices) return HSK_OK;
  if (vertices) {
    if (cap_vertices < verts.size() / 3) return HSK_ERR_ARG;
    if (!verts.empty()) std::memcpy(vertices, verts.data(), verts.size() * 4);
  }
  if (indices && !idx.empty()) std::memcpy(indices, idx.data(), idx.size() * 4);
  return HSK_OK;
}

extern "C" int hsk_write_ply_mesh(const char* path, const float* tri_xyz, size_t n_triangles, size_t* n_vertices_out, size_t* n_faces_out) {
  if (!path || (!tri_xyz && n_triangles)) return HSK_ERR_ARG;
  std::vector<float> verts;
  std::vector<int32_t> idx;
  weld(tri_xyz, n_triangles, verts, idx);
  size_t faces = 0;
  for (size_t t = 0; t < n_triangles; ++t)
    if (idx[3 * t] != idx[3 * t + 1] && idx[3 * t + 1] != idx[3 * t + 2] && idx[3 * t] != idx[3 * t + 2]) ++faces;
  FILE* f = fopen(path, "wb");
  if (!f) return HSK_ERR_STATE;
  fprintf(f,
          "ply\nformat binary_little_endian 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\n"
          "element face %zu\nproperty list uchar int vertex_indices\nend_header\n",
          verts.size() / 3, faces);
  bool ok = verts.empty() || fwrite(verts.data(), 4, verts.size(), f) == verts.size();
  for (size_t t = 0; t < n_triangles && ok; ++t) {
    if (idx[3 * t] == idx[3 * t + 1] || idx[3 * t + 1] == idx[3 * t + 2] || idx[3 * t] == idx[3 * t + 2]) continue;
    const unsigned char three = 3;
    ok = fwrite(&three, 1, 1, f) == 1 && fwrite(&idx[3 * t], 4, 3, f) == 3;
  }
  const int rc = fclose(f);
  if (n_vertices_out) *n_vertices_out = verts.size() / 3;
  if (n_faces_out) *n_faces_out = faces;
  return (ok && rc == 0) ? HSK_OK : HSK_ERR_STATE;
}
