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
