// extract.hip -- read-out kernels for gfx950: zero-crossing cloud extraction, marching tetrahedra and
// marching cubes meshes with their host-built tables.
#pragma clang fp contract(off)
#include "hsk_dev.h"
#include "hsk_launch.h"

// ------------------------------------------------------------------------------------------------------
// extractCloud (A.7): a wave per (y,z) row; pass 1 counts, an exclusive scan orders the rows, pass 2 writes
// the points in voxel order (deterministic, identical to the sequential restatement).
// ------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ int crossing_count(const short2* __restrict__ vol, const VolParams& vp, int x, int y,
                                                     int z, float* pts /* up to 9 floats or null */) {
  const short2 c = vol[hsk_vox_index(vp, x, y, z - vp.zs0)];
  if (c.y == 0 || c.x == HSK_DIVISOR) return 0;
  const float F = (float)c.x / 32767.0f;
  const float V0 = ((float)x + 0.5f) * vp.cell[0], V1 = ((float)y + 0.5f) * vp.cell[1], V2 = ((float)z + 0.5f) * vp.cell[2];
  int n = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int g = k == 0 ? x : (k == 1 ? y : z);
    const int dim = k == 0 ? vp.X : (k == 1 ? vp.Y : vp.Z);
    if (g + 1 >= dim) continue;
    if (k == 2 && (z + 1 - vp.zs0) >= vp.nzs) continue;  // neighbour plane not stored (cannot happen with halo >= 1)
    const short2 nb = vol[hsk_vox_index(vp, x + (k == 0 ? 1 : 0), y + (k == 1 ? 1 : 0), z - vp.zs0 + (k == 2 ? 1 : 0))];
    if (nb.y == 0 || nb.x == HSK_DIVISOR) continue;
    if (!((c.x > 0 && nb.x < 0) || (c.x < 0 && nb.x > 0))) continue;
    if (pts) {
      const float Fn = (float)nb.x / 32767.0f;
      const float cellk = vp.cell[k];
      const float Vk = k == 0 ? V0 : (k == 1 ? V1 : V2);
      const float Vn = Vk + cellk;
      const float d_inv = 1.0f / (fabsf(F) + fabsf(Fn));
      const float pk = (Vk * fabsf(Fn) + Vn * fabsf(F)) * d_inv;
      pts[3 * n + 0] = k == 0 ? pk : V0;
      pts[3 * n + 1] = k == 1 ? pk : V1;
      pts[3 * n + 2] = k == 2 ? pk : V2;
    }
    ++n;
  }
  return n;
}

// A zero crossing, or a cube the level set cuts, needs a NEGATIVE TSDF among the voxels x .. x+1, y .. y+1, z .. z+1.  The
// brick bitfield (integrate: mark_brick_negative; rebuilt on upload) says which bricks have ever held one: a segment of a
// row whose bricks are all clear holds no product -- decided on a few scalar words (the whole field is 4 KiB), without
// touching the volume.  Most of a scanned room is such space: the read-out sweeps follow the surfaces, not the volume
// (round 5; every voxel of the volume was visited three times per cloud before).  Wave-uniform arguments.
// (the row's part of it, once per row: the OR of the brick rows y .. y+1, z .. z+1 as a mask over the x bricks -- up to 64 of
// them; a volume with more bricks along x reports every brick set, i.e. skips nothing)
static __device__ __forceinline__ unsigned long long row_brick_mask(const unsigned* __restrict__ flags, const VolParams& vp, int y, int z) {
  const int bs = vp.bshift, bxn = vp.X >> bs, byn = vp.Y >> bs;
  if (bxn > 64) return ~0ull;
  const int zz0 = z - vp.zs0, zz1 = min(zz0 + 1, vp.nzs - 1);
  const int by0 = y >> bs, by1 = min(y + 1, vp.Y - 1) >> bs;
  const int bz0 = zz0 >> bs, bz1 = zz1 >> bs;
  unsigned long long m = 0ull;
  for (int bz = bz0; bz <= bz1; ++bz)
    for (int by = by0; by <= by1; ++by) {
      const int bit0 = (bz * byn + by) * bxn;  // the brick row's first bit; its bxn bits span at most three words
      // (the words behind the last brick row's own are clamped into the field -- they are super-brick words today, but
      // nothing here depends on what follows the brick bits: whatever is read beyond the row's bxn bits is masked off below)
      const int w0 = bit0 >> 5, sh = bit0 & 31, wlast = hsk_flag_words_total(vp) - 1;
      const unsigned long long lo = (unsigned long long)flags[w0] | ((unsigned long long)flags[min(w0 + 1, wlast)] << 32);
      unsigned long long bits = lo >> sh;
      if (sh != 0 && sh + bxn > 64) bits |= (unsigned long long)flags[min(w0 + 2, wlast)] << (64 - sh);
      m |= bits;
    }
  return bxn == 64 ? m : m & ((1ull << bxn) - 1ull);
}
static __device__ __forceinline__ bool segment_may_hold_negative(unsigned long long row_mask, const VolParams& vp, int xa, int xb) {
  const int bx0 = xa >> vp.bshift, bx1 = min(xb + 1, vp.X - 1) >> vp.bshift;
  if (bx1 >= 64) return true;
  const unsigned long long seg = ((bx1 == 63 ? ~0ull : ((1ull << (bx1 + 1)) - 1ull))) & ~((1ull << bx0) - 1ull);
  return (row_mask & seg) != 0ull;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_extract(const short2* __restrict__ vol, VolParams vp,
                                                 unsigned* __restrict__ row_count,
                                                 const unsigned long long* __restrict__ row_offset,
                                                 float* __restrict__ xyz, unsigned long long cap, const unsigned* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const int nrows = vp.Y * (vp.zo1 - vp.zo0);
  if (row >= nrows) return;
  const int y = row % vp.Y, z = vp.zo0 + row / vp.Y;
  if (WRITE && row_count[row] == 0u) return;  // (the count pass found the row empty)
  const unsigned long long row_mask = row_brick_mask(flags, vp, y, z);
  if (row_mask == 0ull) {
    if (!WRITE && lane == 0) row_count[row] = 0u;
    return;
  }
  unsigned long long base = WRITE ? row_offset[row] : 0;
  unsigned total = 0;
  for (int xb = 0; xb < vp.X; xb += 64) {
    if (!segment_may_hold_negative(row_mask, vp, xb, min(xb + 63, vp.X - 1))) continue;
    const int x = xb + lane;
    float pts[9];
    int n = 0;
    if (x < vp.X) n = crossing_count(vol, vp, x, y, z, WRITE ? pts : nullptr);
    // inclusive wave scan of n
    int scan = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(scan, o, 64);
      if (lane >= o) scan += v;
    }
    const int wave_total = __shfl(scan, 63, 64);
    if (WRITE) {
      unsigned long long at = base + (unsigned long long)(scan - n);
      for (int q = 0; q < n; ++q, ++at)
        if (at < cap) {
          xyz[3 * at] = pts[3 * q];
          xyz[3 * at + 1] = pts[3 * q + 1];
          xyz[3 * at + 2] = pts[3 * q + 2];
        }
      base += wave_total;
    }
    total += wave_total;
  }
  if (!WRITE && lane == 0) row_count[row] = total;
}

// exclusive scan of the row counts (up to a few million rows), three small launches: per block of 1024 rows its sum; the
// scan of those sums and the total (one block); the rows' offsets.  (One block walking all the rows, 1024 at a time with a
// Hillis-Steele scan each, took 0.3 ms of a cloud's 3 ms at 512^3.)
static __device__ __forceinline__ unsigned long long block_scan_1024(unsigned long long v, unsigned long long* sh, unsigned long long* total) {
  // inclusive scan over the block's 1024 threads: inside each wave by shuffles, then across the 16 waves
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long u = __shfl_up(incl, o, 64);
    if (lane >= o) incl += u;
  }
  if (lane == 63) sh[wv] = incl;
  __syncthreads();
  unsigned long long base = 0, all = 0;
  for (int w = 0; w < 16; ++w) {
    const unsigned long long t = sh[w];
    if (w < wv) base += t;
    all += t;
  }
  __syncthreads();
  *total = all;
  return base + incl;
}
__global__ __launch_bounds__(1024) void k_scan_rows_sum(const unsigned* __restrict__ cnt, int n, unsigned long long* __restrict__ bsum) {
  __shared__ unsigned long long sh[16];
  const int i = blockIdx.x * 1024 + threadIdx.x;
  unsigned long long all;
  (void)block_scan_1024(i < n ? cnt[i] : 0u, sh, &all);
  if (threadIdx.x == 0) bsum[blockIdx.x] = all;
}
__global__ __launch_bounds__(1024) void k_scan_rows_top(unsigned long long* __restrict__ bsum, int nb, unsigned long long* __restrict__ total) {
  __shared__ unsigned long long sh[16];
  unsigned long long carry = 0;
  for (int b = 0; b < nb; b += 1024) {  // (block-uniform trip count)
    const int i = b + threadIdx.x;
    const unsigned long long v = i < nb ? bsum[i] : 0;
    unsigned long long all;
    const unsigned long long incl = block_scan_1024(v, sh, &all);
    if (i < nb) bsum[i] = carry + incl - v;
    carry += all;
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(1024) void k_scan_rows_fill(const unsigned* __restrict__ cnt, int n, const unsigned long long* __restrict__ bsum,
                                                         unsigned long long* __restrict__ off) {
  __shared__ unsigned long long sh[16];
  const int i = blockIdx.x * 1024 + threadIdx.x;
  const unsigned long long v = i < n ? cnt[i] : 0u;
  unsigned long long all;
  const unsigned long long incl = block_scan_1024(v, sh, &all);
  if (i < n) off[i] = bsum[blockIdx.x] + incl - v;
}
// (row_offset has room for its n entries and, behind them, the per-block sums: hsk_scan_scratch_entries)
size_t hsk_scan_scratch_entries(int nrows) { return (size_t)nrows + (size_t)(nrows + 1023) / 1024; }
static void launch_scan_rows(hipStream_t s, const unsigned* cnt, unsigned long long* off, int n, unsigned long long* total) {
  const int nb = (n + 1023) / 1024;
  unsigned long long* bsum = off + n;
  hipLaunchKernelGGL(k_scan_rows_sum, dim3(nb), dim3(1024), 0, s, cnt, n, bsum);
  hipLaunchKernelGGL(k_scan_rows_top, dim3(1), dim3(1024), 0, s, bsum, nb, total);
  hipLaunchKernelGGL(k_scan_rows_fill, dim3(nb), dim3(1024), 0, s, cnt, n, (const unsigned long long*)bsum, off);
}

// ------------------------------------------------------------------------------------------------------
// Mesh extraction ("next" row 3): marching tetrahedra over the TSDF, triangle soup in voxel order.
// A cube (x..x+1, y..y+1, z..z+1) is cut into the six Kuhn tetrahedra round its main diagonal (the same cut in
// every cube, so faces of neighbouring cubes agree); corner i sits at offset (i&1, i>>1&1, i>>2&1).  A cube counts
// only when all eight weights are non-zero; a corner is inside when its TSDF is negative.  An edge vertex is
// P = Pa + (Fa / (Fa - Fb)) (Pb - Pa) with a the LOWER corner index, so both cubes that share an edge produce the
// same bits (the mesh can be welded by exact comparison).  Triangles wind so that the normal points to free space.
// ------------------------------------------------------------------------------------------------------
void hsk_build_tet_table(TetTable* tt) {
  static const int tet[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7}, {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};
  for (int t = 0; t < 6; ++t)
    for (int m = 0; m < 16; ++m) {
      int in[4], out[4], ni = 0, no = 0;
      for (int v = 0; v < 4; ++v) {
        if ((m >> v) & 1)
          in[ni++] = tet[t][v];
        else
          out[no++] = tet[t][v];
      }
      int e[2][3][2];
      int nt = 0;
      if (ni == 1 || ni == 3) {
        const int apex = ni == 1 ? in[0] : out[0];
        const int* base = ni == 1 ? out : in;
        for (int q = 0; q < 3; ++q) e[0][q][0] = apex, e[0][q][1] = base[q];
        nt = 1;
      } else if (ni == 2) {
        const int quad[4][2] = {{in[0], out[0]}, {in[0], out[1]}, {in[1], out[1]}, {in[1], out[0]}};
        const int pick[2][3] = {{0, 1, 2}, {0, 2, 3}};
        for (int k = 0; k < 2; ++k)
          for (int q = 0; q < 3; ++q) e[k][q][0] = quad[pick[k][q]][0], e[k][q][1] = quad[pick[k][q]][1];
        nt = 2;
      }
      // orientation: the normal of (p0, p1, p2) (edge midpoints) must point from the inside corners to the outside ones
      double ci[3] = {0, 0, 0}, co[3] = {0, 0, 0};
      for (int v = 0; v < ni; ++v)
        for (int a = 0; a < 3; ++a) ci[a] += ((in[v] >> a) & 1) / (double)(ni ? ni : 1);
      for (int v = 0; v < no; ++v)
        for (int a = 0; a < 3; ++a) co[a] += ((out[v] >> a) & 1) / (double)(no ? no : 1);
      for (int k = 0; k < nt; ++k) {
        double pnt[3][3];
        for (int q = 0; q < 3; ++q)
          for (int a = 0; a < 3; ++a) pnt[q][a] = 0.5 * (((e[k][q][0] >> a) & 1) + ((e[k][q][1] >> a) & 1));
        const double u[3] = {pnt[1][0] - pnt[0][0], pnt[1][1] - pnt[0][1], pnt[1][2] - pnt[0][2]};
        const double w[3] = {pnt[2][0] - pnt[0][0], pnt[2][1] - pnt[0][1], pnt[2][2] - pnt[0][2]};
        const double nrm[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
        const double dir = nrm[0] * (co[0] - ci[0]) + nrm[1] * (co[1] - ci[1]) + nrm[2] * (co[2] - ci[2]);
        if (dir < 0)
          for (int a = 0; a < 2; ++a) {
            const int tmp = e[k][1][a];
            e[k][1][a] = e[k][2][a];
            e[k][2][a] = tmp;
          }
      }
      tt->ntri[t][m] = (unsigned char)nt;
      for (int k = 0; k < 2; ++k)
        for (int q = 0; q < 3; ++q) {
          const int a = k < nt ? e[k][q][0] : 0, b = k < nt ? e[k][q][1] : 0;
          tt->edge[t][m][k][q] = (unsigned char)((a < b ? a : b) | ((a < b ? b : a) << 4));  // low corner first
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Marching cubes (the form PCL's KinFu exports its .ply from, README.md:16-17): one table entry per 8-bit inside mask.
// PCL's 256-case table is not in the reference and cannot be fetched, so the table is GENERATED: on every face of the
// cube the cut edges are joined by segments -- two cut edges: one segment; four (the two diagonal corners inside): two
// segments, each cutting ONE INSIDE corner off, a rule that depends on the face's four signs only, so the two cubes that
// share the face draw the same segments and the surface is closed wherever the cubes are valid.  Every cut edge then has
// exactly two segments: they chain into closed loops, each loop is wound so that its normal points from the inside
// corners to the outside ones and is cut into a fan of triangles from its lowest edge (or the next whose fan keeps out of
// the cube's faces).  820 triangles over the 256
// cases, at most 5 per cube (the classic table's counts).  Vertices as in the tetrahedra form: from the LOWER corner.
// ------------------------------------------------------------------------------------------------------
int hsk_build_cube_table(CubeTable* ct) {
  struct Edge {
    int a, b;  // corners, a < b
  };
  auto code = [](int a, int b) { return a < b ? (a | (b << 4)) : (b | (a << 4)); };
  int worst = 0;
  for (int m = 0; m < 256; ++m) {
    // segments between cut edges, found face by face; link[e][0..1]: the two edges an edge is joined to
    int link[256][2], nlink[256];
    bool cut_edge[256];
    for (int i = 0; i < 256; ++i) nlink[i] = 0, cut_edge[i] = false;
    auto join = [&](int e0, int e1) {
      link[e0][nlink[e0]++] = e1;
      link[e1][nlink[e1]++] = e0;
      cut_edge[e0] = cut_edge[e1] = true;
    };
    for (int ax = 0; ax < 3; ++ax) {
      const int u = ax == 0 ? 1 : 0, v = ax == 2 ? 1 : 2;
      for (int side = 0; side < 2; ++side) {
        int cyc[4];
        const int uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
        for (int i = 0; i < 4; ++i) cyc[i] = (side << ax) | (uv[i][0] << u) | (uv[i][1] << v);
        int fe[4], ncut = 0;
        bool cut[4], in[4];
        for (int i = 0; i < 4; ++i) in[i] = ((m >> cyc[i]) & 1) != 0;
        for (int i = 0; i < 4; ++i) {
          fe[i] = code(cyc[i], cyc[(i + 1) & 3]);
          cut[i] = in[i] != in[(i + 1) & 3];
          ncut += cut[i] ? 1 : 0;
        }
        if (ncut == 2) {
          int e0 = -1, e1 = -1;
          for (int i = 0; i < 4; ++i)
            if (cut[i]) (e0 < 0 ? e0 : e1) = fe[i];
          join(e0, e1);
        } else if (ncut == 4) {
          for (int i = 0; i < 4; ++i)
            if (in[i]) join(fe[(i + 3) & 3], fe[i]);  // the two edges that meet in inside corner i
        }
      }
    }
    int nt = 0;
    bool used[256];
    for (int i = 0; i < 256; ++i) used[i] = false;
    for (int start = 0; start < 256; ++start) {  // (edge codes in ascending order: the loops' order, and each loop's first edge)
      if (!cut_edge[start] || used[start]) continue;
      int loop[12], len = 0, prev = -1, cur = start;
      for (;;) {
        loop[len++] = cur;
        used[cur] = true;
        int next = -1;
        for (int q = 0; q < 2; ++q)
          if (link[cur][q] != prev && !used[link[cur][q]]) {
            next = link[cur][q];
            break;
          }
        if (next < 0) break;
        prev = cur;
        cur = next;
      }
      // winding: Newell normal of the loop of edge midpoints against the summed inside -> outside edge directions
      double mid[12][3], nrm[3] = {0, 0, 0}, dir[3] = {0, 0, 0};
      for (int i = 0; i < len; ++i) {
        const int a = loop[i] & 15, b = loop[i] >> 4;
        const bool a_in = ((m >> a) & 1) != 0;
        for (int k = 0; k < 3; ++k) {
          const double pa = (a >> k) & 1, pb = (b >> k) & 1;
          mid[i][k] = 0.5 * (pa + pb);
          dir[k] += a_in ? pb - pa : pa - pb;
        }
      }
      for (int i = 0; i < len; ++i) {
        const double* p = mid[i];
        const double* q = mid[(i + 1) % len];
        nrm[0] += p[1] * q[2] - p[2] * q[1];
        nrm[1] += p[2] * q[0] - p[0] * q[2];
        nrm[2] += p[0] * q[1] - p[1] * q[0];
      }
      if (nrm[0] * dir[0] + nrm[1] * dir[1] + nrm[2] * dir[2] < 0)
        for (int i = 1, j = len - 1; i < j; ++i, --j) {
          const int t = loop[i];
          loop[i] = loop[j];
          loop[j] = t;
        }
      // the fan's origin: the first edge of the wound loop none of whose diagonals lies IN a face of the cube (both edges on
      // one face: the neighbour across that face could draw the same line, and the welded mesh would use it four times);
      // one of the first three always qualifies
      auto in_one_face = [](int e, int f) {
        for (int k = 0; k < 3; ++k) {
          const int b = ((e & 15) >> k) & 1;
          if ((((e >> 4) >> k) & 1) == b && (((f & 15) >> k) & 1) == b && (((f >> 4) >> k) & 1) == b) return true;
        }
        return false;
      };
      int origin = 0;
      for (int o = 0; o < len; ++o) {
        bool clean = true;
        for (int k = 2; k + 1 < len; ++k) clean = clean && !in_one_face(loop[o], loop[(o + k) % len]);
        if (clean) {
          origin = o;
          break;
        }
      }
      for (int i = 1; i + 1 < len; ++i) {
        if (nt < HSK_MC_MAXT) {
          ct->edge[m][nt][0] = (unsigned char)loop[origin];
          ct->edge[m][nt][1] = (unsigned char)loop[(origin + i) % len];
          ct->edge[m][nt][2] = (unsigned char)loop[(origin + i + 1) % len];
        }
        ++nt;
      }
    }
    worst = nt > worst ? nt : worst;
    ct->ntri[m] = (unsigned char)(nt < HSK_MC_MAXT ? nt : HSK_MC_MAXT);
    for (int t = nt; t < HSK_MC_MAXT; ++t) ct->edge[m][t][0] = ct->edge[m][t][1] = ct->edge[m][t][2] = 0;
  }
  return worst;  // 5: the table's row length (checked by the caller)
}

// triangles of the cube at (x, y, z); when WRITE, stores 9 floats per triangle at tri + 9 * (at + i) while at + i < cap
template <bool WRITE>
static __device__ int cube_triangles(const short2* __restrict__ vol, const VolParams& vp, const TetTable& tt, int x, int y, int z,
                                     float* __restrict__ tri, unsigned long long at, unsigned long long cap) {
  short2 v[8];
  bool ok = true;
  unsigned m8 = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    v[c] = vol[hsk_vox_index(vp, x + (c & 1), y + ((c >> 1) & 1), z + (c >> 2) - vp.zs0)];
    ok = ok && v[c].y != 0;
    m8 |= (v[c].x < 0 ? 1u : 0u) << c;
  }
  if (!ok || m8 == 0u || m8 == 255u) return 0;
  const int tet[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7}, {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};
  int n = 0;
  for (int t = 0; t < 6; ++t) {
    const unsigned m = ((m8 >> tet[t][0]) & 1u) | (((m8 >> tet[t][1]) & 1u) << 1) | (((m8 >> tet[t][2]) & 1u) << 2) |
                       (((m8 >> tet[t][3]) & 1u) << 3);
    const int nt = tt.ntri[t][m];
    if (WRITE) {
      for (int k = 0; k < nt; ++k) {
        const unsigned long long slot = at + (unsigned long long)(n + k);
        if (slot >= cap) continue;
        for (int q = 0; q < 3; ++q) {
          const unsigned code = tt.edge[t][m][k][q];
          const int a = (int)(code & 15u), b = (int)(code >> 4);
          // dynamic corner selection without a scratch array
          short fa = 0, fb = 0;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            fa = c == a ? v[c].x : fa;
            fb = c == b ? v[c].x : fb;
          }
          const float Fa = (float)fa / 32767.0f, Fb = (float)fb / 32767.0f;
          const float w = Fa / (Fa - Fb);
          const int ga[3] = {x + (a & 1), y + ((a >> 1) & 1), z + (a >> 2)};
          const int gb[3] = {x + (b & 1), y + ((b >> 1) & 1), z + (b >> 2)};
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            const float pa = ((float)ga[ax] + 0.5f) * vp.cell[ax];
            const float pb = ((float)gb[ax] + 0.5f) * vp.cell[ax];
            tri[9 * slot + 3 * q + ax] = pa + w * (pb - pa);
          }
        }
      }
    }
    n += nt;
  }
  return n;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_extract_mesh(const short2* __restrict__ vol, VolParams vp, TetTable tt,
                                                      unsigned* __restrict__ row_count,
                                                      const unsigned long long* __restrict__ row_offset,
                                                      float* __restrict__ tri, unsigned long long cap, int z_end, const unsigned* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const int ny = vp.Y - 1;
  const int nrows = ny * (z_end - vp.zo0);
  if (row >= nrows) return;
  const int y = row % ny, z = vp.zo0 + row / ny;
  if (WRITE && row_count[row] == 0u) return;  // (the count pass found the row empty)
  const unsigned long long row_mask = row_brick_mask(flags, vp, y, z);
  if (row_mask == 0ull) {
    if (!WRITE && lane == 0) row_count[row] = 0u;
    return;
  }
  unsigned long long base = WRITE ? row_offset[row] : 0;
  unsigned total = 0;
  for (int xb = 0; xb < vp.X - 1; xb += 64) {
    if (!segment_may_hold_negative(row_mask, vp, xb, min(xb + 63, vp.X - 2))) continue;
    const int x = xb + lane;
    const int n = x < vp.X - 1 ? cube_triangles<false>(vol, vp, tt, x, y, z, nullptr, 0, 0) : 0;
    int scan = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(scan, o, 64);
      if (lane >= o) scan += u;
    }
    const int wave_total = __shfl(scan, 63, 64);
    if (WRITE) {
      if (n) cube_triangles<true>(vol, vp, tt, x, y, z, tri, base + (unsigned long long)(scan - n), cap);
      base += wave_total;
    }
    total += wave_total;
  }
  if (!WRITE && lane == 0) row_count[row] = total;
}

// cubes whose base plane this context owns and whose upper plane is stored
int hsk_mesh_z_end(const VolParams& vp) {
  int z_end = vp.zo1;
  if (z_end > vp.zs0 + vp.nzs - 1) z_end = vp.zs0 + vp.nzs - 1;
  if (z_end > vp.Z - 1) z_end = vp.Z - 1;
  return z_end > vp.zo0 ? z_end : vp.zo0;
}

void launch_extract_mesh(hipStream_t s, const void* vol, const VolParams& vp, const TetTable& tt, unsigned* row_count,
                         unsigned long long* row_offset, unsigned long long* total, float* tri, unsigned long long cap, int pass, const unsigned* flags) {
  const int z_end = hsk_mesh_z_end(vp);
  const int nrows = (vp.Y - 1) * (z_end - vp.zo0);
  if (nrows <= 0) {
    if (pass == 0) (void)hipMemsetAsync(total, 0, 8, s);
    return;
  }
  dim3 block(256), grid((nrows + 3) / 4);
  if (pass == 0) {
    hipLaunchKernelGGL(k_extract_mesh<false>, grid, block, 0, s, (const short2*)vol, vp, tt, row_count, row_offset, tri, cap, z_end, flags);
    launch_scan_rows(s, row_count, row_offset, nrows, total);
  } else {
    hipLaunchKernelGGL(k_extract_mesh<true>, grid, block, 0, s, (const short2*)vol, vp, tt, row_count, row_offset, tri, cap, z_end, flags);
  }
}

// ... and the marching-cubes form: the cube's triangles straight from the table (in device memory: 4 KiB)
template <bool WRITE>
static __device__ int cube_triangles_mc(const short2* __restrict__ vol, const VolParams& vp, const CubeTable* __restrict__ ct, int x, int y,
                                        int z, float* __restrict__ tri, unsigned long long at, unsigned long long cap) {
  short2 v[8];
  bool ok = true;
  unsigned m8 = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    v[c] = vol[hsk_vox_index(vp, x + (c & 1), y + ((c >> 1) & 1), z + (c >> 2) - vp.zs0)];
    ok = ok && v[c].y != 0;
    m8 |= (v[c].x < 0 ? 1u : 0u) << c;
  }
  if (!ok || m8 == 0u || m8 == 255u) return 0;
  const int nt = ct->ntri[m8];
  if (WRITE) {
    for (int k = 0; k < nt; ++k) {
      const unsigned long long slot = at + (unsigned long long)k;
      if (slot >= cap) continue;
      for (int q = 0; q < 3; ++q) {
        const unsigned code = ct->edge[m8][k][q];
        const int a = (int)(code & 15u), b = (int)(code >> 4);
        short fa = 0, fb = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          fa = c == a ? v[c].x : fa;
          fb = c == b ? v[c].x : fb;
        }
        const float Fa = (float)fa / 32767.0f, Fb = (float)fb / 32767.0f;
        const float w = Fa / (Fa - Fb);
        const int ga[3] = {x + (a & 1), y + ((a >> 1) & 1), z + (a >> 2)};
        const int gb[3] = {x + (b & 1), y + ((b >> 1) & 1), z + (b >> 2)};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const float pa = ((float)ga[ax] + 0.5f) * vp.cell[ax];
          const float pb = ((float)gb[ax] + 0.5f) * vp.cell[ax];
          tri[9 * slot + 3 * q + ax] = pa + w * (pb - pa);
        }
      }
    }
  }
  return nt;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_extract_mesh_mc(const short2* __restrict__ vol, VolParams vp, const CubeTable* __restrict__ ct,
                                                         unsigned* __restrict__ row_count,
                                                         const unsigned long long* __restrict__ row_offset,
                                                         float* __restrict__ tri, unsigned long long cap, int z_end, const unsigned* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const int ny = vp.Y - 1;
  const int nrows = ny * (z_end - vp.zo0);
  if (row >= nrows) return;
  const int y = row % ny, z = vp.zo0 + row / ny;
  if (WRITE && row_count[row] == 0u) return;  // (the count pass found the row empty)
  const unsigned long long row_mask = row_brick_mask(flags, vp, y, z);
  if (row_mask == 0ull) {
    if (!WRITE && lane == 0) row_count[row] = 0u;
    return;
  }
  unsigned long long base = WRITE ? row_offset[row] : 0;
  unsigned total = 0;
  for (int xb = 0; xb < vp.X - 1; xb += 64) {
    if (!segment_may_hold_negative(row_mask, vp, xb, min(xb + 63, vp.X - 2))) continue;
    const int x = xb + lane;
    const int n = x < vp.X - 1 ? cube_triangles_mc<false>(vol, vp, ct, x, y, z, nullptr, 0, 0) : 0;
    int scan = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(scan, o, 64);
      if (lane >= o) scan += u;
    }
    const int wave_total = __shfl(scan, 63, 64);
    if (WRITE) {
      if (n) cube_triangles_mc<true>(vol, vp, ct, x, y, z, tri, base + (unsigned long long)(scan - n), cap);
      base += wave_total;
    }
    total += wave_total;
  }
  if (!WRITE && lane == 0) row_count[row] = total;
}

void launch_extract_mesh_mc(hipStream_t s, const void* vol, const VolParams& vp, const CubeTable* ct_dev, unsigned* row_count,
                            unsigned long long* row_offset, unsigned long long* total, float* tri, unsigned long long cap, int pass, const unsigned* flags) {
  const int z_end = hsk_mesh_z_end(vp);
  const int nrows = (vp.Y - 1) * (z_end - vp.zo0);
  if (nrows <= 0) {
    if (pass == 0) (void)hipMemsetAsync(total, 0, 8, s);
    return;
  }
  const dim3 grid((unsigned)((nrows + 3) / 4));
  if (pass == 0) {
    hipLaunchKernelGGL(k_extract_mesh_mc<false>, grid, dim3(256), 0, s, (const short2*)vol, vp, ct_dev, row_count, (const unsigned long long*)nullptr,
                       (float*)nullptr, 0ull, z_end, flags);
    launch_scan_rows(s, row_count, row_offset, nrows, total);
  } else {
    hipLaunchKernelGGL(k_extract_mesh_mc<true>, grid, dim3(256), 0, s, (const short2*)vol, vp, ct_dev, row_count, row_offset, tri, cap, z_end, flags);
  }
}

void launch_extract(hipStream_t s, const void* vol, const VolParams& vp, unsigned* row_count,
                    unsigned long long* row_offset, unsigned long long* total, float* xyz, unsigned long long cap,
                    int pass, const unsigned* flags) {
  const int nrows = vp.Y * (vp.zo1 - vp.zo0);
  dim3 block(256), grid((nrows + 3) / 4);
  if (pass == 0) {
    hipLaunchKernelGGL(k_extract<false>, grid, block, 0, s, (const short2*)vol, vp, row_count, row_offset, xyz, cap, flags);
    launch_scan_rows(s, row_count, row_offset, nrows, total);
  } else {
    hipLaunchKernelGGL(k_extract<true>, grid, block, 0, s, (const short2*)vol, vp, row_count, row_offset, xyz, cap, flags);
  }
}

// The code object of this file is loaded when one of its kernels is first used (deferred loading): 0.7 ms that the first
// product of a process would otherwise pay on whatever thread asks for it.  hsk_prepare_readout asks here instead.
int extract_warm() {
  hipFuncAttributes a;
  hipError_t e = hipFuncGetAttributes(&a, (const void*)k_extract<false>);
  if (e == hipSuccess) e = hipFuncGetAttributes(&a, (const void*)k_extract<true>);
  if (e == hipSuccess) e = hipFuncGetAttributes(&a, (const void*)k_extract_mesh_mc<false>);
  if (e == hipSuccess) e = hipFuncGetAttributes(&a, (const void*)k_extract_mesh_mc<true>);
  if (e == hipSuccess) e = hipFuncGetAttributes(&a, (const void*)k_extract_mesh<false>);
  if (e == hipSuccess) e = hipFuncGetAttributes(&a, (const void*)k_extract_mesh<true>);
  if (e == hipSuccess) e = hipFuncGetAttributes(&a, (const void*)k_scan_rows_sum);
  return (int)e;
}

