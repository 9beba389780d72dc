// kernels_volume.hip -- TSDF volume kernels for gfx950: integrate, raycast (A.6),
// zero-crossing cloud extraction (A.7).  Hand-written for wave64 / 16-B-per-lane HBM access; no MFMA (none
// of these is a contraction).  The volume is an array of (int16 tsdf*32767, int16 weight) pairs, x fastest.
#pragma clang fp contract(off)
#include "hsk_dev.h"
#include "hsk_launch.h"

// ------------------------------------------------------------------------------------------------------
// integrate: each lane owns 4 x-adjacent voxels (one 16-B vector), a wave covers 256 voxels = 1 KiB of a
// row, a block covers 4 consecutive rows, and walks a chunk of z planes.  The depth test comes BEFORE the
// volume access, so only vectors that hold at least one rewritten voxel are ever read or written: HBM
// traffic tracks the algorithmic 8 B x V_upd rather than the 8 B x N^3 sweep.
// ------------------------------------------------------------------------------------------------------
template <bool COUNT_ONLY>
__global__ __launch_bounds__(256) void k_integrate(uint4* __restrict__ vol, const float* __restrict__ scaled,
                                                   const TrackState* __restrict__ st, VolParams vp, int W, int H,
                                                   Intr in, int zchunk, unsigned long long* __restrict__ counter) {
  const int lane = threadIdx.x;
  const int x0 = (blockIdx.x * 64 + lane) * 4;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (!COUNT_ONLY && st->lost) return;
  const bool active = (x0 < vp.X) && (y < vp.Y);
  unsigned long long cnt = 0;
  if (active) {
    const float tx = st->t[0], ty = st->t[1], tz = st->t[2];
    // Rinv = R^T
    const float i00 = st->R[0], i01 = st->R[3], i02 = st->R[6];
    const float i10 = st->R[1], i11 = st->R[4], i12 = st->R[7];
    const float i20 = st->R[2], i21 = st->R[5], i22 = st->R[8];
    const float gy = ((float)y + 0.5f) * vp.cell[1] - ty;
    float ax[4], ay[4], az[4], pn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gx = ((float)(x0 + j) + 0.5f) * vp.cell[0] - tx;
      ax[j] = i00 * gx + i01 * gy;
      ay[j] = i10 * gx + i11 * gy;
      az[j] = i20 * gx + i21 * gy;
      pn[j] = gx * gx + gy * gy;
    }
    const int zbeg = blockIdx.z * zchunk;
    const int zend = min(zbeg + zchunk, vp.nzs);
    const size_t plane_vec = (size_t)vp.X * vp.Y / 4;
    size_t idx = (size_t)zbeg * plane_vec + ((size_t)y * vp.X + x0) / 4;
    for (int zz = zbeg; zz < zend; ++zz, idx += plane_vec) {
      const float gz = ((float)(vp.zs0 + zz) + 0.5f) * vp.cell[2] - tz;
      const float bx = i02 * gz, by = i12 * gz, bz = i22 * gz;
      const float gz2 = gz * gz;
      float F[4];
      unsigned mask = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float camz = az[j] + bz;
        if (camz > 0.0f) {
          const float inv_z = 1.0f / camz;
          const float fu = ((ax[j] + bx) * in.fx) * inv_z + in.cx;
          const float fv = ((ay[j] + by) * in.fy) * inv_z + in.cy;
          int u, v;
          if (hsk_rint_guard(fu, u) && hsk_rint_guard(fv, v) && u >= 0 && v >= 0 && u < W && v < H) {
            const float Ds = scaled[v * W + u];
            const float dist = sqrtf(gz2 + pn[j]);
            const float sdf = Ds - dist;
            if (Ds != 0.0f && sdf >= -vp.tau) {
              const float f = sdf * vp.tau_inv;
              F[j] = f < 1.0f ? f : 1.0f;
              mask |= 1u << j;
            }
          }
        }
      }
      if (mask) {
        if (COUNT_ONLY) {
          cnt += __popc(mask);
        } else {
          uint4 q = vol[idx];
          unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (mask & (1u << j)) {
              const int tp = (int)(short)(w4[j] & 0xffffu);
              const int wp = (int)(short)(w4[j] >> 16);
              const float Fp = (float)tp / 32767.0f;
              const float Wp = (float)wp;
              const float Fn = (Fp * Wp + F[j]) / (Wp + 1.0f);
              int wn = wp + 1;
              wn = wn > HSK_MAX_WEIGHT ? HSK_MAX_WEIGHT : wn;
              int fixed = (int)(Fn * 32767.0f);  // truncation toward zero
              fixed = fixed > HSK_DIVISOR ? HSK_DIVISOR : fixed;
              fixed = fixed < -HSK_DIVISOR ? -HSK_DIVISOR : fixed;
              w4[j] = ((unsigned)fixed & 0xffffu) | ((unsigned)wn << 16);
            }
          }
          vol[idx] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
      }
    }
  }
  if (COUNT_ONLY) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (lane == 0 && cnt) atomicAdd(counter, cnt);
  }
}

void launch_integrate(hipStream_t s, void* vol, const float* scaled, const TrackState* st, const VolParams& vp, int W,
                      int H, Intr in, bool count_only, unsigned long long* counter) {
  const int zchunks = vp.nzs >= 64 ? 8 : 1;
  const int zchunk = (vp.nzs + zchunks - 1) / zchunks;
  dim3 block(64, 4, 1);
  dim3 grid((vp.X / 4 + 63) / 64, (vp.Y + 3) / 4, zchunks);
  if (count_only)
    hipLaunchKernelGGL(k_integrate<true>, grid, block, 0, s, (uint4*)vol, scaled, st, vp, W, H, in, zchunk, counter);
  else
    hipLaunchKernelGGL(k_integrate<false>, grid, block, 0, s, (uint4*)vol, scaled, st, vp, W, H, in, zchunk, counter);
}

// ------------------------------------------------------------------------------------------------------
// raycast (A.6).  One ray per lane; a wave covers an 8x8 pixel tile so that neighbouring rays walk
// neighbouring voxels (L1/L2 locality of the 4-B gathers).  Steps are owned by the slab that contains the
// far sample's z plane; a single-device context owns all of them.
// ------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ int vox_of(float p, float cell) {
  const float q = floorf(p / cell);
  if (!(q >= 0.0f)) return -1;
  if (q > 1.0e6f) return 1000000;
  return (int)q;
}

static __device__ __forceinline__ int raw_at(const short2* __restrict__ vol, const VolParams& vp, int x, int y, int z) {
  const int zz = z - vp.zs0;
  if (zz < 0 || zz >= vp.nzs) return 0;
  return (int)vol[((size_t)zz * vp.Y + y) * vp.X + x].x;
}
static __device__ __forceinline__ float tsdf_at(const short2* __restrict__ vol, const VolParams& vp, int x, int y, int z) {
  return (float)raw_at(vol, vp, x, y, z) / 32767.0f;
}

static __device__ float trilinear(const short2* __restrict__ vol, const VolParams& vp, float px, float py, float pz) {
  int gx = vox_of(px, vp.cell[0]), gy = vox_of(py, vp.cell[1]), gz = vox_of(pz, vp.cell[2]);
  if (gx <= 0 || gx >= vp.X - 1) return HSK_NANF;
  if (gy <= 0 || gy >= vp.Y - 1) return HSK_NANF;
  if (gz <= 0 || gz >= vp.Z - 1) return HSK_NANF;
  if (px < ((float)gx + 0.5f) * vp.cell[0]) gx -= 1;
  if (py < ((float)gy + 0.5f) * vp.cell[1]) gy -= 1;
  if (pz < ((float)gz + 0.5f) * vp.cell[2]) gz -= 1;
  const float a = (px - ((float)gx + 0.5f) * vp.cell[0]) / vp.cell[0];
  const float b = (py - ((float)gy + 0.5f) * vp.cell[1]) / vp.cell[1];
  const float c = (pz - ((float)gz + 0.5f) * vp.cell[2]) / vp.cell[2];
  const float f000 = tsdf_at(vol, vp, gx, gy, gz), f001 = tsdf_at(vol, vp, gx, gy, gz + 1);
  const float f010 = tsdf_at(vol, vp, gx, gy + 1, gz), f011 = tsdf_at(vol, vp, gx, gy + 1, gz + 1);
  const float f100 = tsdf_at(vol, vp, gx + 1, gy, gz), f101 = tsdf_at(vol, vp, gx + 1, gy, gz + 1);
  const float f110 = tsdf_at(vol, vp, gx + 1, gy + 1, gz), f111 = tsdf_at(vol, vp, gx + 1, gy + 1, gz + 1);
  float res = f000 * (1.0f - a) * (1.0f - b) * (1.0f - c);
  res = res + f001 * (1.0f - a) * (1.0f - b) * c;
  res = res + f010 * (1.0f - a) * b * (1.0f - c);
  res = res + f011 * (1.0f - a) * b * c;
  res = res + f100 * a * (1.0f - b) * (1.0f - c);
  res = res + f101 * a * (1.0f - b) * c;
  res = res + f110 * a * b * (1.0f - c);
  res = res + f111 * a * b * c;
  return res;
}

__global__ __launch_bounds__(256) void k_raycast(const short2* __restrict__ vol, const TrackState* __restrict__ st,
                                                 VolParams vp, int W, int H, Intr in, float* __restrict__ vmap,
                                                 float* __restrict__ nmap, int* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int tiles_x = (W + 7) >> 3;
  const int x = (tile % tiles_x) * 8 + (lane & 7);
  const int y = (tile / tiles_x) * 8 + (lane >> 3);
  if (x >= W || y >= H) return;
  if (st->lost) return;
  const size_t P = (size_t)W * H;
  const size_t i = (size_t)y * W + x;
  float vx = HSK_NANF, vy = HSK_NANF, vz = HSK_NANF, nx = HSK_NANF, ny = HSK_NANF, nz = HSK_NANF;
  int key = HSK_KEY_NONE_I;

  const float t0 = st->t[0], t1 = st->t[1], t2 = st->t[2];
  const float rx = ((float)x - in.cx) / in.fx, ry = ((float)y - in.cy) / in.fy;
  float d0 = (st->R[0] * rx + st->R[1] * ry) + st->R[2] * 1.0f;
  float d1 = (st->R[3] * rx + st->R[4] * ry) + st->R[5] * 1.0f;
  float d2 = (st->R[6] * rx + st->R[7] * ry) + st->R[8] * 1.0f;
  const float inv = 1.0f / sqrtf(hsk_dot3(d0, d1, d2, d0, d1, d2));
  d0 = d0 * inv;
  d1 = d1 * inv;
  d2 = d2 * inv;
  if (d0 == 0.0f) d0 = 1e-15f;
  if (d1 == 0.0f) d1 = 1e-15f;
  if (d2 == 0.0f) d2 = 1e-15f;
  const float tmin0 = ((d0 > 0.0f ? 0.0f : vp.size[0]) - t0) / d0, tmax0 = ((d0 > 0.0f ? vp.size[0] : 0.0f) - t0) / d0;
  const float tmin1 = ((d1 > 0.0f ? 0.0f : vp.size[1]) - t1) / d1, tmax1 = ((d1 > 0.0f ? vp.size[1] : 0.0f) - t1) / d1;
  const float tmin2 = ((d2 > 0.0f ? 0.0f : vp.size[2]) - t2) / d2, tmax2 = ((d2 > 0.0f ? vp.size[2] : 0.0f) - t2) / d2;
  float t_start = fmaxf(fmaxf(tmin0, tmin1), tmin2);
  const float t_exit = fminf(fminf(tmax0, tmax1), tmax2);
  t_start = fmaxf(t_start, 0.0f);
  if (t_start < t_exit) {
    const float time_step = vp.tau * 0.8f;
    const float max_time = 3.0f * ((vp.size[0] + vp.size[1]) + vp.size[2]);
    float time_curr = t_start;
    int step = 0;
    for (; time_curr < max_time; time_curr = time_curr + time_step, ++step) {
      const float tn = time_curr + time_step;
      const float pnx = t0 + d0 * tn, pny = t1 + d1 * tn, pnz = t2 + d2 * tn;
      const int gx = vox_of(pnx, vp.cell[0]), gy = vox_of(pny, vp.cell[1]), gz = vox_of(pnz, vp.cell[2]);
      if (gx < 0 || gy < 0 || gz < 0 || gx >= vp.X || gy >= vp.Y || gz >= vp.Z) break;
      if (gz < vp.zo0 || gz >= vp.zo1) continue;
      const float pcx = t0 + d0 * time_curr, pcy = t1 + d1 * time_curr, pcz = t2 + d2 * time_curr;
      const int qx = vox_of(pcx, vp.cell[0]), qy = vox_of(pcy, vp.cell[1]), qz = vox_of(pcz, vp.cell[2]);
      const int cxv = qx < 0 ? 0 : (qx > vp.X - 1 ? vp.X - 1 : qx);
      const int cyv = qy < 0 ? 0 : (qy > vp.Y - 1 ? vp.Y - 1 : qy);
      const int czv = qz < 0 ? 0 : (qz > vp.Z - 1 ? vp.Z - 1 : qz);
      const int raw_prev = raw_at(vol, vp, cxv, cyv, czv);
      const int raw = raw_at(vol, vp, gx, gy, gz);
      if (raw_prev < 0 && raw > 0) {
        key = (step << 1) | 1;
        break;
      }
      if (raw_prev > 0 && raw < 0) {
        key = (step << 1) | 1;
        const float Ftdt = trilinear(vol, vp, pnx, pny, pnz);
        if (!hsk_isnan(Ftdt)) {
          const float Ft = trilinear(vol, vp, pcx, pcy, pcz);
          if (!hsk_isnan(Ft)) {
            const float Ts = time_curr - (time_step * Ft) / (Ftdt - Ft);
            if (Ts >= time_curr - 0.5f * time_step && Ts <= time_curr + 1.5f * time_step) {
              vx = t0 + d0 * Ts;
              vy = t1 + d1 * Ts;
              vz = t2 + d2 * Ts;
              key = (step << 1);
              if (qx > 1 && qy > 1 && qz > 1 && qx < vp.X - 2 && qy < vp.Y - 2 && qz < vp.Z - 2) {
                const float gxn = trilinear(vol, vp, vx + vp.cell[0], vy, vz) - trilinear(vol, vp, vx - vp.cell[0], vy, vz);
                const float gyn = trilinear(vol, vp, vx, vy + vp.cell[1], vz) - trilinear(vol, vp, vx, vy - vp.cell[1], vz);
                const float gzn = trilinear(vol, vp, vx, vy, vz + vp.cell[2]) - trilinear(vol, vp, vx, vy, vz - vp.cell[2]);
                const float ninv = 1.0f / sqrtf(hsk_dot3(gxn, gyn, gzn, gxn, gyn, gzn));
                nx = gxn * ninv;
                ny = gyn * ninv;
                nz = gzn * ninv;
              }
            }
          }
        }
        break;
      }
    }
  }
  vmap[i] = vx;
  vmap[P + i] = vy;
  vmap[2 * P + i] = vz;
  nmap[i] = nx;
  nmap[P + i] = ny;
  nmap[2 * P + i] = nz;
  if (keys) keys[i] = key;
}

void launch_raycast(hipStream_t s, const void* vol, const TrackState* st, const VolParams& vp, int W, int H, Intr in,
                    float* vmap, float* nmap, int* keys) {
  const int tiles = ((W + 7) / 8) * ((H + 7) / 8);
  dim3 block(256);
  dim3 grid((tiles + 3) / 4);
  hipLaunchKernelGGL(k_raycast, grid, block, 0, s, (const short2*)vol, st, vp, W, H, in, vmap, nmap, keys);
}

// ------------------------------------------------------------------------------------------------------
// multi-GPU composite helpers: after the MIN all-reduce of the step keys, a slab keeps its
// maps only where it won; the bit patterns are then SUM-all-reduced as int32 (exact, keeps NaN and -0).
// ------------------------------------------------------------------------------------------------------
__global__ void k_resolve(const int* __restrict__ keys_local, const int* __restrict__ keys_min,
                          const float* __restrict__ vmap, const float* __restrict__ nmap, int* __restrict__ bits, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int kl = keys_local[i], km = keys_min[i];
  const bool mine = (kl == km) && (km != HSK_KEY_NONE_I) && ((km & 1) == 0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    bits[c * P + i] = mine ? __float_as_int(vmap[c * P + i]) : 0;
    bits[(3 + c) * P + i] = mine ? __float_as_int(nmap[c * P + i]) : 0;
  }
}
__global__ void k_adopt(const int* __restrict__ keys_min, const int* __restrict__ bits, float* __restrict__ vmap,
                        float* __restrict__ nmap, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int km = keys_min[i];
  const bool hit = (km != HSK_KEY_NONE_I) && ((km & 1) == 0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    vmap[c * P + i] = hit ? __int_as_float(bits[c * P + i]) : HSK_NANF;
    nmap[c * P + i] = hit ? __int_as_float(bits[(3 + c) * P + i]) : HSK_NANF;
  }
}
void launch_resolve(hipStream_t s, const int* keys_local, const int* keys_min, const float* vmap, const float* nmap,
                    int* bits, int P) {
  hipLaunchKernelGGL(k_resolve, dim3((P + 255) / 256), dim3(256), 0, s, keys_local, keys_min, vmap, nmap, bits, P);
}
void launch_adopt(hipStream_t s, const int* keys_min, const int* bits, float* vmap, float* nmap, int P) {
  hipLaunchKernelGGL(k_adopt, dim3((P + 255) / 256), dim3(256), 0, s, keys_min, bits, vmap, nmap, P);
}

// ------------------------------------------------------------------------------------------------------
// extractCloud (A.7): a wave per (y,z) row; pass 1 counts, an exclusive scan orders the rows, pass 2 writes
// the points in voxel order (deterministic, identical to the sequential restatement).
// ------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ int crossing_count(const short2* __restrict__ vol, const VolParams& vp, int x, int y,
                                                     int z, float* pts /* up to 9 floats or null */) {
  const size_t sx = 1, sy = (size_t)vp.X, sz = (size_t)vp.X * vp.Y;
  const size_t i = ((size_t)(z - vp.zs0) * vp.Y + y) * vp.X + x;
  const short2 c = vol[i];
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
    const short2 nb = vol[i + (k == 0 ? sx : (k == 1 ? sy : sz))];
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

template <bool WRITE>
__global__ __launch_bounds__(256) void k_extract(const short2* __restrict__ vol, VolParams vp,
                                                 unsigned* __restrict__ row_count,
                                                 const unsigned long long* __restrict__ row_offset,
                                                 float* __restrict__ xyz, unsigned long long cap) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nrows = vp.Y * (vp.zo1 - vp.zo0);
  if (row >= nrows) return;
  const int y = row % vp.Y, z = vp.zo0 + row / vp.Y;
  unsigned long long base = WRITE ? row_offset[row] : 0;
  unsigned total = 0;
  for (int xb = 0; xb < vp.X; xb += 64) {
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

// exclusive scan of row counts by one block (rows <= ~1M; not a hot path)
__global__ __launch_bounds__(1024) void k_scan_rows(const unsigned* __restrict__ cnt, unsigned long long* __restrict__ off,
                                                    int n, unsigned long long* __restrict__ total) {
  __shared__ unsigned long long sh[1024];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b = 0; b < n; b += 1024) {
    const int i = b + threadIdx.x;
    const unsigned long long v = i < n ? cnt[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      unsigned long long a = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < n) off[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

void launch_extract(hipStream_t s, const void* vol, const VolParams& vp, unsigned* row_count,
                    unsigned long long* row_offset, unsigned long long* total, float* xyz, unsigned long long cap,
                    int pass) {
  const int nrows = vp.Y * (vp.zo1 - vp.zo0);
  dim3 block(256), grid((nrows + 3) / 4);
  if (pass == 0) {
    hipLaunchKernelGGL(k_extract<false>, grid, block, 0, s, (const short2*)vol, vp, row_count, row_offset, xyz, cap);
    hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, s, row_count, row_offset, nrows, total);
  } else {
    hipLaunchKernelGGL(k_extract<true>, grid, block, 0, s, (const short2*)vol, vp, row_count, row_offset, xyz, cap);
  }
}
