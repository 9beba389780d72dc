// kernels_image.hip -- image-space kernels for gfx950: bilateral + scaleDepth, pyrDown, vertex/normal maps
// (SURVEY.md A.3/A.4), map resize / transform (A.2/A.3), and the projective-association point-to-plane ICP
// with its wavefront reduction and on-device 6x6 solve (A.5).
#pragma clang fp contract(off)
#include "hsk_dev.h"
#ifdef HSK_ICP_TIMING
// timing build (tools/icp_timing.sh): s_memrealtime (100 MHz) stamps of every block of every iteration; defined before
// hsk_icp_dev.h is first seen (through hsk_launch.h), so that the solve step's inner stamps exist too
__device__ unsigned long long g_icp_times[20 * 256 * 10];
extern "C" int hsk_debug_icp_times(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_icp_times), (size_t)n * 8);
}
#define ICP_STAMP(k) do { if (threadIdx.x == 0 && g_icp_iter < 20 && blockIdx.x < 256) g_icp_times[(g_icp_iter * 256 + blockIdx.x) * 10 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#include "hsk_launch.h"

// ------------------------------------------------------------------------------------------------------
// bilateral (13x13) fused with scaleDepth.  16x16 pixel tile per block, 28x28 depth tile + both weight
// tables staged in LDS; every pixel sums its taps in the fixed dy-major / dx-minor order of the spec.
// ------------------------------------------------------------------------------------------------------
#define BIL_R 6
#define BIL_T 16
#define BIL_S (BIL_T + 2 * BIL_R)
#define BIL_OUT (1u << 20)  // a tap outside the image, in the tile's units of a quarter millimetre

struct BilateralWs {
  float w[13 * 13];  // spatial weights: tap-position constants, passed by value so they sit in scalar registers
};
__global__ __launch_bounds__(256) void k_bilateral_scale(const unsigned short* __restrict__ src, int W, int H, Intr in,
                                                         BilateralWs ws,
                                                         const float* __restrict__ wc_tab,
                                                         unsigned short* __restrict__ dst, float* __restrict__ scaled,
                                                         float* __restrict__ tmax, float* __restrict__ tmin, float2* __restrict__ ftab,
                                                         float2* __restrict__ qtab, int fw, int fh, unsigned short* __restrict__ vmask16) {
  // The tile holds 4 x depth: the tap's |difference| then IS the byte offset of its range weight (one v_sad_u32, one
  // v_min_u32, no shift), and sum1 comes out scaled by exactly 4: a power of two passes through the rounding of every
  // normal product and sum, and in the subnormal range (weights down to exp(-145) are in the table) an integer depth times
  // a weight, and any sum of such, is a multiple of 2^-149 and exact either way (binary32 denormals are on: the kernel
  // descriptors' float_denorm_mode_32 = 3).  A tap outside the image is BIL_OUT: its difference from any 16-bit depth is
  // beyond the table, its weight +0, and +0 terms leave both sums as they were -- no test per tap.  Six vector
  // instructions a tap (v_sad_u32, v_min_u32, v_cvt, two v_mul, one v_pk_add for both sums) where there were thirteen and
  // a lane-mask branch.
  __shared__ unsigned tile[BIL_S][BIL_S + 1];
  __shared__ float shx[4], shn[4];
  __shared__ float sct[BIL_T][BIL_T + 1];  // the block's scaled depths (0 outside the image): the finer tile tables come from here
  __shared__ float2 q4[4][4];
  __shared__ float wc[513];  // wc[512] = 0: every difference of 512 and more
  const int tid = threadIdx.y * BIL_T + threadIdx.x;
  const int bx = blockIdx.x * BIL_T, by = blockIdx.y * BIL_T;
  for (int i = tid; i < BIL_S * BIL_S; i += 256) {
    const int ly = i / BIL_S, lx = i % BIL_S;
    const int gx = bx + lx - BIL_R, gy = by + ly - BIL_R;
    // (the windows' upper clip is exclusive of the image's last column and row -- upstream's loop bounds,
    // cx < min(x - 6 + 13, W - 1): those pixels are in no window, not even their own)
    tile[ly][lx] = (gx >= 0 && gy >= 0 && gx < W - 1 && gy < H - 1) ? 4u * (unsigned)src[gy * W + gx] : BIL_OUT;
  }
  for (int i = tid; i < 513; i += 256) wc[i] = i < 512 ? wc_tab[i] : 0.0f;
  __syncthreads();
  const int x = bx + threadIdx.x, y = by + threadIdx.y;
  const bool inside = x < W && y < H;
  const int value = inside ? (int)src[y * W + x] : 0;
  // scaleDepth (A.4), and this 16x16 tile's (max, min) of it for the integrate kernel's group classification:
  // a pixel outside the image or without depth makes the tile minimum 0 ("not all valid")
  float sc = 0.0f;
  if (inside) {
    const float xl = ((float)x - in.cx) / in.fx;
    const float yl = ((float)y - in.cy) / in.fy;
    const float lambda = sqrtf((xl * xl + yl * yl) + 1.0f);
    sc = ((float)value * lambda) / 1000.0f;
    scaled[y * W + x] = sc;
  }
  sct[threadIdx.y][threadIdx.x] = sc;
  {
    // the validity mask (one bit per pixel, 1 = no depth or outside the image): the wave's four rows of 16 pixels are the
    // four halfwords of its ballot
    const unsigned long long inv = __ballot(sc == 0.0f);
    if ((tid & 63) == 0) {
      const int pitch16 = 2 * hsk_mask_pitch32(W);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = by + (tid >> 6) * 4 + r;
        if (yy < H) vmask16[(size_t)yy * pitch16 + blockIdx.x] = (unsigned short)(inv >> (16 * r));
      }
    }
  }
  {
    float mx = sc, mn = sc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      mn = fminf(mn, __shfl_xor(mn, o, 64));
    }
    if ((tid & 63) == 0) {
      shx[tid >> 6] = mx;
      shn[tid >> 6] = mn;
    }
    __syncthreads();
    if (tid == 0) {
      const int tw = gridDim.x;
      tmax[blockIdx.y * tw + blockIdx.x] = fmaxf(fmaxf(shx[0], shx[1]), fmaxf(shx[2], shx[3]));
      tmin[blockIdx.y * tw + blockIdx.x] = fminf(fminf(shn[0], shn[1]), fminf(shn[2], shn[3]));
    }
    // ... and the undilated 4-px and 8-px tables of pass A's pixel-box level (round 5: a launch of their own before, one
    // THREAD per 8-px tile walking 64 floats at a stride of 8 -- 15 us; here the tile is already in the block): the block's
    // sixteen 4-px tiles by sixteen threads, its four 8-px tiles from those (same values, same order-free max / min)
    // (.y: the minimum over the pixels WITH depth, negated when a pixel of the tile has none -- hsk_dev.h, the tile tables)
    if (tid < 16) {
      const int qx = tid & 3, qy = tid >> 2;
      float mx = 0.0f, mn = 1e30f;
      bool hole = false;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
          const float v = sct[qy * 4 + dy][qx * 4 + dx];
          mx = fmaxf(mx, v);
          mn = v != 0.0f ? fminf(mn, v) : mn;
          hole = hole | (v == 0.0f);
        }
      const float2 e = make_float2(mx, hole ? -mn : mn);
      q4[qy][qx] = e;
      const int gx = blockIdx.x * 4 + qx, gy = blockIdx.y * 4 + qy;
      if (gx < 2 * fw && gy < 2 * fh) qtab[(size_t)gy * (2 * fw) + gx] = e;
    }
    __syncthreads();
    if (tid < 4) {
      const int ex = tid & 1, ey = tid >> 1;
      const float2 a = q4[2 * ey][2 * ex], b = q4[2 * ey][2 * ex + 1], c = q4[2 * ey + 1][2 * ex], d = q4[2 * ey + 1][2 * ex + 1];
      const int gx = blockIdx.x * 2 + ex, gy = blockIdx.y * 2 + ey;
      const float mn = fminf(fminf(fabsf(a.y), fabsf(b.y)), fminf(fabsf(c.y), fabsf(d.y)));
      const bool hole = (a.y < 0.0f) | (b.y < 0.0f) | (c.y < 0.0f) | (d.y < 0.0f);
      if (gx < fw && gy < fh) ftab[(size_t)gy * fw + gx] = make_float2(fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)), hole ? -mn : mn);
    }
  }
  if (!inside) return;
  if (value == 0) {
    dst[y * W + x] = 0;
    return;
  }
  float sum1 = 0.0f, sum2 = 0.0f;  // sum1: 4 x the specification's
  const unsigned value4 = 4u * (unsigned)value;
  // (a row of taps per trip: unrolled over all 169 the loop is ONE basic block, the scheduler requests every tap first
  // and the kernel needs 340 registers)
#pragma unroll 1
  for (int dy = 0; dy < 13; ++dy) {
#pragma unroll
    for (int dx = 0; dx < 13; ++dx) {
      const unsigned tmp4 = tile[threadIdx.y + dy][threadIdx.x + dx];
      unsigned ad;  // 4 |value - tmp|
      asm("v_sad_u32 %0, %1, %2, 0" : "=v"(ad) : "v"(value4), "v"(tmp4));
      const unsigned off = min(ad, 2048u);  // 4 x 512 from there on
      const float wcv = *(const float*)((const char*)wc + off);
      const float w = ws.w[dy * 13 + dx] * wcv;
      sum1 = sum1 + (float)tmp4 * w;
      sum2 = sum2 + w;
    }
  }
  sum1 = sum1 * 0.25f;
  int res = sum2 > 0.0f ? __float2int_rn(sum1 / sum2) : 0;  // (0 / 0: a last-column / last-row pixel whose window holds no weight)
  res = res < 0 ? 0 : (res > 32767 ? 32767 : res);
  dst[y * W + x] = (unsigned short)res;
}

// tiles: raw tile maxima then minima (tw*th floats each), as launch_tile_max lays them out; behind them (and the dilated
// table) the raw 8-px and 4-px tables, as launch_tile_fine lays them out
void launch_bilateral_scale(hipStream_t s, const uint16_t* src, int W, int H, Intr in, const float* ws, const float* wc,
                            uint16_t* dst, float* scaled, float* tiles) {
  dim3 block(BIL_T, BIL_T), grid((W + BIL_T - 1) / BIL_T, (H + BIL_T - 1) / BIL_T);
  BilateralWs wsv;
  for (int i = 0; i < 169; ++i) wsv.w[i] = ws[i];  // ws: HOST pointer to the 13x13 table
  const int fw = (W + 7) / 8, fh = (H + 7) / 8;
  float2* ftab = (float2*)(tiles + 4 * grid.x * grid.y);
  float2* qtab = ftab + (size_t)fw * fh;
  hipLaunchKernelGGL(k_bilateral_scale, grid, block, 0, s, src, W, H, in, wsv, wc, dst, scaled, tiles,
                     tiles + grid.x * grid.y, ftab, qtab, fw, fh, (unsigned short*)(tiles + hsk_tiles_mask_offset(W, H)));
}

// scaleDepth alone (stage-level integrate entry point)
__global__ void k_scale_depth(const unsigned short* __restrict__ src, int W, int H, Intr in, float* __restrict__ scaled) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) return;
  const float xl = ((float)x - in.cx) / in.fx;
  const float yl = ((float)y - in.cy) / in.fy;
  const float lambda = sqrtf((xl * xl + yl * yl) + 1.0f);
  scaled[y * W + x] = ((float)src[y * W + x] * lambda) / 1000.0f;
}
void launch_scale_depth(hipStream_t s, const uint16_t* src, int W, int H, Intr in, float* scaled) {
  dim3 block(64, 4), grid((W + 63) / 64, (H + 3) / 4);
  hipLaunchKernelGGL(k_scale_depth, grid, block, 0, s, src, W, H, in, scaled);
}

// ------------------------------------------------------------------------------------------------------
// pyrDown (A.3): integer 5x5 gated mean
// ------------------------------------------------------------------------------------------------------
__global__ void k_pyrdown(const unsigned short* __restrict__ src, int W, int H, unsigned short* __restrict__ dst) {
  const int w2 = W >> 1, h2 = H >> 1;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w2 || y >= h2) return;
  const int center = src[(2 * y) * W + 2 * x];
  const int y0 = max(2 * y - 2, 0), y1 = min(2 * y + 2, H - 2);  // (upper clip exclusive of the last row / column, as the bilateral's)
  const int x0 = max(2 * x - 2, 0), x1 = min(2 * x + 2, W - 2);
  int sum = 0, count = 0;
  for (int cy = y0; cy <= y1; ++cy)
    for (int cx = x0; cx <= x1; ++cx) {
      const int val = src[cy * W + cx];
      const int d = abs(val - center);
      if (d < 90) {
        sum += val;
        ++count;
      }
    }
  dst[y * w2 + x] = (unsigned short)(sum / count);
}
void launch_pyrdown(hipStream_t s, const uint16_t* src, int W, int H, uint16_t* dst) {
  dim3 block(64, 4), grid((W / 2 + 63) / 64, (H / 2 + 3) / 4);
  hipLaunchKernelGGL(k_pyrdown, grid, block, 0, s, src, W, H, dst);
}

// ------------------------------------------------------------------------------------------------------
// vertex + normal maps (A.3), fused: the two forward neighbours are re-projected instead of re-read
// ------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ bool vertex_of(const unsigned short* __restrict__ d, int W, int u, int v, float cx,
                                                 float cy, float fx_inv, float fy_inv, float& X, float& Y, float& Z) {
  const float z = (float)d[v * W + u] / 1000.0f;
  if (z != 0.0f) {
    X = (z * ((float)u - cx)) * fx_inv;
    Y = (z * ((float)v - cy)) * fy_inv;
    Z = z;
    return true;
  }
  X = Y = Z = HSK_NANF;
  return false;
}

struct PyramidArgs {
  const unsigned short* depth[HSK_NLEVELS];
  float* vmap[HSK_NLEVELS];
  float* nmap[HSK_NLEVELS];
  int W[HSK_NLEVELS], H[HSK_NLEVELS], first_block[HSK_NLEVELS + 1];
  Intr in[HSK_NLEVELS];
};
// one launch for the three levels: a block belongs to the level whose block range contains it
__global__ void k_vmap_nmap(PyramidArgs a) {
  int l = 0;
  if ((int)blockIdx.x >= a.first_block[1]) l = 1;
  if ((int)blockIdx.x >= a.first_block[2]) l = 2;
  const unsigned short* __restrict__ depth = a.depth[l];
  float* __restrict__ vmap = a.vmap[l];
  float* __restrict__ nmap = a.nmap[l];
  const int W = a.W[l], H = a.H[l];
  const Intr in = a.in[l];
  const int bw = (W + 63) / 64, b = blockIdx.x - a.first_block[l];
  const int u = (b % bw) * 64 + threadIdx.x, v = (b / bw) * 4 + threadIdx.y;
  if (u >= W || v >= H) return;
  const size_t P = (size_t)W * H, i = (size_t)v * W + u;
  const float fx_inv = 1.0f / in.fx, fy_inv = 1.0f / in.fy;
  float x0, y0, z0;
  const bool ok0 = vertex_of(depth, W, u, v, in.cx, in.cy, fx_inv, fy_inv, x0, y0, z0);
  vmap[i] = x0;
  vmap[P + i] = y0;
  vmap[2 * P + i] = z0;
  float n0 = HSK_NANF, n1 = HSK_NANF, n2 = HSK_NANF;
  if (ok0 && u < W - 1 && v < H - 1) {
    float x1, y1, z1, x2, y2, z2;
    const bool ok1 = vertex_of(depth, W, u + 1, v, in.cx, in.cy, fx_inv, fy_inv, x1, y1, z1);
    const bool ok2 = vertex_of(depth, W, u, v + 1, in.cx, in.cy, fx_inv, fy_inv, x2, y2, z2);
    if (ok1 && ok2) {
      const float ax = x1 - x0, ay = y1 - y0, az = z1 - z0;
      const float bx = x2 - x0, by = y2 - y0, bz = z2 - z0;
      const float r0 = ay * bz - az * by;
      const float r1 = az * bx - ax * bz;
      const float r2 = ax * by - ay * bx;
      const float inv = 1.0f / sqrtf(hsk_dot3(r0, r1, r2, r0, r1, r2));
      n0 = r0 * inv;
      n1 = r1 * inv;
      n2 = r2 * inv;
    }
  }
  nmap[i] = n0;
  nmap[P + i] = n1;
  nmap[2 * P + i] = n2;
}
void launch_vmap_nmap_pyramid(hipStream_t s, uint16_t* const* depth, const ImgLevel* lv, float* const* vmap,
                              float* const* nmap) {
  PyramidArgs a;
  int nb = 0;
  for (int l = 0; l < HSK_NLEVELS; ++l) {
    a.depth[l] = depth[l];
    a.vmap[l] = vmap[l];
    a.nmap[l] = nmap[l];
    a.W[l] = lv[l].W;
    a.H[l] = lv[l].H;
    a.in[l] = lv[l].in;
    a.first_block[l] = nb;
    nb += ((lv[l].W + 63) / 64) * ((lv[l].H + 3) / 4);
  }
  a.first_block[HSK_NLEVELS] = nb;
  hipLaunchKernelGGL(k_vmap_nmap, dim3(nb), dim3(64, 4), 0, s, a);
}

// pyrDown x 2 and the vertex / normal maps of all three levels in ONE launch (round 5; they were three: two pyrDown
// launches of 7 us each and the maps).  A block owns a T x T tile of level 0 = T/2 squared of level 1 = T/4 squared of level 2
// and keeps what they need in LDS: level 2 with its +1 neighbours for the normals is T/4 + 1 squared, whose 5 x 5 windows
// reach T/2 + 5 squared of level 1, whose windows reach T + 13 squared of level 0 (the overlap between blocks is re-read
// from L2, the overlapping pyrDown pixels are computed again: integer arithmetic on the same inputs).  The arithmetic is
// k_pyrdown's and k_vmap_nmap's, on the same values: bit for bit the same images and maps.  T = 16: 1200 blocks for a
// 640 x 480 frame (with T = 32, 300 blocks of four dependent phases each, the launch took 15.8 us).
#ifndef PM_T0
#define PM_T0 16
#endif
#define PM_T1 (PM_T0 / 2)
#define PM_T2 (PM_T0 / 4)
#define PM_S0 (PM_T0 + 13)
#define PM_S1 (PM_T1 + 5)
#define PM_S2 (PM_T2 + 1)
static __device__ __forceinline__ bool pm_vertex(int dmm, int u, int v, float cx, float cy, float fx_inv, float fy_inv, float& X, float& Y, float& Z) {
  const float z = (float)dmm / 1000.0f;
  if (z != 0.0f) {
    X = (z * ((float)u - cx)) * fx_inv;
    Y = (z * ((float)v - cy)) * fy_inv;
    Z = z;
    return true;
  }
  X = Y = Z = HSK_NANF;
  return false;
}
// dst(x, y) of pyrDown from a level held in LDS: src[(sy - oy) * pitch + (sx - ox)] = source pixel (sx, sy); W, H = source size
static __device__ __forceinline__ int pm_pyrdown(const unsigned short* src, int pitch, int ox, int oy, int W, int H, int x, int y) {
  const int center = src[(2 * y - oy) * pitch + (2 * x - ox)];
  const int y0 = max(2 * y - 2, 0), y1 = min(2 * y + 2, H - 2);  // (upper clip exclusive of the last row / column, as the bilateral's)
  const int x0 = max(2 * x - 2, 0), x1 = min(2 * x + 2, W - 2);
  int sum = 0, count = 0;
  for (int cy = y0; cy <= y1; ++cy)
    for (int cx = x0; cx <= x1; ++cx) {
      const int val = src[(cy - oy) * pitch + (cx - ox)];
      const int d = abs(val - center);
      if (d < 90) {
        sum += val;
        ++count;
      }
    }
  return sum / count;
}
static __device__ __forceinline__ void pm_maps(const unsigned short* lv, int pitch, int ox, int oy, int W, int H, const Intr& in, int u, int v,
                                               float* __restrict__ vmap, float* __restrict__ nmap) {
  const size_t P = (size_t)W * H, i = (size_t)v * W + u;
  const float fx_inv = 1.0f / in.fx, fy_inv = 1.0f / in.fy;
  float x0, y0, z0;
  const bool ok0 = pm_vertex(lv[(v - oy) * pitch + (u - ox)], u, v, in.cx, in.cy, fx_inv, fy_inv, x0, y0, z0);
  vmap[i] = x0;
  vmap[P + i] = y0;
  vmap[2 * P + i] = z0;
  float n0 = HSK_NANF, n1 = HSK_NANF, n2 = HSK_NANF;
  if (ok0 && u < W - 1 && v < H - 1) {
    float x1, y1, z1, x2, y2, z2;
    const bool ok1 = pm_vertex(lv[(v - oy) * pitch + (u + 1 - ox)], u + 1, v, in.cx, in.cy, fx_inv, fy_inv, x1, y1, z1);
    const bool ok2 = pm_vertex(lv[(v + 1 - oy) * pitch + (u - ox)], u, v + 1, in.cx, in.cy, fx_inv, fy_inv, x2, y2, z2);
    if (ok1 && ok2) {
      const float ax = x1 - x0, ay = y1 - y0, az = z1 - z0;
      const float bx = x2 - x0, by = y2 - y0, bz = z2 - z0;
      const float r0 = ay * bz - az * by;
      const float r1 = az * bx - ax * bz;
      const float r2 = ax * by - ay * bx;
      const float inv = 1.0f / sqrtf(hsk_dot3(r0, r1, r2, r0, r1, r2));
      n0 = r0 * inv;
      n1 = r1 * inv;
      n2 = r2 * inv;
    }
  }
  nmap[i] = n0;
  nmap[P + i] = n1;
  nmap[2 * P + i] = n2;
}
__global__ __launch_bounds__(256) void k_pyramid_maps(PyramidArgs a, unsigned short* __restrict__ d1, unsigned short* __restrict__ d2) {
  __shared__ unsigned short l0[PM_S0 * PM_S0], l1[PM_S1 * PM_S1], l2[PM_S2 * PM_S2];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  const int W0 = a.W[0], H0 = a.H[0], W1 = a.W[1], H1 = a.H[1], W2 = a.W[2], H2 = a.H[2];
  const int bx = blockIdx.x, by = blockIdx.y;
  // level 0: pixels [T bx - 6, T bx + T + 6] squared (0 outside the image: never used by a pixel inside)
  const int ox0 = PM_T0 * bx - 6, oy0 = PM_T0 * by - 6;
  for (int i = tid; i < PM_S0 * PM_S0; i += 256) {
    const int ly = i / PM_S0, lx = i - ly * PM_S0;
    const int gx = ox0 + lx, gy = oy0 + ly;
    l0[i] = (gx >= 0 && gy >= 0 && gx < W0 && gy < H0) ? a.depth[0][(size_t)gy * W0 + gx] : (unsigned short)0;
  }
  __syncthreads();
  // level 1: [T/2 bx - 2, T/2 bx + T/2 + 2] squared
  const int ox1 = PM_T1 * bx - 2, oy1 = PM_T1 * by - 2;
  for (int i = tid; i < PM_S1 * PM_S1; i += 256) {
    const int ly = i / PM_S1, lx = i - ly * PM_S1;
    const int x = ox1 + lx, y = oy1 + ly;
    int v = 0;
    if (x >= 0 && y >= 0 && x < W1 && y < H1) {
      v = pm_pyrdown(l0, PM_S0, ox0, oy0, W0, H0, x, y);
      if (lx >= 2 && lx < 2 + PM_T1 && ly >= 2 && ly < 2 + PM_T1) d1[(size_t)y * W1 + x] = (unsigned short)v;  // (the block's own tile)
    }
    l1[i] = (unsigned short)v;
  }
  __syncthreads();
  // level 2: [T/4 bx, T/4 bx + T/4] squared
  const int ox2 = PM_T2 * bx, oy2 = PM_T2 * by;
  if (tid < PM_S2 * PM_S2) {
    const int ly = tid / PM_S2, lx = tid - ly * PM_S2;
    const int x = ox2 + lx, y = oy2 + ly;
    int v = 0;
    if (x < W2 && y < H2) {
      v = pm_pyrdown(l1, PM_S1, ox1, oy1, W1, H1, x, y);
      if (lx < PM_T2 && ly < PM_T2) d2[(size_t)y * W2 + x] = (unsigned short)v;
    }
    l2[tid] = (unsigned short)v;
  }
  __syncthreads();
  // the maps: T x T pixels of level 0, a quarter as many of level 1, a sixteenth of level 2
  for (int i = tid; i < PM_T0 * PM_T0; i += 256) {
    const int u = PM_T0 * bx + (i % PM_T0), v = PM_T0 * by + (i / PM_T0);
    if (u < W0 && v < H0) pm_maps(l0, PM_S0, ox0, oy0, W0, H0, a.in[0], u, v, a.vmap[0], a.nmap[0]);
  }
  for (int i = tid; i < PM_T1 * PM_T1; i += 256) {
    const int u = PM_T1 * bx + (i % PM_T1), v = PM_T1 * by + (i / PM_T1);
    if (u < W1 && v < H1) pm_maps(l1, PM_S1, ox1, oy1, W1, H1, a.in[1], u, v, a.vmap[1], a.nmap[1]);
  }
  if (tid < PM_T2 * PM_T2) {
    const int u = PM_T2 * bx + (tid % PM_T2), v = PM_T2 * by + (tid / PM_T2);
    if (u < W2 && v < H2) pm_maps(l2, PM_S2, ox2, oy2, W2, H2, a.in[2], u, v, a.vmap[2], a.nmap[2]);
  }
}
void launch_pyramid_maps(hipStream_t s, uint16_t* const* depth, const ImgLevel* lv, float* const* vmap, float* const* nmap) {
  static_assert(HSK_NLEVELS == 3, "k_pyramid_maps holds three levels");
  PyramidArgs a;
  for (int l = 0; l < HSK_NLEVELS; ++l) {
    a.depth[l] = depth[l];
    a.vmap[l] = vmap[l];
    a.nmap[l] = nmap[l];
    a.W[l] = lv[l].W;
    a.H[l] = lv[l].H;
    a.in[l] = lv[l].in;
    a.first_block[l] = 0;
  }
  a.first_block[HSK_NLEVELS] = 0;
  hipLaunchKernelGGL(k_pyramid_maps, dim3((lv[0].W + PM_T0 - 1) / PM_T0, (lv[0].H + PM_T0 - 1) / PM_T0), dim3(64, 4), 0, s, a, depth[1], depth[2]);
}

// tranformMaps (A.2, first frame): v_g = R v + t, n_g = R n, pose taken from the device state
__global__ void k_transform_maps(const float* __restrict__ vs, const float* __restrict__ ns, int P,
                                 const TrackState* __restrict__ st, float* __restrict__ vd, float* __restrict__ nd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float* R = st->R;
  const float vx = vs[i], vy = vs[P + i], vz = vs[2 * P + i];
  float ox = HSK_NANF, oy = HSK_NANF, oz = HSK_NANF;
  if (!hsk_isnan(vx)) {
    ox = ((R[0] * vx + R[1] * vy) + R[2] * vz) + st->t[0];
    oy = ((R[3] * vx + R[4] * vy) + R[5] * vz) + st->t[1];
    oz = ((R[6] * vx + R[7] * vy) + R[8] * vz) + st->t[2];
  }
  vd[i] = ox;
  vd[P + i] = oy;
  vd[2 * P + i] = oz;
  const float nx = ns[i], ny = ns[P + i], nz = ns[2 * P + i];
  float qx = HSK_NANF, qy = HSK_NANF, qz = HSK_NANF;
  if (!hsk_isnan(nx)) {
    qx = (R[0] * nx + R[1] * ny) + R[2] * nz;
    qy = (R[3] * nx + R[4] * ny) + R[5] * nz;
    qz = (R[6] * nx + R[7] * ny) + R[8] * nz;
  }
  nd[i] = qx;
  nd[P + i] = qy;
  nd[2 * P + i] = qz;
}
void launch_transform_maps(hipStream_t s, const float* vs, const float* ns, int P, const TrackState* st, float* vd,
                           float* nd) {
  hipLaunchKernelGGL(k_transform_maps, dim3((P + 255) / 256), dim3(256), 0, s, vs, ns, P, st, vd, nd);
}

// resizeVMap + resizeNMap (A.3): 2x2 mean, NaN if any tap is NaN; normals renormalised
static __device__ __forceinline__ void resize_tap(const float* __restrict__ src, size_t P, int W, int x, int y, bool normalize,
                                                  float& a, float& b, float& c) {
  const size_t i00 = (size_t)(2 * y) * W + 2 * x, i01 = i00 + 1, i10 = i00 + W, i11 = i10 + 1;
  a = b = c = HSK_NANF;
  if (!(hsk_isnan(src[i00]) || hsk_isnan(src[i01]) || hsk_isnan(src[i10]) || hsk_isnan(src[i11]))) {
    a = (((src[i00] + src[i01]) + src[i10]) + src[i11]) / 4.0f;
    b = (((src[P + i00] + src[P + i01]) + src[P + i10]) + src[P + i11]) / 4.0f;
    c = (((src[2 * P + i00] + src[2 * P + i01]) + src[2 * P + i10]) + src[2 * P + i11]) / 4.0f;
    if (normalize) {
      const float inv = 1.0f / sqrtf(hsk_dot3(a, b, c, a, b, c));
      a = a * inv;
      b = b * inv;
      c = c * inv;
    }
  }
}
// level 0 -> level 1 and level 2 in one launch: blocks [0, nb1) write level 1; the rest write level 2 and
// recompute the four level-1 values they average (same arithmetic, so the same bits as reading them back)
__global__ void k_resize_maps2(const float* __restrict__ v0, const float* __restrict__ n0, int W, int H,
                               float* __restrict__ v1, float* __restrict__ n1, float* __restrict__ v2,
                               float* __restrict__ n2, const TrackState* __restrict__ st, int nb1, RingOut ring) {
  if (ring.slots && blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) {
    // last kernel of a pipelined slab frame: report the tracker state into the host ring (see k_raycast)
    const unsigned n = *ring.seq;
    *ring.seq = n + 1u;
    TrackState* dst = ring.slots + ring.slot_fifo[n % HSK_RING_FIFO];
    const int* src_w = (const int*)st;
    int* dst_w = (int*)dst;
    for (unsigned i = 0; i < (unsigned)(offsetof(TrackState, ring_mark) / 4); ++i) dst_w[i] = src_w[i];
    __threadfence_system();
    __hip_atomic_store(&dst->pose_mark, (n + 1u) | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&dst->ring_mark, (n + 1u) | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (st->lost) return;
  const int w1 = W >> 1, h1 = H >> 1, w2 = W >> 2, h2 = H >> 2;
  const size_t P0 = (size_t)W * H, P1 = (size_t)w1 * h1, P2 = (size_t)w2 * h2;
  if ((int)blockIdx.x < nb1) {
    const int bw = (w1 + 63) / 64;
    const int x = (blockIdx.x % bw) * 64 + threadIdx.x, y = (blockIdx.x / bw) * 4 + threadIdx.y;
    if (x >= w1 || y >= h1) return;
    const size_t o = (size_t)y * w1 + x;
    float a, b, c;
    resize_tap(v0, P0, W, x, y, false, a, b, c);
    v1[o] = a; v1[P1 + o] = b; v1[2 * P1 + o] = c;
    resize_tap(n0, P0, W, x, y, true, a, b, c);
    n1[o] = a; n1[P1 + o] = b; n1[2 * P1 + o] = c;
    return;
  }
  const int bw = (w2 + 63) / 64, bi = blockIdx.x - nb1;
  const int x = (bi % bw) * 64 + threadIdx.x, y = (bi / bw) * 4 + threadIdx.y;
  if (x >= w2 || y >= h2) return;
  const size_t o = (size_t)y * w2 + x;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const float* src = m == 0 ? v0 : n0;
    float* dst = m == 0 ? v2 : n2;
    float t[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) resize_tap(src, P0, W, 2 * x + (q & 1), 2 * y + (q >> 1), m == 1, t[q][0], t[q][1], t[q][2]);
    float a = HSK_NANF, b = HSK_NANF, c = HSK_NANF;
    if (!(hsk_isnan(t[0][0]) || hsk_isnan(t[1][0]) || hsk_isnan(t[2][0]) || hsk_isnan(t[3][0]))) {
      a = (((t[0][0] + t[1][0]) + t[2][0]) + t[3][0]) / 4.0f;
      b = (((t[0][1] + t[1][1]) + t[2][1]) + t[3][1]) / 4.0f;
      c = (((t[0][2] + t[1][2]) + t[2][2]) + t[3][2]) / 4.0f;
      if (m == 1) {
        const float inv = 1.0f / sqrtf(hsk_dot3(a, b, c, a, b, c));
        a = a * inv;
        b = b * inv;
        c = c * inv;
      }
    }
    dst[o] = a;
    dst[P2 + o] = b;
    dst[2 * P2 + o] = c;
  }
}
void launch_resize_maps2(hipStream_t s, const float* v0, const float* n0, int W, int H, float* v1, float* n1, float* v2,
                         float* n2, const TrackState* st, const RingOut* ring) {
  const int nb1 = ((W / 2 + 63) / 64) * ((H / 2 + 3) / 4), nb2 = ((W / 4 + 63) / 64) * ((H / 4 + 3) / 4);
  const RingOut quiet = {nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(k_resize_maps2, dim3(nb1 + nb2), dim3(64, 4), 0, s, v0, n0, W, H, v1, n1, v2, n2, st, nb1,
                     ring ? *ring : quiet);
}

// The END of a z-slab frame in one launch (round 6; it was k_adopt, then k_resize_maps2: 15 us of a multi-GPU frame's critical
// path): the composite of the exchange -- the MIN key of every pixel and, where that key is a hit, the
// winner's vertex / normal bits -- becomes the model maps of all three levels, and the tracker state is reported into the
// host ring.  Blocks [0, nb0) write level 0 (k_adopt's selection), [nb0, nb0 + nb1) level 1, the rest level 2 -- the upper
// levels straight from the composite, with the same selection and k_resize_maps2's arithmetic, operand for operand: the bits
// of adopting first and resizing afterwards, without the launch boundary between them and without re-reading level 0.
struct CompositeIn {
  const int* keys_min;
  const int* bits;
  size_t P;
};
static __device__ __forceinline__ bool comp_hit(const CompositeIn& C, size_t i) {
  const int km = C.keys_min[i];
  return (km != HSK_KEY_NONE_I) && ((km & 1) == 0);
}
// resize_tap with level 0 read through the composite: m = 0 the vertex map, 1 the normal map
static __device__ __forceinline__ void comp_tap(const CompositeIn& C, int m, int W, int x, int y, float& a, float& b, float& c) {
  const size_t i00 = (size_t)(2 * y) * W + 2 * x, i01 = i00 + 1, i10 = i00 + W, i11 = i10 + 1;
  const size_t idx[4] = {i00, i01, i10, i11};
  float t[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool hit = comp_hit(C, idx[q]);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) t[q][ch] = hit ? __int_as_float(C.bits[(size_t)(3 * m + ch) * C.P + idx[q]]) : HSK_NANF;
  }
  a = b = c = HSK_NANF;
  if (!(hsk_isnan(t[0][0]) || hsk_isnan(t[1][0]) || hsk_isnan(t[2][0]) || hsk_isnan(t[3][0]))) {
    a = (((t[0][0] + t[1][0]) + t[2][0]) + t[3][0]) / 4.0f;
    b = (((t[0][1] + t[1][1]) + t[2][1]) + t[3][1]) / 4.0f;
    c = (((t[0][2] + t[1][2]) + t[2][2]) + t[3][2]) / 4.0f;
    if (m == 1) {
      const float inv = 1.0f / sqrtf(hsk_dot3(a, b, c, a, b, c));
      a = a * inv;
      b = b * inv;
      c = c * inv;
    }
  }
}
__global__ void k_adopt_pyramid(CompositeIn C, int W, int H, float* __restrict__ v0, float* __restrict__ n0, float* __restrict__ v1,
                                float* __restrict__ n1, float* __restrict__ v2, float* __restrict__ n2,
                                const TrackState* __restrict__ st, int nb0, int nb1, RingOut ring) {
  if (ring.slots && blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) {
    // last kernel of a pipelined slab frame: report the tracker state into the host ring (see k_raycast)
    const unsigned n = *ring.seq;
    *ring.seq = n + 1u;
    TrackState* dst = ring.slots + ring.slot_fifo[n % HSK_RING_FIFO];
    const int* src_w = (const int*)st;
    int* dst_w = (int*)dst;
    for (unsigned i = 0; i < (unsigned)(offsetof(TrackState, ring_mark) / 4); ++i) dst_w[i] = src_w[i];
    __threadfence_system();
    __hip_atomic_store(&dst->pose_mark, (n + 1u) | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&dst->ring_mark, (n + 1u) | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int w1 = W >> 1, h1 = H >> 1, w2 = W >> 2, h2 = H >> 2;
  const size_t P0 = (size_t)W * H, P1 = (size_t)w1 * h1, P2 = (size_t)w2 * h2;
  if ((int)blockIdx.x < nb0) {  // level 0: k_adopt (whatever the frame's verdict, as before)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x;
    if (i >= P0) return;
    const bool hit = comp_hit(C, i);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      v0[ch * P0 + i] = hit ? __int_as_float(C.bits[(size_t)ch * P0 + i]) : HSK_NANF;
      n0[ch * P0 + i] = hit ? __int_as_float(C.bits[(size_t)(3 + ch) * P0 + i]) : HSK_NANF;
    }
    return;
  }
  if (st->lost) return;
  if ((int)blockIdx.x < nb0 + nb1) {
    const int bw = (w1 + 63) / 64, bi = blockIdx.x - nb0;
    const int x = (bi % bw) * 64 + threadIdx.x, y = (bi / bw) * 4 + threadIdx.y;
    if (x >= w1 || y >= h1) return;
    const size_t o = (size_t)y * w1 + x;
    float a, b, c;
    comp_tap(C, 0, W, x, y, a, b, c);
    v1[o] = a; v1[P1 + o] = b; v1[2 * P1 + o] = c;
    comp_tap(C, 1, W, x, y, a, b, c);
    n1[o] = a; n1[P1 + o] = b; n1[2 * P1 + o] = c;
    return;
  }
  const int bw = (w2 + 63) / 64, bi = blockIdx.x - nb0 - nb1;
  const int x = (bi % bw) * 64 + threadIdx.x, y = (bi / bw) * 4 + threadIdx.y;
  if (x >= w2 || y >= h2) return;
  const size_t o = (size_t)y * w2 + x;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float* dst = m == 0 ? v2 : n2;
    float t[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) comp_tap(C, m, W, 2 * x + (q & 1), 2 * y + (q >> 1), t[q][0], t[q][1], t[q][2]);
    float a = HSK_NANF, b = HSK_NANF, c = HSK_NANF;
    if (!(hsk_isnan(t[0][0]) || hsk_isnan(t[1][0]) || hsk_isnan(t[2][0]) || hsk_isnan(t[3][0]))) {
      a = (((t[0][0] + t[1][0]) + t[2][0]) + t[3][0]) / 4.0f;
      b = (((t[0][1] + t[1][1]) + t[2][1]) + t[3][1]) / 4.0f;
      c = (((t[0][2] + t[1][2]) + t[2][2]) + t[3][2]) / 4.0f;
      if (m == 1) {
        const float inv = 1.0f / sqrtf(hsk_dot3(a, b, c, a, b, c));
        a = a * inv;
        b = b * inv;
        c = c * inv;
      }
    }
    dst[o] = a;
    dst[P2 + o] = b;
    dst[2 * P2 + o] = c;
  }
}
void launch_adopt_pyramid(hipStream_t s, const int* keys_min, const int* bits, int W, int H, float* v0, float* n0, float* v1, float* n1,
                          float* v2, float* n2, const TrackState* st, const RingOut* ring) {
  const size_t P0 = (size_t)W * H;
  const int nb0 = (int)((P0 + 255) / 256), nb1 = ((W / 2 + 63) / 64) * ((H / 2 + 3) / 4), nb2 = ((W / 4 + 63) / 64) * ((H / 4 + 3) / 4);
  const RingOut quiet = {nullptr, nullptr, nullptr};
  const CompositeIn C = {keys_min, bits, P0};
  hipLaunchKernelGGL(k_adopt_pyramid, dim3(nb0 + nb1 + nb2), dim3(64, 4), 0, s, C, W, H, v0, n0, v1, n1, v2, n2, st, nb0, nb1, ring ? *ring : quiet);
}

// ------------------------------------------------------------------------------------------------------
// ICP (A.5).  Per pixel: transform, project into the previous camera, gate, build the 7-vector row.  The
// 27 products are formed in binary64 (exact) and snapped to multiples of 2^-26, which makes every partial
// sum exact: the wave64 shuffle tree, the cross-wave LDS step, the cross-block step and (multi-GPU) the
// RCCL all-reduce all give the same bits as a sequential sum.
// ------------------------------------------------------------------------------------------------------
// ICP_PX pixels per lane (template): loads batched so that the dependent chain is 2 memory round trips
#ifndef ICP_BLOCK
#define ICP_BLOCK 256
#endif
#define ICP_SH_ROWS ((ICP_BLOCK / 64) > 8 ? (ICP_BLOCK / 64) : 8)  // LDS rows: one per wave, at least the 8 slices of the shard reduction
#ifndef ICP_PX_FINE
#define ICP_PX_FINE 5  // pixels per lane at the finest level: 640x480 / (256 x 5) = 240 blocks, one per CU (with 4 the 300
#endif                  // blocks gave 44 CUs a second one and every fine iteration waited for them: 178 -> 169 us of ICP)
#ifndef ICP_PX_MID
#define ICP_PX_MID 2   // 320x240 / (256 x 2) = 150 blocks
#endif

// Exact accumulation.  The spec sums quant26(p) = rint(p * 2^26) * 2^-26 over pixels, p the binary64 product of
// two binary32 row entries.  Scaling one factor by 2^26 first is exact (power of two), so the kernel adds the
// integer-valued rint(a' * b) and leaves the common 2^-26 to the very end: every partial sum is an integer below
// 2^53, hence exact, hence independent of the order of the additions (wave tree, LDS, blocks, GPUs).
#define ICP_SCALE 67108864.0          // 2^26
#define ICP_UNSCALE (1.0 / 67108864.0)

// DPP move of a 64-bit value (two 32-bit halves)
template <int CTRL>
static __device__ __forceinline__ double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double swap16_add_f64(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// wave64 sums of 27 values at once by a halving butterfly: at every step the two partner lanes split the
// remaining values between them, so 16 + 8 + 4 + 2 + 1 + 1 exchanges replace 27 x 6.  On return lane l holds the
// wave total of value (l >> 1) & 31 for l < 64 (values 27..31 are padding); lanes 2k and 2k+1 both hold value k.
// The first two steps (24 of the 32 exchanges) are lane swaps, the last four DPP moves inside a row of 16 lanes;
// the step of distance 4 pairs lane l with l ^ 7 (row_half_mirror): any partner with the opposite bit 2 and the same
// higher bits will do, because the sums are exact (multiples of 2^-26) and so independent of the order of addition.
static __device__ __forceinline__ double wave_sum27(const double* acc, int lane) {
  double v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = i < 27 ? acc[i] : 0.0;
  // distance 32: lanes < 32 keep values 0..15, lanes >= 32 keep 16..31
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = swap32_add_f64(v[i], v[i + 16]);
  // distance 16: even rows keep values i, odd rows i + 8
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = swap16_add_f64(v[i], v[i + 8]);
  {
    const bool up = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double keep = up ? v[i + 4] : v[i], send = up ? v[i] : v[i + 4];
      v[i] = keep + dpp_mov_f64<0x128>(send);  // row_ror:8 == lane ^ 8
    }
  }
  {
    const bool up = lane & 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double keep = up ? v[i + 2] : v[i], send = up ? v[i] : v[i + 2];
      v[i] = keep + dpp_mov_f64<0x141>(send);  // row_half_mirror: lane ^ 7
    }
  }
  {
    const bool up = lane & 2;
    const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
    v[0] = keep + dpp_mov_f64<0x4E>(send);  // quad_perm [2,3,0,1] == lane ^ 2
  }
  v[0] += dpp_mov_f64<0xB1>(v[0]);  // quad_perm [1,0,3,2] == lane ^ 1
  return v[0];
}
// value index held by `lane` after wave_sum27: bit 5 -> +16, bit 4 -> +8, bit 3 -> +4, bit 2 -> +2, bit 1 -> +1
static __device__ __forceinline__ int wave_sum27_index(int lane) {
  return ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
}

// Current-level inputs of one lane (ICP_PX pixels): loaded first, so that the loads overlap whatever comes before
// the pose is known (the solve of the previous iteration in the fused kernel).
template <int ICP_PX>
struct IcpLaneIn {
  float nc[ICP_PX][3], vc[ICP_PX][3];
  bool ok[ICP_PX];
};

template <int ICP_PX>
static __device__ __forceinline__ void icp_load_current(const float* __restrict__ vcur, const float* __restrict__ ncur,
                                                        int W, int H, int row0, int row1, IcpLaneIn<ICP_PX>& L) {
  const int P4 = W * H * 4;  // bytes of one map plane
  const __amdgpu_buffer_rsrc_t nbuf = hsk_buf(ncur), vbuf = hsk_buf(vcur);
  const int npx = (row1 - row0) * W;
  const int base = blockIdx.x * (ICP_BLOCK * ICP_PX) + threadIdx.x;
#pragma unroll
  for (int q = 0; q < ICP_PX; ++q) {
    const int li = base + q * ICP_BLOCK;
    L.ok[q] = li < npx;
    const unsigned i4 = (unsigned)(row0 * W + (L.ok[q] ? li : 0)) * 4u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      L.nc[q][c] = hsk_buf_load_f32(nbuf, i4, (unsigned)(c * P4));
      L.vc[q][c] = hsk_buf_load_f32(vbuf, i4, (unsigned)(c * P4));
    }
  }
}

// Per pixel: transform, project into the previous camera, gate, build the 7-vector row, add the 27 scaled products.
template <int ICP_PX>
static __device__ __forceinline__ void icp_accumulate_pixels(IcpLaneIn<ICP_PX>& L, const float* __restrict__ vprev,
                                                             const float* __restrict__ nprev, int W, int H, const Intr& in,
                                                             const float* R, const float* tt, const float* Rp,
                                                             const float* tp, float dist_thresh, float angle_thresh,
                                                             double* acc) {
  const int P4 = W * H * 4;  // bytes of one map plane
  const __amdgpu_buffer_rsrc_t nbuf = hsk_buf(nprev), vbuf = hsk_buf(vprev);
  const float t0 = tt[0], t1 = tt[1], t2 = tt[2];
  const float p0 = tp[0], p1 = tp[1], p2 = tp[2];
  auto& nc = L.nc;
  auto& vc = L.vc;
  auto& ok = L.ok;
  // phase B: transform + project, then all model-map gathers in flight together
  float g[ICP_PX][3], np_[ICP_PX][3], vp_[ICP_PX][3];
#pragma unroll
  for (int q = 0; q < ICP_PX; ++q) {
    g[q][0] = ((R[0] * vc[q][0] + R[1] * vc[q][1]) + R[2] * vc[q][2]) + t0;
    g[q][1] = ((R[3] * vc[q][0] + R[4] * vc[q][1]) + R[5] * vc[q][2]) + t1;
    g[q][2] = ((R[6] * vc[q][0] + R[7] * vc[q][1]) + R[8] * vc[q][2]) + t2;
    const float dx = g[q][0] - p0, dy = g[q][1] - p1, dz = g[q][2] - p2;
    const float cpx = (Rp[0] * dx + Rp[3] * dy) + Rp[6] * dz;  // Rprev^T * d
    const float cpy = (Rp[1] * dx + Rp[4] * dy) + Rp[7] * dz;
    const float cpz = (Rp[2] * dx + Rp[5] * dy) + Rp[8] * dz;
    const float fu = (cpx * in.fx) / cpz + in.cx;
    const float fv = (cpy * in.fy) / cpz + in.cy;
    // (every test is evaluated for every lane and joined with &: a chain of && becomes a lane-mask branch per link, and
    // the wave that runs this is alone on its SIMD -- its scalar instructions are not hidden behind anybody's)
    const bool inr = (fu > -1.0e6f) & (fu < 1.0e6f) & (fv > -1.0e6f) & (fv < 1.0e6f);  // hsk_rint_guard of both
    const int u = __float2int_rn(inr ? fu : 0.0f), v = __float2int_rn(inr ? fv : 0.0f);
    ok[q] = ok[q] & !hsk_isnan(nc[q][0]) & (cpz > 0.0f) & inr & (u >= 0) & (v >= 0) & (u < W) & (v < H);
    // (buffer loads: the pixel's byte offset in ONE register for all six planes, the plane's offset scalar -- one shift per
    // pixel where six 64-bit address sums were)
    const unsigned j4 = ok[q] ? (unsigned)(v * W + u) * 4u : 0u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      np_[q][c] = hsk_buf_load_f32(nbuf, j4, (unsigned)(c * P4));
      vp_[q][c] = hsk_buf_load_f32(vbuf, j4, (unsigned)(c * P4));
    }
  }
  // phase C: gates, the 7-vector row, 27 scaled products
#pragma unroll
  for (int q = 0; q < ICP_PX; ++q) {
    bool valid = ok[q] & !hsk_isnan(np_[q][0]);
    const float ex = vp_[q][0] - g[q][0], ey = vp_[q][1] - g[q][1], ez = vp_[q][2] - g[q][2];
    // sqrtf(d2) <= dist_thresh of the spec, as d2 <= (largest float whose correctly rounded root is <= the threshold);
    // the launcher converts the thresholds (icp_gate_limits), sqrtf being monotone
    valid = valid & (hsk_dot3(ex, ey, ez, ex, ey, ez) <= dist_thresh);
    const float ngx = (R[0] * nc[q][0] + R[1] * nc[q][1]) + R[2] * nc[q][2];
    const float ngy = (R[3] * nc[q][0] + R[4] * nc[q][1]) + R[5] * nc[q][2];
    const float ngz = (R[6] * nc[q][0] + R[7] * nc[q][1]) + R[8] * nc[q][2];
    const float c0 = ngy * np_[q][2] - ngz * np_[q][1];
    const float c1 = ngz * np_[q][0] - ngx * np_[q][2];
    const float c2 = ngx * np_[q][1] - ngy * np_[q][0];
    valid = valid & (hsk_dot3(c0, c1, c2, c0, c1, c2) < angle_thresh);  // sqrtf(.) < angle_thresh, same conversion
    {
      // Branch-free: a rejected pixel contributes a row of zeros (products exactly +-0, sums unchanged).  With the
      // accumulation inside `if (valid)` the compiler copied the 27 accumulator pairs at every join of the divergent
      // branch (416 v_mov in the 5-pixel kernel).
      float row[7];
      row[0] = g[q][1] * np_[q][2] - g[q][2] * np_[q][1];  // s x n
      row[1] = g[q][2] * np_[q][0] - g[q][0] * np_[q][2];
      row[2] = g[q][0] * np_[q][1] - g[q][1] * np_[q][0];
      row[3] = np_[q][0];
      row[4] = np_[q][1];
      row[5] = np_[q][2];
      row[6] = hsk_dot3(np_[q][0], np_[q][1], np_[q][2], ex, ey, ez);
      double rs[6], rd[7];
#pragma unroll
      for (int a = 0; a < 7; ++a) rd[a] = (double)(valid ? row[a] : 0.0f);
#pragma unroll
      for (int a = 0; a < 6; ++a) rs[a] = rd[a] * ICP_SCALE;
      int k = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 7; ++b, ++k) {
          // (the lane's first pixel starts the sums: no 27 doubles of zeros to write and add to -- 81 instructions of a
          // wave's ~540 at the coarse level.  A rejected first pixel leaves -0 where the sum of zeros was +0: the block
          // sums start from +0 and an all-zero block adds nothing, so no sign of zero gets out)
          const double p = rint(rs[a] * rd[b]);
          acc[k] = q == 0 ? p : acc[k] + p;
        }
    }
  }
}

// block reduction of the per-lane accumulators -> 27 (unscaled) sums in out[0..26], valid after the barrier
static __device__ __forceinline__ void icp_block_sums(const double* acc, double (*sh)[32], double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double v = wave_sum27(acc, lane);
  if ((lane & 1) == 0) sh[wave][wave_sum27_index(lane)] = v;
  __syncthreads();
  if (threadIdx.x < 27) {
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < ICP_BLOCK / 64; ++w) r += sh[w][threadIdx.x];
    out[threadIdx.x] = r * ICP_UNSCALE;
  }
}

template <int ICP_PX>
__global__ __launch_bounds__(ICP_BLOCK) void k_icp_accumulate(const float* __restrict__ vcur,
                                                              const float* __restrict__ ncur,
                                                              const float* __restrict__ vprev,
                                                              const float* __restrict__ nprev, int W, int H, Intr in,
                                                              const TrackState* __restrict__ st, float dist_thresh,
                                                              float angle_thresh, int row0, int row1,
                                                              double* __restrict__ partials) {
  __shared__ double sh[ICP_BLOCK / 64][32];
  double acc[27];
  if (!st->lost) {
    IcpLaneIn<ICP_PX> L;
    icp_load_current<ICP_PX>(vcur, ncur, W, H, row0, row1, L);
    icp_accumulate_pixels<ICP_PX>(L, vprev, nprev, W, H, in, st->R, st->t, st->Rp, st->tp, dist_thresh, angle_thresh, acc);
  } else {
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  }
  icp_block_sums(acc, sh, partials + (size_t)blockIdx.x * 27);
}

// pixels per lane by level width, chosen so that a level's blocks fit the 256 CUs in one round (4:3 images)
static inline int icp_px(int W) { return W >= 512 ? ICP_PX_FINE : (W >= 256 ? ICP_PX_MID : 1); }
int icp_num_blocks(int W, int rows) { return (W * rows + ICP_BLOCK * icp_px(W) - 1) / (ICP_BLOCK * icp_px(W)); }

// The kernels gate on squared quantities.  dist2_max: the largest binary32 y with sqrtf(y) <= dist_thresh;
// sine2_lim: the smallest y with sqrtf(y) >= angle_thresh.  With a correctly rounded, monotone sqrtf:
//   sqrtf(d2) <= dist_thresh  <=>  d2 <= dist2_max      and      sqrtf(c2) < angle_thresh  <=>  c2 < sine2_lim.
static void icp_gate_limits(float dist_thresh, float angle_thresh, float* dist2_max, float* sine2_lim) {
  if (!(dist_thresh >= 0.0f)) {
    *dist2_max = -1.0f;  // nothing passes (also for a NaN threshold)
  } else {
    float y = dist_thresh * dist_thresh;
    if (!(y < INFINITY)) {
      y = INFINITY;
    } else {
      while (sqrtf(y) > dist_thresh) y = nextafterf(y, -INFINITY);
      while (nextafterf(y, INFINITY) < INFINITY && sqrtf(nextafterf(y, INFINITY)) <= dist_thresh) y = nextafterf(y, INFINITY);
    }
    *dist2_max = y;
  }
  if (!(angle_thresh > 0.0f)) {
    *sine2_lim = 0.0f;  // sqrtf(c2) < 0 never holds
  } else {
    float y = angle_thresh * angle_thresh;
    if (!(y < INFINITY)) {
      y = INFINITY;
    } else {
      while (sqrtf(y) < angle_thresh) y = nextafterf(y, INFINITY);
      while (y > 0.0f && sqrtf(nextafterf(y, -INFINITY)) >= angle_thresh) y = nextafterf(y, -INFINITY);
    }
    *sine2_lim = y;
  }
}

void launch_icp_accumulate(hipStream_t s, const float* vcur, const float* ncur, const float* vprev, const float* nprev,
                           int W, int H, Intr in, const TrackState* st, float dist_thresh, float angle_thresh, int row0,
                           int row1, double* partials) {
  const int nb = icp_num_blocks(W, row1 - row0);
  icp_gate_limits(dist_thresh, angle_thresh, &dist_thresh, &angle_thresh);  // the kernels take the squared limits
  if (icp_px(W) == ICP_PX_FINE)
    hipLaunchKernelGGL(k_icp_accumulate<ICP_PX_FINE>, dim3(nb), dim3(ICP_BLOCK), 0, s, vcur, ncur, vprev, nprev, W, H, in, st,
                       dist_thresh, angle_thresh, row0, row1, partials);
  else if (icp_px(W) == ICP_PX_MID)
    hipLaunchKernelGGL(k_icp_accumulate<ICP_PX_MID>, dim3(nb), dim3(ICP_BLOCK), 0, s, vcur, ncur, vprev, nprev, W, H, in, st,
                       dist_thresh, angle_thresh, row0, row1, partials);
  else
    hipLaunchKernelGGL(k_icp_accumulate<1>, dim3(nb), dim3(ICP_BLOCK), 0, s, vcur, ncur, vprev, nprev, W, H, in, st,
                       dist_thresh, angle_thresh, row0, row1, partials);
}

// reduce per-block partials -> 27 sums (one block of 256 threads)
#define ICP_RED_UNROLL 40  // partial rows / 8 slices per trip (600 rows at 640x480 with 2 px per lane: 2 trips)
static __device__ __forceinline__ void block_reduce27(const double* __restrict__ partials, int nblocks,
                                                      double (*sh)[32], double* tot) {
  // thread (slice, k): k = tid & 31 is the sum index (27 used), slice = tid >> 5 takes every 8th block row;
  // consecutive k read consecutive doubles (coalesced) and all of a thread's loads are independent
  const int k = threadIdx.x & 31, slice = threadIdx.x >> 5;
  double v = 0.0;
  if (k < 27) {
    // partials were written by other XCDs: every load is an L2 miss, so put them ALL in flight before adding
    for (int b0 = slice; b0 < nblocks; b0 += 8 * ICP_RED_UNROLL) {
      double w[ICP_RED_UNROLL];
#pragma unroll
      for (int i = 0; i < ICP_RED_UNROLL; ++i) {
        const int b = b0 + 8 * i;
        w[i] = b < nblocks ? partials[(size_t)b * 27 + k] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < ICP_RED_UNROLL; ++i) v += w[i];
    }
  }
  sh[slice][k] = v;
  __syncthreads();
  if (threadIdx.x < 27) {
    double r = sh[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 8; ++q) r += sh[q][threadIdx.x];
    tot[threadIdx.x] = r;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_icp_reduce(const double* __restrict__ partials, int nblocks,
                                                    double* __restrict__ out27) {
  __shared__ double sh[ICP_SH_ROWS][32];
  __shared__ double tot[27];
  block_reduce27(partials, nblocks, sh, tot);
  if (threadIdx.x < 27) out27[threadIdx.x] = tot[threadIdx.x];
}
void launch_icp_reduce(hipStream_t s, const double* partials, int nblocks, double* out27) {
  hipLaunchKernelGGL(k_icp_reduce, dim3(1), dim3(256), 0, s, partials, nblocks, out27);
}

// single-lane kernel: solve the reduced system and refine the pose held in TrackState
__global__ void k_icp_update(const double* __restrict__ sums27, TrackState* __restrict__ st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (st->lost) return;
  double s[27];
  for (int k = 0; k < 27; ++k) {
    s[k] = sums27[k];
    st->sums[k] = s[k];
  }
  float x6[6];
  if (!hsk_solve6(s, x6)) {
    st->lost = 1;
    return;
  }
  float R[9], t[3];
  for (int i = 0; i < 9; ++i) R[i] = st->R[i];
  for (int i = 0; i < 3; ++i) t[i] = st->t[i];
  hsk_pose_update(R, t, x6);
  for (int i = 0; i < 9; ++i) st->R[i] = R[i];
  for (int i = 0; i < 3; ++i) st->t[i] = t[i];
  st->n_iter += 1;
}
void launch_icp_update(hipStream_t s, const double* sums27, TrackState* st) {
  hipLaunchKernelGGL(k_icp_update, dim3(1), dim3(64), 0, s, sums27, st);
}

// ---- fused ICP iteration (single-device path) -------------------------------------------------------------
// Iteration i: every block first reduces the partials of iteration i-1 and solves for the pose increment -- the
// same deterministic arithmetic in every block, so all blocks agree -- then accumulates its pixels with the new
// pose.  One launch per iteration instead of two; the current-map loads are issued before the prologue so their
// latency overlaps the solve.  Poses ping-pong through two IcpPose slots (block 0 publishes the new one).

static __device__ __forceinline__ void shard_reduce27(const double* __restrict__ slot, double (*sh)[32], double* tot) {
  const int k = threadIdx.x & 31, slice = threadIdx.x >> 5;  // 8 slices x 4 shards each
  double v = 0.0;
  if (slice < 8) {
    const double a = slot[(slice)*32 + k], b = slot[(slice + 8) * 32 + k], c = slot[(slice + 16) * 32 + k],
                 d = slot[(slice + 24) * 32 + k];
    v = (a + b) + (c + d);
    sh[slice][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    double r = sh[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 8; ++q) r += sh[q][threadIdx.x];
    tot[threadIdx.x] = r;
  }
  __syncthreads();
}

// block sums -> one f64 atomic add per sum into this block's shard
static __device__ __forceinline__ void icp_block_sums_atomic(const double* acc, double (*sh)[32], double* __restrict__ slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double v = wave_sum27(acc, lane);
  if ((lane & 1) == 0) sh[wave][wave_sum27_index(lane)] = v;
  __syncthreads();
  if (threadIdx.x < 27) {
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < ICP_BLOCK / 64; ++w) r += sh[w][threadIdx.x];
    if (r != 0.0)
      __hip_atomic_fetch_add(slot + (blockIdx.x % ICP_SHARDS) * 32 + threadIdx.x, r * ICP_UNSCALE, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }
}


template <int ICP_PX>
__global__ __launch_bounds__(ICP_BLOCK) void k_icp_iter(double* __restrict__ slots, int iter, const IcpPose* __restrict__ pose_in,
                                                        TrackState* st, const float* __restrict__ vcur,
                                                        const float* __restrict__ ncur, const float* __restrict__ vprev,
                                                        const float* __restrict__ nprev, int W, int H, Intr in,
                                                        float dist_thresh, float angle_thresh, IcpPose* __restrict__ pose_out) {
  __shared__ double sh[ICP_SH_ROWS][32];
  __shared__ double tot[27];
  __shared__ IcpPose sp;
  const int g_icp_iter = iter;
  (void)g_icp_iter;
  ICP_STAMP(0);
  double* __restrict__ slot_add = slots + (size_t)(iter % 3) * ICP_SLOT_DOUBLES;
  const double* __restrict__ slot_read = slots + (size_t)((iter + 2) % 3) * ICP_SLOT_DOUBLES;
  double* __restrict__ slot_clear = slots + (size_t)((iter + 1) % 3) * ICP_SLOT_DOUBLES;
  // the previous iteration's sums FIRST: they are what the iteration's chain waits for, and loads return in issue order --
  // behind the pixel loads below they waited for 30 loads per lane that nobody needs before the solve is done
  double sums_in[16];
  if (threadIdx.x < 64 && iter > 0) shard_load27_wave(slot_read, sums_in);
  IcpLaneIn<ICP_PX> L;
  icp_load_current<ICP_PX>(vcur, ncur, W, H, 0, H, L);  // independent of the pose: in flight during the prologue
  // the previous pose estimate and the model pose come from other launches (L2 misses): fetch them now, not after
  // the reduction's barrier where their latency would sit on the critical path of the solve
  IcpPose p_in;
  float Rp[9], tp[3];
  if (iter == 0) {
    // Start of a tracked frame (what k_begin_frame does for the unfused paths): the estimate starts at the previous
    // frame's pose, which is also the pose the model maps were raycast from.  A frame queued behind a lost one is
    // dropped: it runs with the lost flag set and leaves the state alone.  Block 0 publishes the bookkeeping; the
    // other blocks only read fields that nobody writes here (R, t, need_reset).
#pragma unroll
    for (int i = 0; i < 9; ++i) p_in.R[i] = Rp[i] = st->R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) p_in.t[i] = tp[i] = st->t[i];
    p_in.lost = st->need_reset ? 1 : 0;
    p_in.n_iter = 0;
    p_in.pad[0] = p_in.pad[1] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (p_in.lost) {
        st->lost = 1;
      } else {
        for (int i = 0; i < 9; ++i) st->Rp[i] = Rp[i];
        for (int i = 0; i < 3; ++i) st->tp[i] = tp[i];
        st->lost = 0;
        st->n_iter = 0;
      }
    }
  } else {
    p_in = *pose_in;
#pragma unroll
    for (int i = 0; i < 9; ++i) Rp[i] = st->Rp[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tp[i] = st->tp[i];
  }
  for (int i = blockIdx.x * ICP_BLOCK + threadIdx.x; i < ICP_SLOT_DOUBLES; i += gridDim.x * ICP_BLOCK)
    __hip_atomic_store(slot_clear + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < 64) {  // the first wave (icp_solve_step shares the work among its lanes)
    if (iter > 0) shard_sum27_wave(sums_in, tot);
    ICP_STAMP(1);
    IcpPose p = p_in;
    if (iter > 0) icp_solve_step(tot, p, iter);
    if (threadIdx.x == 0) {
      sp = p;
      if (blockIdx.x == 0) *pose_out = p;
    }
  }
  __syncthreads();
  ICP_STAMP(2);
  // (a lost frame adds nothing: its sums stay the zeros the slot was cleared to.  `sp.lost` is the block's, so the barrier
  // inside the block sums is taken by all of its threads or by none)
  if (!sp.lost) {
    double acc[27];
    icp_accumulate_pixels<ICP_PX>(L, vprev, nprev, W, H, in, sp.R, sp.t, Rp, tp, dist_thresh, angle_thresh, acc);
    ICP_STAMP(3);
    // (no barrier here: `sh` has no other user in this kernel since the sums of the previous iteration are gathered by one wave)
    icp_block_sums_atomic(acc, (double (*)[32])sh, slot_add);
  }
  ICP_STAMP(4);
}

// after the last iteration: final solve, pose and lost flag into the tracker state
__global__ __launch_bounds__(256) void k_icp_final(const IcpPose* __restrict__ pose_in, double* __restrict__ slots, int iter,
                                                   TrackState* __restrict__ st) {
  __shared__ double sh[ICP_SH_ROWS][32];
  __shared__ double tot[27];
  shard_reduce27(slots + (size_t)((iter + 2) % 3) * ICP_SLOT_DOUBLES, sh, tot);
  // slot 0 is where the next frame's first iteration adds: leave it empty (it may be the slot just read)
  for (int i = threadIdx.x; i < ICP_SLOT_DOUBLES; i += blockDim.x)
    __hip_atomic_store(slots + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  IcpPose p = *pose_in;
  if (threadIdx.x < 64) icp_solve_step(tot, p);
  if (threadIdx.x == 0) {
    for (int k = 0; k < 27; ++k) st->sums[k] = tot[k];
    if (p.lost) {
      st->lost = 1;
      st->need_reset = 1;
    } else {
      for (int i = 0; i < 9; ++i) st->R[i] = p.R[i];
      for (int i = 0; i < 3; ++i) st->t[i] = p.t[i];
    }
    st->n_iter = p.n_iter;
  }
}

// two pose slots (ping-pong) followed by the three accumulator slots
#define ICP_POSE_AREA 256
size_t icp_pose_bytes() { return ICP_POSE_AREA + 3 * ICP_SLOT_DOUBLES * sizeof(double); }
static inline double* icp_slots(void* pose_buf) { return (double*)((char*)pose_buf + ICP_POSE_AREA); }

// enqueue the whole ICP of one frame: levels coarse -> fine, iters[l] iterations each
void launch_icp_fused(hipStream_t s, float* const* vcur, float* const* ncur, float* const* vmod, float* const* nmod,
                      const ImgLevel* lv, const int* iters, TrackState* st, float dist_thresh, float angle_thresh,
                      void* pose_buf, double* part_a, double* part_b, IcpFinal* defer_final, hipEvent_t* level_events) {
  (void)part_a;
  (void)part_b;
  static_assert(2 * sizeof(IcpPose) <= ICP_POSE_AREA, "pose ping-pong must fit its area");
  IcpPose* pb = (IcpPose*)pose_buf;
  double* slots = icp_slots(pose_buf);
  icp_gate_limits(dist_thresh, angle_thresh, &dist_thresh, &angle_thresh);  // the kernels take the squared limits
  int i = 0;
  for (int l = HSK_NLEVELS - 1; l >= 0; --l) {
    const int W = lv[l].W, H = lv[l].H;
    const int nb = icp_num_blocks(W, H);
    if (level_events) (void)hipEventRecord(level_events[HSK_NLEVELS - 1 - l], s);  // profiling: the level's iterations start
    for (int it = 0; it < iters[l]; ++it, ++i) {
      if (icp_px(W) == ICP_PX_FINE)
        hipLaunchKernelGGL(k_icp_iter<ICP_PX_FINE>, dim3(nb), dim3(ICP_BLOCK), 0, s, slots, i, pb + (i & 1), st, vcur[l], ncur[l],
                           vmod[l], nmod[l], W, H, lv[l].in, dist_thresh, angle_thresh, pb + ((i + 1) & 1));
      else if (icp_px(W) == ICP_PX_MID)
        hipLaunchKernelGGL(k_icp_iter<ICP_PX_MID>, dim3(nb), dim3(ICP_BLOCK), 0, s, slots, i, pb + (i & 1), st, vcur[l], ncur[l],
                           vmod[l], nmod[l], W, H, lv[l].in, dist_thresh, angle_thresh, pb + ((i + 1) & 1));
      else
        hipLaunchKernelGGL(k_icp_iter<1>, dim3(nb), dim3(ICP_BLOCK), 0, s, slots, i, pb + (i & 1), st, vcur[l], ncur[l], vmod[l], nmod[l],
                           W, H, lv[l].in, dist_thresh, angle_thresh, pb + ((i + 1) & 1));
    }
  }
  if (level_events) (void)hipEventRecord(level_events[HSK_NLEVELS], s);
  if (i > 0 && defer_final) {
    // the caller's next launch (k_column_zrange, first kernel of integrate) does the last solve in its prologue: one
    // launch and one kernel boundary less per frame
    defer_final->pose_in = pb + (i & 1);
    defer_final->slots = slots;
    defer_final->iter = i;
  } else if (i > 0)
    hipLaunchKernelGGL(k_icp_final, dim3(1), dim3(256), 0, s, pb + (i & 1), slots, i, st);
  else
    launch_begin_frame(s, st, nullptr);  // no iteration configured: the frame still starts (previous pose, lost flag)
}

// start of a tracked frame: previous pose <- current pose, clear the lost flag
__global__ void k_begin_frame(TrackState* __restrict__ st, IcpPose* __restrict__ pose0) {
  if (threadIdx.x != 0) return;
  if (st->need_reset) {  // a frame queued behind a lost one (asynchronous submission): dropped, state untouched
    st->lost = 1;
    if (pose0) pose0->lost = 1;
    return;
  }
  for (int i = 0; i < 9; ++i) st->Rp[i] = st->R[i];
  for (int i = 0; i < 3; ++i) st->tp[i] = st->t[i];
  st->lost = 0;
  st->n_iter = 0;
  if (pose0) {  // seed of the fused ICP's pose ping-pong
    IcpPose p;
    for (int i = 0; i < 9; ++i) p.R[i] = st->R[i];
    for (int i = 0; i < 3; ++i) p.t[i] = st->t[i];
    p.lost = 0;
    p.n_iter = 0;
    p.pad[0] = p.pad[1] = 0;
    *pose0 = p;
  }
}
void launch_begin_frame(hipStream_t s, TrackState* st, void* icp_pose_buf) {
  hipLaunchKernelGGL(k_begin_frame, dim3(1), dim3(64), 0, s, st, (IcpPose*)icp_pose_buf);
}


// host mirrors (used by hsk_icp_solve and by tests through the C ABI)
bool host_solve6(const double* in27, float* x6) { return hsk_solve6(in27, x6); }
void host_pose_update(float* R, float* t, const float* x6) { hsk_pose_update(R, t, x6); }
