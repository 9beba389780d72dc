// kernels_volume.hip -- TSDF volume kernels for gfx950: integrate, raycast (A.6),
// zero-crossing cloud extraction (A.7).  Hand-written for wave64 / 16-B-per-lane HBM access; no MFMA (none
// of these is a contraction).  The volume holds (int16 tsdf*32767, int16 weight) pairs in 64-B blocks of one lane-block
// (4 x-voxels x 4 planes: hsk_dev.h, hsk_vox_index); the host's arrays are row-major, x fastest (k_vol_convert).
#pragma clang fp contract(off)
#include "hsk_dev.h"
#include "hsk_launch.h"

// ------------------------------------------------------------------------------------------------------
// integrate (A.4).  Layout: each lane owns 4 x-adjacent voxels (one 16-B vector per plane, the four planes of a group
// consecutive: a 64-B block), a wave covers 64 voxels of 4 consecutive rows, a block 16 rows and a chunk of 8 or 16 planes.
//
// Only vectors that hold at least one rewritten voxel are read or written, so HBM traffic tracks the
// algorithmic 8 B x V_upd, not the 8 B x N^3 sweep.  Three conservative culls keep the
// arithmetic off the voxels that cannot be rewritten (each only ever skips voxels the exact test rejects):
//   1. per-lane z interval of the column inside the (padded) view frustum, computed once per column;
//   2. occlusion: a 16x16-pixel tile table of max scaled depth, 3x3-dilated and staged in LDS; a 4-voxel
//      group whose nearest possible point is farther than that maximum + tau cannot pass sdf >= -tau;
//   3. exact fast paths of the running mean (saturated free space, first observation).
// Bricks (8^3 voxels) that ever received a negative TSDF are flagged for the raycaster's empty-space test.
// ------------------------------------------------------------------------------------------------------
#define HSK_TILE 16
#define HSK_NQUEUES 256       // uncertain lane-blocks are spread over this many queues (pass A -> pass B)
#define HSK_QCOUNT_STRIDE 64  // words between two queue counters (256 B: one counter per memory-side atomic line)
#ifndef INTEGRATE_WPE
#define INTEGRATE_WPE 8  // waves per SIMD the register allocator must leave room for (pass A takes 48 VGPRs: 8 waves fit either
                         // way; told so, the compiler schedules it a little tighter: 69.9 -> 69.5 us at 512^3, 332 -> 327 at 1024^3)
#endif
#ifndef INTEGRATE_WPE_LONG
#define INTEGRATE_WPE_LONG 7  // ... of the form that takes four groups of planes through the stages together (16-plane chunks,
                              // 1024^3): under 64 registers it spilled 36 bytes; with 72 the stage takes 269 us where it took 279
                              // (with 6 waves and 80 registers: 277)
#endif

__global__ void k_tile_max(const float* __restrict__ scaled, int W, int H, float* __restrict__ tmax,
                           float* __restrict__ tmin, int tw) {
  __shared__ float shx[4], shn[4];
  const int tx = blockIdx.x, ty = blockIdx.y;
  const int x = tx * HSK_TILE + (threadIdx.x & 15), y = ty * HSK_TILE + (threadIdx.x >> 4);
  const bool in = x < W && y < H;
  const float v = in ? scaled[y * W + x] : 0.0f;
  float mx = v, mn = v;  // a pixel outside the image or without depth makes the tile minimum 0 ("not all valid")
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    mn = fminf(mn, __shfl_xor(mn, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    shx[threadIdx.x >> 6] = mx;
    shn[threadIdx.x >> 6] = mn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    tmax[ty * tw + tx] = fmaxf(fmaxf(shx[0], shx[1]), fmaxf(shx[2], shx[3]));
    tmin[ty * tw + tx] = fminf(fminf(shn[0], shn[1]), fminf(shn[2], shn[3]));
  }
}
// tiles holds 4 * tw * th floats: raw max, raw min (this kernel, or k_bilateral_scale on the frame path), then the
// interleaved 3x3-dilated (max, min) table that k_column_zrange's tail blocks fill before every integrate
void launch_tile_max(hipStream_t s, const float* scaled, int W, int H, float* tiles) {
  const int tw = (W + HSK_TILE - 1) / HSK_TILE, th = (H + HSK_TILE - 1) / HSK_TILE;
  const int n = tw * th;
  hipLaunchKernelGGL(k_tile_max, dim3(tw, th), dim3(256), 0, s, scaled, W, H, tiles, tiles + n, tw);
}

// Fine tile tables for the second classification level: per 8x8-pixel tile and per 4x4-pixel tile the (max,
// min-if-all-valid) of the scaled depth, NOT dilated (the lookup covers the exact pixel box of a voxel block; the finer
// table serves the blocks whose box spans at most 3 x 3 of its tiles -- most of them from about a metre on).  One thread
// per 8-px tile; it writes its four 4-px quadrants too (qtab: 2 fw x 2 fh tiles).
#define HSK_FTILE 8
__global__ void k_tile_fine(const float* __restrict__ scaled, int W, int H, float2* __restrict__ ftab, int fw, int fh,
                            float2* __restrict__ qtab) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= fw * fh) return;
  const int ty = t / fw, tx = t - ty * fw;
  float mx = 0.0f, mn = 1e30f;
#pragma unroll
  for (int qy = 0; qy < 2; ++qy)
#pragma unroll
    for (int qx = 0; qx < 2; ++qx) {
      float qmx = 0.0f, qmn = 1e30f;
      for (int dy = 0; dy < 4; ++dy)
        for (int dx = 0; dx < 4; ++dx) {
          const int x = tx * HSK_FTILE + qx * 4 + dx, y = ty * HSK_FTILE + qy * 4 + dy;
          const float v = (x < W && y < H) ? scaled[(size_t)y * W + x] : 0.0f;  // outside the image: never "all valid"
          qmx = fmaxf(qmx, v);
          qmn = fminf(qmn, v);
        }
      qtab[(size_t)(2 * ty + qy) * (2 * fw) + (2 * tx + qx)] = make_float2(qmx, qmn);
      mx = fmaxf(mx, qmx);
      mn = fminf(mn, qmn);
    }
  ftab[t] = make_float2(mx, mn);
}

// The second level needs the (max, min) of the depth over the tiles its pixel box touches: nx x ny tiles, nx, ny <= 3,
// from the tile that holds the box's top-left corner.  Those windows are formed here once per frame, one table per
// window shape (nine of them: blockIdx.y = (ny - 1) * 3 + (nx - 1)), so that pass A makes ONE look-up per lane-block
// where it made nine -- pass A and pass B are bound by the CU's vector-memory front end
// (profiles/r02/integrate_analysis.md section 7) -- and over exactly the tiles the box touches, not always 3 x 3.
// Entry (tx, ty) of shape (nx, ny) = over a < ny, b < nx of tab[min(ty + a, last)][min(tx + b, last)].
__global__ void k_tile_window(const float2* __restrict__ tab, int tbw, int tbh, float2* __restrict__ win) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= tbw * tbh) return;
  const int nx = (int)blockIdx.y % 3 + 1, ny = (int)blockIdx.y / 3 + 1;
  const int ty = t / tbw, tx = t - ty * tbw;
  float mx = 0.0f, mn = 1e30f;
  for (int a = 0; a < ny; ++a)
    for (int b = 0; b < nx; ++b) {
      const float2 v = tab[min(ty + a, tbh - 1) * tbw + min(tx + b, tbw - 1)];
      mx = fmaxf(mx, v.x);
      mn = fminf(mn, v.y);
    }
  win[(size_t)blockIdx.y * tbw * tbh + t] = make_float2(mx, mn);
}

// clip [lo,hi] (in gz) with c + m*gz >= 0
static __device__ __forceinline__ void clip_interval(float c, float m, float& lo, float& hi) {
  if (m > 0.0f) {
    lo = fmaxf(lo, -c * __builtin_amdgcn_rcpf(m));
  } else if (m < 0.0f) {
    hi = fminf(hi, -c * __builtin_amdgcn_rcpf(m));
  } else if (c < 0.0f) {
    lo = 1e30f;
    hi = -1e30f;
  }
}

// Per-frame pre-pass: for every lane column (4 x-adjacent voxels at one y) the range of stored planes that can
// project into the padded image [-1.5, W+0.5] x [-1.5, H+0.5] in front of the camera.  The view frustum is
// convex, so each column meets it in one interval; clipping the line cam(gz) = a + gz * c against the five
// half-spaces gives it.  Conservative by 2 planes (float error).  Empty columns get (INT_MAX, INT_MIN).
// (argument order: what a kernel of the frame's chain needs first comes first -- the first 16 dwords of the arguments
// arrive in SGPRs with the wave, -amdgpu-kernarg-preload-count in the Makefile)
__global__ void k_column_zrange(IcpFinal fin, const float* __restrict__ tmax, const float* __restrict__ tmin,
                                float2* __restrict__ dtab, int tw, int th, int dil_blocks, unsigned* __restrict__ qcount,
                                const TrackState* __restrict__ st, VolParams vp, int W, int H, Intr in,
                                int2* __restrict__ zint, TrackState* __restrict__ st_out, int2* __restrict__ wgz, RingOut early) {
  // fin.slots != null: the frame's ICP has left its last solve to this launch (launch_icp_fused).  The first wave of
  // EVERY block reads the sharded sums of the last iteration and solves (deterministic: all blocks get the same pose),
  // the block then works with that pose; block 0 also publishes it -- what k_icp_final does in a launch of its own.
  // (Slot 0 of the accumulators, read here by all blocks, is emptied for the next frame by pass A's first block.)
  __shared__ double fin_tot[27];
  __shared__ IcpPose fin_pose;
  // (the sums are requested first; the table work below, which needs no pose, runs while they travel)
  double sums_in[16];
  if (fin.slots && threadIdx.x < 64) shard_load27_wave(fin.slots + (size_t)((fin.iter + 2) % 3) * ICP_SLOT_DOUBLES, sums_in);
  // The first blocks also dilate the tile table and clear the queue counters of pass A (no extra launch, memset node or
  // extra blocks: at 512^3 the column work alone is exactly one block per CU).
  {
    if (blockIdx.x == 0)
      for (int q = threadIdx.x; q < HSK_NQUEUES; q += blockDim.x) qcount[q * HSK_QCOUNT_STRIDE] = 0u;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tw * th && (int)blockIdx.x < dil_blocks) {
      const int ty = i / tw, tx = i - ty * tw;
      float mx = 0.0f, mn = 1e30f;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = min(max(ty + dy, 0), th - 1), xx = min(max(tx + dx, 0), tw - 1);
          mx = fmaxf(mx, tmax[yy * tw + xx]);
          mn = fminf(mn, tmin[yy * tw + xx]);
        }
      dtab[i] = make_float2(mx, mn);
    }
  }
  if (fin.slots) {
    if (threadIdx.x < 64) {
      shard_sum27_wave(sums_in, fin_tot);
      IcpPose p = *fin.pose_in;
      icp_solve_step(fin_tot, p);
      if (threadIdx.x == 0) {
        fin_pose = p;
        if (blockIdx.x == 0) {
          for (int k = 0; k < 27; ++k) st_out->sums[k] = fin_tot[k];
          if (p.lost) {
            st_out->lost = 1;
            st_out->need_reset = 1;
          } else {
            for (int i = 0; i < 9; ++i) st_out->R[i] = p.R[i];
            for (int i = 0; i < 3; ++i) st_out->t[i] = p.t[i];
          }
          st_out->n_iter = p.n_iter;
          if (early.slots) {
            // The frame's pose and verdict are final HERE -- nothing after the ICP writes them -- so the host is told now,
            // a whole integrate earlier than by the raycast's report (which stays: its mark says that the frame's
            // inputs are consumed).  A caller that takes one frame at a time gets its pose back while the volume work
            // is still running, and its next frame's filtering runs under that.
            const unsigned n = *early.seq;  // (counted up by the raycast's report, not here)
            TrackState* dst = early.slots + early.slot_fifo[n % HSK_RING_FIFO];
            const int* src_w = (const int*)st_out;
            int* dst_w = (int*)dst;
            for (unsigned i = 0; i < (unsigned)(offsetof(TrackState, ring_mark) / 4); ++i) dst_w[i] = src_w[i];
            __threadfence_system();
            __hip_atomic_store(&dst->pose_mark, (n + 1u) | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
    }
    __syncthreads();
  }
  // One block = the x-y footprint of one pass-A workgroup (16 lane columns by 16 rows), so that the block can also leave
  // that workgroup's z range: pass A's workgroups of the chunks outside it (half of its waves lie outside the frustum)
  // then leave on one scalar load instead of a vector load per lane and a wave-wide reduction.
  __shared__ int wg_lo[4], wg_hi[4];
  const int ncol = vp.X / 4;
  const int gxn = (vp.X + 63) / 64, gyn = (vp.Y + 15) / 16;
  const bool fp_block = (int)blockIdx.x < gxn * gyn;
  const int fby = (int)blockIdx.x / gxn, fbx = (int)blockIdx.x - fby * gxn;
  const int y = fby * 16 + (int)(threadIdx.x >> 4), lc = fbx * 16 + (int)(threadIdx.x & 15);
  const bool col_ok = fp_block && y < vp.Y && lc < ncol;
  const int x0 = lc * 4;
  int zl = 0x7fffffff, zh = -0x7fffffff;
  if (col_ok) {
    // (a lost frame keeps the previous pose in st; the solve's own estimate is then meaningless, and nothing integrates)
    const bool own = fin.slots != nullptr && !fin_pose.lost;
    const float* __restrict__ Rm = own ? fin_pose.R : st->R;
    const float* __restrict__ tm = own ? fin_pose.t : st->t;
    const float tx = tm[0], ty = tm[1], tz = tm[2];
    const float i00 = Rm[0], i01 = Rm[3], i02 = Rm[6];
    const float i10 = Rm[1], i11 = Rm[4], i12 = Rm[7];
    const float i20 = Rm[2], i21 = Rm[5], i22 = Rm[8];
    const float gy = ((float)y + 0.5f) * vp.cell[1] - ty;
    float glo = 1e30f, ghi = -1e30f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gx = ((float)(x0 + j) + 0.5f) * vp.cell[0] - tx;
      const float ax = i00 * gx + i01 * gy, ay = i10 * gx + i11 * gy, az = i20 * gx + i21 * gy;
      float lo = -1e30f, hi = 1e30f;
      clip_interval(az, i22, lo, hi);
      const float ul = 1.5f + in.cx, uh = ((float)W + 0.5f) - in.cx;
      const float vl = 1.5f + in.cy, vh = ((float)H + 0.5f) - in.cy;
      clip_interval(ax * in.fx + ul * az, i02 * in.fx + ul * i22, lo, hi);
      clip_interval(uh * az - ax * in.fx, uh * i22 - i02 * in.fx, lo, hi);
      clip_interval(ay * in.fy + vl * az, i12 * in.fy + vl * i22, lo, hi);
      clip_interval(vh * az - ay * in.fy, vh * i22 - i12 * in.fy, lo, hi);
      if (lo <= hi) {
        glo = fminf(glo, lo);
        ghi = fmaxf(ghi, hi);
      }
    }
    if (glo <= ghi) {
      // gz = (z + 0.5) * cell_z - tz  =>  z = (gz + tz) / cell_z - 0.5; pad by 2 planes for float error
      const float inv_cz = __builtin_amdgcn_rcpf(vp.cell[2]);
      const float fl = (glo + tz) * inv_cz - 2.5f, fh = (ghi + tz) * inv_cz + 1.5f;
      const int a = fl < -1e9f ? -1000000000 : (fl > 1e9f ? 1000000000 : (int)floorf(fl));
      const int b = fh < -1e9f ? -1000000000 : (fh > 1e9f ? 1000000000 : (int)ceilf(fh));
      zl = a - vp.zs0;
      zh = b - vp.zs0;
    }
    zint[(size_t)y * ncol + lc] = make_int2(zl, zh);
  }
  if (fp_block) {  // block-uniform
    int lo = zl, hi = zh;
    // ... and the range EVERY column of the wave's footprint covers (largest lower end, smallest upper end; columns
    // outside the volume do not count): a chunk inside it needs no per-lane ranges at all
    int lo_all = col_ok ? zl : -0x7fffffff, hi_all = col_ok ? zh : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, __shfl_xor(lo, o, 64));
      hi = max(hi, __shfl_xor(hi, o, 64));
      lo_all = max(lo_all, __shfl_xor(lo_all, o, 64));
      hi_all = min(hi_all, __shfl_xor(hi_all, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      wg_lo[threadIdx.x >> 6] = lo;
      wg_hi[threadIdx.x >> 6] = hi;
      // this wave's footprint (16 lane columns x 4 rows): union of its columns' ranges, then their intersection
      ((int4*)(wgz + (((size_t)gxn * gyn + 1) & ~(size_t)1)))[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = make_int4(lo, hi, lo_all, hi_all);
    }
    __syncthreads();
    if (threadIdx.x == 0)
      wgz[blockIdx.x] = make_int2(min(min(wg_lo[0], wg_lo[1]), min(wg_lo[2], wg_lo[3])), max(max(wg_hi[0], wg_hi[1]), max(wg_hi[2], wg_hi[3])));
  }
}

// per-lane, z-invariant terms of one column group (4 x-adjacent voxels): the group centre's R^T (gx, gy, 0), x / y terms
// pre-multiplied by fx / fy, and gx^2 + gy^2 (the corner terms of the second level are computed when it is entered)
struct ColumnTerms {
  float axfc, ayfc, azc, pnc;
};
// constants of the launch that depend on the configuration only: computed once on the host (they were ~40 instructions
// with a division in every wave's prologue), passed by value
struct IntegrateConst {
  float hw, hh;
  float rk4, zmin4, cull_thr4, free_thr4;  // first level: a 4-plane block (voxels within 2.2 cells of its centre)
  float cull_thr2, free_thr2;              // second level: against the block's exact distance range (no cell margin)
};
static IntegrateConst integrate_const(const VolParams& vp, int W, int H, const Intr& in) {
  IntegrateConst k;
  // The 4 voxels of a lane lie within 1.5 cells of the group centre; their pixels lie within
  // r = rk / z + 2.5 px of the centre's pixel when z > zmin (z - 2 cells >= z / 1.06).
  const float cellm = fmaxf(vp.cell[0], fmaxf(vp.cell[1], vp.cell[2]));
  const float rk = 1.06f * 2.75f * cellm * fmaxf(in.fx, in.fy);
  k.hw = 0.5f * (float)(W - 1);
  k.hh = 0.5f * (float)(H - 1);
  k.rk4 = rk * 1.47f;  // sqrt(1.5^2 + 1.5^2) / 1.5, rounded up
  k.zmin4 = fmaxf(fmaxf(0.1f, 40.0f * cellm), k.rk4 / ((float)HSK_TILE - 2.5f));
  k.cull_thr4 = vp.tau * 1.001f + 1e-4f + 2.3f * cellm;
  k.free_thr4 = vp.tau * 1.0002f + 1e-4f + 2.3f * cellm;
  k.cull_thr2 = vp.tau * 1.001f + 1e-4f;
  k.free_thr2 = vp.tau * 1.0002f + 1e-4f;
  return k;
}

// A voxel (x, y, stored plane zz) has just been given a negative TSDF: set its brick's bit and, the first time, its
// super-brick's.  Test first: after the first frames the bits are already set and no atomic is issued (a stale read
// only costs a redundant OR).
static __device__ __forceinline__ void mark_brick_negative(unsigned* __restrict__ flags, const VolParams& vp, int x, int y, int zz) {
  const int bs = vp.bshift, bxn = vp.X >> bs, byn = vp.Y >> bs;
  const int bx = x >> bs, by = y >> bs, bz = zz >> bs;
  const int bit = (bz * byn + by) * bxn + bx;
  if ((flags[bit >> 5] >> (bit & 31)) & 1u) return;
  __hip_atomic_fetch_or(&flags[bit >> 5], 1u << (bit & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (hsk_super_ok(vp)) {
    const int sb = ((bz >> HSK_SUPER_SHIFT) * hsk_super_dim(vp.Y, bs) + (by >> HSK_SUPER_SHIFT)) * hsk_super_dim(vp.X, bs) + (bx >> HSK_SUPER_SHIFT);
    __hip_atomic_fetch_or(&flags[hsk_flag_words(vp) + (sb >> 5)], 1u << (sb & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- lane-block summaries: deep free space without volume traffic ---------------------------------------------------
// One byte per lane-block (4 x-voxels x 1 row x 4 planes: the unit pass A classifies):
//   0            nothing known: the block's 16 words are what the volume holds, and must be read
//   1            all 16 voxels are (0, 0): never observed (the state after a reset); the volume agrees
//   2 .. 129     all 16 voxels are (+1, w), w = s - 1 in 1 .. 128.  The volume holds +1 in all 16; its WEIGHTS MAY LAG
//   130 .. 255   all 16 voxels hold +1 with weights >= 1 that differ (the frustum's rim swept through the block); the
//                weights in the volume lag by p = s - 130 observations
// A free-space observation of a block in the last two states moves nothing but the byte: w <- min(w + 1, 128) is s + 1,
// and one more pending observation is s + 1 too.  The volume's weights are brought up to date ("materialised") only
// when somebody is about to look at them: pass A does it for every block it hands to pass B (which therefore knows
// nothing of summaries), k_materialize for the calls that read the volume out (download, cloud, mesh).  The raycast
// reads TSDF values only, and those are always current.  A block leaves state 1 with a store of (+1, 1) and no load;
// state 0 is the read-modify-write of before, after which the block is classified again.  A frame at 512^3 used to move
// 246 MB of its 278 MB through this path.
// What the calls return is bit for bit what it was without the summaries; they are a representation of the weights of
// deep free space, kept by: pass A, hsk_reset (all 1), hsk_upload_tsdf (k_rebuild_uniform).
// Layout: the groups of a pass-A chunk (vp.zchunk planes: 2 or 4 groups) sit side by side, so a pass-A lane fetches
// (and rewrites) all of its summaries with ONE 16- or 32-bit access, and the bytes of a wave (16 lanes in x by 4 rows by
// the chunk's groups) are contiguous.
#define HSK_SUM_RAGGED 130u
#define HSK_SUM_MAX 255u
// (NS = groups, i.e. summary bytes, per lane and chunk: 2 or 4 -- vp.zchunk / 4; a power of two, so shifts and masks)
template <int NS>
static __host__ __device__ __forceinline__ size_t hsk_sum_index_ns(const VolParams& vp, int x0, int y, int zb) {
  static_assert(NS == 2 || NS == 4, "8 or 16 planes per chunk");
  const size_t tiles_x = (size_t)(vp.X + 63) / 64, tiles_y = (size_t)(vp.Y + 3) / 4;
  return (((((size_t)(zb >> (NS == 4 ? 4 : 3)) * tiles_y + (size_t)(y >> 2)) * tiles_x + (size_t)(x0 >> 6)) * 4 + (size_t)(y & 3)) * 16 +
          (size_t)((x0 >> 2) & 15)) * (size_t)NS + (size_t)((zb >> 2) & (NS - 1));
}
static __host__ __device__ __forceinline__ size_t hsk_sum_index(const VolParams& vp, int x0, int y, int zb) {
  return vp.zchunk == 16 ? hsk_sum_index_ns<4>(vp, x0, y, zb) : hsk_sum_index_ns<2>(vp, x0, y, zb);
}
// Where the 16-B vector of voxels x0 .. x0 + 3 (x0 a multiple of 4) of row y, stored plane zb + u (zb a multiple of 4,
// u in 0 .. 3) sits in the volume, in vectors: the four vectors of a lane-block are consecutive (hsk_dev.h: hsk_vox_index).
// VIDX needs `vp`, `x0`, `y` in scope.
static __device__ __forceinline__ size_t hsk_bbase(const VolParams& vp, int x0, int y, int zb) {
  return ((((size_t)(zb >> 2) * vp.Y + (size_t)y) * (size_t)(vp.X >> 2)) + (size_t)(x0 >> 2)) << 2;
}
#define VIDX(zbv, u) (hsk_bbase(vp, x0, y, (zbv)) + (size_t)(u))
static __device__ __forceinline__ unsigned hsk_uniform_code(unsigned word) {
  const unsigned w = word >> 16;
  if (word == 0u) return 1u;
  return ((word & 0xffffu) == (unsigned)HSK_DIVISOR && w >= 1u && w <= (unsigned)HSK_MAX_WEIGHT) ? w + 1u : 0u;
}
static __device__ __forceinline__ bool hsk_vector_is(const uint4& q, unsigned word) {
  return q.x == word && q.y == word && q.z == word && q.w == word;
}
// state of a block from its 16 words (never 130 + p with p > 0: the words are the truth)
static __device__ __forceinline__ unsigned hsk_sum_classify(const uint4 q[4]) {
  const unsigned word = q[0].x;
  if (hsk_vector_is(q[0], word) && hsk_vector_is(q[1], word) && hsk_vector_is(q[2], word) && hsk_vector_is(q[3], word))
    return hsk_uniform_code(word);
  unsigned a = 0xffffffffu, o = 0u, m = 0xffffffffu;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    a &= q[u].x & q[u].y & q[u].z & q[u].w;
    o |= q[u].x | q[u].y | q[u].z | q[u].w;
    m = min(m, min(min(q[u].x, q[u].y), min(q[u].z, q[u].w)));
  }
  // every TSDF half is 0x7fff (all of its low 15 bits set in the AND, bit 15 clear in the OR), every weight >= 1
  return ((a & 0x7fffu) == 0x7fffu && (o & 0x8000u) == 0u && m >= 0x10000u && (o >> 16) <= (unsigned)HSK_MAX_WEIGHT) ? HSK_SUM_RAGGED : 0u;
}
// p more observations of a block whose 16 voxels all hold +1: w <- min(w + p, 128) on the packed words
static __device__ __forceinline__ void hsk_vector_add_weight(uint4& q, unsigned p) {
  const unsigned cap = ((unsigned)HSK_MAX_WEIGHT << 16) | (unsigned)HSK_DIVISOR, inc = p << 16;
  q.x = min(q.x + inc, cap);
  q.y = min(q.y + inc, cap);
  q.z = min(q.z + inc, cap);
  q.w = min(q.w + inc, cap);
}

// All four voxels of the vector observed as free space (F == 1).  Onto a stored +1 the running mean leaves +1
// ((1*W + 1) / (W + 1) == 1 exactly) and only the weight moves, W <- min(W + 1, 128): one add and one min on the packed
// word; an unseen voxel (W == 0) becomes (+1, 1).  Anything else -- a voxel that was inside the band in an earlier
// frame -- takes the running mean on the exact shortcuts (a = Fp * Wp + 1 lies in hsk_div_small_exact's domain:
// |a| <= 129, and a sum of a binary32 product and 1 is 0 or at least 2^-24 in magnitude), behind ONE wave-uniform
// branch.  (Round 2 measured the former per-voxel branches with binary64 arithmetic behind them at 23 of pass A's 71 us:
// one such voxel among a wave's 1024 sent the whole wave through them.)  F == 1 cannot turn a non-negative value
// negative, so no brick flag can newly be due.  Returns whether the vector changed (false: the store is skipped).
static __device__ __forceinline__ bool update_vector_free4(uint4& q) {
  const unsigned cap = ((unsigned)HSK_MAX_WEIGHT << 16) | (unsigned)HSK_DIVISOR;
  const unsigned w4[4] = {q.x, q.y, q.z, q.w};
  unsigned nw[4];
  bool gen[4], gen_any = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool unseen = (w4[j] >> 16) == 0u;
    const bool simple = unseen || (w4[j] & 0xffffu) == (unsigned)HSK_DIVISOR;
    nw[j] = unseen ? (0x10000u | (unsigned)HSK_DIVISOR) : min(w4[j] + 0x10000u, cap);
    gen[j] = !simple;
    gen_any = gen_any || gen[j];
  }
  if (__ballot(gen_any) != 0ull) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tp = (int)(short)(w4[j] & 0xffffu);
      const unsigned wp = w4[j] >> 16;
      const float Wp = (float)wp;
      const float Fn = hsk_div_small_exact(hsk_tsdf_unpack(tp) * Wp + 1.0f, Wp + 1.0f);
      int fixed = (int)(Fn * 32767.0f);  // truncation toward zero
      fixed = min(max(fixed, -HSK_DIVISOR), HSK_DIVISOR);
      const unsigned wg = ((unsigned)fixed & 0xffffu) | (min(wp + 1u, (unsigned)HSK_MAX_WEIGHT) << 16);
      nw[j] = gen[j] ? wg : nw[j];
    }
  }
  q = make_uint4(nw[0], nw[1], nw[2], nw[3]);
  return nw[0] != w4[0] || nw[1] != w4[1] || nw[2] != w4[2] || nw[3] != w4[3];
}

// wave-uniform constants of the per-voxel path: the pose (world -> camera rows) of the frame
struct DetailPose {
  float tx, ty, tz;
  float i00, i01, i02, i10, i11, i12, i20, i21, i22;
};
static __device__ __forceinline__ DetailPose detail_pose(const TrackState* __restrict__ st) {
  DetailPose p;
  p.tx = st->t[0]; p.ty = st->t[1]; p.tz = st->t[2];
  p.i00 = st->R[0]; p.i01 = st->R[3]; p.i02 = st->R[6];
  p.i10 = st->R[1]; p.i11 = st->R[4]; p.i12 = st->R[7];
  p.i20 = st->R[2]; p.i21 = st->R[5]; p.i22 = st->R[8];
  return p;
}

// One lane-block (4 x-voxels at x0, row y, stored planes zb .. zb + 3) through the per-voxel path, U planes per trip.
// `planes`: bit u set = plane zb + u lies in the lane's z range (0: the lane holds no entry and idles inside the
// wave-uniform branches).  Returns the voxels rewritten.
template <bool COUNT_ONLY, int U>
static __device__ __forceinline__ unsigned detail_entry(unsigned planes, int x0, int y, int zb, uint4* __restrict__ vol,
                                                        const float* __restrict__ scaled, const DetailPose& P,
                                                        const VolParams& vp, int W, int H, const Intr& in,
                                                        unsigned* __restrict__ flags) {
  static_assert(U == 1 || U == 2 || U == 4, "planes per trip");
  const int lane = threadIdx.x & 63;
  (void)lane;
  const unsigned cap = ((unsigned)HSK_MAX_WEIGHT << 16) | (unsigned)HSK_DIVISOR;
  unsigned cnt = 0;
  float ax[4], ay[4], az[4], pn[4];
  const float gy = ((float)y + 0.5f) * vp.cell[1] - P.ty;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gx = ((float)(x0 + j) + 0.5f) * vp.cell[0] - P.tx;
    ax[j] = P.i00 * gx + P.i01 * gy;
    ay[j] = P.i10 * gx + P.i11 * gy;
    az[j] = P.i20 * gx + P.i21 * gy;
    pn[j] = gx * gx + gy * gy;
  }
  // The brick flag of the lane-block (its 4 planes lie in one brick: zb is a multiple of 4, a brick's edge of 8 or more) is
  // requested HERE, with the trip's first loads, and acted on once after the last plane.  Looked up where a plane turns
  // out to hold a new negative value, it was a load whose result the very next branch needs: a full drain of the
  // wave's memory queue (the stores of the planes before it included) up to four times a trip.
  unsigned flag_word = 0u;
  bool neg_any = false;
  const int fbit = ((zb >> vp.bshift) * (vp.Y >> vp.bshift) + (y >> vp.bshift)) * (vp.X >> vp.bshift) + (x0 >> vp.bshift);
  if (!COUNT_ONLY && planes != 0u) flag_word = flags[fbit >> 5];
#pragma unroll
  for (int h0 = 0; h0 < 4; h0 += U) {
    bool inr[U];
    bool any_in = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      inr[u] = ((planes >> (h0 + u)) & 1u) != 0u;
      any_in = any_in || inr[u];
    }
    if (__ballot(any_in) == 0ull) continue;
    // 1. the volume vectors of the trip: in flight while the projections run
    uint4 q[U];
    if (!COUNT_ONLY) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        q[u] = make_uint4(0u, 0u, 0u, 0u);
        if (inr[u]) q[u] = vol[VIDX(zb, h0 + u)];
      }
    }
    // 2. projection of the 4U voxel centres (A.4, exact), then their depth gathers
    float D[U][4];  // scaled depth at the voxel's pixel, 0 when it has none; afterwards the observation F
    float gz2[U];
    int pix[U][4];
    // the gathers of the trip's first planes are requested before the later planes are projected: their round trip
    // runs under that arithmetic instead of ahead of a wait
    static_assert(U == 4, "the trip is taken as two halves of two planes");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int u = 2 * half; u < 2 * half + 2; ++u) {
        const float gz = ((float)(vp.zs0 + zb + h0 + u) + 0.5f) * vp.cell[2] - P.tz;
        const float bx = P.i02 * gz, by = P.i12 * gz, bz = P.i22 * gz;
        gz2[u] = gz * gz;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float camz = az[j] + bz;
          const float inv_z = hsk_rcp_exact(camz);
          const float fu = ((ax[j] + bx) * in.fx) * inv_z + in.cx;
          const float fv = ((ay[j] + by) * in.fy) * inv_z + in.cy;
          const int uu = (int)rintf(fu), vv = (int)rintf(fv);
          const bool ok = inr[u] && camz >= 1.17549435e-38f && (unsigned)uu < (unsigned)W && (unsigned)vv < (unsigned)H;
          pix[u][j] = ok ? vv * W + uu : -1;
        }
      }
#pragma unroll
      for (int u = 2 * half; u < 2 * half + 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) D[u][j] = scaled[max(pix[u][j], 0)];
      if (half == 0) asm volatile("" ::: "memory");  // (keeps the first half's gathers ahead of the second half's arithmetic)
    }
    // 3. the observation: F in [-1, 1] for a voxel the rule rewrites, -4 for one it leaves alone
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float Ds = pix[u][j] >= 0 ? D[u][j] : 0.0f;
        const float sdf = Ds - hsk_sqrt_exact(gz2[u] + pn[j]);
        const float f = sdf * vp.tau_inv;
        D[u][j] = (Ds != 0.0f && sdf >= -vp.tau) ? (f < 1.0f ? f : 1.0f) : -4.0f;
      }
    if (COUNT_ONLY) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) cnt += D[u][j] > -2.0f ? 1u : 0u;
      continue;
    }
    // 4. running mean, repack, store
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned w4[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
      unsigned nw[4];
      bool gen[4], gen_any = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool upd = D[u][j] > -2.0f;
        const bool unseen = (w4[j] >> 16) == 0u;
        // free space (F == 1) onto a stored +1, or onto an unseen voxel: the mean is +1 exactly, only the weight moves
        const bool simple = D[u][j] == 1.0f && (unseen || (w4[j] & 0xffffu) == (unsigned)HSK_DIVISOR);
        const unsigned ws = unseen ? (0x10000u | (unsigned)HSK_DIVISOR) : min(w4[j] + 0x10000u, cap);
        nw[j] = (upd && simple) ? ws : w4[j];
        gen[j] = upd && !simple;
        gen_any = gen_any || gen[j];
      }
      bool neg = false;
      if (__ballot(gen_any) != 0ull) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int tp = (int)(short)(w4[j] & 0xffffu);
          const unsigned wp = w4[j] >> 16;
          const float Wp = (float)wp;
          const float Fn = hsk_div_small_exact(hsk_tsdf_unpack(tp) * Wp + D[u][j], Wp + 1.0f);
          int fixed = (int)(Fn * 32767.0f);  // truncation toward zero
          fixed = min(max(fixed, -HSK_DIVISOR), HSK_DIVISOR);
          const unsigned wg = ((unsigned)fixed & 0xffffu) | (min(wp + 1u, (unsigned)HSK_MAX_WEIGHT) << 16);
          nw[j] = gen[j] ? wg : nw[j];
          neg = neg || (gen[j] && fixed < 0);
        }
      }
      // (saturated free space -- +1 at the weight cap -- comes back unchanged: no store)
      if (nw[0] != w4[0] || nw[1] != w4[1] || nw[2] != w4[2] || nw[3] != w4[3])
        vol[VIDX(zb, h0 + u)] = make_uint4(nw[0], nw[1], nw[2], nw[3]);
      neg_any = neg_any || neg;
    }
  }
  if (!COUNT_ONLY && neg_any && ((flag_word >> (fbit & 31)) & 1u) == 0u) mark_brick_negative(flags, vp, x0, y, zb);
  return cnt;
}

#ifdef HSK_PA_TIMING
// timing build (tools/pa_timing.sh): per wave of pass A, s_memrealtime stamps (100 MHz) at the phase boundaries, the
// hardware slot it ran in and what it had to do
__device__ unsigned long long g_pa_times[65536 * 8];
extern "C" int hsk_debug_pa_times(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pa_times), (size_t)n * 8);
}
#define PA_STAMP(k) do { if (!COUNT_ONLY && lane == 0 && pa_wave < 65536u) g_pa_times[pa_wave * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#elif defined(HSK_PA_MARKS)
#define PA_STAMP(k) asm volatile("s_nop 0 ; PA_MARK_" #k ::: "memory")  // static instruction census (tools/pa_census.sh)
#else
#define PA_STAMP(k) do { } while (0)
#endif
// (below, lane predicates are joined by & and |, without short-circuit evaluation: `a && b` on lane-varying conditions is
// compiled into a lane-mask branch round b -- s_and_saveexec / s_cbranch_execz, scalar instructions, which pass A is short of)
// Pass A of integrate (COUNT_ONLY: the same decisions without touching the volume -- V_upd for the roofline).
template <bool COUNT_ONLY, int NS>
__global__ __launch_bounds__(256, NS == 4 ? INTEGRATE_WPE_LONG : INTEGRATE_WPE) void k_integrate(const TrackState* __restrict__ st, const int2* __restrict__ wgz,
                                                   const int2* __restrict__ zint, int zchunk, unsigned gxa, unsigned gmagic,
                                                   double* __restrict__ icp_slot0, unsigned char* __restrict__ uni,
                                                   const float2* __restrict__ dtab, int W, int H, int tw, int th, unsigned gya,
                                                   uint4* __restrict__ vol, const float* __restrict__ scaled, VolParams vp,
                                                   Intr in, unsigned long long* __restrict__ counter,
                                                   unsigned* __restrict__ flags,
                                                   unsigned* __restrict__ queue, unsigned* __restrict__ qcount,
                                                   unsigned qcap, const float2* __restrict__ ftab, int fw, int fh,
                                                   const float2* __restrict__ qtab, IntegrateConst k) {
  // Half of the launch's workgroups lie outside the view frustum, and what they execute before they find that out is a
  // tenth of the kernel's scalar instructions: the test comes FIRST and runs on what arrives with the wave -- the
  // arguments preloaded into SGPRs (the x-block count and its reciprocal for the rotation: `% gridDim.x` was a hidden-
  // argument load and twenty instructions of division) and ONE scalar load, the footprint's z range.
  const unsigned gdx = gxa, gdy = gya;  // the launch's grid (x blocks per row, rows of blocks)
  const unsigned bsum = blockIdx.x + blockIdx.y + blockIdx.z;
  // bsum % gdx: exact for bsum < 2^16 with gmagic = 2^32 / gdx + 1; a single x block (gmagic = 0: the reciprocal does not fit) is 0
  const unsigned bxr = gmagic != 0u ? bsum - __umulhi(bsum, gmagic) * gdx : 0u;
  const unsigned byr = blockIdx.y;  // (rotating the rows with the chunk as well changes nothing: 42.2-42.9 us against 41.7)
  const int zbeg = blockIdx.z * zchunk;
  // (when k_column_zrange has done the frame's last ICP solve: the accumulator slot all its blocks read is emptied here,
  // one launch later, for the next frame's first iteration -- also on a lost frame, hence before the tests below)
  if (!COUNT_ONLY && bsum == 0u) {
    if (icp_slot0)
      for (int i = threadIdx.y * 64 + threadIdx.x; i < ICP_SLOT_DOUBLES; i += 256)
        __hip_atomic_store(icp_slot0 + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  {
    // (zend = min(zbeg + zchunk, nzs) is not needed here: the footprint's range ends below nzs)
    const unsigned long long w2 = *(const unsigned long long*)(wgz + (byr * gdx + bxr));  // (x = low word, y = high word)
    const int wz_x = (int)(unsigned)w2, wz_y = (int)(unsigned)(w2 >> 32);
    if ((zbeg > wz_y) | (zbeg + zchunk - 1 < wz_x)) return;
  }
  const int lane = threadIdx.x;
  if (!COUNT_ONLY && st->lost) return;
#ifdef HSK_PA_TIMING
  const unsigned pa_wave = (((blockIdx.z * gdy + blockIdx.y) * gdx + blockIdx.x) * 4u + threadIdx.y);
  if (!COUNT_ONLY && lane == 0 && pa_wave < 65536u) {
    for (int q = 0; q < 8; ++q) g_pa_times[pa_wave * 8 + q] = 0ull;
    g_pa_times[pa_wave * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }
#endif
  PA_STAMP(0);
  // wave footprint: 64 voxels in x (16 lanes x 16 B = 256 contiguous bytes) by 4 rows in y -- compact, so
  // that the wave-uniform z range and the group classification reject whole planes, not just lanes
  // Which x block a workgroup takes rotates with its row and its chunk (bxr above).  Workgroups go round the eight XCDs in
  // turn, and a volume of 512 voxels has exactly eight blocks per row: unrotated, every workgroup of x block b ran on XCD b,
  // and the frustum covers the middle blocks of a row far more than the outer ones -- the XCDs' loads differed by as much
  // (profiles/r02/raycast_split_experiment.md met the same aliasing).
  const int x0 = (int)(bxr * 16u + (unsigned)(lane & 15)) * 4;
  const int y = (int)(byr * 4u + threadIdx.y) * 4 + (lane >> 4);
  const bool active = (x0 < vp.X) & (y < vp.Y);
  unsigned long long cnt = 0;
  const int zend = min(zbeg + zchunk, vp.nzs);
  // ... and of this wave's own footprint (16 lane columns x 4 rows): the wave-uniform loop bounds, without a wave-wide
  // reduction of the lanes' ranges (min over lanes of max(zl, zbeg) = max(min zl, zbeg)); a wave with nothing to do
  // leaves here, its lanes' ranges never loaded
  const int4 wv = ((const int4*)(wgz + (((size_t)gdx * gdy + 1) & ~(size_t)1)))[(size_t)(byr * gdx + bxr) * 4 +
                                                                                          (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y)];
  const int wl = max(wv.x, zbeg), wh = min(wv.y, zend - 1);
  int zl = 0x7fffffff, zh = -0x7fffffff;  // this lane's stored-plane range inside the padded frustum
  if (wv.z <= zbeg && wv.w >= zend - 1) {
    // every column of the wave covers the whole chunk (the interior of the frustum: most working waves): no lane needs
    // its own range
    if (active) {
      zl = zbeg;
      zh = zend - 1;
    }
  } else if (wl <= wh && active) {
    const int2 zr = zint[(size_t)y * (vp.X / 4) + (x0 >> 2)];  // computed once per frame by k_column_zrange
    zl = max(zr.x, zbeg);
    zh = min(zr.y, zend - 1);
  }
  PA_STAMP(1);
  if (wl <= wh) {
    const float tx = st->t[0], ty = st->t[1];
    // Rinv = R^T
    const float i00 = st->R[0], i01 = st->R[3], i10 = st->R[1], i11 = st->R[4], i20 = st->R[2], i21 = st->R[5];
    const float i02 = st->R[6], i12 = st->R[7], i22 = st->R[8], tz = st->t[2];
    ColumnTerms c;
    const float gy = ((float)y + 0.5f) * vp.cell[1] - ty;
    {
      const float gxc = ((float)x0 + 2.0f) * vp.cell[0] - tx;  // between the lane's second and third voxel
      c.axfc = (i00 * gxc + i01 * gy) * in.fx;
      c.ayfc = (i10 * gxc + i11 * gy) * in.fy;
      c.azc = i20 * gxc + i21 * gy;
      // (gx^2 at the centre is 0.25 cell^2 below the mean of the two middle voxels': the first level's 2.3-cell margin holds it)
      c.pnc = gxc * gxc + gy * gy;
    }
    // Pass A.  A lane's 4(x) x 4(z) block is classified ONCE against the tile table: dead blocks cost nothing
    // more, deep-free-space blocks get four batched vector updates right here (loads in flight together), and the
    // uncertain ones (near a surface, at the frustum rim, close to the camera) are appended to a queue that pass B
    // (k_integrate_detail) walks with the per-voxel path on DENSE waves -- uncertain blocks hug the surfaces and
    // would otherwise drag their whole 64-lane wave through that path (profiles/r01/integrate_analysis.md).
    const int qx = vp.X / 4;
    // The wave's (at most two) groups of 4 planes are taken through the stages TOGETHER: first-level tile lookups of
    // both, second-level lookups of both, free-space loads of both, queue tickets of both.  A wave's life is a chain of
    // dependent memory round trips (tile table -> fine tile table -> volume -> queue counter); stage by stage over both
    // groups the chain is four trips long instead of eight, and pass A is bound by exactly that (its 65 k waves pass
    // through 6 resident slots per SIMD in about eleven rounds).
    static_assert(NS == 2 || NS == 4, "pass A stages the chunk's groups of 4 planes together; a lane's summaries of a chunk are one 16- or 32-bit word");
    const int zb0 = wl & ~3;
    int zbs[NS];
    bool actv[NS], in_all_s[NS], free44_s[NS], other_s[NS];
    float dc_s[NS];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      zbs[sidx] = zb0 + 4 * sidx;
      actv[sidx] = zbs[sidx] <= wh;  // wave-uniform
      free44_s[sidx] = other_s[sidx] = in_all_s[sidx] = false;
      dc_s[sidx] = 0.0f;
    }
    // ---- stage 1: first level (16-px dilated tile table).  Both groups' look-ups are requested first, the lane's summaries
    // behind them: loads return in the order they were issued, and requested ahead of the look-ups the summaries -- an
    // 8 MiB table that misses where the 9.6 KB tile table hits -- made the first level wait for a byte that stage 3 uses.
    float2 Dt_s[NS];
    bool ok_s[NS], in_any_s[NS];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int zb = zbs[sidx];
      in_any_s[sidx] = actv[sidx] & (zb + 3 >= zl) & (zb <= zh) & active;
      in_all_s[sidx] = actv[sidx] & (zb >= zl) & (zb + 3 <= zh) & active;
      const float gz = ((float)(vp.zs0 + zb) + 2.0f) * vp.cell[2] - tz;  // centre of planes zb .. zb+3
      const float czc = c.azc + i22 * gz;
      const float rc = __builtin_amdgcn_rcpf(czc);
      const float uc = (c.axfc + (i02 * gz) * in.fx) * rc + in.cx;
      const float vc = (c.ayfc + (i12 * gz) * in.fy) * rc + in.cy;
      const float r = k.rk4 * rc + 2.5f;
      ok_s[sidx] = (czc > k.zmin4) & (fabsf(uc - k.hw) + r <= k.hw) & (fabsf(vc - k.hh) + r <= k.hh);
      const int tu = min(max((int)uc >> 4, 0), tw - 1), tv = min(max((int)vc >> 4, 0), th - 1);
      Dt_s[sidx] = dtab[tv * tw + tu];
      dc_s[sidx] = __builtin_amdgcn_sqrtf(gz * gz + c.pnc);
    }
    unsigned sum16 = 0u;  // all summaries of the lane's chunk (group g of the chunk in byte g): used in stage 3
    unsigned char* const sum_at = uni + hsk_sum_index_ns<NS>(vp, x0, y, zbeg);
    {
      bool any_actv = false;
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) any_actv = any_actv || actv[sidx];
      if (!COUNT_ONLY && uni != nullptr && active && any_actv) sum16 = NS == 2 ? (unsigned)*(const unsigned short*)sum_at : *(const unsigned*)sum_at;
    }
    PA_STAMP(6);
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const float dc = dc_s[sidx];
      const bool dead4 = ok_s[sidx] & (dc * 0.99999f - Dt_s[sidx].x > k.cull_thr4);
      const bool free44 = in_all_s[sidx] & ok_s[sidx] & (dc * 1.00001f + k.free_thr4 <= Dt_s[sidx].y);
      free44_s[sidx] = free44;
      other_s[sidx] = in_any_s[sidx] & !dead4 & !free44;
    }
    unsigned sum8[NS];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) sum8[sidx] = actv[sidx] ? (sum16 >> (8 * ((zbs[sidx] - zbeg) >> 2))) & 0xffu : 0u;
    unsigned new16 = sum16;
    PA_STAMP(2);
    // ---- stage 2: second level for the still undecided lanes: the 16 voxel centres span a parallelogram in camera
    //      space, whose projection is a convex quadrilateral, so the pixel box of the four projected corners
    //      (+-1 px for rounding) holds all 16 pixels; its exact min / max depth comes from the undilated 8-px
    //      tile table (<= 3x3 tiles).  Every block decided here is one less entry for pass B.
    // (the z-invariant terms of the lane's first and last voxel: only waves with an undecided lane come here)
    float cax[2], cay[2], caz[2], pn_lo, pn_hi;
    {
      const float gx0 = ((float)x0 + 0.5f) * vp.cell[0] - tx, gx3 = ((float)(x0 + 3) + 0.5f) * vp.cell[0] - tx;
      cax[0] = i00 * gx0 + i01 * gy; cax[1] = i00 * gx3 + i01 * gy;
      cay[0] = i10 * gx0 + i11 * gy; cay[1] = i10 * gx3 + i11 * gy;
      caz[0] = i20 * gx0 + i21 * gy; caz[1] = i20 * gx3 + i21 * gy;
      const float p0 = gx0 * gx0 + gy * gy, p3 = gx3 * gx3 + gy * gy;
      pn_hi = fmaxf(p0, p3);  // smallest / largest gx^2 + gy^2 over the lane's x range (not only at its 4 centres)
      pn_lo = (gx0 <= 0.0f && gx3 >= 0.0f) ? gy * gy : fminf(p0, p3);
    }
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      if (!actv[sidx] || __ballot(other_s[sidx]) == 0ull) continue;
      const int zb = zbs[sidx];
      const float gza = ((float)(vp.zs0 + zb) + 0.5f) * vp.cell[2] - tz;
      const float gzb = ((float)(vp.zs0 + zb + 3) + 0.5f) * vp.cell[2] - tz;
      float umin = 1e30f, umax = -1e30f, vmin = 1e30f, vmax = -1e30f, zmn = 1e30f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = q & 1;
        const float gq = (q & 2) ? gzb : gza;
        const float cz = caz[j] + i22 * gq;
        const float rq = __builtin_amdgcn_rcpf(cz);
        const float uq = ((cax[j] + i02 * gq) * in.fx) * rq + in.cx;
        const float vq = ((cay[j] + i12 * gq) * in.fy) * rq + in.cy;
        zmn = fminf(zmn, cz);
        umin = fminf(umin, uq);
        umax = fmaxf(umax, uq);
        vmin = fminf(vmin, vq);
        vmax = fmaxf(vmax, vq);
      }
      umin -= 1.0f; vmin -= 1.0f; umax += 1.0f; vmax += 1.0f;
      // the 4-px table when the box spans at most 3 x 3 of its tiles, else the 8-px one (at most 3 x 3 again, else undecided)
      const bool in_img = (zmn > 0.05f) & (umin >= 0.0f) & (vmin >= 0.0f) & (umax <= (float)(W - 1)) & (vmax <= (float)(H - 1));
      const int iu0 = (int)umin, iv0 = (int)vmin, iu1 = (int)umax, iv1 = (int)vmax;
      const bool fine = ((iu1 >> 2) <= (iu0 >> 2) + 2) & ((iv1 >> 2) <= (iv0 >> 2) + 2);
      const int sh = fine ? 2 : 3;
      const int tu0 = iu0 >> sh, tv0 = iv0 >> sh;
      const int nx = (iu1 >> sh) - tu0, ny = (iv1 >> sh) - tv0;  // tiles spanned, less one
      const bool ok2 = in_img & (nx <= 2) & (ny <= 2);
      const float2* __restrict__ tab = fine ? qtab : ftab;
      const int tbw = fine ? 2 * fw : fw, tbh = fine ? 2 * fh : fh;
      // (one look-up: table (nx, ny) holds the (max, min) of the nx x ny tiles from each tile on, k_tile_window)
      const int shape = min(max(ny, 0), 2) * 3 + min(max(nx, 0), 2);
      const float2 t9 = tab[(size_t)shape * tbw * tbh + min(max(tv0, 0), tbh - 1) * tbw + min(max(tu0, 0), tbw - 1)];
      const float Dx = t9.x, Dn = t9.y;
      // exact distance range of the block: its 16 voxel centres lie in the rectangle [gx0, gx3] x {gy} x [gza, gzb], over
      // which the distance to the camera centre is largest at a corner and smallest where each coordinate is nearest 0
      const float gz2_hi = fmaxf(gza * gza, gzb * gzb);
      const float gz2_lo = (gza <= 0.0f && gzb >= 0.0f) ? 0.0f : fminf(gza * gza, gzb * gzb);
      const float d_hi = __builtin_amdgcn_sqrtf(pn_hi + gz2_hi), d_lo = __builtin_amdgcn_sqrtf(pn_lo + gz2_lo);
      const bool dead2 = ok2 & (d_lo * 0.99999f - Dx > k.cull_thr2);
      const bool free2 = in_all_s[sidx] & ok2 & (d_hi * 1.00001f + k.free_thr2 <= Dn);
      free44_s[sidx] = free44_s[sidx] | (other_s[sidx] & free2);
      other_s[sidx] = other_s[sidx] & !dead2 & !free2;
    }
    PA_STAMP(3);
#ifdef HSK_PA_TIMING
    if (!COUNT_ONLY && pa_wave < 65536u) {
      const unsigned nf = (unsigned)__popcll(__ballot(free44_s[0] && actv[0])) + (unsigned)__popcll(__ballot(free44_s[1] && actv[1]));
      const unsigned no = (unsigned)__popcll(__ballot(other_s[0] && actv[0])) + (unsigned)__popcll(__ballot(other_s[1] && actv[1]));
      if (lane == 0) g_pa_times[pa_wave * 8 + 7] = nf | ((unsigned long long)no << 32);
    }
#endif
    // ---- stage 4: wave-aggregated append of the uncertain lane-blocks: one of HSK_NQUEUES queues (a single counter
    // saturates at ~88 atomics/us chip-wide).  The queue must NOT follow the block's x-y position: surfaces cluster in a
    // few columns, and pass B's time is its longest queue.  Rotate the assignment by the row of HSK_NQUEUES blocks and by
    // the wave: within a row it stays a bijection, so every queue still receives at most one wave-quarter of one block
    // per row and wave index (the capacity bound).  Both groups' tickets are requested before either is used.
    // (the tickets are REQUESTED here, before the free-space loads, and used after the stores: the counters' round trip
    // runs under the volume's -- one dependent round trip less in a wave's life)
    const unsigned lin = (blockIdx.z * gdy + byr) * gdx + bxr;
    const unsigned qi = (lin + (lin / HSK_NQUEUES) * 37u + threadIdx.y * (HSK_NQUEUES / 4)) % HSK_NQUEUES;
    // (ONE ticket for the wave's groups: every vector-memory instruction counts, see the note on k_tile_window)
    unsigned long long bo[NS];
    unsigned n_other = 0u, base_all = 0u;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      bo[sidx] = actv[sidx] ? __ballot(other_s[sidx]) : 0ull;
      n_other += (unsigned)__popcll(bo[sidx]);
    }
    if (n_other != 0u && lane == 0) base_all = atomicAdd(&qcount[qi * HSK_QCOUNT_STRIDE], n_other);
    // ---- stage 3: deep free space -- four batched vector updates per group, the loads of both groups in flight together
    if (COUNT_ONLY) {
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx)
        if (actv[sidx] && free44_s[sidx]) cnt += 16;
    } else {
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      auto store_vec = [&](size_t at, const uint4& q) {
        if (vp.stream_nt) {
          // a volume far larger than the 256 MiB Infinity Cache gains nothing from caching these lines and loses
          // what they evict: non-temporal loads and stores (1024^3: 838 -> 802 us; at 512^3, where half the volume
          // stays cached from frame to frame, they cost 10 us, so the host decides by size)
          const v4u t = {q.x, q.y, q.z, q.w};
          __builtin_nontemporal_store(t, (v4u*)&vol[at]);
        } else {
          vol[at] = q;
        }
      };
      // (a) what needs no load.  Free-space blocks: state 1 -> (+1, 1) stored; uniform or rim blocks -> the byte moves.
      //     Blocks on their way to pass B: a uniform one gets its weights stored, every one loses its summary.
      bool rd[NS];
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) {
        const bool fr = actv[sidx] & free44_s[sidx], ot = actv[sidx] & other_s[sidx];
        const unsigned sm = sum8[sidx];
        const int sbit = 8 * ((zbs[sidx] - zbeg) >> 2);  // where the group's summary sits in new16
        unsigned wstore = 0u;  // weight to store into all 16 voxels (0: none)
        if (fr && sm == 1u) wstore = 1u;
        if (ot && sm >= 2u && sm < HSK_SUM_RAGGED) wstore = sm - 1u;
        if (wstore != 0u) {
          const unsigned word = (wstore << 16) | (unsigned)HSK_DIVISOR;
          const uint4 q = make_uint4(word, word, word, word);
#pragma unroll
          for (int u = 0; u < 4; ++u) store_vec(VIDX(zbs[sidx], u), q);
        }
        // the byte: +1 for a free block in states 1 .. 128 (w + 1 <= 128) and 130 .. 254 (one more pending); 129 stays;
        // 0 for a block on its way to pass B (the word is stored once, after the read path below)
        {
          const bool tick = fr && sm != 0u && sm != (unsigned)HSK_MAX_WEIGHT + 1u && sm != HSK_SUM_MAX;
          if (tick || (ot && sm != 0u)) new16 = (new16 & ~(0xffu << sbit)) | ((tick ? sm + 1u : 0u) << sbit);
        }
        // (b) needs the words: a free block in state 0, or a rim block whose pending count is full; a rim block with
        //     pending observations on its way to pass B
        rd[sidx] = (fr && (sm == 0u || sm == HSK_SUM_MAX)) || (ot && sm > HSK_SUM_RAGGED);
      }
      // (one group at a time: this path is the exception now, and four vectors live instead of eight keep the kernel at
      // seven waves per SIMD)
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) {
        if (__ballot(rd[sidx]) == 0ull) continue;
        uint4 q4[4];
        if (rd[sidx]) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (vp.stream_nt) {
              const v4u t = __builtin_nontemporal_load((const v4u*)&vol[VIDX(zbs[sidx], u)]);
              q4[u] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
              q4[u] = vol[VIDX(zbs[sidx], u)];
            }
          }
          const bool fr = actv[sidx] && free44_s[sidx];
          const unsigned sm = sum8[sidx];
          if (sm == 0u) {  // (free) the update rule on whatever the block holds
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (update_vector_free4(q4[u])) store_vec(VIDX(zbs[sidx], u), q4[u]);
          } else {  // the pending observations (and this frame's, for a free block) onto weights under +1
            const unsigned p = sm - HSK_SUM_RAGGED + (fr ? 1u : 0u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              hsk_vector_add_weight(q4[u], p);
              store_vec(VIDX(zbs[sidx], u), q4[u]);
            }
          }
          if (fr) {
            const int sbit = 8 * ((zbs[sidx] - zbeg) >> 2);
            new16 = (new16 & ~(0xffu << sbit)) | (hsk_sum_classify(q4) << sbit);
          }
        }
      }
      if (new16 != sum16) {
        if (NS == 2)
          *(unsigned short*)sum_at = (unsigned short)new16;
        else
          *(unsigned*)sum_at = new16;
      }
    }
#ifdef HSK_PA_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stamp below sees the loads back and the stores acknowledged
#endif
    PA_STAMP(4);
    unsigned b0 = n_other != 0u ? (unsigned)__builtin_amdgcn_readfirstlane((int)base_all) : 0u;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      if (bo[sidx] == 0ull) continue;
      if (other_s[sidx]) {
        // the entry: lane-block id, and (when it fits: id_mask_shift != 0) which of its 4 planes lie in the lane's z range
        unsigned pm = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) pm |= ((zbs[sidx] + u >= zl && zbs[sidx] + u <= zh) ? 1u : 0u) << u;
        const unsigned id = (unsigned)(((zbs[sidx] >> 2) * vp.Y + y) * qx + (x0 >> 2));
        queue[(size_t)qi * qcap + b0 + (unsigned)__popcll(bo[sidx] & ((1ull << lane) - 1ull))] = id | (pm << 28);
      }
      b0 += (unsigned)__popcll(bo[sidx]);
    }
  }
  PA_STAMP(5);
  if (COUNT_ONLY) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (lane == 0 && cnt) atomicAdd(counter, cnt);
  }
}

// ------------------------------------------------------------------------------------------------------
// Pass B (second form, round 2): the queued lane-blocks, per-voxel path -- built round three facts measured on round 1's
// form (profiles/r01/final_pmc_summary.txt, profiles/r02/integrate_analysis.md):
// it spent half its wave cycles parked on memory and a quarter of its instructions on branches and lane-mask
// bookkeeping (171 basic blocks, 535 s_and/s_or, 37 spilled SGPRs), all of it in service of "approximate first, exact
// where a decision is close".  With hsk_rcp_exact / hsk_sqrt_exact / hsk_div_small_exact the specification's correctly
// rounded operations cost 3 / 5 / 6 instructions, so every voxel simply takes the exact path: no decision-boundary
// tests, no fallback loops.  The U planes of a trip issue their volume vectors FIRST (their addresses need only the
// queue entry), then the 4U depth gathers, so both round trips overlap the projection arithmetic.  Two wave-uniform
// branches remain per plane: the general running mean (skipped when every rewritten voxel of the wave is free space
// onto a stored +1 or an unseen voxel: saturating add on the packed word) and the store (skipped when nothing changed).
// Domain of the shortcuts: hsk_rcp_exact needs a normal camz -- the specification itself asks for camz >= FLT_MIN
// (deviation D6: a voxel whose camera-space depth is a denormal number counts as not in front of the camera); a huge
// reciprocal makes the pixel coordinate overflow the image bounds, as the specification's 1e6 guard does.  The root
// of a value below 2^-102 is inexact but finite and far below half an ulp of any non-zero depth, so sdf is unaffected;
// the root of 0 (voxel centre ON the camera centre: NaN from the shortcut) belongs to a voxel with camz = 0.
// ------------------------------------------------------------------------------------------------------
#ifndef DETAIL2_U
#define DETAIL2_U 4
#endif
#ifndef DETAIL2_WPE
#define DETAIL2_WPE 5
#endif
#ifndef DETAIL2_GX
#define DETAIL2_GX 5  // DETAIL2_GX x 256 queues x 4 waves: one resident round of the chip at 8 waves per SIMD
#endif

#ifdef HSK_PB_TIMING
// timing build (tools/pb_timing.sh): per wave of pass B, s_memrealtime stamps: start, prologue done, after each trip (up to 4)
__device__ unsigned long long g_pb_times[8192 * 8];
extern "C" int hsk_debug_pb_times(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pb_times), (size_t)n * 8);
}
#define PB_STAMP(k) do { if (!COUNT_ONLY && lane == 0 && pb_wave < 8192u) g_pb_times[pb_wave * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PB_STAMP(k) do { } while (0)
#endif
template <bool COUNT_ONLY>
__global__ __launch_bounds__(256, DETAIL2_WPE) void k_integrate_detail2(const TrackState* __restrict__ st,
                                                                        const unsigned* __restrict__ qcount,
                                                                        const unsigned* __restrict__ queue_all, unsigned qcap, int W,
                                                                        int H, uint4* __restrict__ vol,
                                                                        const float* __restrict__ scaled, Intr in, VolParams vp,
                                                                        unsigned long long* __restrict__ counter,
                                                                        unsigned* __restrict__ flags) {
  if (!COUNT_ONLY && st->lost) return;
  const int lane = threadIdx.x & 63;
#ifdef HSK_PB_TIMING
  const unsigned pb_wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (!COUNT_ONLY && lane == 0 && pb_wave < 8192u)
    for (int q = 0; q < 8; ++q) g_pb_times[pb_wave * 8 + q] = 0ull;
  int pb_trip = 0;
#endif
  PB_STAMP(0);
  // The HSK_NQUEUES queues are walked as ONE list (their lengths differ by 1.6x: a grid that strides over each queue
  // by itself ends with the longest queue's last round, a third of the chip idle).  Every block scans the 256 counters
  // once (LDS prefix array); an entry's queue is then found by bisection.
  __shared__ unsigned pre[HSK_NQUEUES + 1];
  {
    static_assert(HSK_NQUEUES == 256, "one counter per thread of the block");
    const unsigned c = qcount[threadIdx.x * HSK_QCOUNT_STRIDE];
    unsigned incl = c;  // inclusive scan inside the wave, then across the four waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned v = (unsigned)__shfl_up((int)incl, o, 64);
      if (lane >= o) incl += v;
    }
    __shared__ unsigned wsum[4];
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned base = 0;
    for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) base += wsum[w];
    pre[threadIdx.x + 1] = base + incl;
    if (threadIdx.x == 0) pre[0] = 0u;
    __syncthreads();
  }
  const unsigned n = pre[HSK_NQUEUES];
  const unsigned stride = gridDim.x * blockDim.x;
  unsigned long long cnt = 0;
  PB_STAMP(1);
  const DetailPose P = detail_pose(st);
  const int qx = vp.X / 4;
  auto entry_at = [&](unsigned g) -> unsigned {  // g-th entry of the concatenated queues (0 beyond the end)
    if (g >= n) return 0u;
    unsigned lo = 0, hi = HSK_NQUEUES;  // pre[lo] <= g < pre[hi]
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const unsigned mid = (lo + hi) >> 1;
      if (pre[mid] <= g) lo = mid; else hi = mid;
    }
    return queue_all[(size_t)lo * qcap + (g - pre[lo])];
  };
  // the entry of the NEXT trip is fetched while the current one is worked on (its address needs nothing but the trip index)
  unsigned e0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u);
  unsigned id_next = entry_at(e0 + lane);
  for (; e0 < n; e0 += stride) {  // wave-uniform trip count
    const unsigned id = id_next;
    id_next = entry_at(e0 + stride + lane);
    const unsigned lb = id & 0x0fffffffu;
    const int x0 = (int)(lb % (unsigned)qx) * 4, y = (int)((lb / (unsigned)qx) % (unsigned)vp.Y);
    const int zb = (int)(lb / ((unsigned)qx * (unsigned)vp.Y)) * 4;
    cnt += detail_entry<COUNT_ONLY, DETAIL2_U>(id >> 28, x0, y, zb, vol, scaled, P, vp, W, H, in, flags);
#ifdef HSK_PB_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pb_trip < 5) PB_STAMP(2 + pb_trip);
    ++pb_trip;
#endif
  }
  if (COUNT_ONLY) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (lane == 0 && cnt) atomicAdd(counter, cnt);
  }
}

// fine (8-px) tile table: depends only on the depth frame, so it belongs to the preprocessing
void launch_tile_fine(hipStream_t s, const float* scaled, int W, int H, float* tiles) {
  const int tw = (W + HSK_TILE - 1) / HSK_TILE, th = (H + HSK_TILE - 1) / HSK_TILE;
  const int fw = (W + HSK_FTILE - 1) / HSK_FTILE, fh = (H + HSK_FTILE - 1) / HSK_FTILE;
  float2* ftab = (float2*)(tiles + 4 * tw * th);
  float2* qtab = ftab + (size_t)fw * fh;
  hipLaunchKernelGGL(k_tile_fine, dim3((fw * fh + 255) / 256), dim3(256), 0, s, scaled, W, H, ftab, fw, fh, qtab);
  // behind them their window forms (what pass A reads): nine shapes of the 8-px table, then nine of the 4-px one
  float2* fwin = qtab + (size_t)4 * fw * fh;
  hipLaunchKernelGGL(k_tile_window, dim3((fw * fh + 255) / 256, 9), dim3(256), 0, s, ftab, fw, fh, fwin);
  hipLaunchKernelGGL(k_tile_window, dim3((4 * fw * fh + 255) / 256, 9), dim3(256), 0, s, qtab, 2 * fw, 2 * fh, fwin + (size_t)9 * fw * fh);
}

// entries of the z-range tables: one int2 per lane column, then one per pass-A workgroup footprint (64 x 16 voxels), then
// an int4 per wave footprint (64 x 4 voxels: four per workgroup)
size_t integrate_zint_entries(const VolParams& vp) {
  const size_t fp = (size_t)((vp.X + 63) / 64) * ((vp.Y + 15) / 16);
  return (size_t)(vp.X / 4) * vp.Y + ((fp + 1) & ~(size_t)1) + 8 * fp;  // (the wave table is int4: kept 16-B aligned)
}
size_t integrate_queue_counter_words() { return (size_t)HSK_NQUEUES * HSK_QCOUNT_STRIDE; }
unsigned long long integrate_queue_entries(const unsigned* counter_words) {
  unsigned long long n = 0;
  for (size_t q = 0; q < HSK_NQUEUES; ++q) n += counter_words[q * HSK_QCOUNT_STRIDE];
  return n;
}
// words of the pass A -> pass B queues: HSK_NQUEUES counters (one 256-B line each) + HSK_NQUEUES queues
size_t integrate_queue_words(const VolParams& vp) {
  const int zchunk = vp.nzs >= vp.zchunk ? vp.zchunk : vp.nzs;
  const size_t nblk = (size_t)((vp.X + 63) / 64) * ((vp.Y + 15) / 16) * ((vp.nzs + zchunk - 1) / zchunk);
  const size_t qcap = ((nblk + HSK_NQUEUES - 1) / HSK_NQUEUES) * 256u * (size_t)((zchunk + 3) / 4);
  return (size_t)HSK_NQUEUES * HSK_QCOUNT_STRIDE + (size_t)HSK_NQUEUES * qcap;
}

void launch_integrate(hipStream_t s, void* vol, const float* scaled, const TrackState* st, const VolParams& vp, int W,
                      int H, Intr in, bool count_only, unsigned long long* counter, unsigned* flags,
                      const float* tmax, int2* zint, unsigned* queue, const IcpFinal* icp_final, unsigned char* uni, const RingOut* early) {
  const int zchunk = vp.nzs >= vp.zchunk ? vp.zchunk : vp.nzs;
  const int zchunks = (vp.nzs + zchunk - 1) / zchunk;
  const int tw = (W + HSK_TILE - 1) / HSK_TILE, th = (H + HSK_TILE - 1) / HSK_TILE;
  const int ncols = (vp.X / 4) * vp.Y;
  const int fw = (W + HSK_FTILE - 1) / HSK_FTILE, fh = (H + HSK_FTILE - 1) / HSK_FTILE;
  // behind the coarse tables (filled by launch_tile_fine): 8-px table, 4-px table, then their window forms (nine shapes
  // each) -- pass A reads the window forms
  const float2* ftab = (const float2*)(tmax + 4 * tw * th) + (size_t)5 * fw * fh;
  const float2* qtab = ftab + (size_t)9 * fw * fh;
  const int col_blocks = ((vp.X + 63) / 64) * ((vp.Y + 15) / 16), dil_blocks = (tw * th + 255) / 256;  // one block per pass-A footprint
  unsigned* qcount = queue;  // HSK_NQUEUES counters, one per 256-B line, cleared by k_column_zrange
  const IcpFinal none = {nullptr, nullptr, 0};
  const IcpFinal fin = (icp_final && !count_only) ? *icp_final : none;
  int2* wgz = zint + ncols;  // behind the column table: one entry per pass-A workgroup footprint (integrate_zint_entries)
  const RingOut quiet_ring = {nullptr, nullptr, nullptr};
  const RingOut early_ring = (early && fin.slots && !count_only) ? *early : quiet_ring;
  hipLaunchKernelGGL(k_column_zrange, dim3(col_blocks > dil_blocks ? col_blocks : dil_blocks), dim3(256), 0, s, fin, tmax,
                     tmax + tw * th, (float2*)(tmax + 2 * tw * th), tw, th, dil_blocks, qcount, st, vp, W, H, in, zint,
                     const_cast<TrackState*>(st), wgz, early_ring);
  dim3 block(64, 4, 1);
  dim3 grid((vp.X + 63) / 64, (vp.Y + 15) / 16, zchunks);
  const float2* dil = (const float2*)(tmax + 2 * tw * th);
  // behind the counters: HSK_NQUEUES queues of qcap entries each; a block of pass A holds at most 4 waves x 64 lanes
  // x (zchunk / 4) blocks and every HSK_NQUEUES-th block shares a queue
  unsigned* qdata = queue + HSK_NQUEUES * HSK_QCOUNT_STRIDE;
  const unsigned nblk = grid.x * grid.y * (unsigned)zchunks;
  const unsigned qcap = ((nblk + HSK_NQUEUES - 1) / HSK_NQUEUES) * 256u * (unsigned)((zchunk + 3) / 4);
  const IntegrateConst kc = integrate_const(vp, W, H, in);
  // n % grid.x = n - mulhi(n, gmagic) * grid.x for n < 2^16; 0 for a single x block (the kernel takes 0 for the remainder)
  const unsigned gmagic = grid.x > 1u ? (unsigned)(0x100000000ull / grid.x) + 1u : 0u;
  const dim3 detail_grid(DETAIL2_GX * HSK_NQUEUES);  // one resident round of the chip, striding over the concatenated queues
  if (count_only) {
    if (vp.zchunk == 16)
      hipLaunchKernelGGL((k_integrate<true, 4>), grid, block, 0, s, st, wgz, zint, zchunk, grid.x, gmagic, (double*)nullptr, (unsigned char*)nullptr, dil,
                         W, H, tw, th, grid.y, (uint4*)vol, scaled, vp, in, counter, flags, qdata, qcount, qcap, ftab, fw, fh, qtab, kc);
    else
      hipLaunchKernelGGL((k_integrate<true, 2>), grid, block, 0, s, st, wgz, zint, zchunk, grid.x, gmagic, (double*)nullptr, (unsigned char*)nullptr, dil,
                         W, H, tw, th, grid.y, (uint4*)vol, scaled, vp, in, counter, flags, qdata, qcount, qcap, ftab, fw, fh, qtab, kc);
    hipLaunchKernelGGL(k_integrate_detail2<true>, detail_grid, dim3(256), 0, s, st, qcount, qdata, qcap, W, H, (uint4*)vol, scaled, in, vp,
                       counter, flags);
  } else {
    if (vp.zchunk == 16)
      hipLaunchKernelGGL((k_integrate<false, 4>), grid, block, 0, s, st, wgz, zint, zchunk, grid.x, gmagic, fin.slots, uni, dil, W, H, tw, th, grid.y,
                         (uint4*)vol, scaled, vp, in, counter, flags, qdata, qcount, qcap, ftab, fw, fh, qtab, kc);
    else
      hipLaunchKernelGGL((k_integrate<false, 2>), grid, block, 0, s, st, wgz, zint, zchunk, grid.x, gmagic, fin.slots, uni, dil, W, H, tw, th, grid.y,
                         (uint4*)vol, scaled, vp, in, counter, flags, qdata, qcount, qcap, ftab, fw, fh, qtab, kc);
    hipLaunchKernelGGL(k_integrate_detail2<false>, detail_grid, dim3(256), 0, s, st, qcount, qdata, qcap, W, H, (uint4*)vol, scaled, in, vp,
                       counter, flags);
  }
}

// rebuild the brick bitfield from a volume that was uploaded rather than integrated
__global__ void k_rebuild_flags(const short2* __restrict__ vol, VolParams vp, unsigned* __restrict__ flags) {
  const size_t n = (size_t)vp.X * vp.Y * vp.nzs;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % vp.X), y = (int)((i / vp.X) % vp.Y), zz = (int)(i / ((size_t)vp.X * vp.Y));
  if (vol[hsk_vox_index(vp, x, y, zz)].x < 0) mark_brick_negative(flags, vp, x, y, zz);
}
// stored planes [zz0, zz0 + nz) between the volume's 64-B blocks and a row-major array (x fastest, then y, then plane):
// the host's view of the volume (hsk_download_tsdf / hsk_upload_tsdf).  One thread per 16-B vector.
template <bool TO_LINEAR>
__global__ void k_vol_convert(uint4* __restrict__ vol, VolParams vp, int zz0, int nz, uint4* __restrict__ lin) {
  const int qx = vp.X >> 2;
  const size_t n = (size_t)qx * vp.Y * nz;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xl = (int)(i % qx), y = (int)((i / qx) % vp.Y), zr = (int)(i / ((size_t)qx * vp.Y));
  const size_t at = hsk_vox_index(vp, xl * 4, y, zz0 + zr) >> 2;
  if (TO_LINEAR)
    lin[i] = vol[at];
  else
    vol[at] = lin[i];
}
void launch_vol_to_linear(hipStream_t s, const void* vol, const VolParams& vp, int zz0, int nz, void* lin) {
  const size_t n = (size_t)(vp.X >> 2) * vp.Y * nz;
  hipLaunchKernelGGL(k_vol_convert<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint4*)const_cast<void*>(vol), vp, zz0, nz, (uint4*)lin);
}
void launch_vol_from_linear(hipStream_t s, void* vol, const VolParams& vp, int zz0, int nz, const void* lin) {
  const size_t n = (size_t)(vp.X >> 2) * vp.Y * nz;
  hipLaunchKernelGGL(k_vol_convert<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint4*)vol, vp, zz0, nz, (uint4*)const_cast<void*>(lin));
}
void launch_rebuild_flags(hipStream_t s, const void* vol, const VolParams& vp, unsigned* flags) {
  const size_t n = (size_t)vp.X * vp.Y * vp.nzs;
  hipLaunchKernelGGL(k_rebuild_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const short2*)vol, vp, flags);
}

// lane-block summaries of a volume that was uploaded rather than integrated (one thread per lane-block; a block that
// reaches beyond the last stored plane has no summary), and the reverse: the volume's weights brought up to date
size_t uniform_bytes(const VolParams& vp) {
  return (size_t)((vp.nzs + vp.zchunk - 1) / vp.zchunk) * ((vp.Y + 3) / 4) * ((vp.X + 63) / 64) * (size_t)(64 * (vp.zchunk >> 2));
}
template <bool MATERIALIZE>
__global__ void k_summaries(uint4* __restrict__ vol, VolParams vp, unsigned char* __restrict__ uni) {
  const int qx = vp.X / 4;
  const size_t n = (size_t)((vp.nzs + 3) / 4) * vp.Y * qx;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xl = (int)(i % qx), y = (int)((i / qx) % vp.Y), zb = (int)(i / ((size_t)qx * vp.Y)) * 4;
  const size_t ui = hsk_sum_index(vp, xl * 4, y, zb);
  const int x0 = xl * 4;
  if (!MATERIALIZE) {
    unsigned code = 0u;
    if (zb + 3 < vp.nzs) {
      uint4 q[4];
      for (int u = 0; u < 4; ++u) q[u] = vol[VIDX(zb, u)];
      code = hsk_sum_classify(q);
    }
    uni[ui] = (unsigned char)code;
    return;
  }
  const unsigned sm = uni[ui];
  if (sm >= 2u && sm < HSK_SUM_RAGGED) {
    const unsigned word = ((sm - 1u) << 16) | (unsigned)HSK_DIVISOR;
    for (int u = 0; u < 4; ++u) vol[VIDX(zb, u)] = make_uint4(word, word, word, word);
  } else if (sm > HSK_SUM_RAGGED) {
    uint4 q[4];
    for (int u = 0; u < 4; ++u) q[u] = vol[VIDX(zb, u)];
    for (int u = 0; u < 4; ++u) {
      hsk_vector_add_weight(q[u], sm - HSK_SUM_RAGGED);
      vol[VIDX(zb, u)] = q[u];
    }
    uni[ui] = (unsigned char)hsk_sum_classify(q);
  }
}
void launch_rebuild_uniform(hipStream_t s, const void* vol, const VolParams& vp, unsigned char* uni) {
  const size_t n = (size_t)((vp.nzs + 3) / 4) * vp.Y * (vp.X / 4);
  hipLaunchKernelGGL(k_summaries<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint4*)const_cast<void*>(vol), vp, uni);
}
void launch_materialize(hipStream_t s, void* vol, const VolParams& vp, unsigned char* uni) {
  const size_t n = (size_t)((vp.nzs + 3) / 4) * vp.Y * (vp.X / 4);
  hipLaunchKernelGGL(k_summaries<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint4*)vol, vp, uni);
}

// ------------------------------------------------------------------------------------------------------
// raycast (A.6).  One ray per lane; a wave covers an 8x8 pixel tile so that neighbouring rays walk
// neighbouring voxels (L1/L2 locality of the 4-B gathers).  Steps are owned by the slab that contains the
// far sample's z plane; a single-device context owns all of them.
// ------------------------------------------------------------------------------------------------------
// voxel index from the quotient q = p / cell (floor, with the spec's range guards)
static __device__ __forceinline__ int vox_of_q(float quot) {
  const float q = floorf(quot);
  if (!(q >= 0.0f)) return -1;
  if (q > 1.0e6f) return 1000000;
  return (int)q;
}

static __device__ __forceinline__ int raw_at(const short2* __restrict__ vol, const VolParams& vp, int x, int y, int z) {
  const int zz = z - vp.zs0;
  if (zz < 0 || zz >= vp.nzs) return 0;
  // (block row and pitch are below 2^24 each: one 24-bit multiply-add forms the row, one widening multiply-add the index --
  // hsk_vox_index in the fewest instructions: these sit on the march's gather chain)
  const unsigned row = __umul24((unsigned)zz >> 2, (unsigned)vp.Y) + (unsigned)y;
  const unsigned low = ((((unsigned)x & ~3u) | ((unsigned)zz & 3u)) << 2) | ((unsigned)x & 3u);
  return (int)vol[(size_t)row * (unsigned)((vp.X >> 2) << 4) + low].x;
}

// trilinear TSDF sample (A.6).  Branch-free: indices are clamped for the loads and the NaN of the spec
// (sample on the outer shell of the grid) is selected at the end, so that the 8 taps of several calls can be
// in flight together.
static __device__ __forceinline__ float trilinear(const short2* __restrict__ vol, const VolParams& vp, float px, float py,
                                                  float pz) {
  // floor(p / cell) and the fractional offsets below are the spec's f32 quotients, obtained as binary64 products
  // (hsk_div_by_const): 3 instructions each instead of a ~10-instruction correctly rounded division
  int gx = vox_of_q(hsk_div_by_const(px, vp.icell[0])), gy = vox_of_q(hsk_div_by_const(py, vp.icell[1])),
      gz = vox_of_q(hsk_div_by_const(pz, vp.icell[2]));
  const bool ok = gx > 0 && gx < vp.X - 1 && gy > 0 && gy < vp.Y - 1 && gz > 0 && gz < vp.Z - 1;
  gx = min(max(gx, 1), vp.X - 2);
  gy = min(max(gy, 1), vp.Y - 2);
  gz = min(max(gz, 1), vp.Z - 2);
  if (px < ((float)gx + 0.5f) * vp.cell[0]) gx -= 1;
  if (py < ((float)gy + 0.5f) * vp.cell[1]) gy -= 1;
  if (pz < ((float)gz + 0.5f) * vp.cell[2]) gz -= 1;
  const float a = hsk_div_by_const(px - ((float)gx + 0.5f) * vp.cell[0], vp.icell[0]);
  const float b = hsk_div_by_const(py - ((float)gy + 0.5f) * vp.cell[1], vp.icell[1]);
  const float c = hsk_div_by_const(pz - ((float)gz + 0.5f) * vp.cell[2], vp.icell[2]);
  // stored planes: a tap outside the slab reads plane 0 of the slab and is discarded (cannot happen when the
  // halo is sized as DESIGN.md prescribes)
  const int z0 = gz - vp.zs0, z1 = z0 + 1;
  const bool in0 = z0 >= 0 && z0 < vp.nzs, in1 = z1 >= 0 && z1 < vp.nzs;
  // (the index is a sum of one term per axis: two terms per axis, eight additions; the two z taps of a cell share a
  // 64-B block three times out of four)
  // the upper neighbours' terms by steps from the lower ones: +1 word in x (or to the next block: +13), one row pitch in y,
  // +4 words in z (or to the next block row of planes: + the plane-group pitch - 12); a z tap outside the stored planes
  // reads plane 0 (term 0: z0 = -1 gives z1 = 0) and is discarded
  const size_t pitch = (size_t)((vp.X >> 2) << 4);
  const size_t tx0 = hsk_vox_xterm(gx), tx1 = tx0 + ((gx & 3) == 3 ? 13u : 1u);
  const size_t ty0 = (size_t)gy * pitch, ty1 = ty0 + pitch;
  const size_t tz0 = in0 ? hsk_vox_zterm(vp, z0) : 0;
  const size_t tz1 = (in0 && in1) ? tz0 + ((z0 & 3) == 3 ? (size_t)vp.Y * pitch - 12u : 4u) : 0;
  const int r000 = vol[tz0 + ty0 + tx0].x, r100 = vol[tz0 + ty0 + tx1].x, r010 = vol[tz0 + ty1 + tx0].x, r110 = vol[tz0 + ty1 + tx1].x;
  const int r001 = vol[tz1 + ty0 + tx0].x, r101 = vol[tz1 + ty0 + tx1].x, r011 = vol[tz1 + ty1 + tx0].x, r111 = vol[tz1 + ty1 + tx1].x;
  const float f000 = hsk_tsdf_unpack(in0 ? r000 : 0), f100 = hsk_tsdf_unpack(in0 ? r100 : 0);
  const float f010 = hsk_tsdf_unpack(in0 ? r010 : 0), f110 = hsk_tsdf_unpack(in0 ? r110 : 0);
  const float f001 = hsk_tsdf_unpack(in1 ? r001 : 0), f101 = hsk_tsdf_unpack(in1 ? r101 : 0);
  const float f011 = hsk_tsdf_unpack(in1 ? r011 : 0), f111 = hsk_tsdf_unpack(in1 ? r111 : 0);
  float res = f000 * (1.0f - a) * (1.0f - b) * (1.0f - c);
  res = res + f001 * (1.0f - a) * (1.0f - b) * c;
  res = res + f010 * (1.0f - a) * b * (1.0f - c);
  res = res + f011 * (1.0f - a) * b * c;
  res = res + f100 * a * (1.0f - b) * (1.0f - c);
  res = res + f101 * a * (1.0f - b) * c;
  res = res + f110 * a * b * (1.0f - c);
  res = res + f111 * a * b * c;
  return ok ? res : HSK_NANF;
}

// floor(p / cell) of the spec without the IEEE division in the common case: q = p * (1/cell) differs from the
// correctly rounded quotient by < 3 * 2^-24 * |q|, so unless q sits within 2.5e-4 of an integer (|q| < 1100)
// both have the same floor; the rare lanes that do sit there take the exact division.
static __device__ __forceinline__ int vox_fast(float p, float cell, float inv_cell) {
  const float q = p * inv_cell;
  float f = floorf(q);
  const float fr = q - f;
  if (!(fr > 2.5e-4f && fr < 0.99975f && q > -1100.0f && q < 1100.0f)) f = floorf(p / cell);
  if (!(f >= 0.0f)) return -1;
  if (f > 1.0e6f) return 1000000;
  return (int)f;
}

// one level of the map pyramid inside a wave that holds an 8x8 pixel tile (lane = y * 8 + x): the lane at the top
// left of each 2x2 group (dx, dy = lane distance to its right / lower neighbour at this level) forms the mean of the
// vertex taps and the renormalised mean of the normal taps, NaN when any tap is NaN; other lanes' results are unused
static __device__ __forceinline__ void pyramid_step(const float* m, int dx, int dy, float* out) {
  float t1[6], t2[6], t3[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    t1[c] = __shfl_down(m[c], dx, 64);
    t2[c] = __shfl_down(m[c], dy, 64);
    t3[c] = __shfl_down(m[c], dx + dy, 64);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int b = 3 * h;
    float a0 = HSK_NANF, a1 = HSK_NANF, a2 = HSK_NANF;
    if (!(hsk_isnan(m[b]) || hsk_isnan(t1[b]) || hsk_isnan(t2[b]) || hsk_isnan(t3[b]))) {
      a0 = (((m[b] + t1[b]) + t2[b]) + t3[b]) / 4.0f;
      a1 = (((m[b + 1] + t1[b + 1]) + t2[b + 1]) + t3[b + 1]) / 4.0f;
      a2 = (((m[b + 2] + t1[b + 2]) + t2[b + 2]) + t3[b + 2]) / 4.0f;
      if (h == 1) {
        const float inv = 1.0f / sqrtf(hsk_dot3(a0, a1, a2, a0, a1, a2));
        a0 = a0 * inv;
        a1 = a1 * inv;
        a2 = a2 * inv;
      }
    }
    out[b] = a0;
    out[b + 1] = a1;
    out[b + 2] = a2;
  }
}

#ifdef HSK_RC_TIMING
__device__ unsigned long long g_rc_times[8192 * 8];  // per tile: 4 stamps, march trips, trips in which a lane gathered
extern "C" int hsk_debug_rc_times(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rc_times), (size_t)n * 8);
}
#define RC_STAMP(k) do { if (lane == 0 && tile_id < 8192) g_rc_times[tile_id * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RC_STAMP(k) do { } while (0)
#endif
#ifndef RC_BLOCK
#define RC_BLOCK 64     // one wave = one 8x8 tile = one workgroup with its own 4 KiB copy of the brick bitfield: the 4800
#endif                  // waves of a 640x480 frame spread evenly over the SIMDs.  Measured 512^3 / 1024^3 (us): 64 threads
                        // 99 / 124, 128: 107 / 126, 256: 99 / 131, 512: 107 / 142.  With 512-thread blocks and a 32 KiB
                        // bitfield 88 of the 256 CUs got a third block and the kernel waited for them (raycast_analysis.md).
#ifndef RC_WPE
#define RC_WPE 5  // waves per SIMD the register allocator must leave room for (96 VGPRs): the 4800 tiles of a 640x480 frame are all resident at five (5120 slots), and six would cost spills
#endif
#ifndef RC_EXT
#define RC_EXT 2       // further clear super-bricks a crossing may run on through
#endif
#define RC_SKIP_MAX (64.0f * (RC_EXT + 1))  // most steps crossed at once
#ifndef RC_MARGIN
#define RC_MARGIN 0.125f  // steps a crossing stops short of the last face (3 mm: the exit times and the accumulated ray parameter are
#endif                    // good to micrometres; two whole steps, the first choice, cost every crossing two steps: 57.3 -> 56.5 us)
#ifndef RC_TIE
#define RC_TIE 0.0625f  // steps by which the runner-up face must lie behind the first for a crossing to run on through it
#endif
#ifndef RC_SKIP
#define RC_SKIP 2      // fewest steps worth crossing at once inside a clear super-brick
#endif
#ifndef RC_GROUP
#define RC_GROUP 4     // march steps located and gathered together (k_raycast)
#endif
#define RC_STAGE_MAX 4  // 16-B loads per thread: 4 KiB / (64 x 16 B); larger bitfields take the loop below
// minimum over the 64 lanes of a wave whose lanes are ALL active, as a wave-uniform value: four DPP steps inside each row of
// 16 lanes, two row broadcasts, one v_readlane (six ds_bpermute round trips through the LDS crossbar before)
static __device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));  // row_half_mirror
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));  // row_mirror: every lane holds its row's minimum
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

// What the kernel needs only AFTER the march (the maps it writes, the pyramid levels): kept out of the march loop's
// scalar registers.  The compiler loads every kernel argument it uses in the entry block and keeps it there; the march
// loop already needs ~100 SGPRs (uniform volume constants plus a saved lane mask per level of divergent control flow),
// so the 16 that these pointers took were spilled into VGPR lanes (v_writelane / v_readlane inside the loop, and any
// further scalar state cost VGPRs the same way: what rounds 2 and 3 took for a wall at 80 VGPRs).  They are therefore
// the LAST member of the argument block and read through the kernarg segment pointer after the loop.
struct RcTail {
  float* vmap;
  float* nmap;
  int* keys;
  MapPyramid pyr;
  int W, H;
};
struct RcArgs {   // (what the kernel needs first comes first: the first 16 dwords arrive in SGPRs with the wave)
  const unsigned* flags;
  int flag_words;
  int W, H;
  const TrackState* st;
  const short2* vol;
  RingOut ring;
  Intr in;
  VolParams vp;
  RcTail tail;   // never touched by name inside the kernel
};
// a member of the argument block fetched where it is used (see RcTail)
#define RC_ARG(type, member) (*(const type*)(rc_kernarg() + offsetof(RcArgs, member)))
static __device__ __forceinline__ const char* rc_kernarg() {
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  return ka;
}
// SLAB: this context stores / owns only part of the z range (multi-GPU).
template <bool SLAB>
__global__ __launch_bounds__(RC_BLOCK, RC_WPE) void k_raycast(RcArgs a) {
  const short2* __restrict__ vol = a.vol;
  const TrackState* __restrict__ st = a.st;
  const VolParams& vp = a.vp;
  const int W = a.W, H = a.H;
  const Intr& in = a.in;
  const unsigned* __restrict__ flags = a.flags;
  const int flag_words = a.flag_words;
  const RingOut& ring = a.ring;
  // the whole brick bitfield ("this brick has held a negative TSDF") lives in LDS: the march then touches
  // global memory only next to surfaces
  extern __shared__ unsigned lflags[];
#ifdef HSK_RC_TIMING
  const int tile_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  RC_STAMP(0);
#endif
  // The bitfield is REQUESTED here -- 16-B loads, all of a thread's loads in flight at once (a one-word-at-a-time staging
  // loop cost 9 us per block: profiles/r01/raycast_analysis.md) -- and put into LDS further down, behind the ray set-up,
  // which needs none of it: at the start of a launch every wave of the chip is at this point at once, and nothing else
  // is there to run under the loads.
  const int nq = (flag_words + HSK_SUPER_WORDS) >> 2;  // brick bits + super-brick bits, both multiples of 4 words
  // (an indexed temporary array here was placed in scratch memory by the compiler: named registers instead)
  const int q0 = threadIdx.x, q1 = q0 + RC_BLOCK, q2 = q1 + RC_BLOCK, q3 = q2 + RC_BLOCK;
  static_assert(RC_STAGE_MAX == 4, "the staging is written for four 16-B loads per thread");
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  const uint4 a0 = q0 < nq ? ((const uint4*)flags)[q0] : zero4;
  const uint4 a1 = q1 < nq ? ((const uint4*)flags)[q1] : zero4;
  const uint4 a2 = q2 < nq ? ((const uint4*)flags)[q2] : zero4;
  const uint4 a3 = q3 < nq ? ((const uint4*)flags)[q3] : zero4;
#ifndef HSK_RC_TIMING
  const int lane = threadIdx.x & 63;
#endif
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3;
  // Tile rows are dispatched from the top and bottom edges of the image inwards (0, last, 1, last - 1, ...): the rays of
  // the border rows meet floor and ceiling at grazing angles and march longest, and a wave dispatched last onto a SIMD
  // that already holds its share of waves finishes last -- with the rows in image order the launch ended with exactly
  // those tiles (tools/rc_timing.sh).  Scheduling only.  Measured 512^3 / 1024^3: 90.6 / 117.9 -> 86.8 / 110.8 us.
  const int ty_lin = tile / tiles_x;
  const int ty = (ty_lin & 1) ? (tiles_y - 1 - (ty_lin >> 1)) : (ty_lin >> 1);
  const int x = (tile % tiles_x) * 8 + (lane & 7);
  const int y = ty * 8 + (lane >> 3);
  if (!SLAB && ring.slots && blockIdx.x == 0 && threadIdx.x == 0) {
    // the tracker state is final once the ICP has ended (nothing after it writes it): report it to the host now, also
    // for a lost or dropped frame, which returns just below
    const unsigned n = *ring.seq;
    *ring.seq = n + 1u;
    TrackState* dst = ring.slots + ring.slot_fifo[n % HSK_RING_FIFO];
    const int* src_w = (const int*)st;
    int* dst_w = (int*)dst;
    for (unsigned i = 0; i < (unsigned)(offsetof(TrackState, ring_mark) / 4); ++i) dst_w[i] = src_w[i];
    __threadfence_system();     // the state words reach the host before the marks that announce them
    // (pose_mark: already there when the integrate's first kernel reported early; set here for the frames it did not)
    __hip_atomic_store(&dst->pose_mark, (n + 1u) | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&dst->ring_mark, (n + 1u) | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // Lanes outside the image (a ragged last tile) and lanes whose ray misses the volume stay in the wave as rays that have
  // ended: every lane is then active at the top of the march loop, which lets its wave-wide decisions use DPP
  // reductions read from a fixed lane, and takes one level of divergent control flow out of the loop.
  const bool in_img = x < W && y < H;
  if (st->lost) return;
  const size_t P = (size_t)W * H;
  const size_t i = in_img ? (size_t)y * W + x : 0;
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
  {
    const float ic0 = 1.0f / vp.cell[0], ic1 = 1.0f / vp.cell[1], ic2 = 1.0f / vp.cell[2];
    const int bs = vp.bshift;
    const int bxn = vp.X >> bs, byn = vp.Y >> bs;
    const float time_step = vp.tau * 0.8f;
    const float max_time = 3.0f * ((vp.size[0] + vp.size[1]) + vp.size[2]);
    float time_curr = t_start;
    int step = 0;
    // near sample of step 0: the entry voxel, clamped into the grid (A.6)
    int qx = vox_fast(t0 + d0 * time_curr, vp.cell[0], ic0);
    int qy = vox_fast(t1 + d1 * time_curr, vp.cell[1], ic1);
    int qz = vox_fast(t2 + d2 * time_curr, vp.cell[2], ic2);
    int px = qx < 0 ? 0 : (qx > vp.X - 1 ? vp.X - 1 : qx);
    int py = qy < 0 ? 0 : (qy > vp.Y - 1 ? vp.Y - 1 : qy);
    int pz = qz < 0 ? 0 : (qz > vp.Z - 1 ? vp.Z - 1 : qz);
    bool crossing = false;
    int nux = 0, nuy = 0, nuz = 0;  // unclamped voxel of the near sample at the crossing
    // brick flag of a voxel inside the grid (0 when its plane is not stored by this slab)
    auto flag_at = [&](int vx_, int vy_, int vz_) -> unsigned {
      const int zz = SLAB ? vz_ - vp.zs0 : vz_;
      const bool stored = !SLAB || (zz >= 0 && zz < vp.nzs);
      const int bf = __mul24(__mul24(stored ? (zz >> bs) : 0, byn) + (vy_ >> bs), bxn) + (vx_ >> bs);
      const unsigned w = lflags[bf >> 5];
      return stored ? ((w >> (bf & 31)) & 1u) : 0u;
    };
    {
      uint4* dst = (uint4*)lflags;
      if (q0 < nq) dst[q0] = a0;
      if (q1 < nq) dst[q1] = a1;
      if (q2 < nq) dst[q2] = a2;
      if (q3 < nq) dst[q3] = a3;
      for (int q = threadIdx.x + RC_STAGE_MAX * RC_BLOCK; q < nq; q += RC_BLOCK) ((uint4*)lflags)[q] = ((const uint4*)flags)[q];
    }
    __syncthreads();
    RC_STAMP(1);
    unsigned fl_prev = flag_at(px, py, pz);  // always the flag of the current near sample
    // Voxel of a sample: the spec's floor(p / cell).  q = p * (1 / cell) differs from the correctly rounded quotient
    // by < 3 * 2^-24 * |q|, so both have the same floor unless q lies within eps of an integer -- for every q inside
    // or within a voxel of the grid; a sample farther out is outside the grid either way (its error is relative).
    const float eps = 3.0e-7f * (float)max(vp.X, max(vp.Y, vp.Z)) + 1.0e-5f;
    bool first = true;  // the near sample of the first step is the (clamped) entry voxel; qx,qy,qz hold it unclamped
    // voxel of the far sample at ray parameter tn (floor(p / cell) of the spec); false when it lies outside the grid
    auto far_voxel = [&](float tn, int& gx, int& gy, int& gz) -> bool {
      const float pnx = t0 + d0 * tn, pny = t1 + d1 * tn, pnz = t2 + d2 * tn;
      const float q0 = pnx * ic0, q1 = pny * ic1, q2 = pnz * ic2;
      const float r0 = __builtin_amdgcn_fractf(q0), r1 = __builtin_amdgcn_fractf(q1), r2 = __builtin_amdgcn_fractf(q2);
      float f0 = q0 - r0, f1 = q1 - r1, f2 = q2 - r2;  // floor
      // distance of the fractional parts from 1/2: far from 1/2 means close to an integer
      const float far_from_half = fmaxf(fmaxf(fabsf(r0 - 0.5f), fabsf(r1 - 0.5f)), fabsf(r2 - 0.5f));
      if (!(far_from_half < 0.5f - eps)) {  // rare (or NaN): the exact floor(p / cell) of the spec
        f0 = floorf(pnx / vp.cell[0]);
        f1 = floorf(pny / vp.cell[1]);
        f2 = floorf(pnz / vp.cell[2]);
      }
      // v_cvt_i32_f32 saturates; a negative or huge index fails the unsigned bound test
      gx = (int)f0;
      gy = (int)f1;
      gz = (int)f2;
      return ((unsigned)gx < (unsigned)vp.X) & ((unsigned)gy < (unsigned)vp.Y) & ((unsigned)gz < (unsigned)vp.Z);  // (no short circuit: no lane-mask branch)
    };
#ifdef HSK_RC_TIMING
    unsigned trips = 0, gtrips = 0;        // acted steps; acted steps that compared voxels (per lane)
    unsigned it_all = 0, it_skip = 0, it_empty = 0;  // loop iterations; crossings; regular trips in which no lane gathered (wave)
#endif
    // The march advances RC_GROUP steps per trip.  A step that lies next to a flagged brick needs its two voxels, and a
    // wave whose lanes reach such bricks at different steps used to stop for a memory round trip (~0.9 us under load) at
    // every step in which ANY lane gathered (tools/rc_timing.sh: march time = 0.06 us x steps + 0.9 us x gather steps +
    // 46 us of waiting for other lanes' gathers).  Here the far samples of the next RC_GROUP steps are located first
    // (voxel + brick flag: arithmetic and LDS only), then every voxel any of those steps will compare is loaded in
    // one batch -- the same voxels the step-by-step march reads, no others -- and the steps are then acted on in order
    // with the values in registers: one round trip per RC_GROUP steps instead of up to RC_GROUP.  Same decisions, same
    // ray parameters ((time_curr + time_step) + time_step ...), so the maps are bit-identical.
    bool ended = !(in_img && t_start < t_exit);
    // Crossing clear super-bricks: when the near sample of EVERY marching lane of the wave sits in a super-brick (4^3
    // bricks) none of whose bricks has held a negative TSDF, and every lane's ray stays inside its super-brick for the
    // next RC_SKIP steps and RC_MARGIN of a step more, none of those steps can gather or end -- their only effect is to
    // advance time_curr and step.  So the wave advances them by the same float additions and looks up the new near
    // sample once.  The decision is wave-wide (the 64 rays of an 8x8 tile are a few centimetres apart, so they cross the
    // same super-bricks together; per-lane skipping made every trip pay for both paths: raycast_analysis.md).
    const bool can_skip = !SLAB && hsk_super_ok(vp);
    const int ss = bs + HSK_SUPER_SHIFT, sxn = hsk_super_dim(vp.X, bs), syn = hsk_super_dim(vp.Y, bs), szn = hsk_super_dim(vp.Z, bs);
    const float s_edge0 = (float)(1 << ss) * vp.cell[0], s_edge1 = (float)(1 << ss) * vp.cell[1], s_edge2 = (float)(1 << ss) * vp.cell[2];
    const float id0 = 1.0f / d0, id1 = 1.0f / d1, id2 = 1.0f / d2;
    const float inv_step = 1.0f / time_step;
    // (a wave-wide loop: lanes whose ray has ended idle inside it, so that the wave-wide minimum below can use shuffles)
    RC_STAMP(6);
    while (__ballot(!ended && time_curr < max_time) != 0ull) {
      const bool act = !ended && time_curr < max_time;
#ifdef HSK_RC_TIMING
      ++it_all;
#endif
      if (can_skip) {
        // Steps every marching lane can cross at once at one level of the block hierarchy (sh: log2 of the block edge in
        // voxels; xn, yn, zn: blocks per axis; woff: where the level's bits start in lflags; half: its edge is half a
        // super-brick's): 0 unless the near sample of EVERY marching lane sits in a clear block.
        auto crossing_steps = [&](const int sh, const int xn, const int yn, const int zn, const int woff, const bool half) -> int {
          const int s0 = px >> sh, s1 = py >> sh, s2 = pz >> sh;
          const int sbit = (s2 * yn + s1) * xn + s0;
          const bool clear = !((lflags[woff + (sbit >> 5)] >> (sbit & 31)) & 1u);
          // (one ballot settles the common "no": the waves that graze a surface for a hundred steps -- the ones the launch
          // ends with -- must not pay for exit distances and a wave-wide minimum at every trip)
          if (__ballot(act && !clear) != 0ull) return 0;
          const float g0 = half ? 0.5f * s_edge0 : s_edge0, g1 = half ? 0.5f * s_edge1 : s_edge1, g2 = half ? 0.5f * s_edge2 : s_edge2;
          // ray parameter at which the ray leaves the block (approximate; RC_MARGIN of a step absorbs the error)
          float e0 = ((float)(s0 + (d0 > 0.0f ? 1 : 0)) * g0 - t0) * id0;
          float e1 = ((float)(s1 + (d1 > 0.0f ? 1 : 0)) * g1 - t1) * id1;
          float e2 = ((float)(s2 + (d2 > 0.0f ? 1 : 0)) * g2 - t2) * id2;
          float te = fminf(fminf(e0, e1), e2);
#if RC_EXT > 0
          // ... and on through up to RC_EXT further blocks while they are clear too (open air: the regular trip that used
          // to carry the march across every face between two clear blocks is most of what a room costs).  The next block
          // is the one behind the face the ray leaves by; that is certain only when the runner-up face lies clearly later
          // (near an edge or corner the float exit times may order wrongly, and the ray could cut through a third, flagged
          // block): RC_TIE = 1/16 step = 1.5 mm, a thousand times what the exit times can be off by (a few ulp of a few
          // metres); otherwise the crossing ends here.  (Two steps, the first choice, ended a fifth of the crossings early:
          // 58.9 -> 57.4 us.)
          {
            int c0 = s0, c1 = s1, c2 = s2;
            bool live = act;
#pragma unroll
            for (int k = 0; k < RC_EXT; ++k) {
              const bool a0 = e0 <= e1 && e0 <= e2, a1 = !a0 && e1 <= e2, a2 = !a0 && !a1;
              const float second = a0 ? fminf(e1, e2) : (a1 ? fminf(e0, e2) : fminf(e0, e1));
              const int n0 = c0 + (a0 ? (d0 > 0.0f ? 1 : -1) : 0), n1 = c1 + (a1 ? (d1 > 0.0f ? 1 : -1) : 0),
                        n2 = c2 + (a2 ? (d2 > 0.0f ? 1 : -1) : 0);
              live = live && (second - te >= RC_TIE * time_step) && (unsigned)n0 < (unsigned)xn && (unsigned)n1 < (unsigned)yn &&
                     (unsigned)n2 < (unsigned)zn;
              const int nb = live ? (n2 * yn + n1) * xn + n0 : 0;
              live = live && !((lflags[woff + (nb >> 5)] >> (nb & 31)) & 1u);
              if (live) {
                c0 = n0; c1 = n1; c2 = n2;
                e0 = a0 ? e0 + g0 * fabsf(id0) : e0;
                e1 = a1 ? e1 + g1 * fabsf(id1) : e1;
                e2 = a2 ? e2 + g2 * fabsf(id2) : e2;
                te = fminf(fminf(e0, e1), e2);
              }
            }
          }
#endif
          const float room = (te - time_curr) * inv_step - RC_MARGIN;
          return wave_min_i32(!act ? 0x7fffffff : (room >= 1.0f ? (int)fminf(room, RC_SKIP_MAX) : 0));
        };
        int n = crossing_steps(ss, sxn, syn, szn, flag_words, false);
        if (n >= RC_SKIP && n != 0x7fffffff) {  // wave-uniform
          float tc = time_curr;
          int i_ = 0;
          for (; i_ + 4 <= n; i_ += 4) tc = (((tc + time_step) + time_step) + time_step) + time_step;  // (the march's own additions, in order)
          for (; i_ < n; ++i_) tc = tc + time_step;
          int nx_, ny_, nz_;
          const bool fine = !act || (far_voxel(tc, nx_, ny_, nz_) && tc < max_time);
          if (__ballot(!fine) == 0ull) {
            if (act) {
              time_curr = tc;
              step += n;
              px = nx_; py = ny_; pz = nz_;
              first = false;
              fl_prev = flag_at(px, py, pz);
            }
#ifdef HSK_RC_TIMING
            ++it_skip;
#endif
            continue;
          }
        }
      }
      if (!act) continue;
      float tt[RC_GROUP];
      int vx_[RC_GROUP], vy_[RC_GROUP], vz_[RC_GROUP];
      bool okv[RC_GROUP], need[RC_GROUP];
      unsigned fl[RC_GROUP];
      bool all_alive;
      {
        float tc = time_curr;
        bool alive = true;
        unsigned fprev = fl_prev;
#pragma unroll
        for (int g = 0; g < RC_GROUP; ++g) {
          alive = alive && (tc < max_time);
          tt[g] = tc + time_step;
          okv[g] = far_voxel(tt[g], vx_[g], vy_[g], vz_[g]);
          alive = alive && okv[g];
          fl[g] = flag_at(alive ? vx_[g] : 0, alive ? vy_[g] : 0, alive ? vz_[g] : 0);  // (looked up whether alive or not: no branch)
          fl[g] = alive ? fl[g] : 0u;
          const bool owned = !SLAB || (vz_[g] >= vp.zo0 && vz_[g] < vp.zo1);
          need[g] = alive && owned && ((fprev | fl[g]) != 0u);
          fprev = fl[g];
          tc = tt[g];
        }
        all_alive = alive;
      }
      bool any_need = false;
#pragma unroll
      for (int g = 0; g < RC_GROUP; ++g) any_need = any_need || need[g];
      // Most trips outside the clear super-bricks still compare nothing (a flagged super-brick is mostly unflagged
      // bricks): when every marching lane's RC_GROUP steps stay inside the grid, before max_time and away from flagged
      // bricks, acting on them one by one comes to this.
      if (__ballot(!(all_alive && !any_need)) == 0ull) {
        px = vx_[RC_GROUP - 1]; py = vy_[RC_GROUP - 1]; pz = vz_[RC_GROUP - 1];
        first = false;
        fl_prev = fl[RC_GROUP - 1];
        time_curr = tt[RC_GROUP - 1];
        step += RC_GROUP;
#ifdef HSK_RC_TIMING
        trips += RC_GROUP;
        ++it_empty;
#endif
        continue;
      }
      int raw[RC_GROUP + 1];  // raw[0]: the near sample of the first step; raw[g + 1]: the far sample of step g
#pragma unroll
      for (int g = 0; g <= RC_GROUP; ++g) raw[g] = 0;
      if (any_need) {
        if (need[0]) raw[0] = raw_at(vol, vp, px, py, pz);
#pragma unroll
        for (int g = 0; g < RC_GROUP; ++g)
          if (need[g] || (g + 1 < RC_GROUP && need[g + 1])) raw[g + 1] = raw_at(vol, vp, vx_[g], vy_[g], vz_[g]);
      }
      // Acting on the RC_GROUP steps in order, without branches: a step halts the lane when the march is past max_time, the
      // far sample lies outside the grid (the ray ends), or the two voxels show a back face or a zero crossing; the steps
      // before the first halt advance the lane.  (With a divergent branch and a break per step this was 85 instructions a
      // step, most of them lane-mask bookkeeping; the same decisions as selects are 15.)
      {
        bool run = true, e_out = false, e_back = false, e_cross = false;
        int adv = 0;
#pragma unroll
        for (int g = 0; g < RC_GROUP; ++g) {
          const float tcur = g == 0 ? time_curr : tt[g - 1];
          const bool on = run && (tcur < max_time);
          const bool back = need[g] && raw[g] < 0 && raw[g + 1] > 0;
          const bool cross = need[g] && raw[g] > 0 && raw[g + 1] < 0;
          e_out = e_out || (on && !okv[g]);
          e_back = e_back || (on && okv[g] && back);
          e_cross = e_cross || (on && okv[g] && cross);
          run = on && okv[g] && !back && !cross;
          // the far sample of an advancing step is the next step's near sample
          px = run ? vx_[g] : px;
          py = run ? vy_[g] : py;
          pz = run ? vz_[g] : pz;
          fl_prev = run ? fl[g] : fl_prev;
          time_curr = run ? tt[g] : time_curr;
          adv += run ? 1 : 0;
#ifdef HSK_RC_TIMING
          trips += on ? 1 : 0;
          gtrips += (on && okv[g] && need[g]) ? 1 : 0;
#endif
        }
        const bool was_first = first && adv == 0;
        first = first && adv == 0;
        step += adv;
        if (e_back) key = (step << 1) | 1;
        if (e_cross) {  // zero crossing: refined below with every lane of the wave; (px, py, pz) is the near sample of its step
          crossing = true;
          nux = was_first ? qx : px;
          nuy = was_first ? qy : py;
          nuz = was_first ? qz : pz;
        }
        ended = ended || e_out || e_back || e_cross;
      }
    }
    // Deferred hit processing: lanes hit at different steps, and refining inside the loop would run these
    // (memory-latency-bound) taps once per distinct step.  Here the wave runs them once, loads batched.
    RC_STAMP(2);
#ifdef HSK_RC_TIMING
    {
      // wave totals: the longest lane's trips, and the number of lanes-trips with gathers (max over lanes)
      unsigned tmax = trips, gmax = gtrips, ia = it_all, is = it_skip, ie = it_empty;
      for (int o = 32; o > 0; o >>= 1) {
        tmax = max(tmax, (unsigned)__shfl_xor((int)tmax, o, 64));
        gmax = max(gmax, (unsigned)__shfl_xor((int)gmax, o, 64));
        ia = max(ia, (unsigned)__shfl_xor((int)ia, o, 64));
        is = max(is, (unsigned)__shfl_xor((int)is, o, 64));
        ie = max(ie, (unsigned)__shfl_xor((int)ie, o, 64));
      }
      if (lane == (int)__builtin_ctzll(__ballot(true)) && tile_id < 8192) {
        g_rc_times[tile_id * 8 + 4] = tmax;
        g_rc_times[tile_id * 8 + 5] = (unsigned long long)(gmax & 0xffffu) | ((unsigned long long)(ia & 0xffffu) << 16) |
                                      ((unsigned long long)(is & 0xffffu) << 32) | ((unsigned long long)(ie & 0xffffu) << 48);
      }
    }
#endif
    if (crossing) {
      key = (step << 1) | 1;
      const float tn = time_curr + time_step;
      const float Ftdt = trilinear(vol, vp, t0 + d0 * tn, t1 + d1 * tn, t2 + d2 * tn);
      const float Ft = trilinear(vol, vp, t0 + d0 * time_curr, t1 + d1 * time_curr, t2 + d2 * time_curr);
      if (!hsk_isnan(Ftdt) && !hsk_isnan(Ft)) {
        const float Ts = time_curr - (time_step * Ft) / (Ftdt - Ft);
        if (Ts >= time_curr - time_step && Ts <= time_curr + 2.0f * time_step) {  // (D3: two steps round the far sample)
          vx = t0 + d0 * Ts;
          vy = t1 + d1 * Ts;
          vz = t2 + d2 * Ts;
          key = (step << 1);
          if (nux > 1 && nuy > 1 && nuz > 1 && nux < vp.X - 2 && nuy < vp.Y - 2 && nuz < vp.Z - 2) {
            const float xp = trilinear(vol, vp, vx + vp.cell[0], vy, vz), xm = trilinear(vol, vp, vx - vp.cell[0], vy, vz);
            const float yp = trilinear(vol, vp, vx, vy + vp.cell[1], vz), ym = trilinear(vol, vp, vx, vy - vp.cell[1], vz);
            const float zp = trilinear(vol, vp, vx, vy, vz + vp.cell[2]), zm = trilinear(vol, vp, vx, vy, vz - vp.cell[2]);
            const float gxn = xp - xm, gyn = yp - ym, gzn = zp - zm;
            const float ninv = 1.0f / sqrtf(hsk_dot3(gxn, gyn, gzn, gxn, gyn, gzn));
            nx = gxn * ninv;
            ny = gyn * ninv;
            nz = gzn * ninv;
          }
        }
      }
    }
  }
  // the tail of the argument block, fetched now (the empty asm hides where the pointer comes from, so the loads cannot
  // be moved up across the march)
  const RcTail tl = RC_ARG(RcTail, tail);
  float* __restrict__ vmap = tl.vmap;
  float* __restrict__ nmap = tl.nmap;
  int* __restrict__ keys = tl.keys;
  const MapPyramid pyr = tl.pyr;
  if (in_img) {
    vmap[i] = vx;
    vmap[P + i] = vy;
    vmap[2 * P + i] = vz;
    nmap[i] = nx;
    nmap[P + i] = ny;
    nmap[2 * P + i] = nz;
    if (keys) keys[i] = key;
  }
  RC_STAMP(3);
  if (!SLAB && pyr.v1) {
    // Model pyramid (resizeVMap / resizeNMap, A.3) from the wave's own 8x8 tile: level 1 is the 2x2 mean held by
    // the even-even lanes, level 2 the 2x2 mean of those -- the arithmetic and its order are k_resize_maps2's, the
    // taps arrive by lane shuffles instead of a second launch reading the maps back.
    float m[6] = {vx, vy, vz, nx, ny, nz};
    float l1[6], l2[6];
    pyramid_step(m, 1, 8, l1);
    pyramid_step(l1, 2, 16, l2);
    const int w1 = W >> 1, w2 = W >> 2;
    const size_t P1 = (size_t)w1 * (H >> 1), P2 = (size_t)w2 * (H >> 2);
    if (((x | y) & 1) == 0) {
      const size_t o = (size_t)(y >> 1) * w1 + (x >> 1);
      pyr.v1[o] = l1[0]; pyr.v1[P1 + o] = l1[1]; pyr.v1[2 * P1 + o] = l1[2];
      pyr.n1[o] = l1[3]; pyr.n1[P1 + o] = l1[4]; pyr.n1[2 * P1 + o] = l1[5];
    }
    if (((x | y) & 3) == 0) {
      const size_t o = (size_t)(y >> 2) * w2 + (x >> 2);
      pyr.v2[o] = l2[0]; pyr.v2[P2 + o] = l2[1]; pyr.v2[2 * P2 + o] = l2[2];
      pyr.n2[o] = l2[3]; pyr.n2[P2 + o] = l2[4]; pyr.n2[2 * P2 + o] = l2[5];
    }
  }
}

void launch_raycast(hipStream_t s, const void* vol, const TrackState* st, const VolParams& vp, int W, int H, Intr in,
                    float* vmap, float* nmap, int* keys, const unsigned* flags, const MapPyramid* pyramid, const RingOut* ring) {
  const int tiles = ((W + 7) / 8) * ((H + 7) / 8);
  dim3 block(RC_BLOCK);
  dim3 grid((tiles + RC_BLOCK / 64 - 1) / (RC_BLOCK / 64));
  const int words = hsk_flag_words(vp);
  const bool slab = vp.zs0 != 0 || vp.nzs != vp.Z || vp.zo0 != 0 || vp.zo1 != vp.Z;
  const MapPyramid none = {nullptr, nullptr, nullptr, nullptr};
  const RingOut quiet = {nullptr, nullptr, nullptr};
  RcArgs a;
  a.vol = (const short2*)vol;
  a.st = st;
  a.vp = vp;
  a.W = W;
  a.H = H;
  a.in = in;
  a.flags = flags;
  a.flag_words = words;
  a.ring = (!slab && ring) ? *ring : quiet;
  a.tail.vmap = vmap;
  a.tail.nmap = nmap;
  a.tail.keys = keys;
  a.tail.pyr = (!slab && pyramid) ? *pyramid : none;
  a.tail.W = W;
  a.tail.H = H;
  if (slab)
    hipLaunchKernelGGL(k_raycast<true>, grid, block, (size_t)(words + HSK_SUPER_WORDS) * 4, s, a);
  else
    hipLaunchKernelGGL(k_raycast<false>, grid, block, (size_t)(words + HSK_SUPER_WORDS) * 4, s, a);
}
// the fused pyramid needs complete 8x8 tiles and a single-device volume
bool raycast_can_fuse_pyramid(const VolParams& vp, int W, int H) {
  const bool slab = vp.zs0 != 0 || vp.nzs != vp.Z || vp.zo0 != 0 || vp.zo1 != vp.Z;
  return !slab && (W % 8) == 0 && (H % 8) == 0;
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
// Direct exchange (hskinfu_group's one-hop form, SURVEY.md 8(e) "xGMI fit"): the slab that won a pixel stores the bit
// patterns of its vertex / normal straight into EVERY device's composite buffer (its own included) -- peer-mapped
// memory, one hop over xGMI -- and nothing where it lost: a pixel has at most one winner among all slabs (a march step is
// owned by exactly one slab), so the writers never collide, and k_adopt reads the composite only where the MIN key says
// "hit".  Replaces the 7.4 MB all-reduce(SUM) by 24 B per won pixel and peer.
__global__ void k_resolve_push(const int* __restrict__ keys_local, const int* __restrict__ keys_min,
                               const float* __restrict__ vmap, const float* __restrict__ nmap, PushDests dst, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int kl = keys_local[i], km = keys_min[i];
  if (!((kl == km) && (km != HSK_KEY_NONE_I) && ((km & 1) == 0))) return;
  int w[6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    w[c] = __float_as_int(vmap[c * P + i]);
    w[3 + c] = __float_as_int(nmap[c * P + i]);
  }
  for (int d = 0; d < dst.n; ++d) {
    int* __restrict__ b = dst.p[d];
#pragma unroll
    for (int c = 0; c < 6; ++c) b[c * P + i] = w[c];
  }
}
void launch_resolve_push(hipStream_t s, const int* keys_local, const int* keys_min, const float* vmap, const float* nmap,
                         const PushDests& dst, int P) {
  hipLaunchKernelGGL(k_resolve_push, dim3((P + 255) / 256), dim3(256), 0, s, keys_local, keys_min, vmap, nmap, dst, P);
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
      unsigned long long a = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
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
                                                      float* __restrict__ tri, unsigned long long cap, int z_end) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int ny = vp.Y - 1;
  const int nrows = ny * (z_end - vp.zo0);
  if (row >= nrows) return;
  const int y = row % ny, z = vp.zo0 + row / ny;
  unsigned long long base = WRITE ? row_offset[row] : 0;
  unsigned total = 0;
  for (int xb = 0; xb < vp.X - 1; xb += 64) {
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
                         unsigned long long* row_offset, unsigned long long* total, float* tri, unsigned long long cap, int pass) {
  const int z_end = hsk_mesh_z_end(vp);
  const int nrows = (vp.Y - 1) * (z_end - vp.zo0);
  if (nrows <= 0) {
    if (pass == 0) (void)hipMemsetAsync(total, 0, 8, s);
    return;
  }
  dim3 block(256), grid((nrows + 3) / 4);
  if (pass == 0) {
    hipLaunchKernelGGL(k_extract_mesh<false>, grid, block, 0, s, (const short2*)vol, vp, tt, row_count, row_offset, tri, cap, z_end);
    hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, s, row_count, row_offset, nrows, total);
  } else {
    hipLaunchKernelGGL(k_extract_mesh<true>, grid, block, 0, s, (const short2*)vol, vp, tt, row_count, row_offset, tri, cap, z_end);
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
                                                         float* __restrict__ tri, unsigned long long cap, int z_end) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int ny = vp.Y - 1;
  const int nrows = ny * (z_end - vp.zo0);
  if (row >= nrows) return;
  const int y = row % ny, z = vp.zo0 + row / ny;
  unsigned long long base = WRITE ? row_offset[row] : 0;
  unsigned total = 0;
  for (int xb = 0; xb < vp.X - 1; xb += 64) {
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
                            unsigned long long* row_offset, unsigned long long* total, float* tri, unsigned long long cap, int pass) {
  const int z_end = hsk_mesh_z_end(vp);
  const int nrows = (vp.Y - 1) * (z_end - vp.zo0);
  if (nrows <= 0) {
    if (pass == 0) (void)hipMemsetAsync(total, 0, 8, s);
    return;
  }
  const dim3 grid((unsigned)((nrows + 3) / 4));
  if (pass == 0) {
    hipLaunchKernelGGL(k_extract_mesh_mc<false>, grid, dim3(256), 0, s, (const short2*)vol, vp, ct_dev, row_count, (const unsigned long long*)nullptr,
                       (float*)nullptr, 0ull, z_end);
    hipLaunchKernelGGL(k_scan_rows, dim3(1), dim3(1024), 0, s, row_count, row_offset, nrows, total);
  } else {
    hipLaunchKernelGGL(k_extract_mesh_mc<true>, grid, dim3(256), 0, s, (const short2*)vol, vp, ct_dev, row_count, row_offset, tri, cap, z_end);
  }
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
