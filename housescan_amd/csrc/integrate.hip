// integrate.hip -- TSDF integration for gfx950: the per-frame tables (tiles, column z ranges, the coarse
// free-space level), pass A (classification of lane-blocks), pass B (per-voxel path), the lane-block summaries and the
// conversion between the volume's 64-B blocks and the host's row-major arrays.  Hand-written for wave64 / 16-B-per-lane
// HBM access; no MFMA (nothing here is a contraction).  The volume holds (int16 tsdf*32767, int16 weight) pairs in 64-B
// blocks of one lane-block (4 x-voxels x 4 planes: hsk_dev.h, hsk_vox_index).
#pragma clang fp contract(off)
#include "hsk_dev.h"
#include "hsk_launch.h"

// ------------------------------------------------------------------------------------------------------
// integrate (A.4).  Layout: each lane owns 4 x-adjacent voxels (one 16-B vector per plane, the four planes of a group
// consecutive: a 64-B block = a lane-block), a wave covers 16 x 16 voxels (4 lanes by 16 rows), a workgroup four waves side
// by side (64 x 16 voxels) and a chunk of 8 or 16 planes.
//
// Only vectors that hold at least one voxel whose bits really change are read or written, so HBM traffic stays BELOW the
// algorithmic 8 B x V_upd.  Conservative classifications -- each only ever replaces work whose outcome it
// has proven -- keep the arithmetic off everything else:
//   0. (round 5) a wave-chunk's box against the 16-px tile range under it: wholly free space (one count in the chunk's
//      byte), wholly occluded or outside the frustum (nothing) -- k_column_zrange, once per frame and chunk;
//   1. per-lane z interval of the column inside the (padded) view frustum, computed once per column;
//   2. a lane-block's tight pixel box against the 4-px / 8-px tile windows and the validity mask (free space: a summary
//      byte moves; occluded: nothing), with the dilated 16-px table as the fall-back near the camera;
//   3. exact fast paths of the running mean (saturated free space, first observation).
// What is left takes pass B, a lane per plane of a queued lane-block: blocks near a surface or at the frustum's rim the exact
// per-voxel rule; (round 6) free-space blocks with a hole in the depth image under them the LIGHT class -- F = 1 where the
// voxel's pixel has depth, nothing where it has none.  Bricks that ever received a negative TSDF are flagged for the
// raycaster's empty-space test.
// ------------------------------------------------------------------------------------------------------
#define HSK_TILE 16
#define HSK_NQUEUES 256       // uncertain lane-blocks are spread over this many queues (pass A -> pass B)
#define HSK_QCOUNT_STRIDE 64  // words between two queue counters (256 B: one counter per memory-side atomic line)
#define HSK_CF_MIXED 0u  // verdicts of the coarse level (one byte per wave-chunk and frame: "the coarse level" below)
#define HSK_CF_SKIP 1u
#define HSK_CF_FREE 2u
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
                           float* __restrict__ tmin, int tw, unsigned short* __restrict__ vmask16) {
  __shared__ float shx[4], shn[4];
  const int tx = blockIdx.x, ty = blockIdx.y;
  const int x = tx * HSK_TILE + (threadIdx.x & 15), y = ty * HSK_TILE + (threadIdx.x >> 4);
  const bool in = x < W && y < H;
  const float v = in ? scaled[y * W + x] : 0.0f;
  {  // the validity mask, as k_bilateral_scale writes it (1 = no depth or outside the image)
    const unsigned long long inv = __ballot(v == 0.0f);
    if ((threadIdx.x & 63) == 0) {
      const int pitch16 = 2 * hsk_mask_pitch32(W);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = ty * HSK_TILE + (int)(threadIdx.x >> 6) * 4 + r;
        if (yy < H) vmask16[(size_t)yy * pitch16 + tx] = (unsigned short)(inv >> (16 * r));
      }
    }
  }
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
  hipLaunchKernelGGL(k_tile_max, dim3(tw, th), dim3(256), 0, s, scaled, W, H, tiles, tiles + n, tw, (unsigned short*)(tiles + hsk_tiles_mask_offset(W, H)));
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
  // (.y: the minimum over the pixels WITH depth, negated when a pixel of the tile has none -- hsk_dev.h, the tile tables)
  float mx = 0.0f, mn = 1e30f;
  bool hole = false;
#pragma unroll
  for (int qy = 0; qy < 2; ++qy)
#pragma unroll
    for (int qx = 0; qx < 2; ++qx) {
      float qmx = 0.0f, qmn = 1e30f;
      bool qhole = false;
      for (int dy = 0; dy < 4; ++dy)
        for (int dx = 0; dx < 4; ++dx) {
          const int x = tx * HSK_FTILE + qx * 4 + dx, y = ty * HSK_FTILE + qy * 4 + dy;
          const float v = (x < W && y < H) ? scaled[(size_t)y * W + x] : 0.0f;  // outside the image: never "all valid"
          qmx = fmaxf(qmx, v);
          qmn = v != 0.0f ? fminf(qmn, v) : qmn;
          qhole = qhole | (v == 0.0f);
        }
      qtab[(size_t)(2 * ty + qy) * (2 * fw) + (2 * tx + qx)] = make_float2(qmx, qhole ? -qmn : qmn);
      mx = fmaxf(mx, qmx);
      mn = fminf(mn, qmn);
      hole = hole | qhole;
    }
  ftab[t] = make_float2(mx, hole ? -mn : mn);
}

// The second level needs the (max, min) of the depth over the tiles its pixel box touches: nx x ny tiles, nx, ny <= 3,
// from the tile that holds the box's top-left corner.  Those windows are formed here once per frame, one table per
// window shape (nine of them: blockIdx.y = (ny - 1) * 3 + (nx - 1)), so that pass A makes ONE look-up per lane-block
// where it made nine -- pass A and pass B are bound by the CU's vector-memory front end
// (profiles/r02/integrate_analysis.md section 7) -- and over exactly the tiles the box touches, not always 3 x 3.
// Entry (tx, ty) of shape (nx, ny) = over a < ny, b < nx of tab[min(ty + a, last)][min(tx + b, last)].

// The coarse level (k_column_zrange) asks for the (max, min) of the depth over the 16-px tiles a wave-chunk's pixel box
// touches -- up to 16 x 16 of them for a chunk next to the camera.  A sparse table answers any such range with FOUR
// look-ups: level (kx, ky), kx, ky in 0 .. 3, holds for every tile the (max, min) over the 2^kx x 2^ky tiles from it on
// (clipped to the table), and a range of nx x ny tiles is the union of the four blocks of the largest powers of two not
// above nx, ny that sit in its corners.  Walked tile by tile the look-up was a chain of up to 144 dependent round trips in a
// kernel of one wave per SIMD: 12 us of the frame's critical path.
#define HSK_SPARSE_LEVELS 4

// The derived tables of a frame in ONE launch (round 5; they were three): blockIdx.y 0 .. 8 the nine window shapes of the
// 8-px table, 9 .. 17 those of the 4-px table, 18 .. 33 the sixteen levels of the 16-px sparse table.  All of them read only
// the raw tables that k_bilateral_scale (or k_tile_max + k_tile_fine on the stage paths) has written.
__global__ void k_tile_tables(const float* __restrict__ tmax, const float* __restrict__ tmin, int tw, int th, const float2* __restrict__ ftab,
                              const float2* __restrict__ qtab, int fw, int fh, float2* __restrict__ fwin, float2* __restrict__ sparse) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int job = blockIdx.y;
  if (job < 18) {
    const bool fine = job >= 9;
    const float2* __restrict__ tab = fine ? qtab : ftab;
    const int tbw = fine ? 2 * fw : fw, tbh = fine ? 2 * fh : fh;
    if (t >= tbw * tbh) return;
    const int shape = fine ? job - 9 : job;
    const int nx = shape % 3 + 1, ny = shape / 3 + 1;
    const int ty = t / tbw, tx = t - ty * tbw;
    float mx = 0.0f, mn = 1e30f;
    bool hole = false;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        const float2 v = tab[min(ty + a, tbh - 1) * tbw + min(tx + b, tbw - 1)];
        mx = fmaxf(mx, v.x);
        mn = fminf(mn, fabsf(v.y));
        hole = hole | (v.y < 0.0f);
      }
    float2* __restrict__ win = fine ? fwin + (size_t)9 * fw * fh : fwin;
    win[(size_t)shape * tbw * tbh + t] = make_float2(mx, hole ? -mn : mn);
  } else {
    if (t >= tw * th) return;
    const int lv = job - 18;
    const int kx = lv % HSK_SPARSE_LEVELS, ky = lv / HSK_SPARSE_LEVELS;
    const int ty = t / tw, tx = t - ty * tw;
    float mx = 0.0f, mn = 1e30f;
    for (int a = ty; a < min(ty + (1 << ky), th); ++a)
      for (int b = tx; b < min(tx + (1 << kx), tw); ++b) {
        mx = fmaxf(mx, tmax[a * tw + b]);
        mn = fminf(mn, tmin[a * tw + b]);
      }
    sparse[(size_t)lv * tw * th + t] = make_float2(mx, mn);
  }
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

// what the coarse level needs (integrate.hip: "the coarse level"): the verdict bytes of the frame (one per wave-chunk), the
// chunks' pending bytes (null: a count-only launch, which bumps nothing) and the thresholds of the box test
struct CoarseArgs {
  unsigned char* cflag;
  unsigned char* cs;
  const float2* sparse;  // the sparse table (k_tile_tables) of the frame's 16-px tiles
  unsigned dirty_off;    // the chunks' dirty bytes sit this far behind their pending bytes
  int zchunk;  // planes per chunk of THIS launch (vp.zchunk, or all stored planes of a thinner slab)
  float free_thr, cull_thr;
};
// Per-frame pre-pass: for every lane column (4 x-adjacent voxels at one y) the range of stored planes that can
// project into the padded image [-1.5, W+0.5] x [-1.5, H+0.5] in front of the camera.  The view frustum is
// convex, so each column meets it in one interval; clipping the line cam(gz) = a + gz * c against the five
// half-spaces gives it.  Conservative by 2 planes (float error).  Empty columns get (INT_MAX, INT_MIN).
// (argument order: what a kernel of the frame's chain needs first comes first -- the first 16 dwords of the arguments
// arrive in SGPRs with the wave, -amdgpu-kernarg-preload-count in the Makefile)
__global__ void k_column_zrange(IcpFinal fin, const float* __restrict__ tmax, const float* __restrict__ tmin,
                                float2* __restrict__ dtab, int tw, int th, int dil_blocks, unsigned* __restrict__ qcount,
                                const TrackState* __restrict__ st, VolParams vp, int W, int H, Intr in,
                                int2* __restrict__ zint, TrackState* __restrict__ st_out, int2* __restrict__ wgz, RingOut early,
                                CoarseArgs ca) {
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
      for (int q = threadIdx.x; q < HSK_NQUEUES; q += blockDim.x) *(unsigned long long*)&qcount[q * HSK_QCOUNT_STRIDE] = 0ull;  // (heavy count, light count)
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
        }
        if (early.slots && blockIdx.x == gridDim.x - 1) {
          // The frame's pose and verdict are final HERE -- nothing after the ICP writes them -- so the host is told now, a
          // whole integrate earlier than by the raycast's report (which stays: its mark says that the frame's inputs are
          // consumed).  A caller that takes one frame at a time gets its pose back while the volume work is still running,
          // and its next frame's filtering runs under that.  By a block of its OWN, launched for nothing else (round 5):
          // the copy into the pinned ring slot and its system-scope fence take 2-3 us, and in block 0's path they kept
          // this launch -- which pass A waits for -- open that much longer than its slowest block needed; a block that
          // only solves and reports is done before the others have classified their chunks.  It composes the state from its
          // own solve and from the fields of `st` that nobody writes in this launch (block 0 is writing the others).
          const unsigned n = *early.seq;  // (counted up by the raycast's report, not here)
          TrackState* dst = early.slots + early.slot_fifo[n % HSK_RING_FIFO];
          const bool use_new = !p.lost;
          for (int i = 0; i < 9; ++i) dst->R[i] = use_new ? p.R[i] : st->R[i];
          for (int i = 0; i < 3; ++i) dst->t[i] = use_new ? p.t[i] : st->t[i];
          for (int i = 0; i < 9; ++i) dst->Rp[i] = st->Rp[i];
          for (int i = 0; i < 3; ++i) dst->tp[i] = st->tp[i];
          dst->lost = p.lost ? 1 : st->lost;
          dst->frame = st->frame;
          dst->n_iter = p.n_iter;
          dst->need_reset = p.lost ? 1 : st->need_reset;
          for (int k = 0; k < 27; ++k) dst->sums[k] = fin_tot[k];
          __threadfence_system();
          __hip_atomic_store(&dst->pose_mark, (n + 1u) | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    __syncthreads();
    if (early.slots && blockIdx.x == gridDim.x - 1) return;  // (the reporting block has no footprint of its own)
  }
  // One block = the x-y footprint of one pass-A workgroup (16 lane columns by 16 rows), so that the block can also leave
  // that workgroup's z range: pass A's workgroups of the chunks outside it (half of its waves lie outside the frustum)
  // then leave on one scalar load instead of a vector load per lane and a wave-wide reduction.
  __shared__ int4 wv_sh[4];
  const int ncol = vp.X / 4;
  const int gxn = (vp.X + 63) / 64, gyn = (vp.Y + 15) / 16;
  const bool fp_block = (int)blockIdx.x < gxn * gyn;
  const int fby = (int)blockIdx.x / gxn, fbx = (int)blockIdx.x - fby * gxn;
  // (a wave = the footprint of a pass-A wave: 4 lane columns by 16 rows)
  const int y = fby * 16 + (int)((threadIdx.x & 63) >> 2), lc = fbx * 16 + (int)(threadIdx.x >> 6) * 4 + (int)(threadIdx.x & 3);
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
      // this wave's footprint (4 lane columns x 16 rows): union of its columns' ranges, then their intersection
      ((int4*)(wgz + (((size_t)gxn * gyn + 1) & ~(size_t)1)))[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = make_int4(lo, hi, lo_all, hi_all);
      wv_sh[threadIdx.x >> 6] = make_int4(lo, hi, lo_all, hi_all);
    }
    __syncthreads();
    // (the footprint-wide range that pass A's workgroups used to leave by is gone: they go by the coarse level's verdicts)
    // ---- the coarse level: one verdict per wave-chunk of this footprint (4 waves x the launch's chunks of planes).  The
    // chunk's voxel centres fill a box; a box in front of the camera projects into the pixel box of its 8 corners (+-1 px
    // for the rounding to a pixel), and its distances to the camera centre lie between the box's nearest and farthest
    // point.  Against the 16-px tile table (raw: a tile with a pixel outside the image or without depth has minimum 0):
    //   free   every voxel has a pixel inside the image, with depth, and sdf >= tau: the rule writes F = 1 into all of them
    //   dead   no voxel can pass sdf >= -tau (or has a pixel with depth): the rule writes nothing
    // Both are SUFFICIENT conditions of the exact per-voxel rule with the margins of pass A's second level -- a verdict only
    // ever replaces work whose outcome it has proven.
    if (ca.cflag != nullptr) {
      const bool own = fin.slots != nullptr && !fin_pose.lost;
      const bool lost = fin.slots != nullptr ? fin_pose.lost != 0 : st->lost != 0;
      const float* __restrict__ Rm = own ? fin_pose.R : st->R;
      const float* __restrict__ tm = own ? fin_pose.t : st->t;
      const int zchunks = (vp.nzs + ca.zchunk - 1) / ca.zchunk;
      const int tw16 = tw;
      for (int item = threadIdx.x; item < 4 * zchunks; item += blockDim.x) {
        const int w = item & 3, zc = item >> 2;
        const int4 r = wv_sh[w];
        const int zbeg = zc * ca.zchunk, zend = min(zbeg + ca.zchunk, vp.nzs);
        const size_t ci = (((size_t)zc * gyn + fby) * gxn + fbx) * 4 + w;
        // (the chunk's byte is requested now: it travels under the geometry and the table look-ups instead of behind them)
        const unsigned c_now = ca.cs != nullptr ? (unsigned)ca.cs[ci] : 0u;
        unsigned verdict = HSK_CF_MIXED;
        if ((zbeg > r.y) | (zend - 1 < r.x)) {
          verdict = HSK_CF_SKIP;  // outside the padded frustum
        } else {
          const int xa = fbx * 64 + w * 16, ya = fby * 16;
          const int xb = min(xa + 15, vp.X - 1), yb = min(ya + 15, vp.Y - 1);
          const bool full = (xa + 15 < vp.X) & (ya + 15 < vp.Y) & (zbeg + vp.zchunk <= vp.nzs) & (ca.zchunk == vp.zchunk);
          const float g0[3] = {((float)xa + 0.5f) * vp.cell[0] - tm[0], ((float)ya + 0.5f) * vp.cell[1] - tm[1],
                               ((float)(vp.zs0 + zbeg) + 0.5f) * vp.cell[2] - tm[2]};
          const float g1[3] = {((float)xb + 0.5f) * vp.cell[0] - tm[0], ((float)yb + 0.5f) * vp.cell[1] - tm[1],
                               ((float)(vp.zs0 + zend - 1) + 0.5f) * vp.cell[2] - tm[2]};
          float umin = 1e30f, umax = -1e30f, vmin = 1e30f, vmax = -1e30f, zmn = 1e30f;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float gx = (c & 1) ? g1[0] : g0[0], gy = (c & 2) ? g1[1] : g0[1], gz = (c & 4) ? g1[2] : g0[2];
            const float cxm = (Rm[0] * gx + Rm[3] * gy) + Rm[6] * gz, cym = (Rm[1] * gx + Rm[4] * gy) + Rm[7] * gz;
            const float czm = (Rm[2] * gx + Rm[5] * gy) + Rm[8] * gz;
            const float rq = __builtin_amdgcn_rcpf(czm);
            const float uq = (cxm * in.fx) * rq + in.cx, vq = (cym * in.fy) * rq + in.cy;
            zmn = fminf(zmn, czm);
            umin = fminf(umin, uq);
            umax = fmaxf(umax, uq);
            vmin = fminf(vmin, vq);
            vmax = fmaxf(vmax, vq);
          }
          if (zmn > 0.05f) {
            umin -= 1.0f; vmin -= 1.0f; umax += 1.0f; vmax += 1.0f;
            float d_hi2 = 0.0f, d_lo2 = 0.0f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const float p0 = g0[a] * g0[a], p1 = g1[a] * g1[a];
              d_hi2 += fmaxf(p0, p1);
              d_lo2 += (g0[a] <= 0.0f && g1[a] >= 0.0f) ? 0.0f : fminf(p0, p1);
            }
            const float d_hi = __builtin_amdgcn_sqrtf(d_hi2), d_lo = __builtin_amdgcn_sqrtf(d_lo2);
            const bool in_img = (umin >= 0.0f) & (vmin >= 0.0f) & (umax <= (float)(W - 1)) & (vmax <= (float)(H - 1));
            const bool off_img = (umax < 0.0f) | (vmax < 0.0f) | (umin > (float)(W - 1)) | (vmin > (float)(H - 1));
            if (off_img) {
              verdict = HSK_CF_SKIP;  // no voxel has a pixel
            } else {
              // (the box clamped to the image: a voxel whose pixel lies outside is not rewritten whatever the depth there)
              const int tu0 = (int)fminf(fmaxf(umin, 0.0f), (float)(W - 1)) >> 4, tu1 = (int)fminf(fmaxf(umax, 0.0f), (float)(W - 1)) >> 4;
              const int tv0 = (int)fminf(fmaxf(vmin, 0.0f), (float)(H - 1)) >> 4, tv1 = (int)fminf(fmaxf(vmax, 0.0f), (float)(H - 1)) >> 4;
              const int nx = tu1 - tu0 + 1, ny = tv1 - tv0 + 1;
              if ((nx <= (2 << (HSK_SPARSE_LEVELS - 1))) & (ny <= (2 << (HSK_SPARSE_LEVELS - 1)))) {
                // (k_tile_tables: the blocks of 2^kx x 2^ky tiles in the range's four corners cover it)
                const int kx = min(31 - __clz(nx), HSK_SPARSE_LEVELS - 1), ky = min(31 - __clz(ny), HSK_SPARSE_LEVELS - 1);
                const float2* __restrict__ lv = ca.sparse + (size_t)(ky * HSK_SPARSE_LEVELS + kx) * tw16 * th;
                const int ub = tu1 - (1 << kx) + 1, vb = tv1 - (1 << ky) + 1;
                const float2 q00 = lv[tv0 * tw16 + tu0], q01 = lv[tv0 * tw16 + ub], q10 = lv[vb * tw16 + tu0], q11 = lv[vb * tw16 + ub];
                const float Dx = fmaxf(fmaxf(q00.x, q01.x), fmaxf(q10.x, q11.x)), Dn = fminf(fminf(q00.y, q01.y), fminf(q10.y, q11.y));
                const bool dead = d_lo * 0.99999f - Dx > ca.cull_thr;
                const bool in_all = (r.z <= zbeg) & (r.w >= zend - 1);
                const bool fre = full & in_img & in_all & (d_hi * 1.00001f + ca.free_thr <= Dn);
                verdict = dead ? HSK_CF_SKIP : (fre ? HSK_CF_FREE : HSK_CF_MIXED);
              }
            }
          }
        }
        if (verdict == HSK_CF_FREE && ca.cs != nullptr && !lost) {
          const unsigned c = c_now;
          if (c >= 1u && c < 255u) {  // one more pending observation of the whole chunk: the frame's work on it is done
            ca.cs[ci] = (unsigned char)(c + 1u);
            ca.cs[ca.dirty_off + ci] = 1;
            verdict = HSK_CF_SKIP;
          }
        }
        ca.cflag[ci] = (unsigned char)verdict;
      }
    }
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
// (and rewrites) all of its summaries with ONE 16- or 32-bit access, and the bytes of a wave -- round 5: 4 lanes in x by 16
// rows by the chunk's groups, a footprint of 16 x 16 voxels (94 mm square at 512^3) where it was 64 x 4 (375 x 23 mm): a
// compact wave-chunk is far more often all free space or all occluded -- are contiguous.
//
// ---- the coarse level (round 5): one byte per WAVE-CHUNK (16 x 16 voxels x vp.zchunk planes: what one pass-A wave owns) ----
// behind the lane-block bytes, index = lane-block byte index / (64 NS):
//   0          nothing deferred; the chunk's lane-block bytes are current, and nothing is known about them
//   1 + k      k free-space observations of the WHOLE chunk are pending on top of its lane-block bytes, k in 0 .. 254, and
//              every lane-block byte of the chunk is "quiet": 2 .. 255 (all 16 voxels hold +1 with weight >= 1), so that
//              one more free-space observation of the whole chunk is one more count here and nothing else.
// k_column_zrange classifies every wave-chunk of the frame ONCE, conservatively, by its box (coarse_classify): all free space
// in front of the surface (-> the byte is bumped there and then, when it is 1 .. 254), all occluded or outside the frustum
// (-> nothing to do), or mixed.  Pass A's waves read that verdict with a scalar load and leave at once unless it is
// "mixed" (or free with a byte that cannot take the bump): the classification of lane-blocks is O(surface + rim), not
// O(volume).  A wave that does work its chunk first pushes the pending count down into its lanes' bytes
// (hsk_sum_push), and at its end marks the chunk quiet (1) or not (0).  Readers: k_summaries<true> (flush), which pushes
// down as well; hsk_reset (0: the blocks are in state 1, not quiet); hsk_upload_tsdf (k_summaries<false> rebuilds both levels).
#define HSK_SUM_RAGGED 130u
#define HSK_SUM_MAX 255u
#define HSK_LIGHT_PEND 15u  // pending observations of a rim block that a light queue entry can carry (four bits)
// (NS = groups, i.e. summary bytes, per lane and chunk: 2 or 4 -- vp.zchunk / 4; a power of two, so shifts and masks)
static __host__ __device__ __forceinline__ size_t hsk_chunk_count(const VolParams& vp) {  // wave-chunks (4 per workgroup-chunk)
  return (size_t)((vp.nzs + vp.zchunk - 1) / vp.zchunk) * ((vp.Y + 15) / 16) * ((vp.X + 63) / 64) * 4;
}
static __host__ __device__ __forceinline__ size_t hsk_lane_sum_bytes(const VolParams& vp) {  // lane-block bytes; the chunk bytes follow
  return hsk_chunk_count(vp) * (size_t)(64 * (vp.zchunk >> 2));
}
// ... and behind the chunk bytes one DIRTY byte per wave-chunk: 1 = a weight of the chunk may be ahead of the volume's copy
// (a byte was ticked, or an observation is pending in the chunk byte) since the last flush -- the flush (k_summaries<true>)
// rewrites only such chunks: a host that reads a product after every frame pays for the frame's free space, not for the volume
static __host__ __device__ __forceinline__ size_t hsk_chunk_bytes_padded(const VolParams& vp) { return (hsk_chunk_count(vp) + 255) & ~(size_t)255; }
template <int NS>
static __host__ __device__ __forceinline__ size_t hsk_sum_index_ns(const VolParams& vp, int x0, int y, int zb) {
  static_assert(NS == 2 || NS == 4, "8 or 16 planes per chunk");
  const size_t tiles_x = (size_t)(vp.X + 63) / 64, tiles_y = (size_t)(vp.Y + 15) / 16;
  const size_t wg = ((size_t)(zb >> (NS == 4 ? 4 : 3)) * tiles_y + (size_t)(y >> 4)) * tiles_x + (size_t)(x0 >> 6);
  const size_t wave = (size_t)((x0 >> 4) & 3), lane = (size_t)(((y & 15) << 2) | ((x0 >> 2) & 3));
  return ((wg * 4 + wave) * 64 + lane) * (size_t)NS + (size_t)((zb >> 2) & (NS - 1));
}
static __host__ __device__ __forceinline__ size_t hsk_sum_index(const VolParams& vp, int x0, int y, int zb) {
  return vp.zchunk == 16 ? hsk_sum_index_ns<4>(vp, x0, y, zb) : hsk_sum_index_ns<2>(vp, x0, y, zb);
}
// k pending free-space observations of a quiet block (byte s >= 2) pushed into its byte: a uniform block's weight
// saturates at 128 (s = 129); a rim block's pending count simply grows -- and may leave the byte's range, which the callers
// treat as "the words must be rewritten now" (>= HSK_SUM_MAX where a further observation is due, > HSK_SUM_MAX otherwise)
static __host__ __device__ __forceinline__ unsigned hsk_sum_push(unsigned s, unsigned k) {
  return s < HSK_SUM_RAGGED ? (s + k < (unsigned)HSK_MAX_WEIGHT + 1u ? s + k : (unsigned)HSK_MAX_WEIGHT + 1u) : s + k;
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

#ifdef HSK_PA_TIMING
// timing build (tools/pa_timing.sh): per working wave of pass A (slot = its wave-chunk), s_memrealtime stamps (100 MHz) at
// the phase boundaries; slot 7 = free | uncertain << 32 lane-blocks of the chunk
#define HSK_PA_SLOTS 65536u
__device__ unsigned long long g_pa_times[HSK_PA_SLOTS * 8];
extern "C" int hsk_debug_pa_times(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pa_times), (size_t)n * 8);
}
extern "C" int hsk_debug_pa_clear() {
  static unsigned long long zeros[HSK_PA_SLOTS * 8];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pa_times), zeros, sizeof(zeros));
}
#define PA_STAMP(k) do { if (!COUNT_ONLY && lane == 0 && pa_slot < HSK_PA_SLOTS) g_pa_times[pa_slot * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PA_STAMP(k) do { } while (0)  // (phase boundaries of pass A)
#endif
// (below, lane predicates are joined by & and |, without short-circuit evaluation: `a && b` on lane-varying conditions is
// compiled into a lane-mask branch round b -- s_and_saveexec / s_cbranch_execz, scalar instructions, which pass A is short of)
// Pass A of integrate (COUNT_ONLY: the same decisions without touching the volume -- V_upd for the roofline).
template <bool COUNT_ONLY, int NS>
__global__ __launch_bounds__(256, NS == 4 ? INTEGRATE_WPE_LONG : INTEGRATE_WPE) void k_integrate(const TrackState* __restrict__ st, const unsigned* __restrict__ cflag4,
                                                   const int4* __restrict__ wvz, int zchunk, unsigned gxa, unsigned gmagic,
                                                   double* __restrict__ icp_slot0, unsigned char* __restrict__ uni,
                                                   const float2* __restrict__ dtab, int W, int H, int tw, int th, unsigned gya,
                                                   uint4* __restrict__ vol, const float* __restrict__ scaled, VolParams vp,
                                                   Intr in, unsigned long long* __restrict__ counter,
                                                   unsigned* __restrict__ flags,
                                                   unsigned* __restrict__ queue, unsigned* __restrict__ qcount,
                                                   unsigned qcap, const float2* __restrict__ ftab, int fw, int fh,
                                                   const float2* __restrict__ qtab, IntegrateConst k, const int2* __restrict__ zint,
                                                   const unsigned* __restrict__ vmask, int mpitch) {
  // Most of the launch's waves have nothing to do -- their wave-chunk lies outside the view frustum, is wholly occluded, or
  // is wholly free space already recorded in its chunk byte -- and what they execute before they find that out is pure
  // overhead: the test comes FIRST and runs on what arrives with the wave -- the arguments preloaded into SGPRs (the
  // x-block count and its reciprocal for the rotation: `% gridDim.x` was a hidden-argument load and twenty instructions of
  // division) and ONE scalar load, the coarse level's verdicts of the workgroup's four wave-chunks (k_column_zrange).
  // (Round 5 also built the launch as a persistent one over compacted work lists of the chunks that do need work, with
  // static and with ticketed distribution: 23.6 / 31 us against 21.8 -- the loop keeps every argument live (52 scalar
  // registers spilled into vector lanes), every wave starts behind two more dependent round trips, and the dispatcher
  // balances better than either scheme: profiles/r05/integrate_notes.md.)
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
  const unsigned lin = (blockIdx.z * gdy + byr) * gdx + bxr;
  const unsigned wq = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y);  // the wave's quarter of the footprint
  const unsigned verdict = (cflag4[lin] >> (8u * wq)) & 0xffu;
  if (verdict == HSK_CF_SKIP) return;
  const int lane = threadIdx.x;
  unsigned long long cnt = 0;
  if (COUNT_ONLY && verdict == HSK_CF_FREE) {  // every voxel of the chunk is rewritten (F = 1)
    if (lane == 0) atomicAdd(counter, (unsigned long long)(64 * NS * 16));
    return;
  }
  if (!COUNT_ONLY && st->lost) return;
#ifdef HSK_PA_TIMING
  const unsigned pa_slot = lin * 4u + wq;
  if (!COUNT_ONLY && lane == 0 && pa_slot < HSK_PA_SLOTS) {
    for (int q = 0; q < 8; ++q) g_pa_times[pa_slot * 8 + q] = 0ull;
  }
#endif
  PA_STAMP(0);
  // wave footprint: 64 voxels in x (16 lanes x 16 B = 256 contiguous bytes) by 4 rows in y -- compact, so
  // that the wave-uniform z range and the group classification reject whole planes, not just lanes
  // Which x block a workgroup takes rotates with its row and its chunk (bxr above).  Workgroups go round the eight XCDs in
  // turn, and a volume of 512 voxels has exactly eight blocks per row: unrotated, every workgroup of x block b ran on XCD b,
  // and the frustum covers the middle blocks of a row far more than the outer ones -- the XCDs' loads differed by as much
  // (profiles/r02/raycast_split_experiment.md met the same aliasing).
  const int x0 = (int)(bxr * 16u + wq * 4u + (unsigned)(lane & 3)) * 4;
  const int y = (int)(byr * 16u) + (lane >> 2);
  const bool active = (x0 < vp.X) & (y < vp.Y);
  const int zend = min(zbeg + zchunk, vp.nzs);
  // ... and of this wave's own footprint (16 lane columns x 4 rows): the wave-uniform loop bounds, without a wave-wide
  // reduction of the lanes' ranges (min over lanes of max(zl, zbeg) = max(min zl, zbeg)); a wave with nothing to do
  // leaves here, its lanes' ranges never loaded
  const int4 wv = wvz[(size_t)(byr * gdx + bxr) * 4 + wq];
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
    // (slot sidx IS group sidx of the chunk -- round 5: the pending count of the coarse level is pushed into every group's
    // byte, whether the wave's z range reaches the group or not)
    const int zb0 = zbeg;
    int zbs[NS];
    bool actv[NS], in_all_s[NS], free44_s[NS], other_s[NS], light_s[NS];
    float dc_s[NS];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      zbs[sidx] = zb0 + 4 * sidx;
      actv[sidx] = (zbs[sidx] + 3 >= wl) & (zbs[sidx] <= wh);  // wave-uniform
      free44_s[sidx] = other_s[sidx] = in_all_s[sidx] = light_s[sidx] = false;
      dc_s[sidx] = 0.0f;
    }
    // ---- stage 1 (round 5: the levels changed places).  With the coarse level above it, pass A only meets chunks that hold
    // a surface, the frustum's rim or the camera's neighbourhood: the 16-px level, built to discard the bulk of free and
    // occluded space cheaply, decided little there that the pixel-box level did not decide as well (its table is coarser
    // and dilated, its distance bound looser: whenever the pixel-box level applies -- ok2 -- it decides at least as much),
    // and every mixed wave went through both.  The pixel-box level now comes first, for every lane-block inside the lane's
    // z range; the 16-px level is the fall-back for the blocks it cannot look up (a pixel box wider than 3 x 3 tiles of
    // 8 px: blocks within half a metre of the camera, or one that leaves the image).
    //      The 16 voxel centres span a parallelogram in camera space, whose projection is a convex quadrilateral, so the
    //      pixel box of the four projected corners (+-1 px for rounding) holds all 16 pixels; its exact min / max depth
    //      comes from the undilated 4-px or 8-px tile table (<= 3x3 tiles).  Every block decided is one less entry for pass B.
    bool in_any_s[NS], ok2_s[NS];
    unsigned box_s[NS];
    float2 t9_s[NS];
    float dlo_s[NS], dhi_s[NS];
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
    // (both groups' look-ups are requested before either is used, the lane's summaries behind them: loads return in the
    // order they were issued, and the summaries -- an 8 MiB table that misses where the tile tables hit -- are not needed
    // before stage 3)
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int zb = zbs[sidx];
      in_any_s[sidx] = actv[sidx] & (zb + 3 >= zl) & (zb <= zh) & active;
      in_all_s[sidx] = actv[sidx] & (zb >= zl) & (zb + 3 <= zh) & active;
      ok2_s[sidx] = false;
      box_s[sidx] = 0u;
      t9_s[sidx] = make_float2(0.0f, 0.0f);
      dlo_s[sidx] = dhi_s[sidx] = 0.0f;
      if (!actv[sidx]) continue;  // wave-uniform
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
      // The PIXELS of the 16 voxels: a voxel's pixel is the nearest integer to its projection, the projections lie between
      // the corners', and these corner values differ from what the exact rule computes by rounding errors of 1e-4 px at
      // most -- so the pixels lie in [rint(umin - 1/64), rint(umax + 1/64)].  (Round 5: the box was floor(umin - 1) ..
      // floor(umax + 1), two pixels more in either direction, which for the typical block -- 4 voxels across, ONE row -- is
      // a box of 11 x 4 pixels where 9 x 2 hold them all: more tiles under it, fewer decisions, and with holes in the depth
      // image three times the chance of meeting one.)
      const float ue = 0.015625f;
      const bool front = zmn > 0.05f;
      const int iu0 = (int)rintf(fminf(fmaxf(front ? umin - ue : -8.0f, -8.0f), 65536.0f)), iu1 = (int)rintf(fminf(fmaxf(front ? umax + ue : -8.0f, -8.0f), 65536.0f));
      const int iv0 = (int)rintf(fminf(fmaxf(front ? vmin - ue : -8.0f, -8.0f), 65536.0f)), iv1 = (int)rintf(fminf(fmaxf(front ? vmax + ue : -8.0f, -8.0f), 65536.0f));
      // the 4-px table when the box spans at most 3 x 3 of its tiles, else the 8-px one (at most 3 x 3 again, else undecided)
      const bool in_img = front & (iu0 >= 0) & (iv0 >= 0) & (iu1 <= W - 1) & (iv1 <= H - 1);
      // (the box, for the validity mask below: 11 + 11 bits of its corner, 5 + 2 of its extent; wider or taller, or an image
      // beyond 2048 pixels a side: no look-up)
      box_s[sidx] = ((iu1 - iu0 < 32) & (iv1 - iv0 < 4) & in_img & (W <= 2048) & (H <= 2048)) ? ((unsigned)iu0 | ((unsigned)iv0 << 11) | ((unsigned)(iu1 - iu0) << 22) | ((unsigned)(iv1 - iv0) << 27) | (1u << 29)) : 0u;
      const bool fine = ((iu1 >> 2) <= (iu0 >> 2) + 2) & ((iv1 >> 2) <= (iv0 >> 2) + 2);
      const int sh = fine ? 2 : 3;
      const int tu0 = iu0 >> sh, tv0 = iv0 >> sh;
      const int nx = (iu1 >> sh) - tu0, ny = (iv1 >> sh) - tv0;  // tiles spanned, less one
      ok2_s[sidx] = in_img & (nx <= 2) & (ny <= 2);
      const float2* __restrict__ tab = fine ? qtab : ftab;
      const int tbw = fine ? 2 * fw : fw, tbh = fine ? 2 * fh : fh;
      // (one look-up: table (nx, ny) holds the (max, min) of the nx x ny tiles from each tile on, k_tile_tables)
      const int shape = min(max(ny, 0), 2) * 3 + min(max(nx, 0), 2);
      t9_s[sidx] = tab[(size_t)shape * tbw * tbh + min(max(tv0, 0), tbh - 1) * tbw + min(max(tu0, 0), tbw - 1)];
      // exact distance range of the block: its 16 voxel centres lie in the rectangle [gx0, gx3] x {gy} x [gza, gzb], over
      // which the distance to the camera centre is largest at a corner and smallest where each coordinate is nearest 0
      const float gz2_hi = fmaxf(gza * gza, gzb * gzb);
      const float gz2_lo = (gza <= 0.0f && gzb >= 0.0f) ? 0.0f : fminf(gza * gza, gzb * gzb);
      dhi_s[sidx] = __builtin_amdgcn_sqrtf(pn_hi + gz2_hi);
      dlo_s[sidx] = __builtin_amdgcn_sqrtf(pn_lo + gz2_lo);
    }
    unsigned sum16 = 0u;  // all summaries of the lane's chunk (group g of the chunk in byte g): used in stage 3
    unsigned char* const sum_at = uni + ((size_t)(lin * 4u + wq) * 64u + (unsigned)lane) * (unsigned)NS;  // (hsk_sum_index_ns<NS>(vp, x0, y, zbeg))
    // ... and the chunk's byte of the coarse level (its workgroup's four as one scalar word): 1 + k = k observations pending
    unsigned cbyte = 0u;
    if (!COUNT_ONLY && uni != nullptr) {
      if (active) sum16 = NS == 2 ? (unsigned)*(const unsigned short*)sum_at : *(const unsigned*)sum_at;
      cbyte = (((const unsigned*)(uni + hsk_lane_sum_bytes(vp)))[lin] >> (8u * wq)) & 0xffu;
    }
    const unsigned kpend = cbyte >= 2u ? cbyte - 1u : 0u;
    PA_STAMP(6);
    bool need1 = false;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      bool dead2 = ok2_s[sidx] & (dlo_s[sidx] * 0.99999f - t9_s[sidx].x > k.cull_thr2);
      // free space: every pixel of the box WITH depth lies far enough behind the block (.y: the minimum over those,
      // negated when a pixel under the tiles has none) -- and every pixel of the box has depth: by the tiles when they say
      // so, else by the validity mask over the box itself.  A depth image with holes (a real sensor's: 2 % of the pixels
      // in SURVEY.md 8(d)'s noise run) left no tile window whole, and every block of free space went to pass B.
      const bool free2v = in_all_s[sidx] & ok2_s[sidx] & (dhi_s[sidx] * 1.00001f + k.free_thr2 <= fabsf(t9_s[sidx].y));
      bool whole = t9_s[sidx].y > 0.0f;
      const bool ask = free2v & !whole & (box_s[sidx] != 0u);
      if (__ballot(ask) != 0ull) {
        unsigned bad = 1u, allbad = 0u;
        const unsigned bx = box_s[sidx];
        const int iu0 = (int)(bx & 2047u), iv0 = (int)((bx >> 11) & 2047u), wd = (int)((bx >> 22) & 31u), ht = (int)((bx >> 27) & 3u);
        // (round 6) A box of at most 17 columns -- nearly every one: four voxels across -- lies inside the 32 pixels that start
        // at the 16-px boundary below it, and the mask's rows are arrays of 16-bit halves: ONE 32-bit load at a halfword
        // address per row of the box, and only the rows the box has (it was two words for each of four rows, the last row
        // repeated: eight loads a block where two or three do -- under SURVEY.md 8(d)'s 2 % dropout nearly every free block asks)
        const bool narrow = ask & (wd <= 16);
        if (__ballot(narrow) != 0ull) {
          const unsigned short* __restrict__ row16 = (const unsigned short*)vmask + (size_t)iv0 * (size_t)(2 * mpitch) + (iu0 >> 4);
          const unsigned sel = (unsigned)(((1ull << (wd + 1)) - 1ull) << (iu0 & 15));
          unsigned any = 0u, all = ~0u;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const bool rowh = narrow & (ht >= h);
            if (h != 0 && __ballot(rowh) == 0ull) break;  // wave-uniform
            if (rowh) {
              unsigned w2;
              __builtin_memcpy(&w2, row16 + (size_t)h * (size_t)(2 * mpitch), 4);  // (one global_load_dword at a 2-byte-aligned address)
              any |= w2 & sel;
              all &= w2 | ~sel;
            }
          }
          if (narrow) {
            bad = any != 0u ? 1u : 0u;
            allbad = all == ~0u ? 1u : 0u;
          }
        }
        const bool wide = ask & (wd > 16);
        if (__ballot(wide) != 0ull) {
          if (wide) {
            const unsigned* __restrict__ row = vmask + (size_t)iv0 * mpitch + (iu0 >> 5);
            const unsigned long long sel = ((1ull << (wd + 1)) - 1ull) << (iu0 & 31);  // (wd + 1 <= 32 columns from bit iu0 & 31: never beyond the row's two words)
            unsigned long long any = 0ull, all = ~0ull;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
              const unsigned* __restrict__ rp = row + (size_t)min(h, ht) * mpitch;
              const unsigned long long w2 = (unsigned long long)rp[0] | ((unsigned long long)rp[1] << 32);
              any |= w2 & sel;
              all &= w2 | ~sel;
            }
            bad = any != 0ull ? 1u : 0u;
            allbad = all == ~0ull ? 1u : 0u;
          }
        }
        whole = whole | (ask & (bad == 0u));
        // (round 6) every pixel of the box is a hole -- the inside of a shadow, of an absorbing surface's silhouette: no voxel
        // of the block has a pixel with depth, the rule writes nothing
        dead2 = dead2 | (ask & (allbad != 0u));
      }
      const bool free2 = free2v & whole;
      free44_s[sidx] = free2;
      // (round 6) THE LIGHT CLASS: every pixel of the box that HAS depth lies a truncation distance or more behind the block
      // (free2v), but the box holds holes.  Each voxel is then either rewritten with F = 1 (its pixel has depth) or left alone
      // (it has none): pass B needs its pixel and one bit -- no distance, no square root, no division -- where the per-voxel
      // path spends ~60 vector instructions a voxel.  A depth image with holes (SURVEY.md 8(d)'s noise run: every third
      // lane-block's box holds one) sent all of these through that path.
      const bool light2 = free2v & !whole & !dead2;
      light_s[sidx] = light2;
      other_s[sidx] = in_any_s[sidx] & !dead2 & !free2 & !light2;
      need1 = need1 | (in_any_s[sidx] & !ok2_s[sidx]);
    }
    // the lane's summaries with the chunk's pending observations pushed into them (kpend > 0 only over quiet bytes); a rim
    // block's count may leave the byte's range: its words are rewritten below
    unsigned sum8[NS];
    unsigned new16 = 0u;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      sum8[sidx] = hsk_sum_push((sum16 >> (8 * sidx)) & 0xffu, kpend);
      new16 |= min(sum8[sidx], HSK_SUM_MAX) << (8 * sidx);  // (an out-of-range count is replaced by the block's new state below)
    }
    PA_STAMP(2);
    // ---- stage 2: the 16-px level (dilated tile table against the block's centre +- radius) for the blocks the pixel-box
    // level could not look up -- only in waves that hold such a block
    if (__ballot(need1) != 0ull) {
      float2 Dt_s[NS];
      bool ok_s[NS];
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) {
        const int zb = zbs[sidx];
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
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) {
        const float dc = dc_s[sidx];
        const bool dead4 = ok_s[sidx] & (dc * 0.99999f - Dt_s[sidx].x > k.cull_thr4);
        const bool free44 = in_all_s[sidx] & ok_s[sidx] & (dc * 1.00001f + k.free_thr4 <= Dt_s[sidx].y);
        free44_s[sidx] = free44_s[sidx] | (other_s[sidx] & free44);
        other_s[sidx] = other_s[sidx] & !dead4 & !free44;
      }
    }
    PA_STAMP(3);
#ifdef HSK_PA_TIMING
    if (!COUNT_ONLY && pa_slot < HSK_PA_SLOTS) {
      unsigned nf = 0, no = 0;
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) {
        nf += (unsigned)__popcll(__ballot(free44_s[sidx] && actv[sidx]));
        no += (unsigned)__popcll(__ballot(other_s[sidx] && actv[sidx]));
      }
      if (lane == 0) g_pa_times[pa_slot * 8 + 7] = nf | ((unsigned long long)no << 32);
    }
#endif
    // ---- stage 4: wave-aggregated append of the uncertain lane-blocks: one of HSK_NQUEUES queues (a single counter
    // saturates at ~88 atomics/us chip-wide).  The queue must NOT follow the block's x-y position: surfaces cluster in a
    // few columns, and pass B's time is its longest queue.  Rotate the assignment by the row of HSK_NQUEUES blocks and by
    // the wave: within a row it stays a bijection, so every queue still receives at most one wave-quarter of one block
    // per row and wave index (the capacity bound).  Both groups' tickets are requested before either is used.
    // (the tickets are REQUESTED here, before the free-space loads, and used after the stores: the counters' round trip
    // runs under the volume's -- one dependent round trip less in a wave's life)
    const unsigned qi = (lin + (lin / HSK_NQUEUES) * 37u + wq * (HSK_NQUEUES / 4)) % HSK_NQUEUES;
    // (ONE ticket for the wave's groups: every vector-memory instruction counts, see the note on k_tile_tables)
    // (round 6: a queue is filled from both ends -- the per-voxel entries from its head, the light ones from its tail; the
    // two counts are the halves of ONE 64-bit counter, so a wave still draws one ticket)
    unsigned long long bo[NS], bl[NS];
    unsigned lcode[NS];  // what a light entry carries in its top four bits: pending observations pass B is to add (0: none)
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) lcode[sidx] = 0u;
    unsigned n_other = 0u, n_light = 0u;
    unsigned long long base_all = 0ull;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      bo[sidx] = actv[sidx] ? __ballot(other_s[sidx]) : 0ull;
      bl[sidx] = actv[sidx] ? __ballot(light_s[sidx]) : 0ull;
      n_other += (unsigned)__popcll(bo[sidx]);
      n_light += (unsigned)__popcll(bl[sidx]);
    }
    if ((n_other | n_light) != 0u && lane == 0)
      base_all = atomicAdd((unsigned long long*)&qcount[qi * HSK_QCOUNT_STRIDE], (unsigned long long)n_other | ((unsigned long long)n_light << 32));
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
        const bool fr = actv[sidx] & free44_s[sidx], lt = actv[sidx] & light_s[sidx], ot = (actv[sidx] & other_s[sidx]) | lt;
        const unsigned sm = sum8[sidx];
        const int sbit = 8 * sidx;  // where the group's summary sits in new16
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
          const bool tick = fr && sm != 0u && sm != (unsigned)HSK_MAX_WEIGHT + 1u && sm < HSK_SUM_MAX;
          if (tick || (ot && sm != 0u)) new16 = (new16 & ~(0xffu << sbit)) | ((tick ? sm + 1u : 0u) << sbit);
        }
        // (b) needs the words: a free block in state 0, or a rim block whose pending count is full; a rim block with
        //     pending observations on its way to pass B; a rim block the frame does not touch whose count, with the chunk's
        //     pushed into it, no longer fits the byte
        // (round 6: a LIGHT rim block with up to HSK_LIGHT_PEND pending observations keeps them: its queue entry carries the
        // count and pass B, which reads and rewrites the words anyway, adds it -- the block's read-modify-write happens once,
        // on a dense wave, instead of twice, the first time inside a mixed wave of this kernel)
        lcode[sidx] = (lt && sm > HSK_SUM_RAGGED && sm <= HSK_SUM_RAGGED + HSK_LIGHT_PEND) ? sm - HSK_SUM_RAGGED : 0u;
        rd[sidx] = (fr && (sm == 0u || sm >= HSK_SUM_MAX)) || (ot && sm > HSK_SUM_RAGGED && lcode[sidx] == 0u) || sm > HSK_SUM_MAX;
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
          if (!(actv[sidx] && (other_s[sidx] | light_s[sidx]))) {  // (a block on its way to pass B has lost its summary above)
            const int sbit = 8 * sidx;
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
      // the chunk's byte: nothing pending any more; quiet (1) when the chunk is whole and every one of its blocks now holds
      // +1 throughout with a weight (bytes 2 .. 255), so that a free-space observation of all of it is one count -- else 0
      if (uni != nullptr) {
        bool quiet = active;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) quiet = quiet & (((new16 >> (8 * sidx)) & 0xffu) >= 2u);
        const bool whole = (zbeg + vp.zchunk <= vp.nzs) & (zchunk == vp.zchunk);
        const unsigned cnew = (whole && __ballot(quiet) == ~0ull) ? 1u : 0u;
        if (cnew != cbyte && lane == 0) (uni + hsk_lane_sum_bytes(vp))[lin * 4u + wq] = (unsigned char)cnew;
        // (a summary of the chunk moved: some weight may now be ahead of the volume's copy -- the flush's business)
        if (__ballot(new16 != sum16) != 0ull && lane == 0) (uni + hsk_lane_sum_bytes(vp) + hsk_chunk_bytes_padded(vp))[lin * 4u + wq] = 1;
      }
    }
#ifdef HSK_PA_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stamp below sees the loads back and the stores acknowledged
#endif
    PA_STAMP(4);
    unsigned b0 = 0u, l0 = 0u;
    if ((n_other | n_light) != 0u) {
      b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base_all);
      l0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base_all >> 32));
    }
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      if ((bo[sidx] | bl[sidx]) == 0ull) continue;
      const unsigned id = (unsigned)(((zbs[sidx] >> 2) * vp.Y + y) * qx + (x0 >> 2));
      if (other_s[sidx]) {
        // the entry: lane-block id, and (when it fits: id_mask_shift != 0) which of its 4 planes lie in the lane's z range
        unsigned pm = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) pm |= ((zbs[sidx] + u >= zl && zbs[sidx] + u <= zh) ? 1u : 0u) << u;
        queue[(size_t)qi * qcap + b0 + (unsigned)__popcll(bo[sidx] & ((1ull << lane) - 1ull))] = id | (pm << 28);
      }
      // (a light block lies with all four planes in the lane's range -- in_all is part of its class: the id alone, from the tail)
      if (light_s[sidx]) queue[(size_t)qi * qcap + (qcap - 1u) - (l0 + (unsigned)__popcll(bl[sidx] & ((1ull << lane) - 1ull)))] = id | (lcode[sidx] << 28);
      b0 += (unsigned)__popcll(bo[sidx]);
      l0 += (unsigned)__popcll(bl[sidx]);
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
// Pass B, third form (round 6): A LANE IS ONE PLANE OF ONE LANE-BLOCK -- four voxels, one 16-B vector -- and the four lanes
// of a quad are the four planes of one queue entry.  The second form gave a lane the whole lane-block: every load or store
// instruction of a wave then touched 64 different 64-B blocks, 16 bytes of each, four instructions in a row going back to
// the same 64 blocks -- "the memory side's acceptance rate for scattered 64-B blocks" was half of the kernel
// (profiles/r04/integrate_notes.md), 93 vector registers held it at five waves per SIMD, and knocking the arithmetic out of
// its light class bought nothing (profiles/r06/integrate_notes.md: the light entries cost what the per-voxel ones do).
// Here an instruction of a wave covers 16 blocks WHOLE (a quad's four vectors are 64 contiguous bytes: one request where
// there were four quarter requests), a trip is a quarter as long and there are four times as many, and the register count
// falls to what eight waves per SIMD allow.  What the four planes of an entry share -- the x-y terms of the projection --
// is computed by each of the four lanes (a dozen multiply-adds of ~200); what they must agree on -- the brick flag, the
// light class's summary byte -- is settled by quad shuffles.  The arithmetic per voxel is the second form's, expression for
// expression: the same bits.
// ------------------------------------------------------------------------------------------------------
#ifndef DETAIL3_WPE
#define DETAIL3_WPE 8
#endif
#ifndef DETAIL3_GX
#define DETAIL3_GX 16  // DETAIL3_GX x 256 blocks of 4 waves: TWO resident rounds of the chip at 8 waves per SIMD -- the second round
#endif                 // rebalances the first (tools/experiments/detail3_ab.sh, pass B medians at 8 / 12 / 16 / 24 / 32 / 48: 512^3 21.6 /
                       // 21.4 / 21.7-22.0 / 23.1 / 23.2 / 25.7 us, noise 49.4 / 50.0 / 47.2-47.5 / 48.8 / 48.7 / 51.6, 1024^3 121.7-122.8 /
                       // 115.3 / 114.5-116.5 / 113.7 / 113.3 / 114.0)
static __device__ __forceinline__ unsigned quad_or(unsigned v) {
  v |= (unsigned)__shfl_xor((int)v, 1, 64);
  return v | (unsigned)__shfl_xor((int)v, 2, 64);
}
static __device__ __forceinline__ unsigned quad_and(unsigned v) {
  v &= (unsigned)__shfl_xor((int)v, 1, 64);
  return v & (unsigned)__shfl_xor((int)v, 2, 64);
}
static __device__ __forceinline__ unsigned quad_min(unsigned v) {
  v = min(v, (unsigned)__shfl_xor((int)v, 1, 64));
  return min(v, (unsigned)__shfl_xor((int)v, 2, 64));
}
template <bool COUNT_ONLY>
__global__ __launch_bounds__(256, DETAIL3_WPE) void k_integrate_detail3(const TrackState* __restrict__ st, const unsigned* __restrict__ qcount,
                                                                        const unsigned* __restrict__ queue_all, unsigned qcap, int W, int H,
                                                                        uint4* __restrict__ vol, const float* __restrict__ scaled, Intr in,
                                                                        VolParams vp, unsigned long long* __restrict__ counter,
                                                                        unsigned* __restrict__ flags, unsigned long long qmag_x,
                                                                        unsigned long long qmag_y, unsigned char* __restrict__ uni) {
  if (!COUNT_ONLY && st->lost) return;
  const int lane = threadIdx.x & 63;
  // the two lists (per-voxel entries at the heads of the queues, light ones at their tails): prefix sums of both counts
  __shared__ unsigned pre[HSK_NQUEUES + 1], pre2[HSK_NQUEUES + 1];
  {
    static_assert(HSK_NQUEUES == 256, "one counter per thread of the block");
    const uint2 c = *(const uint2*)&qcount[threadIdx.x * HSK_QCOUNT_STRIDE];
    unsigned incl = c.x, incl2 = c.y;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned v = (unsigned)__shfl_up((int)incl, o, 64), v2 = (unsigned)__shfl_up((int)incl2, o, 64);
      if (lane >= o) {
        incl += v;
        incl2 += v2;
      }
    }
    __shared__ unsigned wsum[4], wsum2[4];
    if (lane == 63) {
      wsum[threadIdx.x >> 6] = incl;
      wsum2[threadIdx.x >> 6] = incl2;
    }
    __syncthreads();
    unsigned base = 0, base2 = 0;
    for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) {
      base += wsum[w];
      base2 += wsum2[w];
    }
    pre[threadIdx.x + 1] = base + incl;
    pre2[threadIdx.x + 1] = base2 + incl2;
    if (threadIdx.x == 0) pre[0] = pre2[0] = 0u;
    __syncthreads();
  }
  const unsigned n = pre[HSK_NQUEUES], n2 = pre2[HSK_NQUEUES];
  const unsigned nwaves = gridDim.x * (blockDim.x >> 6), wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned stride = nwaves * 16u;  // entries a round of the grid takes: 16 per wave
  const unsigned cap = ((unsigned)HSK_MAX_WEIGHT << 16) | (unsigned)HSK_DIVISOR;
  const DetailPose P = detail_pose(st);
  const int qx = vp.X / 4, u = lane & 3;
  const int last = W * H - 1;
  unsigned long long cnt = 0;
  // g-th entry of a concatenated list (0 beyond its end): its queue by bisection over the prefix sums, heads count up, tails down
  auto entry_at = [&](const unsigned* __restrict__ pr, unsigned total, bool tail, unsigned g) -> unsigned {
    if (g >= total) return 0u;
    unsigned lo = 0, hi = HSK_NQUEUES;  // pr[lo] <= g < pr[hi]
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const unsigned mid = (lo + hi) >> 1;
      if (pr[mid] <= g) lo = mid; else hi = mid;
    }
    const unsigned r = g - pr[lo];
    return queue_all[(size_t)lo * qcap + (tail ? (qcap - 1u) - r : r)];
  };
  // one trip: the lane's plane of its quad's entry.  LIGHT: the class of pass A's free blocks over holes (every plane in
  // range, F = 1 wherever the pixel has depth, `code` = pending observations of a rim block to add first; the block's
  // summary byte is rewritten from the new words); else the specification's per-voxel rule, `code` = the planes in range.
  auto work = [&](bool have, unsigned id, bool LIGHT) {
    const unsigned lb = id & 0x0fffffffu, code = id >> 28;
    // lb = ((zb / 4) * Y + y) * qx + x0 / 4, taken apart by multiplications (qmag = floor(2^40 / d) + 1: exact for every id)
    const unsigned row = (unsigned)(((unsigned long long)lb * qmag_x) >> 40);
    const unsigned zq = (unsigned)(((unsigned long long)row * qmag_y) >> 40);
    const int x0 = (int)(lb - row * (unsigned)qx) * 4, y = (int)(row - zq * (unsigned)vp.Y);
    const int zb = (int)zq * 4;
    const bool inr = have && (LIGHT || ((code >> u) & 1u) != 0u);
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    if (!COUNT_ONLY && inr) q = vol[VIDX(zb, u)];
    unsigned flag_word = 0u;
    const int fbit = ((zb >> vp.bshift) * (vp.Y >> vp.bshift) + (y >> vp.bshift)) * (vp.X >> vp.bshift) + (x0 >> vp.bshift);
    if (!COUNT_ONLY && !LIGHT && inr) flag_word = flags[fbit >> 5];
    const float gy = ((float)y + 0.5f) * vp.cell[1] - P.ty;
    const float gz = ((float)(vp.zs0 + zb + u) + 0.5f) * vp.cell[2] - P.tz;
    const float bx = P.i02 * gz, by = P.i12 * gz, bz = P.i22 * gz;
    const float gz2 = gz * gz;
    int pix[4];
    float pn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gx = ((float)(x0 + j) + 0.5f) * vp.cell[0] - P.tx;
      const float axj = P.i00 * gx + P.i01 * gy, ayj = P.i10 * gx + P.i11 * gy, azj = P.i20 * gx + P.i21 * gy;
      pn[j] = gx * gx + gy * gy;
      const float camz = azj + bz;
      const float inv_z = hsk_rcp_exact(camz);
      const float fu = ((axj + bx) * in.fx) * inv_z + in.cx;
      const float fv = ((ayj + by) * in.fy) * inv_z + in.cy;
      const int uu = (int)rintf(fu), vv = (int)rintf(fv);
      // (the light class has every pixel inside the image and in front of the camera by its classification: the clamp only
      // guards the address)
      const bool ok = LIGHT ? true : (inr && camz >= 1.17549435e-38f && (unsigned)uu < (unsigned)W && (unsigned)vv < (unsigned)H);
      pix[j] = ok ? min(max(vv * W + uu, 0), last) : -1;
    }
    float D[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) D[j] = scaled[max(pix[j], 0)];
    // the observation: F in [-1, 1] for a voxel the rule rewrites, -4 for one it leaves alone
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float Ds = pix[j] >= 0 ? D[j] : 0.0f;
      if (LIGHT) {
        D[j] = (inr && Ds != 0.0f) ? 1.0f : -4.0f;
      } else {
        const float sdf = Ds - hsk_sqrt_exact(gz2 + pn[j]);
        const float f = sdf * vp.tau_inv;
        D[j] = (Ds != 0.0f && sdf >= -vp.tau) ? (f < 1.0f ? f : 1.0f) : -4.0f;
      }
    }
    if (COUNT_ONLY) {
#pragma unroll
      for (int j = 0; j < 4; ++j) cnt += D[j] > -2.0f ? 1u : 0u;
      return;
    }
    // running mean, repack, store
    uint4 q1 = q;
    if (LIGHT && code != 0u) hsk_vector_add_weight(q1, code);  // (a rim block's pending observations: all 16 voxels hold +1 with a weight)
    const unsigned w4[4] = {q1.x, q1.y, q1.z, q1.w};
    unsigned nw[4];
    bool gen[4], gen_any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool upd = D[j] > -2.0f;
      const bool unseen = (w4[j] >> 16) == 0u;
      // free space (F == 1) onto a stored +1, or onto an unseen voxel: the mean is +1 exactly, only the weight moves
      const bool simple = D[j] == 1.0f && (unseen || (w4[j] & 0xffffu) == (unsigned)HSK_DIVISOR);
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
        const float Fn = hsk_div_small_exact(hsk_tsdf_unpack(tp) * Wp + D[j], Wp + 1.0f);
        int fixed = (int)(Fn * 32767.0f);  // truncation toward zero
        fixed = min(max(fixed, -HSK_DIVISOR), HSK_DIVISOR);
        const unsigned wg = ((unsigned)fixed & 0xffffu) | (min(wp + 1u, (unsigned)HSK_MAX_WEIGHT) << 16);
        nw[j] = gen[j] ? wg : nw[j];
        neg = neg || (gen[j] && fixed < 0);
      }
    }
    // (saturated free space -- +1 at the weight cap -- comes back unchanged: no store)
    if (inr && (nw[0] != q.x || nw[1] != q.y || nw[2] != q.z || nw[3] != q.w)) vol[VIDX(zb, u)] = make_uint4(nw[0], nw[1], nw[2], nw[3]);
    if (!LIGHT) {
      // the brick of the lane-block (its 4 planes lie in one brick): one lane of the quad marks it when any plane went negative
      const unsigned negq = quad_or(neg ? 1u : 0u), fw = quad_or(flag_word);
      if (negq != 0u && u == 0 && ((fw >> (fbit & 31)) & 1u) == 0u) mark_brick_negative(flags, vp, x0, y, zb);
    } else if (uni != nullptr) {
      // the block's state from its 16 new words (hsk_sum_classify across the quad): all equal -> the uniform code; all +1 with
      // a weight -> a rim block without pending observations; else nothing is known (pass A left the byte at 0)
      const unsigned word = nw[0];
      const unsigned same = (nw[1] == word && nw[2] == word && nw[3] == word) ? 1u : 0u;
      const unsigned w_first = (unsigned)__shfl((int)word, lane & ~3, 64);
      const unsigned all_same = quad_and((same != 0u && word == w_first) ? 1u : 0u);
      const unsigned a = quad_and(nw[0] & nw[1] & nw[2] & nw[3]), o = quad_or(nw[0] | nw[1] | nw[2] | nw[3]);
      const unsigned m = quad_min(min(min(nw[0], nw[1]), min(nw[2], nw[3])));
      const unsigned ragged = ((a & 0x7fffu) == 0x7fffu && (o & 0x8000u) == 0u && m >= 0x10000u && (o >> 16) <= (unsigned)HSK_MAX_WEIGHT) ? HSK_SUM_RAGGED : 0u;
      const unsigned state = all_same != 0u ? hsk_uniform_code(w_first) : ragged;
      if (have && u == 0 && state != 0u) uni[hsk_sum_index(vp, x0, y, zb)] = (unsigned char)state;
    }
  };
  // ---- the per-voxel list: wave w takes entries 16 w .. 16 w + 15 of every round; the next trip's entry is fetched under
  // the current one (its address needs nothing but the trip index)
  {
    unsigned e0 = wave * 16u;
    unsigned id_next = entry_at(pre, n, false, e0 + (unsigned)(lane >> 2));
    for (; e0 < n; e0 += stride) {  // wave-uniform trip count
      const unsigned id = id_next;
      id_next = entry_at(pre, n, false, e0 + stride + (unsigned)(lane >> 2));
      work(e0 + (unsigned)(lane >> 2) < n, id, false);
    }
  }
  // ---- the light list: the waves that came last out of the list above go first into this one
  if (n2 != 0u) {
    unsigned g0 = (nwaves - 1u - wave) * 16u;
    unsigned id_next = entry_at(pre2, n2, true, g0 + (unsigned)(lane >> 2));
    for (; g0 < n2; g0 += stride) {
      const unsigned id = id_next;
      id_next = entry_at(pre2, n2, true, g0 + stride + (unsigned)(lane >> 2));
      work(g0 + (unsigned)(lane >> 2) < n2, id, true);
    }
  }
  if (COUNT_ONLY) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (lane == 0 && cnt) atomicAdd(counter, cnt);
  }
}

// the derived tables behind the raw ones (what pass A and the coarse level read): the window forms of the 8-px and 4-px
// tables (nine shapes each), then the sparse table of the 16-px tiles -- one launch
void launch_tile_tables(hipStream_t s, int W, int H, float* tiles) {
  const int tw = (W + HSK_TILE - 1) / HSK_TILE, th = (H + HSK_TILE - 1) / HSK_TILE;
  const int fw = (W + HSK_FTILE - 1) / HSK_FTILE, fh = (H + HSK_FTILE - 1) / HSK_FTILE;
  float2* ftab = (float2*)(tiles + 4 * tw * th);
  float2* qtab = ftab + (size_t)fw * fh;
  float2* fwin = qtab + (size_t)4 * fw * fh;
  float2* sparse = fwin + (size_t)45 * fw * fh;
  hipLaunchKernelGGL(k_tile_tables, dim3((4 * fw * fh + 255) / 256, 18 + HSK_SPARSE_LEVELS * HSK_SPARSE_LEVELS), dim3(256), 0, s, tiles, tiles + tw * th, tw,
                     th, (const float2*)ftab, (const float2*)qtab, fw, fh, fwin, sparse);
}
// stage paths (the scaled depth alone, no bilateral launch): the raw 8-px and 4-px tables, then the derived ones
void launch_tile_fine(hipStream_t s, const float* scaled, int W, int H, float* tiles) {
  const int tw = (W + HSK_TILE - 1) / HSK_TILE, th = (H + HSK_TILE - 1) / HSK_TILE;
  const int fw = (W + HSK_FTILE - 1) / HSK_FTILE, fh = (H + HSK_FTILE - 1) / HSK_FTILE;
  float2* ftab = (float2*)(tiles + 4 * tw * th);
  float2* qtab = ftab + (size_t)fw * fh;
  hipLaunchKernelGGL(k_tile_fine, dim3((fw * fh + 255) / 256), dim3(256), 0, s, scaled, W, H, ftab, fw, fh, qtab);
  launch_tile_tables(s, W, H, tiles);
}
// bytes of a frame's tile tables (layout: hsk_dev.h)
size_t tile_table_bytes(int W, int H) { return hsk_tiles_floats(W, H) * sizeof(float); }

// entries of the z-range tables: one int2 per lane column, then one per pass-A workgroup footprint (64 x 16 voxels), then
// an int4 per wave footprint (64 x 4 voxels: four per workgroup)
size_t integrate_zint_entries(const VolParams& vp) {
  const size_t fp = (size_t)((vp.X + 63) / 64) * ((vp.Y + 15) / 16);
  // (the wave table is int4: kept 16-B aligned; behind it the coarse level's verdict bytes, one per wave-chunk)
  return (size_t)(vp.X / 4) * vp.Y + ((fp + 1) & ~(size_t)1) + 8 * fp + (hsk_chunk_count(vp) + 7) / 8;
}
// where the coarse level's verdict bytes of the last frame sit in launch_integrate's zint buffer, and how many there are
size_t integrate_cflag_offset_bytes(const VolParams& vp) {
  const size_t fp = (size_t)((vp.X + 63) / 64) * ((vp.Y + 15) / 16);
  return ((size_t)(vp.X / 4) * vp.Y + ((fp + 1) & ~(size_t)1) + 8 * fp) * sizeof(int2);
}
size_t integrate_chunk_count(const VolParams& vp) { return hsk_chunk_count(vp); }
size_t integrate_queue_counter_words() { return (size_t)HSK_NQUEUES * HSK_QCOUNT_STRIDE; }
unsigned long long integrate_queue_entries(const unsigned* counter_words) {
  unsigned long long n = 0;
  for (size_t q = 0; q < HSK_NQUEUES; ++q) n += counter_words[q * HSK_QCOUNT_STRIDE];
  return n;
}
unsigned long long integrate_queue_light_entries(const unsigned* counter_words) {  // (round 6: the light class, counted in word 1)
  unsigned long long n = 0;
  for (size_t q = 0; q < HSK_NQUEUES; ++q) n += counter_words[q * HSK_QCOUNT_STRIDE + 1];
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
  const int4* wvz = (const int4*)(wgz + (((size_t)col_blocks + 1) & ~(size_t)1));  // ... one per wave footprint ...
  unsigned char* cflag = (unsigned char*)(wvz + (size_t)col_blocks * 4);           // ... and the frame's verdict per wave-chunk
  const IntegrateConst kc = integrate_const(vp, W, H, in);
  const float2* sparse = (const float2*)(tmax + 4 * tw * th) + (size_t)50 * fw * fh;
  const unsigned* vmask = (const unsigned*)(tmax + hsk_tiles_mask_offset(W, H));  // validity of every pixel (k_bilateral_scale / k_tile_max)
  const CoarseArgs ca = {cflag, (count_only || uni == nullptr) ? nullptr : uni + hsk_lane_sum_bytes(vp), sparse, (unsigned)hsk_chunk_bytes_padded(vp), zchunk, kc.free_thr2, kc.cull_thr2};
  const RingOut quiet_ring = {nullptr, nullptr, nullptr};
  const RingOut early_ring = (early && fin.slots && !count_only) ? *early : quiet_ring;
  // (one block more when the launch reports the pose: that block solves, reports and leaves)
  hipLaunchKernelGGL(k_column_zrange, dim3((col_blocks > dil_blocks ? col_blocks : dil_blocks) + (early_ring.slots ? 1 : 0)), dim3(256), 0, s, fin, tmax,
                     tmax + tw * th, (float2*)(tmax + 2 * tw * th), tw, th, dil_blocks, qcount, st, vp, W, H, in, zint,
                     const_cast<TrackState*>(st), wgz, early_ring, ca);
  dim3 block(64, 4, 1);
  const dim3 grid((vp.X + 63) / 64, (vp.Y + 15) / 16, zchunks);
  const float2* dil = (const float2*)(tmax + 2 * tw * th);
  // behind the counters: HSK_NQUEUES queues of qcap entries each; a block of pass A holds at most 4 waves x 64 lanes
  // x (zchunk / 4) blocks and every HSK_NQUEUES-th block shares a queue
  unsigned* qdata = queue + HSK_NQUEUES * HSK_QCOUNT_STRIDE;
  const unsigned nblk = grid.x * grid.y * (unsigned)zchunks;
  const unsigned qcap = ((nblk + HSK_NQUEUES - 1) / HSK_NQUEUES) * 256u * (unsigned)((zchunk + 3) / 4);
  // n % grid.x = n - mulhi(n, gmagic) * grid.x for n < 2^16; 0 for a single x block (the kernel takes 0 for the remainder)
  const unsigned gmagic = grid.x > 1u ? (unsigned)(0x100000000ull / grid.x) + 1u : 0u;
  // (pass B takes a queued lane-block id apart by multiplications: floor(2^40 / d) + 1 for d = X / 4 and d = Y)
  const unsigned long long qmag_x = (1ull << 40) / (unsigned long long)(vp.X / 4) + 1ull, qmag_y = (1ull << 40) / (unsigned long long)vp.Y + 1ull;
  const dim3 detail_grid(DETAIL3_GX * HSK_NQUEUES);  // one resident round of the chip, striding over the concatenated queues
  if (count_only) {
    if (vp.zchunk == 16)
      hipLaunchKernelGGL((k_integrate<true, 4>), grid, block, 0, s, st, (const unsigned*)cflag, wvz, zchunk, grid.x, gmagic, (double*)nullptr, (unsigned char*)nullptr, dil,
                         W, H, tw, th, grid.y, (uint4*)vol, scaled, vp, in, counter, flags, qdata, qcount, qcap, ftab, fw, fh, qtab, kc, (const int2*)zint, vmask, hsk_mask_pitch32(W));
    else
      hipLaunchKernelGGL((k_integrate<true, 2>), grid, block, 0, s, st, (const unsigned*)cflag, wvz, zchunk, grid.x, gmagic, (double*)nullptr, (unsigned char*)nullptr, dil,
                         W, H, tw, th, grid.y, (uint4*)vol, scaled, vp, in, counter, flags, qdata, qcount, qcap, ftab, fw, fh, qtab, kc, (const int2*)zint, vmask, hsk_mask_pitch32(W));
    hipLaunchKernelGGL(k_integrate_detail3<true>, detail_grid, dim3(256), 0, s, st, qcount, qdata, qcap, W, H, (uint4*)vol, scaled, in, vp,
                       counter, flags, qmag_x, qmag_y, (unsigned char*)nullptr);
  } else {
    if (vp.zchunk == 16)
      hipLaunchKernelGGL((k_integrate<false, 4>), grid, block, 0, s, st, (const unsigned*)cflag, wvz, zchunk, grid.x, gmagic, fin.slots, uni, dil, W, H, tw, th, grid.y,
                         (uint4*)vol, scaled, vp, in, counter, flags, qdata, qcount, qcap, ftab, fw, fh, qtab, kc, (const int2*)zint, vmask, hsk_mask_pitch32(W));
    else
      hipLaunchKernelGGL((k_integrate<false, 2>), grid, block, 0, s, st, (const unsigned*)cflag, wvz, zchunk, grid.x, gmagic, fin.slots, uni, dil, W, H, tw, th, grid.y,
                         (uint4*)vol, scaled, vp, in, counter, flags, qdata, qcount, qcap, ftab, fw, fh, qtab, kc, (const int2*)zint, vmask, hsk_mask_pitch32(W));
    hipLaunchKernelGGL(k_integrate_detail3<false>, detail_grid, dim3(256), 0, s, st, qcount, qdata, qcap, W, H, (uint4*)vol, scaled, in, vp,
                       counter, flags, qmag_x, qmag_y, uni);
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

// Summaries of a volume that was uploaded rather than integrated (both levels), and the reverse: the volume's weights
// brought up to date.  One WAVE per wave-chunk, lane = the pass-A lane that owns the blocks (a block that reaches beyond
// the last stored plane has no summary): the chunk's byte is read by every lane before lane 0 rewrites it.
size_t uniform_bytes(const VolParams& vp) { return hsk_lane_sum_bytes(vp) + 2 * hsk_chunk_bytes_padded(vp); }
size_t uniform_lane_bytes(const VolParams& vp) { return hsk_lane_sum_bytes(vp); }
template <bool MATERIALIZE, int NS>
__global__ __launch_bounds__(256) void k_summaries(uint4* __restrict__ vol, VolParams vp, unsigned char* __restrict__ uni) {
  const size_t cw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // wave-chunk
  if (cw >= hsk_chunk_count(vp)) return;
  const int lane = threadIdx.x & 63, w = (int)(cw & 3);
  const int tiles_x = (vp.X + 63) / 64, tiles_y = (vp.Y + 15) / 16;
  const size_t wg = cw >> 2;
  const int fbx = (int)(wg % tiles_x), fby = (int)((wg / tiles_x) % tiles_y), zc = (int)(wg / ((size_t)tiles_x * tiles_y));
  const int x0 = (fbx * 16 + w * 4 + (lane & 3)) * 4, y = fby * 16 + (lane >> 2);
  const bool active = x0 < vp.X && y < vp.Y;
  unsigned char* const cs = uni + hsk_lane_sum_bytes(vp);
  unsigned char* const dirty = cs + hsk_chunk_bytes_padded(vp);
  unsigned char* const sum_at = uni + (cw * 64 + (size_t)lane) * NS;
  if (MATERIALIZE && dirty[cw] == 0) return;  // nothing of this chunk is ahead of the volume (wave-uniform)
  if (!MATERIALIZE) {
    bool quiet = active;
#pragma unroll
    for (int g = 0; g < NS; ++g) {
      const int zb = zc * vp.zchunk + 4 * g;
      unsigned code = 0u;
      if (active && zb + 3 < vp.nzs) {
        uint4 q[4];
        for (int u = 0; u < 4; ++u) q[u] = vol[VIDX(zb, u)];
        code = hsk_sum_classify(q);
      }
      sum_at[g] = (unsigned char)code;
      quiet = quiet && code >= 2u;
    }
    const bool all_quiet = __ballot(quiet) == ~0ull;
    if (lane == 0) {
      cs[cw] = (zc * vp.zchunk + vp.zchunk <= vp.nzs && all_quiet) ? 1 : 0;
      dirty[cw] = 0;  // (the summaries were just read off the volume)
    }
    return;
  }
  const unsigned c = cs[cw];
  const unsigned kpend = c >= 2u ? c - 1u : 0u;
#pragma unroll
  for (int g = 0; g < NS; ++g) {
    const int zb = zc * vp.zchunk + 4 * g;
    if (!active || zb >= vp.nzs) continue;
    const unsigned s0 = sum_at[g], sm = hsk_sum_push(s0, kpend);
    if (sm >= 2u && sm < HSK_SUM_RAGGED) {
      const unsigned word = ((sm - 1u) << 16) | (unsigned)HSK_DIVISOR;
      for (int u = 0; u < 4; ++u) vol[VIDX(zb, u)] = make_uint4(word, word, word, word);
      if (sm != s0) sum_at[g] = (unsigned char)sm;
    } else if (sm > HSK_SUM_RAGGED) {
      uint4 q[4];
      for (int u = 0; u < 4; ++u) q[u] = vol[VIDX(zb, u)];
      for (int u = 0; u < 4; ++u) {
        hsk_vector_add_weight(q[u], sm - HSK_SUM_RAGGED);
        vol[VIDX(zb, u)] = q[u];
      }
      sum_at[g] = (unsigned char)hsk_sum_classify(q);
    }
  }
  // (the blocks stay quiet: a uniform block stays uniform, a rim block becomes 130 or 129)
  if (lane == 0) {
    if (kpend != 0u) cs[cw] = 1;
    dirty[cw] = 0;
  }
}
void launch_rebuild_uniform(hipStream_t s, const void* vol, const VolParams& vp, unsigned char* uni) {
  const unsigned nb = (unsigned)((hsk_chunk_count(vp) + 3) / 4);
  if (vp.zchunk == 16)
    hipLaunchKernelGGL((k_summaries<false, 4>), dim3(nb), dim3(256), 0, s, (uint4*)const_cast<void*>(vol), vp, uni);
  else
    hipLaunchKernelGGL((k_summaries<false, 2>), dim3(nb), dim3(256), 0, s, (uint4*)const_cast<void*>(vol), vp, uni);
}
void launch_materialize(hipStream_t s, void* vol, const VolParams& vp, unsigned char* uni) {
  const unsigned nb = (unsigned)((hsk_chunk_count(vp) + 3) / 4);
  if (vp.zchunk == 16)
    hipLaunchKernelGGL((k_summaries<true, 4>), dim3(nb), dim3(256), 0, s, (uint4*)vol, vp, uni);
  else
    hipLaunchKernelGGL((k_summaries<true, 2>), dim3(nb), dim3(256), 0, s, (uint4*)vol, vp, uni);
}
