// raycast.hip -- TSDF raycast for gfx950: the synthetic model frame, its step keys and levels 1 and 2 of
// the model maps.  (Split out of kernels_volume.hip in round 5; no behaviour change.)
#pragma clang fp contract(off)
#include "hsk_dev.h"
#include "hsk_launch.h"

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
// A wave's pixel tile is TW x (64 / TW), a template parameter of the kernel (round 5; profiles/r05/raycast_notes.md): 8 x 8,
// or 16 x 4 for volumes far beyond the Infinity Cache -- x-adjacent rays gather x-adjacent voxels, four of which share a
// 64-B block: 1024^3 71.6 -> 66.3 us, 512^3 56.8 -> 57.1 (kept at 8 x 8); 4 x 16: 60.8 / 84.6 us.
#define RC_TH (64 / RC_TW)
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
template <bool SLAB, int RC_TW>
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
  const int tiles_x = (W + RC_TW - 1) / RC_TW, tiles_y = (H + RC_TH - 1) / RC_TH;
  // Tile rows are dispatched from the top and bottom edges of the image inwards (0, last, 1, last - 1, ...): the rays of
  // the border rows meet floor and ceiling at grazing angles and march longest, and a wave dispatched last onto a SIMD
  // that already holds its share of waves finishes last -- with the rows in image order the launch ended with exactly
  // those tiles (tools/rc_timing.sh).  Scheduling only.  Measured 512^3 / 1024^3: 90.6 / 117.9 -> 86.8 / 110.8 us.
  const int ty_lin = tile / tiles_x;
  const int ty = (ty_lin & 1) ? (tiles_y - 1 - (ty_lin >> 1)) : (ty_lin >> 1);
  const int x = (tile % tiles_x) * RC_TW + (lane % RC_TW);
  const int y = ty * RC_TH + (lane / RC_TW);
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
    pyramid_step(m, 1, RC_TW, l1);
    pyramid_step(l1, 2, 2 * RC_TW, l2);
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
  // (the wide tile wants whole tiles: the fused pyramid's shuffles assume them)
  const int tw_px = (vp.stream_nt && (W % 16) == 0 && (H % 4) == 0) ? 16 : 8;
  const int tiles = ((W + tw_px - 1) / tw_px) * ((H + 64 / tw_px - 1) / (64 / tw_px));
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
  const size_t lds = (size_t)(words + HSK_SUPER_WORDS) * 4;
  if (slab && tw_px == 16)
    hipLaunchKernelGGL((k_raycast<true, 16>), grid, block, lds, s, a);
  else if (slab)
    hipLaunchKernelGGL((k_raycast<true, 8>), grid, block, lds, s, a);
  else if (tw_px == 16)
    hipLaunchKernelGGL((k_raycast<false, 16>), grid, block, lds, s, a);
  else
    hipLaunchKernelGGL((k_raycast<false, 8>), grid, block, lds, s, a);
}
// the fused pyramid needs complete 8x8 tiles and a single-device volume
bool raycast_can_fuse_pyramid(const VolParams& vp, int W, int H) {
  const bool slab = vp.zs0 != 0 || vp.nzs != vp.Z || vp.zo0 != 0 || vp.zo1 != vp.Z;
  return !slab && (W % 8) == 0 && (H % 8) == 0;
}
