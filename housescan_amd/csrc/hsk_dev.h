// hsk_dev.h -- shared device/host structures of the KinectFusion core (gfx950 only).
//
// Numerical contract: IEEE binary32/binary64, one rounding per written operator, NO fused multiply-add
// (the translation units are built with -ffp-contract=off), correctly rounded '/' and sqrtf
// (-fhip-fp32-correctly-rounded-divide-sqrt).  Expression trees are parenthesised deliberately; do not
// "simplify" them -- bit-parity with the CPU oracle depends on the association order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HSK_DIVISOR 32767
#define HSK_MAX_WEIGHT 128
#define HSK_NLEVELS 3
#define HSK_KEY_NONE_I 0x7fffffff

struct Intr {
  float fx, fy, cx, cy;
};

// Device-resident tracker state.  Kernels read the pose from here (never from kernel arguments) so that a
// whole frame can be replayed as one hipGraph with no host round trip.
struct TrackState {
  float R[9], t[3];    // pose estimate, cam->world; refined in place by the ICP iterations
  float Rp[9], tp[3];  // pose of the previous frame (the model maps were raycast from it)
  int lost;            // set by the solve when the 6x6 system is singular (A.5)
  int frame;
  int n_iter;
  int need_reset;      // sticky after a tracking loss: later frames already in flight are dropped until the host resets
  double sums[27];     // last reduced normal equations (debug / hsk_icp_accumulate)
  unsigned ring_mark;  // in a host ring slot: sequence number of the frame that wrote it, stored last (RingOut): the frame's inputs are consumed
  unsigned pose_mark;  // in a host ring slot: the same number, stored as soon as the frame's ICP has ended (its pose and verdict are final)
};

struct VolParams {
  int X, Y, Z;     // full volume
  int zs0, nzs;    // stored planes [zs0, zs0+nzs)
  int zo0, zo1;    // owned planes (raycast step ownership)
  float cell[3];
  float size[3];
  float tau, tau_inv;
  int bshift;      // log2 of the brick edge of the "has held a negative TSDF" bitfield (3 => 8^3 voxels)
  double icell[3]; // correctly rounded binary64 reciprocals of cell[] (hsk_div_by_cell)
  int stream_nt;   // free-space updates stream the volume with non-temporal accesses (volumes >> Infinity Cache)
  int zchunk;      // planes a pass-A workgroup takes: 8, or 16 for volumes whose launch stays large with half the workgroups
                   // (hsk_pass_a_zchunk); also the unit of the lane-block summaries' layout (integrate.hip)
};

// ---- where a voxel lives (round 4) ------------------------------------------------------------------------------------
// The volume is stored in 64-B BLOCKS of one lane-block each -- 4 x-adjacent voxels by 4 consecutive stored planes, the
// unit the integrate's two passes classify and rewrite: word index of voxel (x, y, stored plane zz)
//     ((((zz >> 2) * Y + y) * (X / 4) + (x >> 2)) * 16) + (zz & 3) * 4 + (x & 3)
// so a queued lane-block is ONE sector where the row-major layout had four, 1 MiB apart; x-adjacent lane-blocks stay
// contiguous (a pass-A wave row: 16 lanes x 64 B = 1 KiB).  The index is a sum of one term per axis, so the raycast's
// 2 x 2 x 2 taps cost two terms per axis and eight additions.  Stored planes are padded to a multiple of 4 (the padding is
// never observed: weight 0).  Host arrays (hsk_download_tsdf / hsk_upload_tsdf) stay row-major, x fastest: the
// conversion kernels are k_vol_to_linear / k_vol_from_linear.
__host__ __device__ static inline size_t hsk_vox_xterm(int x) { return ((size_t)(x >> 2) << 4) + (size_t)(x & 3); }
__host__ __device__ static inline size_t hsk_vox_yterm(const VolParams& vp, int y) { return (size_t)y * ((size_t)(vp.X >> 2) << 4); }
__host__ __device__ static inline size_t hsk_vox_zterm(const VolParams& vp, int zz) {
  return (size_t)(zz >> 2) * (size_t)vp.Y * ((size_t)(vp.X >> 2) << 4) + ((size_t)(zz & 3) << 2);
}
__host__ __device__ static inline size_t hsk_vox_index(const VolParams& vp, int x, int y, int zz) {
  return hsk_vox_zterm(vp, zz) + hsk_vox_yterm(vp, y) + hsk_vox_xterm(x);
}
__host__ __device__ static inline size_t hsk_vol_words(const VolParams& vp) {  // allocation, in voxels (4 B each)
  return (size_t)vp.X * vp.Y * (size_t)((vp.nzs + 3) & ~3);
}

// Pass A's chunk of planes per workgroup.  Sixteen planes halve the waves and what each of them spends before its
// first useful instruction (pose, footprint tables, column terms, queue ticket): -13 us at 1024^3 (65 k workgroups left),
// +2.6 us at 512^3 (8 k left: the launch's tail grows) -- so the size decides (profiles/r04/integrate_notes.md).
#ifndef HSK_ZCHUNK16_MIN_WGS
#define HSK_ZCHUNK16_MIN_WGS 16384
#endif
__host__ static inline int hsk_pass_a_zchunk(int X, int Y, int nzs) {
  const long wgs16 = (long)((X + 63) / 64) * ((Y + 15) / 16) * ((nzs + 15) / 16);
  return wgs16 >= HSK_ZCHUNK16_MIN_WGS ? 16 : 8;
}

#ifndef HSK_FLAG_WORDS_MAX
#define HSK_FLAG_WORDS_MAX 1024  // 4 KiB
#endif
// words of the brick bitfield; the brick edge is chosen (hsk_create) so that it fits HSK_FLAG_WORDS_MAX
__host__ __device__ static inline int hsk_flag_words(const VolParams& vp) {
  const long bits = (long)(vp.X >> vp.bshift) * (vp.Y >> vp.bshift) * ((vp.nzs + (1 << vp.bshift) - 1) >> vp.bshift);
  return (int)(((bits + 31) / 32 + 3) / 4 * 4);  // multiple of 4 words: staged into LDS with 16-B loads
}
// Behind the brick bitfield: one bit per SUPER-brick of (2^HSK_SUPER_SHIFT)^3 bricks ("one of my bricks has held a negative TSDF"),
// the raycast's licence to cross such a block without looking at its steps.  Always HSK_SUPER_WORDS words (a multiple
// of 4: staged with the same 16-B loads); a volume with more super-bricks than bits does not use it (hsk_super_ok).
#ifndef HSK_SUPER_SHIFT
#define HSK_SUPER_SHIFT 2  // log2 of the super-brick edge in bricks
#endif
#define HSK_SUPER_WORDS (HSK_SUPER_SHIFT == 2 ? 32 : 128)
__host__ __device__ static inline int hsk_super_dim(int voxels, int bshift) {
  return ((voxels >> bshift) + (1 << HSK_SUPER_SHIFT) - 1) >> HSK_SUPER_SHIFT;
}
__host__ __device__ static inline bool hsk_super_ok(const VolParams& vp) {
  const int zb = (vp.nzs + (1 << vp.bshift) - 1) >> vp.bshift;
  return (long)hsk_super_dim(vp.X, vp.bshift) * hsk_super_dim(vp.Y, vp.bshift) * ((zb + (1 << HSK_SUPER_SHIFT) - 1) >> HSK_SUPER_SHIFT) <=
         HSK_SUPER_WORDS * 32;
}
__host__ __device__ static inline int hsk_flag_words_total(const VolParams& vp) { return hsk_flag_words(vp) + HSK_SUPER_WORDS; }

// marching-tetrahedra lookup: per Kuhn tetrahedron and 4-bit inside mask, 0..2 triangles; each triangle corner is an
// edge of the cube coded (low corner) | (high corner << 4)
struct TetTable {
  unsigned char ntri[6][16];
  unsigned char edge[6][16][2][3];
};

// marching-cubes lookup (hsk_build_cube_table): per 8-bit inside mask 0..5 triangles; triangle corners coded like TetTable's
#define HSK_MC_MAXT 5
struct CubeTable {
  unsigned char ntri[256];
  unsigned char edge[256][HSK_MC_MAXT][3];
};

// levels 1 and 2 of the model maps, written by the raycast itself when it can (launch_raycast)
struct MapPyramid {
  float *v1, *n1, *v2, *n2;
};

// Pose read-back without a copy node: the last kernel of a pipelined frame writes the tracker state straight into a
// pinned host ring slot.  `slot_fifo` (pinned host memory) holds the slot the host assigned to each submitted frame,
// `seq` (device memory) counts the frames that have written; all null when the launch does not report.
#define HSK_RING_FIFO 8
struct RingOut {
  TrackState* slots;
  const int* slot_fifo;
  unsigned* seq;
};

// destinations of a direct (peer-write) exchange: one buffer per device of the group
#define HSK_PUSH_MAX 16
struct PushDests {
  int* p[HSK_PUSH_MAX];
  int n;
};

#define HSK_NANF (__builtin_nanf(""))

static __device__ __forceinline__ bool hsk_isnan(float x) { return x != x; }

// round-to-nearest-even with the +-1e6 range guard of the spec
static __device__ __forceinline__ bool hsk_rint_guard(float f, int& out) {
  if (!(f > -1.0e6f && f < 1.0e6f)) return false;
  out = __float2int_rn(f);
  return true;
}

// (float)raw / 32767.0f of the specification, for an integer raw in [-32768, 32767], without the ~10-instruction
// correctly-rounded f32 division: the product with the binary64 reciprocal, rounded to binary32, equals the binary32
// quotient for EVERY such raw (checked exhaustively in tests/test_host_logic.py).
static __device__ __forceinline__ float hsk_tsdf_unpack(int raw) {
  return (float)((double)raw * (1.0 / 32767.0));
}

// x / c of the specification for a fixed binary32 divisor c, as a binary64 product with the correctly rounded binary64
// reciprocal rc: a binary32 quotient of two binary32 numbers is either exact or at least 2^-48 (relative) away from a
// rounding boundary (ties need c to be a power of two, where rc is exact), and the product is within 2^-52 of it.
static __device__ __forceinline__ float hsk_div_by_const(float x, double rc) { return (float)((double)x * rc); }

// ---- correctly rounded 1/x, sqrt(x) and a/n in a handful of instructions -------------------------------------------
// The specification asks for the IEEE-754 correctly rounded results; the compiler's sequences for them cost 10-14
// instructions.  On gfx950 the one-ulp hardware approximations corrected by ONE fused multiply-add step give the same
// bits -- not by a theorem but by exhaustive comparison on the hardware itself: every binary32 value for 1/x and
// sqrt(x), every (a, n) pair of the domain for a/n (hsk_selftest_exact_ops; tests/test_gpu_exact_ops.py runs it on
// the GPU under test, tools/exact/rcp_sqrt_check.hip is the stand-alone form).  The FMAs below are explicit builtins:
// the translation units are compiled with -ffp-contract=off, nothing else is ever fused.
//   hsk_rcp_exact : exact for every normal x whose reciprocal is normal              (2^-126 <= |x| <= 2^126)
//   hsk_sqrt_exact: exact for every x >= 2^-102 (below that the residual underflows)
//   hsk_div_small_exact: exact for 2^-100 <= |a| < 512 or a = +0, and an integer 1 <= n <= 129 held in a float
//                        (a = -0 gives +0: immaterial, the quotient is only ever converted to an integer)
static __device__ __forceinline__ float hsk_rcp_exact(float x) {
  const float r0 = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r0, 1.0f);
  return __builtin_fmaf(r0, e, r0);
}
static __device__ __forceinline__ float hsk_sqrt_exact(float x) {
  const float s0 = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(x);
  const float d = __builtin_fmaf(-s0, s0, x);  // exact: s0 is within one ulp of the root
  return __builtin_fmaf(d, h, s0);
}
static __device__ __forceinline__ float hsk_div_small_exact(float a, float n) {
  const float y = hsk_rcp_exact(n);
  const float q0 = a * y;
  const float r = __builtin_fmaf(-q0, n, a);
  return __builtin_fmaf(r, y, q0);
}

// ---- buffer loads (gfx950): address = descriptor base (scalar) + byte offset (one 32-bit register) + scalar offset ------
// For planes of one array read at the same pixel: the pixel's offset is computed once, the plane's offset is scalar, and no
// 64-bit address is formed per load (a global load takes a 64-bit address register pair: two instructions per plane).
// Raw descriptor, no stride, no bound to rely on (the callers clamp their indices as they did for global loads).
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t hsk_buf(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);  // (32-bit data format: gfx90a / gfx94x / gfx950)
}
static __device__ __forceinline__ float hsk_buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned scalar_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)scalar_off, 0));
}

static __device__ __forceinline__ float hsk_dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return (ax * bx + ay * by) + az * bz;
}

// ---- the tile tables of a frame (one allocation per image-buffer set: launch_tile_max / launch_bilateral_scale fill the raw
// tables, launch_tile_tables the derived ones); offsets in floats.  n16 / n8: tiles of 16 / 8 pixels.
//   [0, n16) raw 16-px max, [n16, 2 n16) raw 16-px min (0 when a pixel of the tile has no depth), [2 n16, 4 n16) the dilated table
//   then, as float2: the 8-px table (n8), the 4-px table (4 n8), their nine window shapes each (45 n8), the sparse 16-px table
//   (16 n16) -- in the 8-px / 4-px tables and their windows .y is the minimum over the pixels WITH depth, negated when a
//   pixel of the tile (window) has none
//   then the validity mask (round 5): one bit per pixel, 1 = no depth (or outside the image), rows of hsk_mask_pitch32 words
__host__ __device__ static inline size_t hsk_tiles_n16(int W, int H) { return (size_t)((W + 15) / 16) * ((H + 15) / 16); }
__host__ __device__ static inline size_t hsk_tiles_n8(int W, int H) { return (size_t)((W + 7) / 8) * ((H + 7) / 8); }
__host__ __device__ static inline int hsk_mask_pitch32(int W) { return (W + 31) / 32 + 1; }  // (+1: a box's two words are always there)
__host__ __device__ static inline size_t hsk_tiles_mask_offset(int W, int H) { return 4 * hsk_tiles_n16(W, H) + 2 * 50 * hsk_tiles_n8(W, H) + 2 * 16 * hsk_tiles_n16(W, H); }
__host__ __device__ static inline size_t hsk_tiles_floats(int W, int H) { return hsk_tiles_mask_offset(W, H) + (size_t)hsk_mask_pitch32(W) * (size_t)(((H + 15) / 16) * 16); }

// launchers implemented in the kernel translation units
struct ImgLevel {
  int W, H;
  Intr in;
};
