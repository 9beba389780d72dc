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
};

struct VolParams {
  int X, Y, Z;     // full volume
  int zs0, nzs;    // stored planes [zs0, zs0+nzs)
  int zo0, zo1;    // owned planes (raycast step ownership)
  float cell[3];
  float size[3];
  float tau, tau_inv;
  int bshift;      // log2 of the brick edge of the "has held a negative TSDF" bitfield (3 => 8^3 voxels)
};

// words of the brick bitfield; the brick edge is chosen so that it fits 32 KiB of LDS
static inline int hsk_flag_words(const VolParams& vp) {
  const long bits = (long)(vp.X >> vp.bshift) * (vp.Y >> vp.bshift) * ((vp.nzs + (1 << vp.bshift) - 1) >> vp.bshift);
  return (int)(((bits + 31) / 32 + 3) / 4 * 4);  // multiple of 4 words: staged into LDS with 16-B loads
}

// marching-tetrahedra lookup: per Kuhn tetrahedron and 4-bit inside mask, 0..2 triangles; each triangle corner is an
// edge of the cube coded (low corner) | (high corner << 4)
struct TetTable {
  unsigned char ntri[6][16];
  unsigned char edge[6][16][2][3];
};

#define HSK_NANF (__builtin_nanf(""))

static __device__ __forceinline__ bool hsk_isnan(float x) { return x != x; }

// round-to-nearest-even with the +-1e6 range guard of the spec
static __device__ __forceinline__ bool hsk_rint_guard(float f, int& out) {
  if (!(f > -1.0e6f && f < 1.0e6f)) return false;
  out = __float2int_rn(f);
  return true;
}

static __device__ __forceinline__ float hsk_dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return (ax * bx + ay * by) + az * bz;
}

// launchers implemented in the kernel translation units
struct ImgLevel {
  int W, H;
  Intr in;
};
