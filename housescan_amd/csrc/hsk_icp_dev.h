// hsk_icp_dev.h -- device pieces of the fused ICP that more than one translation unit needs: the 6x6 solve and pose
// update (also the host mirror), the sharded-sum read-back and the solve step of one wave.  kernels_image.hip owns the
// iterations; integrate.hip runs the LAST solve of a frame in the prologue of its first integrate kernel.
#pragma once
#include "hsk_dev.h"

#ifndef ICP_STAMP
#define ICP_STAMP(k) do { } while (0)  // the timing build of kernels_image.hip defines it before this header
#endif

// gfx950 lane swaps of a 64-bit pair: v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31
// of its second, v_permlane16_swap the odd 16-lane rows of the first with the even rows of the second -- one VALU
// instruction per 32-bit half moves BOTH directions of a butterfly step (no selects, no trip through the LDS crossbar).
static __device__ __forceinline__ double swap32_add_f64(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// ---- 6x6 solve + pose update, shared by the device kernel and the host mirror (hsk_icp_solve) ----
__host__ __device__ static inline void hsk_sincos(double x_in, double* s, double* c) {
  // (no early return and no branch on the quadrant: selects -- see hsk_solve6)
  const bool in_range = fabs(x_in) < 1.0e5;  // otherwise (also NaN): sine 0, cosine 1
  const double x = in_range ? x_in : 0.0;
  const double two_over_pi = 0.63661977236758134308;
  const double pio2_hi = 1.57079632673412561417e+00;
  const double pio2_lo = 6.07710050650619224932e-11;
  const double kf = rint(x * two_over_pi);
  const double r = (x - kf * pio2_hi) - kf * pio2_lo;
  const double r2 = r * r;
  double S = -1.0 / 1307674368000.0;
  S = S * r2 + 1.0 / 6227020800.0;
  S = S * r2 - 1.0 / 39916800.0;
  S = S * r2 + 1.0 / 362880.0;
  S = S * r2 - 1.0 / 5040.0;
  S = S * r2 + 1.0 / 120.0;
  S = S * r2 - 1.0 / 6.0;
  const double sr = r + (r * r2) * S;
  double C = 1.0 / 20922789888000.0;
  C = C * r2 - 1.0 / 87178291200.0;
  C = C * r2 + 1.0 / 479001600.0;
  C = C * r2 - 1.0 / 3628800.0;
  C = C * r2 + 1.0 / 40320.0;
  C = C * r2 - 1.0 / 720.0;
  C = C * r2 + 1.0 / 24.0;
  const double cr = (1.0 - 0.5 * r2) + (r2 * r2) * C;
  const int q = ((int)kf) & 3;
  // q: 0 -> (sr, cr), 1 -> (cr, -sr), 2 -> (-sr, -cr), 3 -> (-cr, sr)
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  const double s1 = (q & 2) ? -s0 : s0, c1 = ((q + 1) & 2) ? -c0 : c0;
  *s = in_range ? s1 : 0.0;
  *c = in_range ? c1 : 1.0;
}

__host__ __device__ static inline bool hsk_solve6(const double* in27, float* x6) {
  double A[6][6], b[6], L[6][6], D[6];
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      const double v = in27[k++];
      if (j == 6)
        b[i] = v;
      else {
        A[i][j] = v;
        A[j][i] = v;
      }
    }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) L[i][j] = 0.0;
  // LDL^T (unit lower L, diagonal D) with one reciprocal per pivot: no square roots on the dependent chain
  // (the failure tests are gathered and taken at the end: on the device this runs on one lane of a wave that has its SIMD
  // to itself, where every early return is a compare, a lane-mask update and a branch on the dependent chain; after a
  // failed test the arithmetic goes on with whatever it has and its results are not used)
  double det = 1.0;
  double dinv[6];
  bool good = true;
  for (int j = 0; j < 6; ++j) {
    double dj = A[j][j];
    for (int q = 0; q < j; ++q) dj = dj - (L[j][q] * L[j][q]) * D[q];
    good = good & (dj > 0.0);
    D[j] = dj;
    dinv[j] = 1.0 / dj;
    det = det * dj;
    for (int i = j + 1; i < 6; ++i) {
      double r = A[i][j];
      for (int q = 0; q < j; ++q) r = r - (L[i][q] * L[j][q]) * D[q];
      L[i][j] = r * dinv[j];
    }
  }
  good = good & (det >= 1e-15);
  double yv[6], xv[6];
  for (int i = 0; i < 6; ++i) {  // L y = b
    double r = b[i];
    for (int q = 0; q < i; ++q) r = r - L[i][q] * yv[q];
    yv[i] = r;
  }
  for (int i = 5; i >= 0; --i) {  // L^T x = D^-1 y
    double r = yv[i] * dinv[i];
    for (int q = i + 1; q < 6; ++q) r = r - L[q][i] * xv[q];
    xv[i] = r;
  }
  for (int q = 0; q < 6; ++q) good = good & (xv[q] == xv[q]) & (fabs(xv[q]) < 1e30);
  if (good)  // (x6 is left alone on failure, as the callers expect)
    for (int q = 0; q < 6; ++q) x6[q] = (float)xv[q];
  return good;
}

__host__ __device__ static inline void hsk_mat3mul(const float* A, const float* B, float* O) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O[i * 3 + j] = (A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j];
}

// pose refinement from the solved increment, given the sines and cosines of its three angles
__host__ __device__ static inline void hsk_pose_update_sc(float* R, float* t, const float* x6, float sa, float ca, float sb,
                                                          float cb, float sg, float cg) {
  const float Rx[9] = {1.0f, 0.0f, 0.0f, 0.0f, ca, -sa, 0.0f, sa, ca};
  const float Ry[9] = {cb, 0.0f, sb, 0.0f, 1.0f, 0.0f, -sb, 0.0f, cb};
  const float Rz[9] = {cg, -sg, 0.0f, sg, cg, 0.0f, 0.0f, 0.0f, 1.0f};
  float Rzy[9], Rinc[9], Rn[9];
  hsk_mat3mul(Rz, Ry, Rzy);
  hsk_mat3mul(Rzy, Rx, Rinc);
  const float n0 = ((Rinc[0] * t[0] + Rinc[1] * t[1]) + Rinc[2] * t[2]) + x6[3];
  const float n1 = ((Rinc[3] * t[0] + Rinc[4] * t[1]) + Rinc[5] * t[2]) + x6[4];
  const float n2 = ((Rinc[6] * t[0] + Rinc[7] * t[1]) + Rinc[8] * t[2]) + x6[5];
  t[0] = n0;
  t[1] = n1;
  t[2] = n2;
  hsk_mat3mul(Rinc, R, Rn);
  for (int i = 0; i < 9; ++i) R[i] = Rn[i];
}

__host__ __device__ static inline void hsk_pose_update(float* R, float* t, const float* x6) {
  double sd, cd;
  hsk_sincos((double)x6[0], &sd, &cd);
  const float sa = (float)sd, ca = (float)cd;
  hsk_sincos((double)x6[1], &sd, &cd);
  const float sb = (float)sd, cb = (float)cd;
  hsk_sincos((double)x6[2], &sd, &cd);
  const float sg = (float)sd, cg = (float)cd;
  hsk_pose_update_sc(R, t, x6, sa, ca, sb, cb, sg, cg);
}


struct IcpPose {
  float R[9], t[3];
  int lost, n_iter, pad[2];
};



// lane Q of every quad to the quad's four lanes (DPP quad_perm [Q, Q, Q, Q]: a plain vector move, no scalar register)
template <int Q>
static __device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), Q * 0x55, 0xf, 0xf, true));
}
// Row `i` of hsk_pose_update_sc: the same expressions in the same order, element by element (the three 3x3 products
// are row-separable), so three lanes produce the rows of the new rotation and translation with the bits one lane would.
static __device__ __forceinline__ void hsk_pose_update_row(int i, const float* R, const float* t, const float* x6, float sa,
                                                           float ca, float sb, float cb, float sg, float cg, float* rn,
                                                           float* tn) {
  const float z0 = i == 0 ? cg : (i == 1 ? sg : 0.0f), z1 = i == 0 ? -sg : (i == 1 ? cg : 0.0f), z2 = i == 2 ? 1.0f : 0.0f;
  const float Ry[9] = {cb, 0.0f, sb, 0.0f, 1.0f, 0.0f, -sb, 0.0f, cb};
  const float Rx[9] = {1.0f, 0.0f, 0.0f, 0.0f, ca, -sa, 0.0f, sa, ca};
  float zy[3], inc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) zy[j] = (z0 * Ry[j] + z1 * Ry[3 + j]) + z2 * Ry[6 + j];
#pragma unroll
  for (int j = 0; j < 3; ++j) inc[j] = (zy[0] * Rx[j] + zy[1] * Rx[3 + j]) + zy[2] * Rx[6 + j];
  const float xi = i == 0 ? x6[3] : (i == 1 ? x6[4] : x6[5]);
  *tn = ((inc[0] * t[0] + inc[1] * t[1]) + inc[2] * t[2]) + xi;
#pragma unroll
  for (int j = 0; j < 3; ++j) rn[j] = (inc[0] * R[j] + inc[1] * R[3 + j]) + inc[2] * R[6 + j];
}

// Executed by the whole first wave: every lane solves; the lanes of a quad evaluate one sine/cosine pair each (the three
// polynomial evaluations are the longest serial piece after the factorisation) and then one row each of the pose update
// (three 3x3 products on one lane were 150 dependent instructions); the rows travel inside the quad, so every lane of the
// wave leaves with the whole new pose.
static __device__ __forceinline__ void icp_solve_step(const double* tot, IcpPose& p, int g_icp_iter = 99) {
  (void)g_icp_iter;  // only the timing build's stamps use it
  const int lane = threadIdx.x & 63;
  float x6[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  int go = 0;
  // EVERY lane solves (the same LDS words, the same arithmetic: a wave's instruction costs the same for one lane as for
  // 64), so that nothing has to be masked, branched around or broadcast afterwards
  if (!__builtin_amdgcn_readfirstlane(p.lost)) {  // (the same in every lane: a scalar branch)
    double s[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) s[k] = tot[k];
    ICP_STAMP(5);
    go = hsk_solve6(s, x6) ? 1 : 0;
    if (!go) p.lost = 1;
    ICP_STAMP(6);
  }
  // Every QUAD of lanes does what lanes 0..2 would: lane q of a quad (the fourth doubles as the third) evaluates one
  // sine / cosine pair and one row of the pose update, and the values travel inside the quad by DPP moves -- the 18
  // broadcasts were v_readlane, each a trip through a scalar register with its wait states on the solve's chain.
  const int qi = (lane & 3) < 3 ? (lane & 3) : 2;
  double sd, cd;
  hsk_sincos((double)(qi == 0 ? x6[0] : (qi == 1 ? x6[1] : x6[2])), &sd, &cd);
  const float sf = (float)sd, cf = (float)cd;
  const float sa = quad_bcast<0>(sf), ca = quad_bcast<0>(cf);
  const float sb = quad_bcast<1>(sf), cb = quad_bcast<1>(cf);
  const float sg = quad_bcast<2>(sf), cg = quad_bcast<2>(cf);
  ICP_STAMP(7);
  if (go) {  // wave-uniform
    float rn[3], tn;
    hsk_pose_update_row(qi, p.R, p.t, x6, sa, ca, sb, cb, sg, cg, rn, &tn);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      p.R[0 * 3 + j] = quad_bcast<0>(rn[j]);
      p.R[1 * 3 + j] = quad_bcast<1>(rn[j]);
      p.R[2 * 3 + j] = quad_bcast<2>(rn[j]);
    }
    p.t[0] = quad_bcast<0>(tn);
    p.t[1] = quad_bcast<1>(tn);
    p.t[2] = quad_bcast<2>(tn);
    p.n_iter += 1;
  }
  ICP_STAMP(8);
}

// The 27 sums travel between launches through sharded accumulators instead of per-block partial rows: the
// products are integer multiples of 2^-26 (exact in binary64), so hardware f64 atomic adds give the same bits in
// any arrival order, and the next launch reads ICP_SHARDS x 27 doubles instead of (blocks) x 27 (19 MB of L2 reads
// per fine iteration before).  Three slots rotate: iteration i adds into slot i % 3, reads slot (i - 1) % 3 and
// clears slot (i + 1) % 3 for its successor; slot 0 is empty at the start of a frame (cleared at creation and by
// k_icp_final).
#define ICP_SHARDS 32
#define ICP_SLOT_DOUBLES (ICP_SHARDS * 32)
// The same totals gathered by ONE wave (the one that solves): lane l adds 16 of the 32 shards of sum l & 31, one lane
// swap joins the halves, lanes 0..26 leave the totals in tot[].  No block barrier, no second LDS stage -- the other
// waves of the block have nothing to do before the pose is known anyway.  (Any order of addition gives the same bits.)
// (in two parts, so that the caller can request the loads before anything else it has to fetch: loads return in the order
// they were issued, and the sums are what the chain of an iteration waits for)
static __device__ __forceinline__ void shard_load27_wave(const double* __restrict__ slot, double* a) {
  const int lane = threadIdx.x & 63, k = lane & 31, half = lane >> 5;
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = slot[(half * 16 + j) * 32 + k];
}
static __device__ __forceinline__ void shard_sum27_wave(double* a, double* tot) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int w = 8; w > 0; w >>= 1)
#pragma unroll
    for (int j = 0; j < w; ++j) a[j] = a[j] + a[j + w];
  const double v = swap32_add_f64(a[0], a[0]);  // every lane: its half + the other half
  if (lane < 27) tot[lane] = v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
static __device__ __forceinline__ void shard_reduce27_wave(const double* __restrict__ slot, double* tot) {
  const int lane = threadIdx.x & 63, k = lane & 31, half = lane >> 5;
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = slot[(half * 16 + j) * 32 + k];
#pragma unroll
  for (int w = 8; w > 0; w >>= 1)
#pragma unroll
    for (int j = 0; j < w; ++j) a[j] = a[j] + a[j + w];
  const double v = swap32_add_f64(a[0], a[0]);  // every lane: its half + the other half
  if (lane < 27) tot[lane] = v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// What the first kernel of integrate needs to finish a frame's ICP itself (launch_icp_fused hands it out instead of
// launching k_icp_final): where the last pose estimate is, the accumulator slots, and the iteration count.
struct IcpFinal {
  const IcpPose* pose_in;
  double* slots;
  int iter;
};
