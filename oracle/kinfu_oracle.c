/*
 * kinfu_oracle.c -- CPU restatement of the KinectFusion hot path.  TEST INFRASTRUCTURE ONLY; see the
 * header of kinfu_oracle.h ("PARITY UNPINNED").  Every stage cites the SURVEY.md Appendix A paragraph it
 * follows (the reference repository itself holds no KinFu source: /root/reference/README.md:13-14).
 *
 * Numerical contract (shared with the HIP kernels; the 6x6 solve, sin/cos and the integration gate exist as two
 * separately written texts, here and in housescan_amd/csrc, that must agree bit for bit):
 *   - IEEE-754 binary32/binary64, round-to-nearest-even, one rounding per written operator;
 *     no fused multiply-add anywhere (build with -ffp-contract=off), no fast-math;
 *   - sqrtf and '/' are the correctly rounded operations;
 *   - the only transcendental on data is the bilateral range weight, taken from a host-computed table;
 *     sin/cos of the ICP increment use ora_sincos (fixed polynomial in binary64) so that the whole
 *     tracker is bit-reproducible across CPU and GPU;
 *   - expression trees are evaluated exactly as parenthesised below.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math [-fopenmp]).
 *
 * ORA_LITERAL (libkinfu_oracle_literal.so; oracle/Makefile): the SAME pipeline with this build's deliberate deviations
 * from Appendix-A-literal arithmetic taken back out -- FMA contraction allowed (-ffp-contract=fast -mfma: what nvcc does
 * by default), one expf per bilateral tap with a zero centre filtered like any other (D1; the window's exclusive clip
 * is the specification's own since round 4, like pyrDown's), plain binary64 products and sums in the ICP (no 2^-26 snap,
 * D4), LLT Cholesky with square roots (D4), libm sinf / cosf for the pose increment, every interpolated hit time accepted
 * (D3: the specification keeps [t - step, t + 2 step]), "1 / z < 0" as the only in-front test
 * (D6), and integrate's camera coordinates advanced incrementally along z as upstream does (A.4's closing note).  It is the yardstick for the north_star's "within a stated tolerance" (tools/spec_vs_literal.py,
 * tests/test_spec_vs_literal.py): how far the bit-reproducible specification moves TSDF values and poses from the
 * PCL-form arithmetic.  Still a recollection of PCL (parity unpinned), still test infrastructure.
 */
#include "kinfu_oracle.h"

/* ORA_LITERAL = every deviation taken back out; the single switches exist so that tools/spec_vs_literal.py can attribute
 * the difference to its causes (one variant library per switch, oracle/Makefile: variant) */
#ifdef ORA_LITERAL
#define ORA_LIT_D1 1 /* bilateral: one expf per tap, zero centre filtered (the exclusive window clip is the specification's too) */
#define ORA_LIT_D3 1 /* raycast: extrapolated hit times accepted */
#define ORA_LIT_D4 1 /* ICP: plain binary64 sums, LLT Cholesky, libm sinf / cosf */
#define ORA_LIT_D6 1 /* integrate: "1 / z < 0" as the in-front test */
#define ORA_LIT_INC 1 /* integrate: camera coordinates advanced incrementally along z (A.4's closing note) */
#endif

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif
#define NANF (__builtin_nanf(""))

/* threads of the OpenMP builds (bench.py's cpu_baseline times the same build on all cores and on one); no effect otherwise */
void ora_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------ */
/* A.1 constants                                                                                     */
/* ------------------------------------------------------------------------------------------------ */
void ora_default_config(ora_config* c, int vol_n) {
  memset(c, 0, sizeof(*c));
  c->vol[0] = c->vol[1] = c->vol[2] = vol_n;
  c->size[0] = c->size[1] = c->size[2] = 3.0f;
  c->trunc = 0.03f;
  c->W = 640;
  c->H = 480;
  c->fx = c->fy = 525.0f;
  c->cx = 319.5f;
  c->cy = 239.5f;
  c->icp_iters[0] = 10;
  c->icp_iters[1] = 5;
  c->icp_iters[2] = 4;
  c->dist_thresh = 0.10f;
  c->angle_thresh = 0.3420201433256687f; /* sin(20 deg) */
  c->move_thresh = 0.0f;
  c->init_R[0] = c->init_R[4] = c->init_R[8] = 1.0f;
  /* t = size/2 - (0,0,1.2*size_z/2)  (A.1) */
  c->init_t[0] = c->size[0] / 2.0f;
  c->init_t[1] = c->size[1] / 2.0f;
  c->init_t[2] = c->size[2] / 2.0f - 1.2f * c->size[2] / 2.0f;
}

/* truncation distance clamped to >= 2.1 * max cell (A.1) */
float ora_tau(const ora_config* c) {
  float cx = c->size[0] / (float)c->vol[0];
  float cy = c->size[1] / (float)c->vol[1];
  float cz = c->size[2] / (float)c->vol[2];
  float m = cx > cy ? cx : cy;
  m = m > cz ? m : cz;
  float lo = 2.1f * m;
  return c->trunc > lo ? c->trunc : lo;
}

/* round-to-nearest-even to int with a range guard (values outside +-1e6 or NaN are "no pixel") */
static inline int rint_guard(float f, int* out) {
  if (!(f > -1.0e6f && f < 1.0e6f)) return 0;
  *out = (int)lrintf(f);
  return 1;
}

/* ------------------------------------------------------------------------------------------------ */
/* A.4 scaleDepth: z-depth in mm -> ray length in metres                                             */
/* ------------------------------------------------------------------------------------------------ */
void ora_scale_depth(const uint16_t* depth, int W, int H, float fx, float fy, float cx, float cy, float* out) {
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      float xl = ((float)u - cx) / fx;
      float yl = ((float)v - cy) / fy;
      float lambda = sqrtf((xl * xl + yl * yl) + 1.0f);
      out[v * W + u] = ((float)depth[v * W + u] * lambda) / 1000.0f;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* A.4 integrate (tsdf23), direct (non-incremental) form                                             */
/* ------------------------------------------------------------------------------------------------ */
#ifdef ORA_LIT_INC
/* A.4 as upstream walks it: one (x, y) column at a time, z innermost; the camera-space x and y (pre-multiplied by fx, fy)
 * and the depth term are ADVANCED by float additions from plane to plane instead of being formed from the voxel index --
 * differences of an ulp or so, which flip a round-to-nearest pixel now and then (SURVEY.md A.4's closing note: "count and
 * report such voxels").  The in-front test is upstream's "1 / z < 0"; the update rule is the same. */
uint64_t ora_integrate(int16_t* vol, const int dims[3], const float size[3], float tau, int zs0, int nzs,
                       const float* scaled, int W, int H, float fx, float fy, float cx, float cy,
                       const float R[9], const float t[3]) {
  const int X = dims[0], Y = dims[1];
  const float cellx = size[0] / (float)dims[0], celly = size[1] / (float)dims[1], cellz = size[2] / (float)dims[2];
  const float tau_inv = 1.0f / tau;
  const float i00 = R[0], i01 = R[3], i02 = R[6], i10 = R[1], i11 = R[4], i12 = R[7], i20 = R[2], i21 = R[5], i22 = R[8];
  uint64_t n_upd = 0;
  /* (walked plane by plane for the memory's sake: the per-column accumulators are kept in arrays, the terms that are the
   * same for every column -- gz, z_scaled -- as scalars; the additions are the ones the column-wise loop would make) */
  const size_t ncol = (size_t)X * Y;
  float* acc = (float*)malloc(ncol * 4 * sizeof(float)); /* vx, vy (advanced), vz, part_norm (fixed) per column */
  float *avx = acc, *avy = acc + ncol, *avz = acc + 2 * ncol, *apn = acc + 3 * ncol;
  const float gz0 = ((float)zs0 + 0.5f) * cellz - t[2];
  const float dx = i02 * cellz * fx, dy = i12 * cellz * fy;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      const float gx = ((float)x + 0.5f) * cellx - t[0];
      const float gy = ((float)y + 0.5f) * celly - t[1];
      const size_t c = (size_t)y * X + x;
      avx[c] = (i00 * gx + i01 * gy + i02 * gz0) * fx;
      avy[c] = (i10 * gx + i11 * gy + i12 * gz0) * fy;
      avz[c] = i20 * gx + i21 * gy + i22 * gz0;
      apn[c] = gx * gx + gy * gy;
    }
  float gz = gz0, z_scaled = 0.0f;
  for (int zz = 0; zz < nzs; ++zz, gz += cellz, z_scaled += cellz) {
    const float zterm = i22 * z_scaled, gz2 = gz * gz;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : n_upd) schedule(static)
#endif
    for (int y = 0; y < Y; ++y)
      for (int x = 0; x < X; ++x) {
        const size_t c = (size_t)y * X + x;
        const float vx = avx[c], vy = avy[c];
        avx[c] = vx + dx; /* the advance of the loop header: made whether or not the voxel is updated */
        avy[c] = vy + dy;
        const float inv_z = 1.0f / (avz[c] + zterm);
        if (inv_z < 0.0f) continue;
        int u, v;
        if (!rint_guard(vx * inv_z + cx, &u) || !rint_guard(vy * inv_z + cy, &v)) continue;
        if (u < 0 || v < 0 || u >= W || v >= H) continue;
        const float Ds = scaled[v * W + u];
        const float sdf = Ds - sqrtf(gz2 + apn[c]);
        if (Ds != 0.0f && sdf >= -tau) {
          float F = sdf * tau_inv;
          F = F < 1.0f ? F : 1.0f;
          int16_t* vox = vol + 2 * ((size_t)zz * ncol + c);
          const float Fp = (float)vox[0] / 32767.0f, Wp = (float)vox[1];
          const float Fn = (Fp * Wp + F) / (Wp + 1.0f);
          int wn = vox[1] + 1;
          if (wn > ORA_MAX_WEIGHT) wn = ORA_MAX_WEIGHT;
          int fixed = (int)(Fn * 32767.0f);
          if (fixed > ORA_DIVISOR) fixed = ORA_DIVISOR;
          if (fixed < -ORA_DIVISOR) fixed = -ORA_DIVISOR;
          vox[0] = (int16_t)fixed;
          vox[1] = (int16_t)wn;
          ++n_upd;
        }
      }
  }
  free(acc);
  return n_upd;
}
#else
uint64_t ora_integrate(int16_t* vol, const int dims[3], const float size[3], float tau, int zs0, int nzs,
                       const float* scaled, int W, int H, float fx, float fy, float cx, float cy,
                       const float R[9], const float t[3]) {
  const int X = dims[0], Y = dims[1];
  const float cellx = size[0] / (float)dims[0];
  const float celly = size[1] / (float)dims[1];
  const float cellz = size[2] / (float)dims[2];
  const float tau_inv = 1.0f / tau;
  /* Rinv = R^T */
  const float i00 = R[0], i01 = R[3], i02 = R[6];
  const float i10 = R[1], i11 = R[4], i12 = R[7];
  const float i20 = R[2], i21 = R[5], i22 = R[8];
  uint64_t n_upd = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : n_upd) schedule(static)
#endif
  for (int zz = 0; zz < nzs; ++zz) {
    const int z = zs0 + zz;
    const float gz = ((float)z + 0.5f) * cellz - t[2];
    for (int y = 0; y < Y; ++y) {
      const float gy = ((float)y + 0.5f) * celly - t[1];
      int16_t* row = vol + 2 * ((size_t)zz * Y + y) * X;
      for (int x = 0; x < X; ++x) {
        const float gx = ((float)x + 0.5f) * cellx - t[0];
        const float camx = (i00 * gx + i01 * gy) + i02 * gz;
        const float camy = (i10 * gx + i11 * gy) + i12 * gz;
        const float camz = (i20 * gx + i21 * gy) + i22 * gz;
#ifdef ORA_LIT_D6
        const float inv_z = 1.0f / camz;
        if (inv_z < 0.0f) continue; /* A.4: "implemented as 1 / v_z < 0"; z = 0 or NaN falls to the range guard below */
#else
        if (!(camz >= 1.17549435e-38f)) continue; /* in front of the camera (D6: a denormal depth counts as not in front) */
        const float inv_z = 1.0f / camz;
#endif
        const float fu = (camx * fx) * inv_z + cx;
        const float fv = (camy * fy) * inv_z + cy;
        int u, v;
        if (!rint_guard(fu, &u) || !rint_guard(fv, &v)) continue;
        if (u < 0 || v < 0 || u >= W || v >= H) continue;
        const float Ds = scaled[v * W + u];
        const float dist = sqrtf(gz * gz + (gx * gx + gy * gy));
        const float sdf = Ds - dist;
        if (Ds != 0.0f && sdf >= -tau) {
          float F = sdf * tau_inv;
          F = F < 1.0f ? F : 1.0f;
          const int16_t tp = row[2 * x], wp = row[2 * x + 1];
          const float Fp = (float)tp / 32767.0f;
          const float Wp = (float)wp;
          const float Fn = (Fp * Wp + F) / (Wp + 1.0f);
          int wn = wp + 1;
          if (wn > ORA_MAX_WEIGHT) wn = ORA_MAX_WEIGHT;
          int fixed = (int)(Fn * 32767.0f); /* truncation toward zero */
          if (fixed > ORA_DIVISOR) fixed = ORA_DIVISOR;
          if (fixed < -ORA_DIVISOR) fixed = -ORA_DIVISOR;
          row[2 * x] = (int16_t)fixed;
          row[2 * x + 1] = (int16_t)wn;
          ++n_upd;
        }
      }
    }
  }
  return n_upd;
}
#endif

/* ------------------------------------------------------------------------------------------------ */
/* A.3 bilateral filter: 13x13, sigma_space 4.5 px, sigma_color 30 mm                                 */
/* weight = ws(dx,dy) * wc(|dd|), each factor (float)exp((double)arg); taps summed dy-major, dx-minor */
/* ------------------------------------------------------------------------------------------------ */
#define BIL_R 6
#define BIL_LUT 512
#ifndef ORA_LIT_D1
static float g_ws[13][13];
static float g_wc[BIL_LUT];
static int g_bil_init = 0;
static void bil_init(void) {
  if (g_bil_init) return;
  const float sig_s = 4.5f, sig_c = 30.0f;
  const float s2 = 0.5f / (sig_s * sig_s);
  const float c2 = 0.5f / (sig_c * sig_c);
  for (int dy = -BIL_R; dy <= BIL_R; ++dy)
    for (int dx = -BIL_R; dx <= BIL_R; ++dx) g_ws[dy + BIL_R][dx + BIL_R] = (float)exp(-(double)((float)(dx * dx + dy * dy) * s2));
  for (int k = 0; k < BIL_LUT; ++k) g_wc[k] = (float)exp(-(double)((float)(k * k) * c2));
  g_bil_init = 1;
}
#endif

#ifdef ORA_LIT_D1
/* A.3 as written: window [x - 6, min(x + 7, W - 1)) x [y - 6, min(y + 7, H - 1)) (the upper clip is EXCLUSIVE of the last
 * column / row), one expf of the summed exponent per tap, a zero centre filtered like any other value; an empty or
 * all-underflowed window (0 / 0) gives 0, as __float2int_rn(NaN) does */
void ora_bilateral(const uint16_t* src, int W, int H, uint16_t* dst) {
  const float sig_s = 4.5f, sig_c = 30.0f;
  const float s2 = 0.5f / (sig_s * sig_s), c2 = 0.5f / (sig_c * sig_c);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const int value = src[y * W + x];
      const int tx = x - BIL_R + 13 < W - 1 ? x - BIL_R + 13 : W - 1;
      const int ty = y - BIL_R + 13 < H - 1 ? y - BIL_R + 13 : H - 1;
      float sum1 = 0.0f, sum2 = 0.0f;
      for (int cy = y - BIL_R < 0 ? 0 : y - BIL_R; cy < ty; ++cy)
        for (int cx = x - BIL_R < 0 ? 0 : x - BIL_R; cx < tx; ++cx) {
          const int tmp = src[cy * W + cx];
          const float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
          const float color2 = (float)((value - tmp) * (value - tmp));
          const float w = expf(-(space2 * s2 + color2 * c2));
          sum1 += (float)tmp * w;
          sum2 += w;
        }
      int res = sum2 > 0.0f ? (int)lrintf(sum1 / sum2) : 0;
      if (res < 0) res = 0;
      if (res > 32767) res = 32767;
      dst[y * W + x] = (uint16_t)res;
    }
}
#else
void ora_bilateral(const uint16_t* src, int W, int H, uint16_t* dst) {
  bil_init();
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const int value = src[y * W + x];
      if (value == 0) {
        dst[y * W + x] = 0;
        continue;
      }
      /* the window's upper clip is EXCLUSIVE of the image's last column / row, as upstream's loop bounds are
       * (cx < min(x - 6 + 13, W - 1)): column W - 1 and row H - 1 are in no window, not even their own pixels' */
      const int y0 = y - BIL_R < 0 ? 0 : y - BIL_R, y1 = y + BIL_R > H - 2 ? H - 2 : y + BIL_R;
      const int x0 = x - BIL_R < 0 ? 0 : x - BIL_R, x1 = x + BIL_R > W - 2 ? W - 2 : x + BIL_R;
      float sum1 = 0.0f, sum2 = 0.0f;
      for (int cy = y0; cy <= y1; ++cy)
        for (int cx = x0; cx <= x1; ++cx) {
          const int tmp = src[cy * W + cx];
          int dd = value - tmp;
          if (dd < 0) dd = -dd;
          const float wc = dd < BIL_LUT ? g_wc[dd] : 0.0f;
          const float w = g_ws[cy - y + BIL_R][cx - x + BIL_R] * wc;
          sum1 = sum1 + (float)tmp * w;
          sum2 = sum2 + w;
        }
      int res = sum2 > 0.0f ? (int)lrintf(sum1 / sum2) : 0; /* (0 / 0: a last-column / last-row pixel whose window holds no weight) */
      if (res < 0) res = 0;
      if (res > 32767) res = 32767;
      dst[y * W + x] = (uint16_t)res;
    }
}
#endif

/* A.3 pyrDown: 5x5 window mean of depths within 3*sigma_color of the centre (integer arithmetic) */
void ora_pyrdown(const uint16_t* src, int W, int H, uint16_t* dst) {
  const int w2 = W / 2, h2 = H / 2;
  for (int y = 0; y < h2; ++y)
    for (int x = 0; x < w2; ++x) {
      const int center = src[(2 * y) * W + 2 * x];
      /* upper clip exclusive of the last column / row, as in the bilateral (upstream: cx < min(2x - 2 + 5, W - 1)); the
       * centre (2y, 2x) <= (H - 2, W - 2) is always inside.  Integer arithmetic: this IS the Appendix-A-literal form too. */
      const int y0 = 2 * y - 2 < 0 ? 0 : 2 * y - 2, y1 = 2 * y + 2 > H - 2 ? H - 2 : 2 * y + 2;
      const int x0 = 2 * x - 2 < 0 ? 0 : 2 * x - 2, x1 = 2 * x + 2 > W - 2 ? W - 2 : 2 * x + 2;
      int sum = 0, count = 0;
      for (int cy = y0; cy <= y1; ++cy)
        for (int cx = x0; cx <= x1; ++cx) {
          const int val = src[cy * W + cx];
          int d = val - center;
          if (d < 0) d = -d;
          if (d < 90) { /* 3 * sigma_color */
            sum += val;
            ++count;
          }
        }
      dst[y * w2 + x] = (uint16_t)(sum / count);
    }
}

/* A.3 vertex map: SoA planes x,y,z; NaN = invalid */
void ora_vmap(const uint16_t* depth, int W, int H, float fx, float fy, float cx, float cy, float* vmap) {
  const float fx_inv = 1.0f / fx, fy_inv = 1.0f / fy;
  const size_t P = (size_t)W * H;
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      const size_t i = (size_t)v * W + u;
      const float z = (float)depth[i] / 1000.0f;
      if (z != 0.0f) {
        vmap[i] = (z * ((float)u - cx)) * fx_inv;
        vmap[P + i] = (z * ((float)v - cy)) * fy_inv;
        vmap[2 * P + i] = z;
      } else {
        vmap[i] = vmap[P + i] = vmap[2 * P + i] = NANF;
      }
    }
}

static inline float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void rot3(const float R[9], const float v[3], float o[3]) {
  o[0] = (R[0] * v[0] + R[1] * v[1]) + R[2] * v[2];
  o[1] = (R[3] * v[0] + R[4] * v[1]) + R[5] * v[2];
  o[2] = (R[6] * v[0] + R[7] * v[1]) + R[8] * v[2];
}

/* A.3 normal map: normalised cross product of forward differences */
void ora_nmap(const float* vmap, int W, int H, float* nmap) {
  const size_t P = (size_t)W * H;
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      const size_t i = (size_t)v * W + u;
      float n[3] = {NANF, NANF, NANF};
      if (u < W - 1 && v < H - 1) {
        const size_t i01 = i + 1, i10 = i + W;
        if (!isnan(vmap[i]) && !isnan(vmap[i01]) && !isnan(vmap[i10])) {
          const float a[3] = {vmap[i01] - vmap[i], vmap[P + i01] - vmap[P + i], vmap[2 * P + i01] - vmap[2 * P + i]};
          const float b[3] = {vmap[i10] - vmap[i], vmap[P + i10] - vmap[P + i], vmap[2 * P + i10] - vmap[2 * P + i]};
          float r[3];
          cross3(a, b, r);
          const float inv = 1.0f / sqrtf(dot3(r, r));
          n[0] = r[0] * inv;
          n[1] = r[1] * inv;
          n[2] = r[2] * inv;
        }
      }
      nmap[i] = n[0];
      nmap[P + i] = n[1];
      nmap[2 * P + i] = n[2];
    }
}

/* A.2 tranformMaps: v_g = R v + t, n_g = R n */
void ora_transform_maps(const float* vsrc, const float* nsrc, int W, int H, const float R[9], const float t[3],
                        float* vdst, float* ndst) {
  const size_t P = (size_t)W * H;
  for (size_t i = 0; i < P; ++i) {
    float v[3] = {vsrc[i], vsrc[P + i], vsrc[2 * P + i]};
    float o[3] = {NANF, NANF, NANF};
    if (!isnan(v[0])) {
      rot3(R, v, o);
      o[0] = o[0] + t[0];
      o[1] = o[1] + t[1];
      o[2] = o[2] + t[2];
    }
    vdst[i] = o[0];
    vdst[P + i] = o[1];
    vdst[2 * P + i] = o[2];
    float n[3] = {nsrc[i], nsrc[P + i], nsrc[2 * P + i]};
    float q[3] = {NANF, NANF, NANF};
    if (!isnan(n[0])) rot3(R, n, q);
    ndst[i] = q[0];
    ndst[P + i] = q[1];
    ndst[2 * P + i] = q[2];
  }
}

/* A.3 resizeVMap / resizeNMap: 2x2 mean, NaN if any tap is NaN; normals renormalised */
static void resize_map(const float* src, int W, int H, float* dst, int normalize) {
  const int w2 = W / 2, h2 = H / 2;
  const size_t P = (size_t)W * H, P2 = (size_t)w2 * h2;
  for (int y = 0; y < h2; ++y)
    for (int x = 0; x < w2; ++x) {
      const size_t i00 = (size_t)(2 * y) * W + 2 * x, i01 = i00 + 1, i10 = i00 + W, i11 = i10 + 1;
      const size_t o = (size_t)y * w2 + x;
      if (isnan(src[i00]) || isnan(src[i01]) || isnan(src[i10]) || isnan(src[i11])) {
        dst[o] = dst[P2 + o] = dst[2 * P2 + o] = NANF;
        continue;
      }
      float n[3];
      for (int k = 0; k < 3; ++k)
        n[k] = (((src[k * P + i00] + src[k * P + i01]) + src[k * P + i10]) + src[k * P + i11]) / 4.0f;
      if (normalize) {
        const float inv = 1.0f / sqrtf(dot3(n, n));
        n[0] = n[0] * inv;
        n[1] = n[1] * inv;
        n[2] = n[2] * inv;
      }
      dst[o] = n[0];
      dst[P2 + o] = n[1];
      dst[2 * P2 + o] = n[2];
    }
}
void ora_resize_vmap(const float* src, int W, int H, float* dst) { resize_map(src, W, H, dst, 0); }
void ora_resize_nmap(const float* src, int W, int H, float* dst) { resize_map(src, W, H, dst, 1); }

/* ------------------------------------------------------------------------------------------------ */
/* A.5 ICP estimateCombined                                                                           */
/* Products are formed in binary64 (exact for binary32 factors) and snapped to multiples of 2^-26      */
/* before summation; every partial sum is then exactly representable while |sum| < 2^27, which makes   */
/* the 27 sums independent of summation order (CPU loop == GPU tree == multi-GPU all-reduce).          */
/* ------------------------------------------------------------------------------------------------ */
#ifdef ORA_LIT_D4
static inline double quant26(double x) { return x; } /* plain binary64 products (A.5: float_type = double) */
#else
static inline double quant26(double x) { return rint(x * 67108864.0) * (1.0 / 67108864.0); }
#endif

uint64_t ora_icp_accumulate(const float* vcur, const float* ncur, const float* vprev_g, const float* nprev_g,
                            int W, int H, float fx, float fy, float cx, float cy,
                            const float R[9], const float t[3], const float Rprev[9], const float tprev[3],
                            float dist_thresh, float angle_thresh, int row0, int row1, double out27[27]) {
  const size_t P = (size_t)W * H;
  /* Rprev_inv = Rprev^T */
  const float Ri[9] = {Rprev[0], Rprev[3], Rprev[6], Rprev[1], Rprev[4], Rprev[7], Rprev[2], Rprev[5], Rprev[8]};
  double acc[27];
  for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  uint64_t n_valid = 0;
  /* the snapped sums are exact, hence independent of how the rows are split over threads (the literal form's plain
   * binary64 sums are not: it adds the pixels in image order on one thread) */
#if defined(_OPENMP) && !defined(ORA_LIT_D4)
#pragma omp parallel for reduction(+ : n_valid) reduction(+ : acc[:27]) schedule(static)
#endif
  for (int y = row0; y < row1; ++y)
    for (int x = 0; x < W; ++x) {
      const size_t i = (size_t)y * W + x;
      const float nc[3] = {ncur[i], ncur[P + i], ncur[2 * P + i]};
      if (isnan(nc[0])) continue;
      const float vc[3] = {vcur[i], vcur[P + i], vcur[2 * P + i]};
      float vg[3];
      rot3(R, vc, vg);
      vg[0] = vg[0] + t[0];
      vg[1] = vg[1] + t[1];
      vg[2] = vg[2] + t[2];
      const float dv[3] = {vg[0] - tprev[0], vg[1] - tprev[1], vg[2] - tprev[2]};
      float vcp[3];
      rot3(Ri, dv, vcp);
      if (!(vcp[2] > 0.0f)) continue;
      const float fu = (vcp[0] * fx) / vcp[2] + cx;
      const float fv = (vcp[1] * fy) / vcp[2] + cy;
      int u, v;
      if (!rint_guard(fu, &u) || !rint_guard(fv, &v)) continue;
      if (u < 0 || v < 0 || u >= W || v >= H) continue;
      const size_t j = (size_t)v * W + u;
      const float np[3] = {nprev_g[j], nprev_g[P + j], nprev_g[2 * P + j]};
      if (isnan(np[0])) continue;
      const float vp[3] = {vprev_g[j], vprev_g[P + j], vprev_g[2 * P + j]};
      const float df[3] = {vp[0] - vg[0], vp[1] - vg[1], vp[2] - vg[2]};
      const float dist = sqrtf(dot3(df, df));
      if (!(dist <= dist_thresh)) continue;
      float ng[3];
      rot3(R, nc, ng);
      float cr[3];
      cross3(ng, np, cr);
      const float sine = sqrtf(dot3(cr, cr));
      if (!(sine < angle_thresh)) continue;
      float row[7];
      cross3(vg, np, row); /* s x n */
      row[3] = np[0];
      row[4] = np[1];
      row[5] = np[2];
      row[6] = dot3(np, df); /* n . (d - s) */
      int k = 0;
      for (int a = 0; a < 6; ++a)
        for (int b = a; b < 7; ++b) acc[k++] += quant26((double)row[a] * (double)row[b]);
      ++n_valid;
    }
  for (int k = 0; k < 27; ++k) out27[k] = acc[k];
  return n_valid;
}

/* A.5 host side.  The 27 sums are the rows of the upper triangle of the augmented system [A | b], packed:
 * (i, j), i <= j <= 6, sits at 7 i - i (i - 1) / 2 + (j - i).  A.5 names a Cholesky solve; this build's specification
 *  is the square-root-free form A = L D L^T, L unit lower triangular, with ONE reciprocal per pivot:
 *   d_c   = A_cc - sum_{q<c} (L_cq L_cq) d_q                      (q ascending)
 *   L_rc  = (A_rc - sum_{q<c} (L_rq L_cq) d_q) * (1 / d_c)
 *   det   = d_0 d_1 ... d_5                                        (left to right)
 *   L y = b forwards, then x_r = y_r (1 / d_r) - sum_{q>r} L_qr x_q backwards (q ascending).
 * Lost (return 0) when a pivot is not positive, det < 1e-15 or NaN, or a component is NaN / not below 1e30.
 * Written here from that specification; the device solve in housescan_amd/csrc is a separate text. */
static int tri_at(int i, int j) { return 7 * i - (i * (i - 1)) / 2 + (j - i); }

static int solve_lost(float x6[6]) {
  for (int q = 0; q < 6; ++q) x6[q] = 0.0f;
  return 0;
}

#ifdef ORA_LIT_D4
/* A.2 step (3) as written: A symmetric from the packed triangle, |det A| < 1e-15 or NaN => lost, x = A.llt().solve(b)
 * (Cholesky A = L L^T with square roots, forward and back substitution), cast to float.  det A = (prod L_ii)^2. */
int ora_icp_solve(const double in27[27], float x6[6]) {
  double L[6][6];
  double det = 1.0;
  for (int c = 0; c < 6; ++c) {
    double d = in27[tri_at(c, c)];
    for (int q = 0; q < c; ++q) d -= L[c][q] * L[c][q];
    if (!(d > 0.0)) return solve_lost(x6); /* not positive definite: Eigen's LLT would return garbage; det test below would catch most */
    L[c][c] = sqrt(d);
    det *= d;
    for (int r = c + 1; r < 6; ++r) {
      double a = in27[tri_at(c, r)];
      for (int q = 0; q < c; ++q) a -= L[r][q] * L[c][q];
      L[r][c] = a / L[c][c];
    }
  }
  if (!(fabs(det) >= 1e-15)) return solve_lost(x6);
  double y[6], x[6];
  for (int r = 0; r < 6; ++r) {
    double acc = in27[tri_at(r, 6)];
    for (int q = 0; q < r; ++q) acc -= L[r][q] * y[q];
    y[r] = acc / L[r][r];
  }
  for (int r = 5; r >= 0; --r) {
    double acc = y[r];
    for (int q = r + 1; q < 6; ++q) acc -= L[q][r] * x[q];
    x[r] = acc / L[r][r];
  }
  for (int q = 0; q < 6; ++q)
    if (isnan(x[q]) || !(fabs(x[q]) < 1e30)) return solve_lost(x6);
  for (int q = 0; q < 6; ++q) x6[q] = (float)x[q];
  return 1;
}
#else
int ora_icp_solve(const double in27[27], float x6[6]) {
  double low[6][6]; /* strictly lower part of L; only [r][c] with c < r is ever read */
  double piv[6], rpiv[6];
  double det = 1.0;
  for (int c = 0; c < 6; ++c) {
    double d = in27[tri_at(c, c)];
    for (int q = 0; q < c; ++q) d -= (low[c][q] * low[c][q]) * piv[q];
    if (!(d > 0.0)) return solve_lost(x6);
    piv[c] = d;
    rpiv[c] = 1.0 / d;
    det *= d;
    for (int r = c + 1; r < 6; ++r) {
      double a = in27[tri_at(c, r)]; /* A_rc = A_cr */
      for (int q = 0; q < c; ++q) a -= (low[r][q] * low[c][q]) * piv[q];
      low[r][c] = a * rpiv[c];
    }
  }
  if (!(det >= 1e-15)) return solve_lost(x6); /* NaN fails the comparison too */
  double y[6], x[6];
  for (int r = 0; r < 6; ++r) {
    double acc = in27[tri_at(r, 6)];
    for (int q = 0; q < r; ++q) acc -= low[r][q] * y[q];
    y[r] = acc;
  }
  for (int r = 5; r >= 0; --r) {
    double acc = y[r] * rpiv[r];
    for (int q = r + 1; q < 6; ++q) acc -= low[q][r] * x[q];
    x[r] = acc;
  }
  for (int q = 0; q < 6; ++q)
    if (isnan(x[q]) || !(fabs(x[q]) < 1e30)) return solve_lost(x6);
  for (int q = 0; q < 6; ++q) x6[q] = (float)x[q];
  return 1;
}
#endif

/* sin and cos of the ICP increment, bit-reproducible on every machine: Cody-Waite reduction by pi/2 in two pieces
 * (k = rint(x 2/pi), r = (x - k hi) - k lo), then the Taylor polynomials to r^15 / r^16 in Horner form, binary64:
 *   sin r = r + (r r^2) S(r^2),  S = ((((((-1/15! r2 + 1/13!) r2 - 1/11!) r2 + 1/9!) r2 - 1/7!) r2 + 1/5!) r2 - 1/3!)
 *   cos r = (1 - r^2 / 2) + (r^2 r^2) C(r^2),  C likewise from 1/16! down to 1/4!
 * and the quadrant k mod 4 picks (s, c) from the cycle sr, cr, -sr, -cr.  |x| >= 1e5 or NaN gives (0, 1). */
static const double kSinTail[7] = {-(1.0 / 1307674368000.0), 1.0 / 6227020800.0, -(1.0 / 39916800.0), 1.0 / 362880.0,
                                   -(1.0 / 5040.0),          1.0 / 120.0,        -(1.0 / 6.0)};
static const double kCosTail[7] = {1.0 / 20922789888000.0, -(1.0 / 87178291200.0), 1.0 / 479001600.0, -(1.0 / 3628800.0),
                                   1.0 / 40320.0,          -(1.0 / 720.0),         1.0 / 24.0};

void ora_sincos(double x, double* s, double* c) {
  *s = 0.0;
  *c = 1.0;
  if (!(fabs(x) < 1.0e5)) return;
  const double k = rint(x * 0.63661977236758134308);
  double r = x - k * 1.57079632673412561417e+00; /* the first 33 bits of pi/2: k times it is exact */
  r = r - k * 6.07710050650619224932e-11;
  const double r2 = r * r;
  double ps = kSinTail[0], pc = kCosTail[0];
  for (int i = 1; i < 7; ++i) {
    ps = ps * r2 + kSinTail[i];
    pc = pc * r2 + kCosTail[i];
  }
  const double sr = r + (r * r2) * ps;
  const double cr = (1.0 - 0.5 * r2) + (r2 * r2) * pc;
  const double cycle[4] = {sr, cr, -sr, -cr};
  const int quad = ((int)k) & 3;
  *s = cycle[quad];
  *c = cycle[(quad + 1) & 3];
}

/* C = A B for row-major 3x3 matrices, each element (a0 b0 + a1 b1) + a2 b2 */
static void mul33(const float A[9], const float B[9], float C[9]) {
  for (int e = 0; e < 9; ++e) {
    const float* a = A + 3 * (e / 3);
    const float* b = B + (e % 3);
    C[e] = (a[0] * b[0] + a[1] * b[3]) + a[2] * b[6];
  }
}

/* A.2 step (3): R_inc = Rz(gamma) Ry(beta) Rx(alpha) from x6 = (alpha, beta, gamma, tx, ty, tz);
 * t <- R_inc t + t_inc; R <- R_inc R.  The sines and cosines are ora_sincos's, rounded to binary32. */
void ora_pose_update(float R[9], float t[3], const float x6[6]) {
  float sn[3], cs[3];
  for (int a = 0; a < 3; ++a) {
#ifdef ORA_LIT_D4
    sn[a] = sinf(x6[a]); /* Eigen's AngleAxisf: the C library's single-precision sine and cosine */
    cs[a] = cosf(x6[a]);
#else
    double sd, cd;
    ora_sincos((double)x6[a], &sd, &cd);
    sn[a] = (float)sd;
    cs[a] = (float)cd;
#endif
  }
  const float about_x[9] = {1.0f, 0.0f, 0.0f, 0.0f, cs[0], -sn[0], 0.0f, sn[0], cs[0]};
  const float about_y[9] = {cs[1], 0.0f, sn[1], 0.0f, 1.0f, 0.0f, -sn[1], 0.0f, cs[1]};
  const float about_z[9] = {cs[2], -sn[2], 0.0f, sn[2], cs[2], 0.0f, 0.0f, 0.0f, 1.0f};
  float zy[9], inc[9], turned[9];
  mul33(about_z, about_y, zy);
  mul33(zy, about_x, inc);
  float moved[3];
  for (int r = 0; r < 3; ++r) moved[r] = ((inc[3 * r] * t[0] + inc[3 * r + 1] * t[1]) + inc[3 * r + 2] * t[2]) + x6[3 + r];
  mul33(inc, R, turned);
  memcpy(t, moved, sizeof(moved));
  memcpy(R, turned, sizeof(turned));
}

#ifdef ORA_D3_STATS
/* study build (tools/d3_study.py): where the interpolated hit time of a zero crossing falls relative to the march step
 * [t, t + step] that found it, in steps: bins (-inf,-4) [-4,-2) [-2,-1) [-1,-0.5) [-0.5,0) [0,1] (1,1.5] (1.5,2] (2,3] (3,5] (5,inf) NaN */
static long long g_d3_bins[12];
static void ora_d3_note(float r) {
  int b;
  if (isnan(r)) b = 11;
  else if (r < -4.0f) b = 0;
  else if (r < -2.0f) b = 1;
  else if (r < -1.0f) b = 2;
  else if (r < -0.5f) b = 3;
  else if (r < 0.0f) b = 4;
  else if (r <= 1.0f) b = 5;
  else if (r <= 1.5f) b = 6;
  else if (r <= 2.0f) b = 7;
  else if (r <= 3.0f) b = 8;
  else if (r <= 5.0f) b = 9;
  else b = 10;
#ifdef _OPENMP
#pragma omp atomic
#endif
  g_d3_bins[b] += 1;
}
void ora_d3_stats(long long out[12], int reset) {
  for (int i = 0; i < 12; ++i) {
    out[i] = g_d3_bins[i];
    if (reset) g_d3_bins[i] = 0;
  }
}
#endif

/* ------------------------------------------------------------------------------------------------ */
/* A.6 raycast                                                                                        */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
  const int16_t* vol;
  int X, Y, Z;
  int zs0, nzs;
  float cell[3];
} rc_vol;

/* voxel index of a coordinate: floor(p / cell), saturated so that the int conversion is defined */
static inline int vox_of(float p, float cell) {
  const float q = floorf(p / cell);
  if (!(q >= 0.0f)) return -1; /* negative or NaN */
  if (q > 1.0e6f) return 1000000;
  return (int)q;
}

static inline int16_t rc_raw(const rc_vol* V, int x, int y, int z) {
  const int zz = z - V->zs0;
  if (zz < 0 || zz >= V->nzs) return 0; /* outside the stored slab: never reached when the halo is sized right */
  return V->vol[2 * (((size_t)zz * V->Y + y) * V->X + x)];
}
static inline float rc_tsdf(const rc_vol* V, int x, int y, int z) { return (float)rc_raw(V, x, y, z) / 32767.0f; }

static float rc_trilinear(const rc_vol* V, const float p[3]) {
  int g[3] = {vox_of(p[0], V->cell[0]), vox_of(p[1], V->cell[1]), vox_of(p[2], V->cell[2])};
  if (g[0] <= 0 || g[0] >= V->X - 1) return NANF;
  if (g[1] <= 0 || g[1] >= V->Y - 1) return NANF;
  if (g[2] <= 0 || g[2] >= V->Z - 1) return NANF;
  float a[3];
  for (int k = 0; k < 3; ++k) {
    const float vc = ((float)g[k] + 0.5f) * V->cell[k];
    if (p[k] < vc) g[k] -= 1;
    a[k] = (p[k] - ((float)g[k] + 0.5f) * V->cell[k]) / V->cell[k];
  }
  const float A = a[0], B = a[1], C = a[2];
  const int x = g[0], y = g[1], z = g[2];
  float res = rc_tsdf(V, x, y, z) * (1.0f - A) * (1.0f - B) * (1.0f - C);
  res = res + rc_tsdf(V, x, y, z + 1) * (1.0f - A) * (1.0f - B) * C;
  res = res + rc_tsdf(V, x, y + 1, z) * (1.0f - A) * B * (1.0f - C);
  res = res + rc_tsdf(V, x, y + 1, z + 1) * (1.0f - A) * B * C;
  res = res + rc_tsdf(V, x + 1, y, z) * A * (1.0f - B) * (1.0f - C);
  res = res + rc_tsdf(V, x + 1, y, z + 1) * A * (1.0f - B) * C;
  res = res + rc_tsdf(V, x + 1, y + 1, z) * A * B * (1.0f - C);
  res = res + rc_tsdf(V, x + 1, y + 1, z + 1) * A * B * C;
  return res;
}

void ora_raycast(const int16_t* vol, const int dims[3], const float size[3], float tau, int zs0, int nzs,
                 int zo0, int zo1, int W, int H, float fx, float fy, float cx, float cy,
                 const float R[9], const float t[3], float* vmap, float* nmap, int32_t* keys,
                 uint64_t* n_steps_out) {
  rc_vol V;
  V.vol = vol;
  V.X = dims[0];
  V.Y = dims[1];
  V.Z = dims[2];
  V.zs0 = zs0;
  V.nzs = nzs;
  V.cell[0] = size[0] / (float)dims[0];
  V.cell[1] = size[1] / (float)dims[1];
  V.cell[2] = size[2] / (float)dims[2];
  const size_t P = (size_t)W * H;
  const float time_step = tau * 0.8f;
  const float max_time = 3.0f * ((size[0] + size[1]) + size[2]);
  uint64_t n_steps = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : n_steps) schedule(dynamic, 4)
#endif
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const size_t i = (size_t)y * W + x;
      vmap[i] = vmap[P + i] = vmap[2 * P + i] = NANF;
      nmap[i] = nmap[P + i] = nmap[2 * P + i] = NANF;
      if (keys) keys[i] = ORA_KEY_NONE;
      const float rn[3] = {((float)x - cx) / fx, ((float)y - cy) / fy, 1.0f};
      float nx[3];
      rot3(R, rn, nx);
      const float inv = 1.0f / sqrtf(dot3(nx, nx));
      float dir[3] = {nx[0] * inv, nx[1] * inv, nx[2] * inv};
      for (int k = 0; k < 3; ++k)
        if (dir[k] == 0.0f) dir[k] = 1e-15f;
      float tmin[3], tmax[3];
      for (int k = 0; k < 3; ++k) {
        tmin[k] = ((dir[k] > 0.0f ? 0.0f : size[k]) - t[k]) / dir[k];
        tmax[k] = ((dir[k] > 0.0f ? size[k] : 0.0f) - t[k]) / dir[k];
      }
      float t_start = fmaxf(fmaxf(tmin[0], tmin[1]), tmin[2]);
      const float t_exit = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
      t_start = fmaxf(t_start, 0.0f);
      if (!(t_start < t_exit)) continue;
      float time_curr = t_start;
      int step = 0;
      for (; time_curr < max_time; time_curr = time_curr + time_step, ++step) {
        const float tn = time_curr + time_step;
        const float pn[3] = {t[0] + dir[0] * tn, t[1] + dir[1] * tn, t[2] + dir[2] * tn};
        const int gx = vox_of(pn[0], V.cell[0]), gy = vox_of(pn[1], V.cell[1]), gz = vox_of(pn[2], V.cell[2]);
        if (gx < 0 || gy < 0 || gz < 0 || gx >= V.X || gy >= V.Y || gz >= V.Z) break;
        if (gz < zo0 || gz >= zo1) continue; /* step owned by another slab */
        const float pc[3] = {t[0] + dir[0] * time_curr, t[1] + dir[1] * time_curr, t[2] + dir[2] * time_curr};
        int px = vox_of(pc[0], V.cell[0]), py = vox_of(pc[1], V.cell[1]), pz = vox_of(pc[2], V.cell[2]);
        px = px < 0 ? 0 : (px > V.X - 1 ? V.X - 1 : px);
        py = py < 0 ? 0 : (py > V.Y - 1 ? V.Y - 1 : py);
        pz = pz < 0 ? 0 : (pz > V.Z - 1 ? V.Z - 1 : pz);
        const int16_t raw_prev = rc_raw(&V, px, py, pz);
        const int16_t raw = rc_raw(&V, gx, gy, gz);
        ++n_steps;
        if (raw_prev < 0 && raw > 0) { /* back face */
          if (keys) keys[i] = (step << 1) | 1;
          break;
        }
        if (raw_prev > 0 && raw < 0) { /* zero crossing */
          int32_t key = (step << 1) | 1;
          const float Ftdt = rc_trilinear(&V, pn);
          if (!isnan(Ftdt)) {
            const float Ft = rc_trilinear(&V, pc);
            if (!isnan(Ft)) {
              const float Ts = time_curr - (time_step * Ft) / (Ftdt - Ft);
#ifdef ORA_D3_STATS
              ora_d3_note((Ts - time_curr) / time_step);
#endif
              /* deviation from A.6: reject an interpolated time outside [t - step, t + 2 step] -- two
               * steps round the far sample, which bounds the taps to the slab halo; 99.8 % of the crossings of the
               * scripted stream fall inside (tools/d3_study.py), A.6 as written keeps the extrapolated rest too. */
#ifdef ORA_LIT_D3
              if (!isnan(Ts)) { /* A.6 as written: whatever the interpolation gives (also an extrapolated time) */
#else
#ifndef ORA_D3_LO /* (study builds: tools/spec_vs_literal.py --d3-window) */
#define ORA_D3_LO 1.0f
#define ORA_D3_HI 2.0f
#endif
              if (Ts >= time_curr - ORA_D3_LO * time_step && Ts <= time_curr + ORA_D3_HI * time_step) {
#endif
                const float vtx[3] = {t[0] + dir[0] * Ts, t[1] + dir[1] * Ts, t[2] + dir[2] * Ts};
                vmap[i] = vtx[0];
                vmap[P + i] = vtx[1];
                vmap[2 * P + i] = vtx[2];
                key = (step << 1);
                const int qx = vox_of(pc[0], V.cell[0]), qy = vox_of(pc[1], V.cell[1]), qz = vox_of(pc[2], V.cell[2]);
                if (qx > 1 && qy > 1 && qz > 1 && qx < V.X - 2 && qy < V.Y - 2 && qz < V.Z - 2) {
                  float n[3];
                  for (int k = 0; k < 3; ++k) {
                    float p1[3] = {vtx[0], vtx[1], vtx[2]}, p2[3] = {vtx[0], vtx[1], vtx[2]};
                    p1[k] = p1[k] + V.cell[k];
                    p2[k] = p2[k] - V.cell[k];
                    n[k] = rc_trilinear(&V, p1) - rc_trilinear(&V, p2);
                  }
                  const float ninv = 1.0f / sqrtf(dot3(n, n));
                  nmap[i] = n[0] * ninv;
                  nmap[P + i] = n[1] * ninv;
                  nmap[2 * P + i] = n[2] * ninv;
                }
              }
            }
          }
          if (keys) keys[i] = key;
          break;
        }
      }
    }
  if (n_steps_out) *n_steps_out = n_steps;
}

/* ------------------------------------------------------------------------------------------------ */
/* A.7 extractCloud: zero crossings between +x/+y/+z neighbours, linear interpolation, voxel order    */
/* ------------------------------------------------------------------------------------------------ */
size_t ora_extract_cloud(const int16_t* vol, const int dims[3], const float size[3], float* xyz, size_t cap) {
  const int X = dims[0], Y = dims[1], Z = dims[2];
  const float cell[3] = {size[0] / (float)X, size[1] / (float)Y, size[2] / (float)Z};
  size_t n = 0;
  for (int z = 0; z < Z; ++z)
    for (int y = 0; y < Y; ++y)
      for (int x = 0; x < X; ++x) {
        const size_t i = ((size_t)z * Y + y) * X + x;
        const int16_t w = vol[2 * i + 1];
        const int16_t fr = vol[2 * i];
        if (w == 0 || fr == ORA_DIVISOR) continue;
        const float F = (float)fr / 32767.0f;
        const float Vc[3] = {((float)x + 0.5f) * cell[0], ((float)y + 0.5f) * cell[1], ((float)z + 0.5f) * cell[2]};
        const int g[3] = {x, y, z};
        const size_t stride[3] = {1, (size_t)X, (size_t)X * Y};
        for (int k = 0; k < 3; ++k) {
          if (g[k] + 1 >= dims[k]) continue;
          const size_t j = i + stride[k];
          const int16_t wn = vol[2 * j + 1];
          const int16_t fnr = vol[2 * j];
          if (wn == 0 || fnr == ORA_DIVISOR) continue;
          if (!((fr > 0 && fnr < 0) || (fr < 0 && fnr > 0))) continue;
          const float Fn = (float)fnr / 32767.0f;
          float p[3] = {Vc[0], Vc[1], Vc[2]};
          const float Vn = Vc[k] + cell[k];
          const float d_inv = 1.0f / (fabsf(F) + fabsf(Fn));
          p[k] = (Vc[k] * fabsf(Fn) + Vn * fabsf(F)) * d_inv;
          if (n < cap) {
            xyz[3 * n] = p[0];
            xyz[3 * n + 1] = p[1];
            xyz[3 * n + 2] = p[2];
          }
          ++n;
        }
      }
  return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* A.8 extractMesh: marching tetrahedra (this build's own specification -- PCL's marching-cubes tables are not  */
/* in /root/reference).  Cube corner i at offset (i&1, i>>1&1, i>>2&1); six Kuhn tetrahedra round the diagonal   */
/* 0-7; cube valid when all 8 weights != 0; inside = TSDF < 0; edge vertex from the LOWER corner index a:        */
/* P = Pa + (Fa/(Fa-Fb)) (Pb-Pa); triangles wind so that the normal points to free space; voxel order, then     */
/* tetrahedron order, then triangle order.                                                                      */
/* ------------------------------------------------------------------------------------------------ */
static const int ora_tet[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7}, {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};

/* triangles (pairs of cube corners per triangle corner) of tetrahedron t under inside-mask m */
static int ora_tet_case(int t, int m, int e[2][3][2]) {
  int in[4], out[4], ni = 0, no = 0, nt = 0;
  for (int v = 0; v < 4; ++v) {
    if ((m >> v) & 1) in[ni++] = ora_tet[t][v];
    else out[no++] = ora_tet[t][v];
  }
  if (ni == 1 || ni == 3) {
    const int apex = ni == 1 ? in[0] : out[0];
    const int* base = ni == 1 ? out : in;
    for (int q = 0; q < 3; ++q) { e[0][q][0] = apex; e[0][q][1] = base[q]; }
    nt = 1;
  } else if (ni == 2) {
    const int quad[4][2] = {{in[0], out[0]}, {in[0], out[1]}, {in[1], out[1]}, {in[1], out[0]}};
    const int pick[2][3] = {{0, 1, 2}, {0, 2, 3}};
    for (int k = 0; k < 2; ++k)
      for (int q = 0; q < 3; ++q) { e[k][q][0] = quad[pick[k][q]][0]; e[k][q][1] = quad[pick[k][q]][1]; }
    nt = 2;
  }
  double ci[3] = {0, 0, 0}, co[3] = {0, 0, 0};
  for (int v = 0; v < ni; ++v)
    for (int a = 0; a < 3; ++a) ci[a] += ((in[v] >> a) & 1) / (double)ni;
  for (int v = 0; v < no; ++v)
    for (int a = 0; a < 3; ++a) co[a] += ((out[v] >> a) & 1) / (double)no;
  for (int k = 0; k < nt; ++k) {
    double p[3][3];
    for (int q = 0; q < 3; ++q)
      for (int a = 0; a < 3; ++a) p[q][a] = 0.5 * (((e[k][q][0] >> a) & 1) + ((e[k][q][1] >> a) & 1));
    const double u[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]};
    const double w[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
    const double n[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
    if (n[0] * (co[0] - ci[0]) + n[1] * (co[1] - ci[1]) + n[2] * (co[2] - ci[2]) < 0)
      for (int a = 0; a < 2; ++a) { const int tmp = e[k][1][a]; e[k][1][a] = e[k][2][a]; e[k][2][a] = tmp; }
  }
  return nt;
}

size_t ora_extract_mesh(const int16_t* vol, const int dims[3], const float size[3], float* tri, size_t cap) {
  const int X = dims[0], Y = dims[1], Z = dims[2];
  const float cell[3] = {size[0] / (float)X, size[1] / (float)Y, size[2] / (float)Z};
  size_t n = 0;
  for (int z = 0; z + 1 < Z; ++z)
    for (int y = 0; y + 1 < Y; ++y)
      for (int x = 0; x + 1 < X; ++x) {
        int16_t f[8];
        int ok = 1, m8 = 0;
        for (int c = 0; c < 8; ++c) {
          const size_t i = ((size_t)(z + (c >> 2)) * Y + (y + ((c >> 1) & 1))) * X + (x + (c & 1));
          f[c] = vol[2 * i];
          if (vol[2 * i + 1] == 0) ok = 0;
          if (f[c] < 0) m8 |= 1 << c;
        }
        if (!ok || m8 == 0 || m8 == 255) continue;
        for (int t = 0; t < 6; ++t) {
          int m = 0;
          for (int v = 0; v < 4; ++v) m |= ((m8 >> ora_tet[t][v]) & 1) << v;
          int e[2][3][2];
          const int nt = ora_tet_case(t, m, e);
          for (int k = 0; k < nt; ++k) {
            if (n < cap)
              for (int q = 0; q < 3; ++q) {
                int a = e[k][q][0], b = e[k][q][1];
                if (a > b) { const int tmp = a; a = b; b = tmp; }
                const float Fa = (float)f[a] / 32767.0f, Fb = (float)f[b] / 32767.0f;
                const float w = Fa / (Fa - Fb);
                const int ga[3] = {x + (a & 1), y + ((a >> 1) & 1), z + (a >> 2)};
                const int gb[3] = {x + (b & 1), y + ((b >> 1) & 1), z + (b >> 2)};
                for (int ax = 0; ax < 3; ++ax) {
                  const float pa = ((float)ga[ax] + 0.5f) * cell[ax];
                  const float pb = ((float)gb[ax] + 0.5f) * cell[ax];
                  tri[9 * n + 3 * q + ax] = pa + w * (pb - pa);
                }
              }
            ++n;
          }
        }
      }
  return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* A.8b extractMesh, marching cubes (the form upstream's .ply export has; PCL's table is not in /root/reference, so   */
/* this build generates one and states the rule): same cube validity, inside test, edge vertices and voxel order as  */
/* above.  Per cube: every cut edge is a node; on each of the six faces the cut edges are paired -- 2 cut edges: with  */
/* each other; 4 cut edges: each with its neighbour ACROSS AN INSIDE CORNER -- so every node has two partners and the   */
/* nodes fall into closed loops.  Loops are taken in order of their smallest edge code (low corner | high corner << 4), */
/* start there, run so that the Newell normal of the edge midpoints has a positive product with the sum of the          */
/* (outside end - inside end) vectors of its edges, and are fanned (e0, e_i, e_i+1) from that first edge -- or from the   */
/* next one along the loop none of whose fan diagonals lies in a face of the cube (one of the first three does).        */
/* (Written independently of the product's hsk_build_cube_table: there the faces are walked as corner cycles, here a   */
/* node's partners are found by flipping axis bits of its end corners.)                                                */
/* ------------------------------------------------------------------------------------------------ */
static int ora_mc_is_in(int m, int c) { return (m >> c) & 1; }

/* partner of cut edge (a, b) [b = a ^ (1 << ax)] on the face spanned by axes ax and w, on the side both corners lie on */
static int ora_mc_partner(int m, int a, int b, int w) {
  const int a2 = a ^ (1 << w), b2 = b ^ (1 << w); /* the face's other two corners, next to a and b */
  const int ia = ora_mc_is_in(m, a), ia2 = ora_mc_is_in(m, a2), ib2 = ora_mc_is_in(m, b2);
  /* the face's edges besides (a, b): (a, a2), (b, b2), (a2, b2) */
  const int cut_a = ia != ia2, cut_b = (!ia) != ib2, cut_far = ia2 != ib2; /* (b is the opposite of a: the edge is cut) */
  int lo, hi;
  if (cut_a && cut_b && cut_far) { /* all four cut: across the inside corner of this edge */
    if (ia) { lo = a; hi = a2; } else { lo = b; hi = b2; }
  } else if (cut_a) { lo = a; hi = a2; }
  else if (cut_b) { lo = b; hi = b2; }
  else { lo = a2; hi = b2; }
  if (lo > hi) { const int t = lo; lo = hi; hi = t; }
  return lo | (hi << 4);
}

/* triangles of inside-mask m as edge codes; returns their number (at most 5) */
static int ora_mc_case(int m, int tri[5][3]) {
  int nodes[12], n_nodes = 0, done[12] = {0}, nt = 0;
  for (int a = 0; a < 8; ++a)
    for (int ax = 0; ax < 3; ++ax) {
      const int b = a | (1 << ax);
      if (b == a) continue; /* a has the bit: the edge is listed from its lower corner */
      if (ora_mc_is_in(m, a) != ora_mc_is_in(m, b)) nodes[n_nodes++] = a | (b << 4);
    }
  /* ascending edge code */
  for (int i = 0; i < n_nodes; ++i)
    for (int j = i + 1; j < n_nodes; ++j)
      if (nodes[j] < nodes[i]) { const int t = nodes[i]; nodes[i] = nodes[j]; nodes[j] = t; }
  for (int s0 = 0; s0 < n_nodes; ++s0) {
    if (done[s0]) continue;
    /* walk the cycle: leave every node by the partner it was not entered from (the first by its w1 partner: the winding
     * rule below fixes the direction afterwards, so either way round is the same loop) */
    int loop[12], len = 0, cur = nodes[s0], from = -1;
    do {
      loop[len++] = cur;
      for (int i = 0; i < n_nodes; ++i)
        if (nodes[i] == cur) done[i] = 1;
      const int a = cur & 15, b = cur >> 4;
      int ax = 0;
      while (((a ^ b) >> ax) != 1) ++ax;
      const int p1 = ora_mc_partner(m, a, b, (ax + 1) % 3), p2 = ora_mc_partner(m, a, b, (ax + 2) % 3);
      const int next = p1 != from ? p1 : p2;
      from = cur;
      cur = next;
    } while (cur != loop[0] && len < 12);
    double mid[12][3], nrm[3] = {0, 0, 0}, dir[3] = {0, 0, 0};
    for (int i = 0; i < len; ++i) {
      const int a = loop[i] & 15, b = loop[i] >> 4;
      for (int k = 0; k < 3; ++k) {
        const double pa = (a >> k) & 1, pb = (b >> k) & 1;
        mid[i][k] = (pa + pb) / 2;
        dir[k] += ora_mc_is_in(m, a) ? pb - pa : pa - pb;
      }
    }
    for (int i = 0; i < len; ++i) {
      const int j = i + 1 == len ? 0 : i + 1;
      nrm[0] += mid[i][1] * mid[j][2] - mid[i][2] * mid[j][1];
      nrm[1] += mid[i][2] * mid[j][0] - mid[i][0] * mid[j][2];
      nrm[2] += mid[i][0] * mid[j][1] - mid[i][1] * mid[j][0];
    }
    /* the loop as it is wound: from its first edge onwards, or backwards */
    int w[12];
    const int flip = nrm[0] * dir[0] + nrm[1] * dir[1] + nrm[2] * dir[2] < 0;
    for (int i = 0; i < len; ++i) w[i] = loop[flip ? (len - i) % len : i];
    /* fan origin: the first position whose diagonals all leave the cube's faces (two edges lie in one face when some
     * coordinate is the same at all four of their ends) */
    int o0 = 0;
    for (int o = 0; o < len; ++o) {
      int bad = 0;
      for (int k = 2; k + 1 < len; ++k) {
        const int e = w[o], f = w[(o + k) % len];
        const int all_and = (e & 15) & (e >> 4) & (f & 15) & (f >> 4), all_or = (e & 15) | (e >> 4) | (f & 15) | (f >> 4);
        if ((all_and & 7) != 0 || (all_or & 7) != 7) bad = 1; /* a bit set in all four corners, or clear in all four */
      }
      if (!bad) { o0 = o; break; }
    }
    for (int i = 1; i + 1 < len; ++i) {
      if (nt < 5) { tri[nt][0] = w[o0]; tri[nt][1] = w[(o0 + i) % len]; tri[nt][2] = w[(o0 + i + 1) % len]; }
      ++nt;
    }
  }
  return nt;
}

size_t ora_extract_mesh_mc(const int16_t* vol, const int dims[3], const float size[3], float* tri, size_t cap) {
  const int X = dims[0], Y = dims[1], Z = dims[2];
  const float cell[3] = {size[0] / (float)X, size[1] / (float)Y, size[2] / (float)Z};
  static int table_n[256], table[256][5][3], have = 0;
  if (!have) {
    for (int m = 0; m < 256; ++m) table_n[m] = ora_mc_case(m, table[m]);
    have = 1;
  }
  size_t n = 0;
  for (int z = 0; z + 1 < Z; ++z)
    for (int y = 0; y + 1 < Y; ++y)
      for (int x = 0; x + 1 < X; ++x) {
        int16_t f[8];
        int ok = 1, m8 = 0;
        for (int c = 0; c < 8; ++c) {
          const size_t i = ((size_t)(z + (c >> 2)) * Y + (y + ((c >> 1) & 1))) * X + (x + (c & 1));
          f[c] = vol[2 * i];
          if (vol[2 * i + 1] == 0) ok = 0;
          if (f[c] < 0) m8 |= 1 << c;
        }
        if (!ok || m8 == 0 || m8 == 255) continue;
        for (int k = 0; k < table_n[m8]; ++k) {
          if (n < cap)
            for (int q = 0; q < 3; ++q) {
              const int a = table[m8][k][q] & 15, b = table[m8][k][q] >> 4;
              const float Fa = (float)f[a] / 32767.0f, Fb = (float)f[b] / 32767.0f;
              const float w = Fa / (Fa - Fb);
              const int ga[3] = {x + (a & 1), y + ((a >> 1) & 1), z + (a >> 2)};
              const int gb[3] = {x + (b & 1), y + ((b >> 1) & 1), z + (b >> 2)};
              for (int ax = 0; ax < 3; ++ax) {
                const float pa = ((float)ga[ax] + 0.5f) * cell[ax];
                const float pb = ((float)gb[ax] + 0.5f) * cell[ax];
                tri[9 * n + 3 * q + ax] = pa + w * (pb - pa);
              }
            }
          ++n;
        }
      }
  return n;
}
/* the table itself, for the tests: ntri[256], codes[256][5][3] */
void ora_mc_table(int* ntri, int* codes) {
  for (int m = 0; m < 256; ++m) {
    int t[5][3] = {{0}};
    ntri[m] = ora_mc_case(m, t);
    for (int k = 0; k < 5; ++k)
      for (int q = 0; q < 3; ++q) codes[(m * 5 + k) * 3 + q] = k < ntri[m] ? t[k][q] : 0;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* A.2 tracker state machine                                                                          */
/* ------------------------------------------------------------------------------------------------ */
struct ora_tracker {
  ora_config c;
  float tau;
  int16_t* vol;
  int frame;
  float R[9], t[3];
  uint16_t* dep[ORA_LEVELS];
  float *vcur[ORA_LEVELS], *ncur[ORA_LEVELS], *vmod[ORA_LEVELS], *nmod[ORA_LEVELS];
  float* scaled;
  double secs[4];
  uint64_t last_vupd;
};

ora_tracker* ora_tracker_create(const ora_config* c) {
  ora_tracker* k = (ora_tracker*)calloc(1, sizeof(*k));
  k->c = *c;
  k->tau = ora_tau(c);
  const size_t nvox = (size_t)c->vol[0] * c->vol[1] * c->vol[2];
  k->vol = (int16_t*)calloc(nvox * 2, sizeof(int16_t));
  for (int l = 0; l < ORA_LEVELS; ++l) {
    const size_t P = (size_t)(c->W >> l) * (c->H >> l);
    k->dep[l] = (uint16_t*)calloc(P, sizeof(uint16_t));
    k->vcur[l] = (float*)calloc(3 * P, sizeof(float));
    k->ncur[l] = (float*)calloc(3 * P, sizeof(float));
    k->vmod[l] = (float*)calloc(3 * P, sizeof(float));
    k->nmod[l] = (float*)calloc(3 * P, sizeof(float));
  }
  k->scaled = (float*)calloc((size_t)c->W * c->H, sizeof(float));
  ora_tracker_reset(k);
  return k;
}

void ora_tracker_destroy(ora_tracker* k) {
  if (!k) return;
  free(k->vol);
  for (int l = 0; l < ORA_LEVELS; ++l) {
    free(k->dep[l]);
    free(k->vcur[l]);
    free(k->ncur[l]);
    free(k->vmod[l]);
    free(k->nmod[l]);
  }
  free(k->scaled);
  free(k);
}

void ora_tracker_reset(ora_tracker* k) {
  const size_t nvox = (size_t)k->c.vol[0] * k->c.vol[1] * k->c.vol[2];
  memset(k->vol, 0, nvox * 2 * sizeof(int16_t));
  memcpy(k->R, k->c.init_R, sizeof(k->R));
  memcpy(k->t, k->c.init_t, sizeof(k->t));
  k->frame = 0;
}

static void pose_to16(const float R[9], const float t[3], float m[16]) {
  for (int i = 0; i < 3; ++i) {
    m[i * 4] = R[i * 3];
    m[i * 4 + 1] = R[i * 3 + 1];
    m[i * 4 + 2] = R[i * 3 + 2];
    m[i * 4 + 3] = t[i];
  }
  m[12] = m[13] = m[14] = 0.0f;
  m[15] = 1.0f;
}

/* integration gate (A.2 step 5); only evaluated when move_thresh > 0.  M = R^-1 R_prev = R^T R_prev; the norm of its
 * Rodrigues vector is the rotation angle acos((trace M - 1) / 2), clamped into [-1, 1]; the frame integrates iff
 * (angle + |t - t_prev|) / 2 >= threshold.  Only the diagonal of M is formed:
 * M_ii = (R_0i Rp_0i + R_1i Rp_1i) + R_2i Rp_2i, trace = (M_00 + M_11) + M_22; acosf is the C library's. */
static int gate_passes(const float R[9], const float t[3], const float Rp[9], const float tp[3], float thr) {
  if (!(thr > 0.0f)) return 1;
  float diag[3];
  for (int i = 0; i < 3; ++i) diag[i] = (R[i] * Rp[i] + R[3 + i] * Rp[3 + i]) + R[6 + i] * Rp[6 + i];
  const float trace = (diag[0] + diag[1]) + diag[2];
  float cosine = (trace - 1.0f) / 2.0f;
  if (cosine > 1.0f) cosine = 1.0f;
  if (cosine < -1.0f) cosine = -1.0f;
  const float angle = acosf(cosine);
  const float d[3] = {t[0] - tp[0], t[1] - tp[1], t[2] - tp[2]};
  const float moved = sqrtf(dot3(d, d));
  return (angle + moved) / 2.0f >= thr;
}

int ora_tracker_process(ora_tracker* k, const uint16_t* depth, float pose16[16]) {
  const ora_config* c = &k->c;
  const int W = c->W, H = c->H;
  double t0 = now_s();
  /* (1) bilateral -> pyramid -> vertex / normal maps */
  ora_bilateral(depth, W, H, k->dep[0]);
  for (int l = 1; l < ORA_LEVELS; ++l) ora_pyrdown(k->dep[l - 1], W >> (l - 1), H >> (l - 1), k->dep[l]);
  for (int l = 0; l < ORA_LEVELS; ++l) {
    const float s = (float)(1 << l);
    ora_vmap(k->dep[l], W >> l, H >> l, c->fx / s, c->fy / s, c->cx / s, c->cy / s, k->vcur[l]);
    ora_nmap(k->vcur[l], W >> l, H >> l, k->ncur[l]);
  }
  ora_scale_depth(depth, W, H, c->fx, c->fy, c->cx, c->cy, k->scaled);
  double t1 = now_s();
  k->secs[0] += t1 - t0;

  if (k->frame == 0) { /* (2) first frame */
    k->last_vupd = ora_integrate(k->vol, c->vol, c->size, k->tau, 0, c->vol[2], k->scaled, W, H, c->fx, c->fy, c->cx,
                                 c->cy, k->R, k->t);
    for (int l = 0; l < ORA_LEVELS; ++l)
      ora_transform_maps(k->vcur[l], k->ncur[l], W >> l, H >> l, k->R, k->t, k->vmod[l], k->nmod[l]);
    k->secs[2] += now_s() - t1;
    k->frame = 1;
    pose_to16(k->R, k->t, pose16);
    return 0;
  }

  /* (3) ICP, coarse to fine */
  float R[9], t[3];
  memcpy(R, k->R, sizeof(R));
  memcpy(t, k->t, sizeof(t));
  for (int l = ORA_LEVELS - 1; l >= 0; --l) {
    const float s = (float)(1 << l);
    for (int it = 0; it < c->icp_iters[l]; ++it) {
      double sums[27];
      float x6[6];
      ora_icp_accumulate(k->vcur[l], k->ncur[l], k->vmod[l], k->nmod[l], W >> l, H >> l, c->fx / s, c->fy / s,
                         c->cx / s, c->cy / s, R, t, k->R, k->t, c->dist_thresh, c->angle_thresh, 0, H >> l, sums);
      if (!ora_icp_solve(sums, x6)) {
        k->secs[1] += now_s() - t1;
        ora_tracker_reset(k);
        pose_to16(k->R, k->t, pose16);
        return 0;
      }
      ora_pose_update(R, t, x6);
    }
  }
  double t2 = now_s();
  k->secs[1] += t2 - t1;
  /* (4)-(6) store pose, gate, integrate raw depth */
  const int do_integrate = gate_passes(R, t, k->R, k->t, c->move_thresh);
  memcpy(k->R, R, sizeof(R));
  memcpy(k->t, t, sizeof(t));
  if (do_integrate)
    k->last_vupd = ora_integrate(k->vol, c->vol, c->size, k->tau, 0, c->vol[2], k->scaled, W, H, c->fx, c->fy, c->cx,
                                 c->cy, k->R, k->t);
  double t3 = now_s();
  k->secs[2] += t3 - t2;
  /* (7) raycast -> model level 0, resize -> levels 1, 2 */
  ora_raycast(k->vol, c->vol, c->size, k->tau, 0, c->vol[2], 0, c->vol[2], W, H, c->fx, c->fy, c->cx, c->cy, k->R,
              k->t, k->vmod[0], k->nmod[0], NULL, NULL);
  for (int l = 1; l < ORA_LEVELS; ++l) {
    ora_resize_vmap(k->vmod[l - 1], W >> (l - 1), H >> (l - 1), k->vmod[l]);
    ora_resize_nmap(k->nmod[l - 1], W >> (l - 1), H >> (l - 1), k->nmod[l]);
  }
  k->secs[3] += now_s() - t3;
  k->frame += 1;
  pose_to16(k->R, k->t, pose16);
  return 1;
}

const int16_t* ora_tracker_volume(const ora_tracker* k) { return k->vol; }
const float* ora_tracker_model_vmap(const ora_tracker* k, int level) { return k->vmod[level]; }
const float* ora_tracker_model_nmap(const ora_tracker* k, int level) { return k->nmod[level]; }
void ora_tracker_stage_seconds(const ora_tracker* k, double out4[4]) { memcpy(out4, k->secs, sizeof(k->secs)); }
uint64_t ora_tracker_last_vupd(const ora_tracker* k) { return k->last_vupd; }
