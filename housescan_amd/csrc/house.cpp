// house.cpp -- host-side room stitching behind include/hshouse.h (BASELINE configs[0]).
//
// A restatement of what HouseScan does with KinFu's products, written from the behaviour of the reference's
// functions (cited per function), not from their text.  Room geometry is binary32 like the reference's
// Data.Vect.Float; the cuboid fit and the least squares are binary64.  Vectors are ROW vectors and 3x3 / 4x4
// matrices act from the right (p' = p M), the convention of the reference's `vect` package; the C ABI exports
// the transposed, left-multiplicative form.
#include "../../include/hshouse.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <sys/stat.h>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_err;

struct V3 {
  float x, y, z;
};
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float norm(V3 a) { return std::sqrt(dot(a, a)); }
static inline float comp(V3 a, int axis) { return axis == 0 ? a.x : axis == 1 ? a.y : a.z; }
static inline V3 along(float d, int axis) { return {axis == 0 ? d : 0.f, axis == 1 ? d : 0.f, axis == 2 ? d : 0.f}; }
static inline V3 normalized(V3 a) { return (1.0f / norm(a)) * a; }

struct M3 {  // rows; p' = p M
  float m[3][3];
};
static inline V3 rmul(V3 p, const M3& M) {
  return {p.x * M.m[0][0] + p.y * M.m[1][0] + p.z * M.m[2][0], p.x * M.m[0][1] + p.y * M.m[1][1] + p.z * M.m[2][1],
          p.x * M.m[0][2] + p.y * M.m[1][2] + p.z * M.m[2][2]};
}
static inline V3 rotate_around(V3 c, const M3& R, V3 p) { return rmul(p - c, R) + c; }  // Main.hs:1581-1582

struct M4 {
  float m[4][4];
};
static M4 identity4() {
  M4 r{};
  for (int i = 0; i < 4; ++i) r.m[i][i] = 1.f;
  return r;
}
static M4 mul4(const M4& a, const M4& b) {
  M4 r{};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
// apply "then translate by v" to a right-multiplicative affine map: only the last row moves
static void translate_after(M4& P, V3 v) {
  P.m[3][0] += v.x;
  P.m[3][1] += v.y;
  P.m[3][2] += v.z;
}
static M4 embed(const M3& R) {
  M4 r = identity4();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = R.m[i][j];
  return r;
}

// right-multiplicative rotation by `a` about the unit axis v: (1-c) v v^T + [[c, s z, -s y], [-s z, c, s x], [s y, -s x, c]]
static M3 rot_axis_angle(V3 v, float a) {
  const float c = std::cos(a), s = std::sin(a), k = 1.f - c;
  M3 r;
  const float u[3] = {v.x, v.y, v.z};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = k * (u[i] * u[j]);
  r.m[0][0] += c;
  r.m[0][1] += s * v.z;
  r.m[0][2] += -s * v.y;
  r.m[1][0] += -s * v.z;
  r.m[1][1] += c;
  r.m[1][2] += s * v.x;
  r.m[2][0] += s * v.y;
  r.m[2][1] += -s * v.x;
  r.m[2][2] += c;
  return r;
}

struct PlaneEq {
  V3 n;
  float d;
};
static PlaneEq mk_plane_eq(V3 abc, float d) {  // Main.hs:1360-1361
  const float l = norm(abc);
  return {(1.0f / l) * abc, d / l};
}
static float signed_distance(const PlaneEq& e, V3 p) { return dot(e.n, p) - e.d; }  // Main.hs:1371-1372

static PlaneEq rotate_eq_around(V3 c, const M3& R, const PlaneEq& e) {  // Main.hs:1571-1578
  const V3 n2 = rmul(e.n, R);
  const V3 foot = e.d * e.n;
  const V3 foot2 = rotate_around(c, R, foot);
  return mk_plane_eq(n2, dot(foot2, n2));
}
static PlaneEq translate_eq(V3 off, const PlaneEq& e) {  // Main.hs:1681-1688
  const V3 foot = (e.d * e.n) + off;
  return mk_plane_eq(e.n, dot(foot, e.n));
}

static V3 point_mean(const std::vector<V3>& ps) {  // Main.hs:1597-1602: sequential binary32 sum, then * (1/n)
  V3 s{0.f, 0.f, 0.f};
  for (const V3& p : ps) s = s + p;
  return (1.0f / (float)ps.size()) * s;
}

struct Plane {
  uint32_t id;
  PlaneEq eq;
  std::vector<V3> bounds;
};
struct Room {
  uint32_t id;
  std::vector<Plane> planes;
  std::vector<V3> cloud;
  std::vector<std::pair<uint32_t, V3>> corners, suggested;
  M4 proj;  // right-multiplicative
  std::string name;
};
struct WallLink {
  int axis, relation;
  float thickness;
  uint32_t p1, p2;
};

// ---------------------------------------------------------------------------------------------------------------
// dense helpers (binary64)
// ---------------------------------------------------------------------------------------------------------------
// Gaussian elimination with partial pivoting; false when a pivot vanishes relative to the matrix scale
// (stands in for the missing HmatrixUtils.safeLinearSolve: Nothing on a singular system).
static bool solve_square(int n, std::vector<double> A, std::vector<double> b, std::vector<double>& x) {
  double scale = 0.0;
  for (double v : A) scale = std::max(scale, std::fabs(v));
  if (!(scale > 0.0)) return false;
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(A[i * n + k]) > std::fabs(A[p * n + k])) p = i;
    if (!(std::fabs(A[p * n + k]) > 1e-13 * scale)) return false;
    if (p != k) {
      for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[p * n + j]);
      std::swap(b[k], b[p]);
    }
    for (int i = k + 1; i < n; ++i) {
      const double f = A[i * n + k] / A[k * n + k];
      for (int j = k; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
      b[i] -= f * b[k];
    }
  }
  x.assign(n, 0.0);
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * x[j];
    x[i] = s / A[i * n + i];
  }
  for (double v : x)
    if (!std::isfinite(v)) return false;
  return true;
}

// least squares min |A x - b| for an m x n matrix (m >= n) by Householder QR; false when rank deficient
// (stands in for HmatrixUtils.safeLinearSolveLS over LAPACK dgels).
static bool solve_least_squares(int m, int n, std::vector<double> A, std::vector<double> b, std::vector<double>& x) {
  if (m < n || n == 0) return false;
  double scale = 0.0;
  for (double v : A) scale = std::max(scale, std::fabs(v));
  for (int k = 0; k < n; ++k) {
    double nrm = 0.0;
    for (int i = k; i < m; ++i) nrm += A[i * n + k] * A[i * n + k];
    nrm = std::sqrt(nrm);
    if (!(nrm > 1e-13 * scale)) return false;
    const double alpha = A[k * n + k] > 0 ? -nrm : nrm;
    std::vector<double> v(m - k);
    for (int i = k; i < m; ++i) v[i - k] = A[i * n + k];
    v[0] -= alpha;
    double vn = 0.0;
    for (double t : v) vn += t * t;
    if (vn > 0.0) {
      for (int j = k; j < n; ++j) {
        double s = 0.0;
        for (int i = k; i < m; ++i) s += v[i - k] * A[i * n + j];
        s = 2.0 * s / vn;
        for (int i = k; i < m; ++i) A[i * n + j] -= s * v[i - k];
      }
      double s = 0.0;
      for (int i = k; i < m; ++i) s += v[i - k] * b[i];
      s = 2.0 * s / vn;
      for (int i = k; i < m; ++i) b[i] -= s * v[i - k];
    }
  }
  x.assign(n, 0.0);
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * x[j];
    x[i] = s / A[i * n + i];
  }
  return true;
}

// symmetric 3x3 eigen decomposition (cyclic Jacobi); eigenvalues descending like hmatrix eigSH
static void eig_sym3(const double S[3][3], double w[3], double V[3][3]) {
  double a[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a[i][j] = S[i][j];
      V[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  std::sort(idx, idx + 3, [&](int i, int j) { return a[i][i] > a[j][j]; });
  double Vs[3][3];
  for (int c = 0; c < 3; ++c) {
    w[c] = a[idx[c]][idx[c]];
    for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][idx[c]];
  }
  std::memcpy(V, Vs, sizeof(Vs));
}

// ---------------------------------------------------------------------------------------------------------------
// Nelder-Mead: GSL multimin nmsimplex2 under hmatrix-gsl's `minimize` loop
// ---------------------------------------------------------------------------------------------------------------
struct Simplex {
  int n, P;                 // dimension, corners (n + 1)
  std::vector<double> x;    // P rows of n
  std::vector<double> y;    // value per corner
  std::vector<double> center;
  double S2;                // mean squared corner distance from the centre
  hsh_objective f;
  void* user;

  double* row(int i) { return &x[(size_t)i * n]; }
  double eval(const double* p) { return f(p, n, user); }

  void recompute_center() {
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int i = 0; i < P; ++i) s += row(i)[j];
      center[j] = s / P;
    }
  }
  double recompute_size() {
    double ss = 0.0;
    for (int i = 0; i < P; ++i) {
      double t = 0.0;
      for (int j = 0; j < n; ++j) {
        const double dlt = row(i)[j] - center[j];
        t += dlt * dlt;
      }
      const double r = std::sqrt(t);
      ss += r * r;
    }
    S2 = ss / P;
    return std::sqrt(S2);
  }
  // corner moved along the line through the centroid of the OTHER corners; coeff < 0 mirrors
  double trial(double coeff, int corner, std::vector<double>& xc) {
    const double alpha = (1.0 - coeff) * P / (P - 1.0);
    const double beta = (P * coeff - 1.0) / (P - 1.0);
    for (int j = 0; j < n; ++j) xc[j] = alpha * center[j] + beta * row(corner)[j];
    return eval(xc.data());
  }
  void replace(int i, const std::vector<double>& xn, double val) {
    double d2 = 0.0, xmcd = 0.0;
    for (int j = 0; j < n; ++j) {
      const double delta = xn[j] - row(i)[j];
      d2 += delta * delta;
      xmcd += (row(i)[j] - center[j]) * delta;
    }
    const double d = std::sqrt(d2);
    S2 += (2.0 / P) * xmcd + ((P - 1.0) / P) * (d * d / P);
    const double a = 1.0 / P;
    for (int j = 0; j < n; ++j) center[j] = (center[j] - a * row(i)[j]) + a * xn[j];
    for (int j = 0; j < n; ++j) row(i)[j] = xn[j];
    y[i] = val;
  }
  void shrink_towards(int best) {
    for (int i = 0; i < P; ++i) {
      if (i == best) continue;
      for (int j = 0; j < n; ++j) row(i)[j] = 0.5 * (row(i)[j] + row(best)[j]);
      y[i] = eval(row(i));
    }
    recompute_center();
    recompute_size();
  }
};

static int nm_minimize(hsh_objective f, void* user, int n, const double* start, const double* steps, double eps, int maxit,
                       double* x_out, double* f_out, int* iterations) {
  Simplex s;
  s.n = n;
  s.P = n + 1;
  s.x.assign((size_t)s.P * n, 0.0);
  s.y.assign(s.P, 0.0);
  s.center.assign(n, 0.0);
  s.f = f;
  s.user = user;
  for (int i = 0; i < s.P; ++i) {
    for (int j = 0; j < n; ++j) s.row(i)[j] = start[j];
    if (i > 0) s.row(i)[i - 1] = start[i - 1] + steps[i - 1];
    s.y[i] = s.eval(s.row(i));
  }
  s.recompute_center();
  double size = s.recompute_size();
  std::vector<double> xc(n), xc2(n);
  int it = 0;
  int best = 0;
  while (it < maxit) {
    ++it;
    int hi = 0, lo = 0;
    double dhi = s.y[0], dlo = s.y[0], dshi = s.y[1];
    for (int i = 1; i < s.P; ++i) {
      const double v = s.y[i];
      if (v < dlo) {
        dlo = v;
        lo = i;
      } else if (v > dhi) {
        dshi = dhi;
        dhi = v;
        hi = i;
      } else if (v > dshi) {
        dshi = v;
      }
    }
    const double val = s.trial(-1.0, hi, xc);
    if (std::isfinite(val) && val < dlo) {
      const double val2 = s.trial(-2.0, hi, xc2);
      if (std::isfinite(val2) && val2 < dlo)
        s.replace(hi, xc2, val2);
      else
        s.replace(hi, xc, val);
    } else if (!std::isfinite(val) || val > dshi) {
      if (std::isfinite(val) && val <= dhi) s.replace(hi, xc, val);
      const double val2 = s.trial(0.5, hi, xc2);
      if (std::isfinite(val2) && val2 <= dhi)
        s.replace(hi, xc2, val2);
      else
        s.shrink_towards(lo);
    } else {
      s.replace(hi, xc, val);
    }
    best = (int)(std::min_element(s.y.begin(), s.y.end()) - s.y.begin());
    size = s.S2 > 0 ? std::sqrt(s.S2) : s.recompute_size();
    if (size < eps) break;
  }
  if (it == 0) best = (int)(std::min_element(s.y.begin(), s.y.end()) - s.y.begin());
  for (int j = 0; j < n; ++j) x_out[j] = s.row(best)[j];
  if (f_out) *f_out = s.y[best];
  if (iterations) *iterations = it;
  return HSH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// cuboid model (binary64)
// ---------------------------------------------------------------------------------------------------------------
struct D3 {
  double x, y, z;
};
// rotation of the scalar-first quaternion q (normalised first), applied to row vectors from the right
static void quat_right_matrix(const double q[4], double R[3][3]) {
  const double l = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double a = q[0] / l, b = q[1] / l, c = q[2] / l, d = q[3] / l;
  // standard (column-vector) matrix L; the right-multiplicative one is its transpose
  const double L[3][3] = {{a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)},
                          {2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)},
                          {2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = L[j][i];
}
static void cuboid_from_params(const double p[10], D3 out[8]) {  // FitCuboidBFGS.hs:98-112
  double R[3][3];
  quat_right_matrix(p + 6, R);
  const double ha = p[3] / 2, hb = p[4] / 2, hc = p[5] / 2;
  for (int k = 0; k < 8; ++k) {
    const double lx = (k & 4) ? ha : -ha, ly = (k & 2) ? hb : -hb, lz = (k & 1) ? hc : -hc;
    out[k] = {lx * R[0][0] + ly * R[1][0] + lz * R[2][0] + p[0], lx * R[0][1] + ly * R[1][1] + lz * R[2][1] + p[1],
              lx * R[0][2] + ly * R[1][2] + lz * R[2][2] + p[2]};
  }
}
static inline double normsqr(D3 a, D3 b) {
  const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return dx * dx + dy * dy + dz * dz;
}
static double errfun_ordered(const D3 pts[8], const double p[10]) {  // FitCuboidBFGS.hs:51-65
  D3 est[8];
  cuboid_from_params(p, est);
  double s = 0.0;
  for (int k = 0; k < 8; ++k) s += normsqr(pts[k], est[k]);
  return s;
}
static double errfun_closest(const D3* pts, int npts, const double p[10]) {  // FitCuboidBFGS.hs:73-76
  D3 est[8];
  cuboid_from_params(p, est);
  double s = 0.0;
  for (int i = 0; i < npts; ++i) {
    int bestk = 0;
    double bestd = std::sqrt(normsqr(pts[i], est[0]));
    for (int k = 1; k < 8; ++k) {
      const double dk = std::sqrt(normsqr(pts[i], est[k]));
      if (dk < bestd) {  // first minimum wins on ties (minimumBy)
        bestd = dk;
        bestk = k;
      }
    }
    s += normsqr(pts[i], est[bestk]);
  }
  return s;
}
static void guess_dims(const D3 pts[8], double abc[3]) {  // FitCuboidBFGS.hs:247-252
  double dist[7];
  for (int k = 1; k < 8; ++k) dist[k - 1] = std::sqrt(normsqr(pts[0], pts[k]));
  std::sort(dist, dist + 7);
  abc[0] = dist[0];
  abc[1] = dist[1];
  abc[2] = std::sqrt(dist[6] * dist[6] - dist[0] * dist[0] - dist[1] * dist[1]);
}

struct FitCtx {
  const D3* pts;
  double center[3];
};
static double obj_center_fixed(const double* x, int, void* u) {
  const FitCtx* c = (const FitCtx*)u;
  double p[10] = {c->center[0], c->center[1], c->center[2], x[0], x[1], x[2], x[3], x[4], x[5], x[6]};
  return errfun_closest(c->pts, 8, p);
}
static double obj_closest(const double* x, int, void* u) { return errfun_closest(((const FitCtx*)u)->pts, 8, x); }
static double obj_ordered(const double* x, int, void* u) { return errfun_ordered(((const FitCtx*)u)->pts, x); }

static const double kFitEps = 1e-8;
static const int kFitMaxIt = 2000;

// fitCuboidFromCenter (FitCuboidBFGS.hs:172-184): centre pinned to the corner mean, closest-corner association
static void fit_from_center(const D3 pts[8], int arg_order, double params[10], int* steps, double* err) {
  FitCtx c{pts, {0, 0, 0}};
  for (int k = 0; k < 8; ++k) {
    c.center[0] += pts[k].x;
    c.center[1] += pts[k].y;
    c.center[2] += pts[k].z;
  }
  for (double& v : c.center) v *= (1.0 / 8.0);
  double abc[3];
  guess_dims(pts, abc);
  const double a = abc[0];
  const double named_start[7] = {a, a, a, 0.1, 0.1, 0.1, 0.1};
  const double named_box[7] = {a / 10, a / 10, a / 10, 0.1, 0.1, 0.1, 0.1};
  const double* start = arg_order == HSH_FIT_AS_NAMED ? named_start : named_box;
  const double* box = arg_order == HSH_FIT_AS_NAMED ? named_box : named_start;
  double sol[7], fv;
  nm_minimize(obj_center_fixed, &c, 7, start, box, kFitEps, kFitMaxIt, sol, &fv, steps);
  for (int i = 0; i < 3; ++i) params[i] = c.center[i];
  for (int i = 0; i < 7; ++i) params[3 + i] = sol[i];
  if (err) *err = obj_center_fixed(sol, 7, &c);
}
// fitCuboidFromCenterFirst (FitCuboidBFGS.hs:188-201): the above, then all ten parameters free
static void fit_from_center_first(const D3 pts[8], int arg_order, double params[10], int* steps, double* err) {
  double first[10];
  int steps1 = 0;
  fit_from_center(pts, arg_order, first, &steps1, nullptr);
  double abc[3];
  guess_dims(pts, abc);
  const double a = abc[0];
  const double named_box[10] = {0.01, 0.01, 0.01, a / 10, a / 10, a / 10, 0.1, 0.1, 0.1, 0.1};
  const double* start = arg_order == HSH_FIT_AS_NAMED ? first : named_box;
  const double* box = arg_order == HSH_FIT_AS_NAMED ? named_box : first;
  FitCtx c{pts, {0, 0, 0}};
  int steps2 = 0;
  double fv;
  nm_minimize(obj_closest, &c, 10, start, box, kFitEps, kFitMaxIt, params, &fv, &steps2);
  if (steps) *steps = steps1 + steps2;
  if (err) *err = errfun_closest(pts, 8, params);
}
// fitCuboid (FitCuboidBFGS.hs:205-233): corners given in cuboidFromParams order
static void fit_ordered(const D3 pts[8], int arg_order, double params[10], int* steps, double* err) {
  double abc[3];
  guess_dims(pts, abc);
  double m[3] = {0, 0, 0};
  for (int k = 0; k < 8; ++k) {
    m[0] += pts[k].x;
    m[1] += pts[k].y;
    m[2] += pts[k].z;
  }
  const double named_start[10] = {m[0] / 8, m[1] / 8, m[2] / 8, abc[0], abc[1], abc[2], 0.1, 0.1, 0.1, 0.1};
  const double named_box[10] = {0.01, 0.01, 0.01, abc[0] / 10, abc[0] / 10, abc[0] / 10, 0.1, 0.1, 0.1, 0.1};
  const double* start = arg_order == HSH_FIT_AS_NAMED ? named_start : named_box;
  const double* box = arg_order == HSH_FIT_AS_NAMED ? named_box : named_start;
  FitCtx c{pts, {0, 0, 0}};
  double fv;
  nm_minimize(obj_ordered, &c, 10, start, box, kFitEps, kFitMaxIt, params, &fv, steps);
  if (err) *err = errfun_ordered(pts, params);
}

// ---------------------------------------------------------------------------------------------------------------
// graph + 1-D placement
// ---------------------------------------------------------------------------------------------------------------
// ids -> dense indices in order of first appearance (Bijection.hs:16-23)
struct Bijection {
  std::map<uint32_t, int> index_of;
  std::vector<uint32_t> id_of;
  int add(uint32_t id) {
    auto it = index_of.find(id);
    if (it != index_of.end()) return it->second;
    const int k = (int)id_of.size();
    index_of.emplace(id, k);
    id_of.push_back(id);
    return k;
  }
};

static void connected_components(const uint32_t* a, const uint32_t* b, int m, std::vector<int>& comp_of_edge, int& n_comp) {
  Bijection bj;
  std::vector<std::pair<int, int>> e(m);
  for (int i = 0; i < m; ++i) {
    e[i].first = bj.add(a[i]);
    e[i].second = bj.add(b[i]);
  }
  std::vector<int> parent(bj.id_of.size());
  for (size_t i = 0; i < parent.size(); ++i) parent[i] = (int)i;
  auto find = [&](int v) {
    while (parent[v] != v) v = parent[v] = parent[parent[v]];
    return v;
  };
  for (auto& p : e) {
    const int ra = find(p.first), rb = find(p.second);
    if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);  // root = smallest index = first appearance
  }
  std::map<int, int> label;  // root (ascending = order of first appearance) -> component number
  for (size_t v = 0; v < parent.size(); ++v) label.emplace(find((int)v), 0);
  int k = 0;
  for (auto& kv : label) kv.second = k++;
  n_comp = k;
  comp_of_edge.resize(m);
  for (int i = 0; i < m; ++i) comp_of_edge[i] = label[find(e[i].first)];
}

struct Placement {
  std::vector<uint32_t> nodes;
  std::vector<double> pos;
  double rmse;
};
// TranslationOptimizer.hs:36-72
static bool lstsq_distances(const std::map<std::pair<uint32_t, uint32_t>, double>& dist, Placement& out) {
  Bijection bj;
  for (auto& kv : dist) {
    bj.add(kv.first.first);
    bj.add(kv.first.second);
  }
  const int n = (int)bj.id_of.size(), m = (int)dist.size();
  if (m == 0 || n < 2) return false;
  std::map<std::pair<int, int>, double> rows;  // re-keyed by index pair; duplicates cannot arise (bijection)
  for (auto& kv : dist) rows[{bj.index_of[kv.first.first], bj.index_of[kv.first.second]}] = kv.second;
  const int cols = n - 1;  // x_0 = 0: its column is dropped
  std::vector<double> A((size_t)m * cols, 0.0), b(m);
  int r = 0;
  for (auto& kv : rows) {
    const int i = kv.first.first, j = kv.first.second;
    // a row holds -1 at i and then +1 at j; for i == j the -1 wins, as in the reference's guard order
    if (j > 0 && j != i) A[(size_t)r * cols + (j - 1)] = 1.0;
    if (i > 0) A[(size_t)r * cols + (i - 1)] = -1.0;
    b[r++] = kv.second;
  }
  std::vector<double> x;
  if (!solve_least_squares(m, cols, A, b, x)) return false;
  double ss = 0.0;
  for (int k = 0; k < m; ++k) {
    double s = 0.0;
    for (int j = 0; j < cols; ++j) s += A[(size_t)k * cols + j] * x[j];
    ss += (s - b[k]) * (s - b[k]);
  }
  out.nodes = bj.id_of;
  out.pos.assign(n, 0.0);
  for (int j = 0; j < cols; ++j) out.pos[j + 1] = x[j];
  out.rmse = std::sqrt(std::sqrt(ss) / m);  // the reference divides the 2-NORM (not its square) by m (:70)
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Haskell `show :: Float -> String`
// ---------------------------------------------------------------------------------------------------------------
static std::string show_float(float v) {
  if (std::isnan(v)) return "NaN";
  if (std::isinf(v)) return v < 0 ? "-Infinity" : "Infinity";
  std::string sign = std::signbit(v) ? "-" : "";
  const float x = std::fabs(v);
  if (x == 0.f) return sign + "0.0";
  char buf[64];
  std::string digits;
  int e10 = 0;  // x = 0.d1 d2 ... * 10^e10
  for (int p = 1; p <= 9; ++p) {
    std::snprintf(buf, sizeof buf, "%.*e", p - 1, (double)x);
    if (std::strtof(buf, nullptr) == x) {
      const char* ep = std::strchr(buf, 'e');
      digits.clear();
      for (const char* c = buf; c < ep; ++c)
        if (*c >= '0' && *c <= '9') digits.push_back(*c);
      e10 = std::atoi(ep + 1) + 1;
      break;
    }
  }
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  std::string out = sign;
  if (x >= 0.1f && x < 1.0e7f) {
    if (e10 <= 0) {
      out += "0." + std::string((size_t)(-e10), '0') + digits;
    } else {
      std::string ip = digits.substr(0, std::min((size_t)e10, digits.size()));
      ip += std::string((size_t)e10 - ip.size(), '0');
      std::string fp = digits.size() > (size_t)e10 ? digits.substr((size_t)e10) : "0";
      out += ip + "." + fp;
    }
  } else {
    out += digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : "0") + "e" + std::to_string(e10 - 1);
  }
  return out;
}

// ---------------------------------------------------------------------------------------------------------------
// files
// ---------------------------------------------------------------------------------------------------------------
static bool read_pcd_xyz(const std::string& path, std::vector<V3>& out, std::string& err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) {
    err = "cannot open " + path;
    return false;
  }
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  std::vector<char> types;
  size_t points = 0, width = 0, height = 1;
  std::string data;
  char line[1024];
  while (std::fgets(line, sizeof line, f)) {
    if (line[0] == '#') continue;
    char key[64] = {0};
    if (std::sscanf(line, "%63s", key) != 1) continue;
    std::vector<std::string> tok;
    for (char* t = std::strtok(line, " \t\r\n"); t; t = std::strtok(nullptr, " \t\r\n")) tok.push_back(t);
    const std::string k = tok[0];
    if (k == "FIELDS") fields.assign(tok.begin() + 1, tok.end());
    else if (k == "SIZE") for (size_t i = 1; i < tok.size(); ++i) sizes.push_back(std::atoi(tok[i].c_str()));
    else if (k == "TYPE") for (size_t i = 1; i < tok.size(); ++i) types.push_back(tok[i][0]);
    else if (k == "COUNT") for (size_t i = 1; i < tok.size(); ++i) counts.push_back(std::atoi(tok[i].c_str()));
    else if (k == "WIDTH" && tok.size() > 1) width = std::strtoull(tok[1].c_str(), nullptr, 10);
    else if (k == "HEIGHT" && tok.size() > 1) height = std::strtoull(tok[1].c_str(), nullptr, 10);
    else if (k == "POINTS" && tok.size() > 1) points = std::strtoull(tok[1].c_str(), nullptr, 10);
    else if (k == "DATA" && tok.size() > 1) {
      data = tok[1];
      break;
    }
  }
  if (points == 0) points = width * height;
  if (counts.empty()) counts.assign(fields.size(), 1);
  if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size() || counts.size() != fields.size()) {
    std::fclose(f);
    err = "malformed PCD header in " + path;
    return false;
  }
  int col[3] = {-1, -1, -1}, off[3] = {0, 0, 0};
  int stride = 0, ncols = 0;
  for (size_t i = 0; i < fields.size(); ++i) {
    for (int a = 0; a < 3; ++a)
      if (fields[i] == std::string(1, "xyz"[a])) {
        if (sizes[i] != 4 || types[i] != 'F') {
          std::fclose(f);
          err = "PCD x/y/z must be float32 in " + path;
          return false;
        }
        col[a] = ncols;
        off[a] = stride;
      }
    stride += sizes[i] * counts[i];
    ncols += counts[i];
  }
  if (col[0] < 0 || col[1] < 0 || col[2] < 0) {
    std::fclose(f);
    err = "PCD without x y z fields: " + path;
    return false;
  }
  out.clear();
  out.reserve(points);
  if (data == "ascii") {
    std::vector<double> vals(ncols);
    for (size_t p = 0; p < points; ++p) {
      bool ok = true;
      for (int c = 0; c < ncols; ++c)
        if (std::fscanf(f, "%lf", &vals[c]) != 1) {
          ok = false;
          break;
        }
      if (!ok) break;
      out.push_back({(float)vals[col[0]], (float)vals[col[1]], (float)vals[col[2]]});
    }
  } else if (data == "binary") {
    std::vector<unsigned char> rec(stride);
    for (size_t p = 0; p < points; ++p) {
      if (std::fread(rec.data(), 1, stride, f) != (size_t)stride) break;
      V3 v;
      std::memcpy(&v.x, &rec[off[0]], 4);
      std::memcpy(&v.y, &rec[off[1]], 4);
      std::memcpy(&v.z, &rec[off[2]], 4);
      out.push_back(v);
    }
  } else {
    std::fclose(f);
    err = "unsupported PCD DATA '" + data + "' in " + path;
    return false;
  }
  std::fclose(f);
  if (out.size() != points) {
    err = "truncated PCD " + path;
    return false;
  }
  return true;
}

static bool read_planes_txt(const std::string& path, std::vector<PlaneEq>& out, std::string& err) {  // Main.hs:1379-1389
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) {
    err = "cannot open " + path;
    return false;
  }
  out.clear();
  double a, b, c, d;
  while (std::fscanf(f, "%lf %lf %lf %lf", &a, &b, &c, &d) == 4) out.push_back(mk_plane_eq({(float)a, (float)b, (float)c}, -(float)d));
  std::fclose(f);
  if (out.empty()) {
    err = "Could not load planes: " + path;
    return false;
  }
  return true;
}

static std::string base_of_parent(const std::string& path) {  // takeFileName . takeDirectory
  std::string p = path;
  const size_t cut = p.find_last_of('/');
  p = cut == std::string::npos ? std::string() : p.substr(0, cut);
  const size_t cut2 = p.find_last_of('/');
  return cut2 == std::string::npos ? p : p.substr(cut2 + 1);
}

}  // namespace

// =================================================================================================================
// the house
// =================================================================================================================
struct hsh_house {
  std::map<uint32_t, Room> rooms;   // Data.Map ID Room: iteration in ascending id
  std::vector<WallLink> links;      // newest first (connectWalls conses)
  uint32_t next_id = 1;
  mutable std::string err;

  uint32_t gen_id() { return next_id++; }
  int fail(int code, const std::string& msg) const {
    err = msg;
    g_err = msg;
    return code;
  }
  Room* room(uint32_t id) {
    auto it = rooms.find(id);
    return it == rooms.end() ? nullptr : &it->second;
  }
  const Room* room(uint32_t id) const {
    auto it = rooms.find(id);
    return it == rooms.end() ? nullptr : &it->second;
  }
  Room* room_of_plane(uint32_t pid, Plane** plane = nullptr) {  // findRoomContainingPlane, Main.hs:1617-1618
    for (auto& kv : rooms)
      for (Plane& p : kv.second.planes)
        if (p.id == pid) {
          if (plane) *plane = &p;
          return &kv.second;
        }
    return nullptr;
  }
};

namespace {

static int fail0(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

static void rotate_room_around(Room& r, V3 c, const M3& R) {  // Main.hs:1664-1675
  for (Plane& p : r.planes) {
    p.eq = rotate_eq_around(c, R, p.eq);
    for (V3& b : p.bounds) b = rotate_around(c, R, b);
  }
  for (V3& p : r.cloud) p = rotate_around(c, R, p);
  for (auto& kv : r.corners) kv.second = rotate_around(c, R, kv.second);
  for (auto& kv : r.suggested) kv.second = rotate_around(c, R, kv.second);
  M4 P = r.proj;
  translate_after(P, {-c.x, -c.y, -c.z});
  P = mul4(P, embed(R));
  translate_after(P, c);
  r.proj = P;
}
static void translate_room(Room& r, V3 off) {  // Main.hs:1700-1709
  for (Plane& p : r.planes) {
    p.eq = translate_eq(off, p.eq);
    for (V3& b : p.bounds) b = off + b;
  }
  for (V3& p : r.cloud) p = off + p;
  for (auto& kv : r.corners) kv.second = kv.second + off;
  for (auto& kv : r.suggested) kv.second = kv.second + off;
  translate_after(r.proj, off);
}
static V3 corner_mean(const Room& r) {
  std::vector<V3> ps;
  for (auto& kv : r.corners) ps.push_back(kv.second);
  return point_mean(ps);
}

// rotation taking direction n1 onto n2 (both unit): axis n1 x n2 (normalised), angle acos(n1.n2) -- Main.hs:1553-1560.
// The reference is undefined for parallel normals (0/0 axis); here: identity when they agree, an error when opposed.
static bool rotation_between(V3 n1, V3 n2, M3& R, std::string& err) {
  const V3 ax = cross(n1, n2);
  const float l = norm(ax);
  const float costheta = dot(n1, n2) / (norm(n1) * norm(n2));
  if (!(l > 0.f)) {
    if (costheta > 0.f) {
      R = M3{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
      return true;
    }
    err = "rotationBetweenPlaneEqs: normals are opposed, the rotation axis is undefined";
    return false;
  }
  R = rot_axis_angle((1.0f / l) * ax, std::acos(std::min(1.0f, std::max(-1.0f, costheta))));
  return true;
}

static bool plane_corner(const PlaneEq& e1, const PlaneEq& e2, const PlaneEq& e3, V3& out) {  // Main.hs:1413-1430
  std::vector<double> A = {e1.n.x, e1.n.y, e1.n.z, e2.n.x, e2.n.y, e2.n.z, e3.n.x, e3.n.y, e3.n.z};
  std::vector<double> b = {e1.d, e2.d, e3.d}, x;
  if (!solve_square(3, A, b, x)) return false;
  out = {(float)x[0], (float)x[1], (float)x[2]};
  return true;
}

}  // namespace

extern "C" {

hsh_house* hsh_create(void) {
  return new hsh_house();
}
void hsh_destroy(hsh_house* h) { delete h; }
const char* hsh_last_error(const hsh_house* h) { return h ? h->err.c_str() : g_err.c_str(); }

int hsh_add_room(hsh_house* h, const char* name, const float* cloud_xyz, size_t n_points, const float* planes_abcd, int n_planes,
                 const float* hull_xyz, const int* hull_offsets, uint32_t* room_id) {
  if (!h || !cloud_xyz || n_points == 0 || n_planes < 0 || (n_planes > 0 && (!planes_abcd || !hull_xyz || !hull_offsets)))
    return h ? h->fail(HSH_ERR_ARG, "hsh_add_room: bad arguments (a room needs a non-empty cloud)") : fail0(HSH_ERR_ARG, "null house");
  Room r;
  h->gen_id();  // the cloud's own id (cloudFromFile, Main.hs:1336) -- keeps id numbering in step with the reference
  r.cloud.resize(n_points);
  for (size_t i = 0; i < n_points; ++i) r.cloud[i] = {cloud_xyz[3 * i], cloud_xyz[3 * i + 1], cloud_xyz[3 * i + 2]};
  const V3 center = point_mean(r.cloud);
  for (int k = 0; k < n_planes; ++k) {
    Plane p;
    p.eq = mk_plane_eq({planes_abcd[4 * k], planes_abcd[4 * k + 1], planes_abcd[4 * k + 2]}, -planes_abcd[4 * k + 3]);
    for (int i = hull_offsets[k]; i < hull_offsets[k + 1]; ++i) p.bounds.push_back({hull_xyz[3 * i], hull_xyz[3 * i + 1], hull_xyz[3 * i + 2]});
    if (p.bounds.empty()) return h->fail(HSH_ERR_ARG, "hsh_add_room: plane without hull points (planeMean of nothing)");
    p.id = h->gen_id();
    // makeInwardFacing (Main.hs:1746-1752): the normal must point from the hull's mean towards the cloud's mean
    const V3 inward = center - point_mean(p.bounds);
    if (!(dot(inward, p.eq.n) > 0.f)) p.eq = {{-p.eq.n.x, -p.eq.n.y, -p.eq.n.z}, -p.eq.d};
    r.planes.push_back(std::move(p));
  }
  r.id = h->gen_id();
  r.proj = identity4();
  r.name = name ? name : "";
  if (room_id) *room_id = r.id;
  h->rooms[r.id] = std::move(r);
  return HSH_OK;
}

int hsh_load_room(hsh_house* h, const char* dir, uint32_t* room_id) {
  if (!h || !dir) return h ? h->fail(HSH_ERR_ARG, "hsh_load_room: null argument") : fail0(HSH_ERR_ARG, "null house");
  const std::string d(dir);
  const std::string cloud_path = d + "/cloud_downsampled.pcd";
  std::vector<V3> cloud;
  std::string err;
  if (!read_pcd_xyz(cloud_path, cloud, err)) return h->fail(HSH_ERR_IO, err);
  if (cloud.empty()) return h->fail(HSH_ERR_IO, "File " + cloud_path + " contains no points!");
  std::vector<PlaneEq> eqs;
  if (!read_planes_txt(d + "/planes.txt", eqs, err)) return h->fail(HSH_ERR_IO, err);
  std::vector<float> abcd, hull;
  std::vector<int> offs{0};
  for (size_t k = 0; k < eqs.size(); ++k) {
    std::vector<V3> pts;
    if (!read_pcd_xyz(d + "/cloud_plane_hull" + std::to_string(k) + ".pcd", pts, err)) return h->fail(HSH_ERR_IO, err);
    for (const V3& p : pts) hull.insert(hull.end(), {p.x, p.y, p.z});
    offs.push_back((int)(hull.size() / 3));
    abcd.insert(abcd.end(), {eqs[k].n.x, eqs[k].n.y, eqs[k].n.z, -eqs[k].d});
  }
  return hsh_add_room(h, cloud_path.c_str(), &cloud[0].x, cloud.size(), abcd.data(), (int)eqs.size(), hull.data(), offs.data(), room_id);
}

int hsh_room_ids(const hsh_house* h, uint32_t* ids, int cap, int* n) {
  if (!h || !n) return fail0(HSH_ERR_ARG, "hsh_room_ids: null argument");
  *n = (int)h->rooms.size();
  if (!ids) return HSH_OK;
  if (cap < *n) return h->fail(HSH_ERR_CAPACITY, "hsh_room_ids: capacity");
  int i = 0;
  for (auto& kv : h->rooms) ids[i++] = kv.first;
  return HSH_OK;
}

int hsh_room_planes(const hsh_house* h, uint32_t room, uint32_t* plane_ids, float* eq_nd, int cap, int* n) {
  const Room* r = h ? h->room(room) : nullptr;
  if (!r || !n) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  *n = (int)r->planes.size();
  if (!plane_ids && !eq_nd) return HSH_OK;
  if (cap < *n) return h->fail(HSH_ERR_CAPACITY, "hsh_room_planes: capacity");
  for (int i = 0; i < *n; ++i) {
    if (plane_ids) plane_ids[i] = r->planes[i].id;
    if (eq_nd) {
      const PlaneEq& e = r->planes[i].eq;
      eq_nd[4 * i] = e.n.x, eq_nd[4 * i + 1] = e.n.y, eq_nd[4 * i + 2] = e.n.z, eq_nd[4 * i + 3] = e.d;
    }
  }
  return HSH_OK;
}

int hsh_plane_bounds(const hsh_house* h, uint32_t plane, float* xyz, int cap_points, int* n) {
  if (!h || !n) return fail0(HSH_ERR_ARG, "hsh_plane_bounds: null argument");
  Plane* p = nullptr;
  if (!const_cast<hsh_house*>(h)->room_of_plane(plane, &p)) return h->fail(HSH_ERR_ARG, "no such plane");
  *n = (int)p->bounds.size();
  if (!xyz) return HSH_OK;
  if (cap_points < *n) return h->fail(HSH_ERR_CAPACITY, "hsh_plane_bounds: capacity");
  std::memcpy(xyz, p->bounds.data(), sizeof(V3) * p->bounds.size());
  return HSH_OK;
}

int hsh_room_corners(const hsh_house* h, uint32_t room, int suggested, uint32_t* ids, float* xyz, int cap, int* n) {
  const Room* r = h ? h->room(room) : nullptr;
  if (!r || !n) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  const auto& cs = suggested ? r->suggested : r->corners;
  *n = (int)cs.size();
  if (!ids && !xyz) return HSH_OK;
  if (cap < *n) return h->fail(HSH_ERR_CAPACITY, "hsh_room_corners: capacity");
  for (int i = 0; i < *n; ++i) {
    if (ids) ids[i] = cs[i].first;
    if (xyz) xyz[3 * i] = cs[i].second.x, xyz[3 * i + 1] = cs[i].second.y, xyz[3 * i + 2] = cs[i].second.z;
  }
  return HSH_OK;
}

int hsh_room_cloud(const hsh_house* h, uint32_t room, float* xyz, size_t cap_points, size_t* n) {
  const Room* r = h ? h->room(room) : nullptr;
  if (!r || !n) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  *n = r->cloud.size();
  if (!xyz) return HSH_OK;
  if (cap_points < *n) return h->fail(HSH_ERR_CAPACITY, "hsh_room_cloud: capacity");
  std::memcpy(xyz, r->cloud.data(), sizeof(V3) * r->cloud.size());
  return HSH_OK;
}

int hsh_room_means(const hsh_house* h, uint32_t room, float cloud_mean[3], float corner_mean_out[3]) {
  const Room* r = h ? h->room(room) : nullptr;
  if (!r) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  if (cloud_mean) {
    if (r->cloud.empty()) return h->fail(HSH_ERR_STATE, "pointMean: empty");
    const V3 m = point_mean(r->cloud);
    cloud_mean[0] = m.x, cloud_mean[1] = m.y, cloud_mean[2] = m.z;
  }
  if (corner_mean_out) {
    if (r->corners.empty()) return h->fail(HSH_ERR_STATE, "pointMean: empty");
    const V3 m = corner_mean(*r);
    corner_mean_out[0] = m.x, corner_mean_out[1] = m.y, corner_mean_out[2] = m.z;
  }
  return HSH_OK;
}

int hsh_set_room_corners(hsh_house* h, uint32_t room, const float* xyz, int n) {
  Room* r = h ? h->room(room) : nullptr;
  if (!r || n < 0 || (n > 0 && !xyz)) return h ? h->fail(HSH_ERR_ARG, "hsh_set_room_corners: bad arguments") : fail0(HSH_ERR_ARG, "null house");
  r->corners.clear();
  for (int i = 0; i < n; ++i) r->corners.push_back({h->gen_id(), V3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}});
  return HSH_OK;
}

int hsh_accept_corner_suggestion(hsh_house* h, uint32_t room, uint32_t suggestion_id) {
  Room* r = h ? h->room(room) : nullptr;
  if (!r) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  for (auto& s : r->suggested)
    if (s.first == suggestion_id) {
      if (r->corners.size() >= 8) return h->fail(HSH_ERR_STATE, "room already has 8 corners");
      r->corners.push_back(s);
      return HSH_OK;
    }
  return h->fail(HSH_ERR_ARG, "no such corner suggestion");
}

int hsh_translate_room(hsh_house* h, uint32_t room, const float off[3]) {
  Room* r = h ? h->room(room) : nullptr;
  if (!r || !off) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  translate_room(*r, {off[0], off[1], off[2]});
  return HSH_OK;
}

int hsh_rotate_room(hsh_house* h, uint32_t room, const float rot_right[9]) {
  Room* r = h ? h->room(room) : nullptr;
  if (!r || !rot_right) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  M3 R;
  std::memcpy(R.m, rot_right, sizeof(R.m));
  rotate_room_around(*r, point_mean(r->cloud), R);
  return HSH_OK;
}

int hsh_rotate_kinfu_room(hsh_house* h, uint32_t room) {
  Room* r = h ? h->room(room) : nullptr;
  if (!r) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  const float a = 180.0f / 180.0f * 3.14159265358979323846f;  // toRad 180 in binary32
  const float c = std::cos(a), s = std::sin(a);
  const M3 R{{{1, 0, 0}, {0, c, s}, {0, -s, c}}};  // right-multiplicative rotation about X
  rotate_room_around(*r, point_mean(r->cloud), R);
  return HSH_OK;
}

int hsh_room_auto_align_axis(hsh_house* h, uint32_t room, const float axis[3]) {
  Room* r = h ? h->room(room) : nullptr;
  if (!r || !axis) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  if (r->planes.empty()) return h->fail(HSH_ERR_STATE, "room has no planes");
  const V3 ax{axis[0], axis[1], axis[2]};
  size_t best = 0;
  for (size_t i = 1; i < r->planes.size(); ++i)  // maximumBy keeps the LAST of equal maxima
    if (dot(ax, r->planes[i].eq.n) >= dot(ax, r->planes[best].eq.n)) best = i;
  M3 R;
  std::string err;
  if (!rotation_between(r->planes[best].eq.n, normalized(ax), R, err)) return h->fail(HSH_ERR_SINGULAR, err);
  rotate_room_around(*r, point_mean(r->cloud), R);
  return HSH_OK;
}

int hsh_auto_align_floor(hsh_house* h, uint32_t room) {
  const float y[3] = {0.f, 1.f, 0.f};
  return hsh_room_auto_align_axis(h, room, y);
}

int hsh_remove_ceiling(hsh_house* h, uint32_t room) {
  Room* r = h ? h->room(room) : nullptr;
  if (!r) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  const size_t n = r->cloud.size();
  if (n == 0) return HSH_OK;
  const size_t discard = n / 5;
  if (discard < 1) return h->fail(HSH_ERR_STATE, "kLargestBy: k must be >= 1 if the vector is not empty");
  std::vector<float> ys(n);
  for (size_t i = 0; i < n; ++i) ys[i] = r->cloud[i].y;
  std::nth_element(ys.begin(), ys.begin() + (discard - 1), ys.end(), std::greater<float>());
  const float limit = ys[discard - 1];  // the discard-th largest y; points AT the limit survive
  std::vector<V3> kept;
  for (const V3& p : r->cloud)
    if (p.y <= limit) kept.push_back(p);
  r->cloud.swap(kept);
  return HSH_OK;
}

int hsh_suggest_points(hsh_house* h, uint32_t room, float cutoff_factor, int* n_suggested, int* adopted) {
  Room* r = h ? h->room(room) : nullptr;
  if (!r) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  if (r->cloud.empty()) return h->fail(HSH_ERR_STATE, "pointMean: empty");
  const V3 mean = point_mean(r->cloud);
  float far = 0.f;
  for (const V3& p : r->cloud) far = std::max(far, norm(p - mean));
  const float cutoff = cutoff_factor * far;
  std::vector<std::pair<uint32_t, V3>> sugg;
  const auto& ps = r->planes;
  // triples in LIST order, filtered to id-ascending ones (the derived Ord on Plane compares planeID first)
  for (size_t i = 0; i < ps.size(); ++i)
    for (size_t j = 0; j < ps.size(); ++j)
      for (size_t k = 0; k < ps.size(); ++k) {
        if (!(ps[i].id < ps[j].id && ps[j].id < ps[k].id)) continue;
        V3 c;
        if (!plane_corner(ps[i].eq, ps[j].eq, ps[k].eq, c)) continue;
        if (norm(c - mean) <= cutoff) sugg.push_back({0u, c});
      }
  for (auto& s : sugg) s.first = h->gen_id();
  const bool take = r->corners.empty() && sugg.size() == 8;
  if (take)
    r->corners = sugg;
  else
    r->suggested = sugg;
  if (n_suggested) *n_suggested = (int)sugg.size();
  if (adopted) *adopted = take ? 1 : 0;
  return HSH_OK;
}

int hsh_fit_cuboid_to_room(hsh_house* h, uint32_t room, int arg_order, int* steps, double* rmse, double params_out[10]) {
  Room* r = h ? h->room(room) : nullptr;
  if (!r) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  if (r->corners.size() < 8) return h->fail(HSH_ERR_STATE, "not enough room corners; need 8");
  if (r->corners.size() > 8) return h->fail(HSH_ERR_STATE, "too many room corners; the cuboid fit takes exactly 8");
  D3 pts[8];
  for (int k = 0; k < 8; ++k) pts[k] = {r->corners[k].second.x, r->corners[k].second.y, r->corners[k].second.z};
  double params[10], err = 0.0;
  int st = 0;
  fit_from_center_first(pts, arg_order, params, &st, &err);
  for (int i = 0; i < 10; ++i)
    if (!std::isfinite(params[i])) return h->fail(HSH_ERR_SINGULAR, "cuboid fit diverged");
  D3 cub[8];
  cuboid_from_params(params, cub);
  V3 cpts[8];
  for (int k = 0; k < 8; ++k) cpts[k] = {(float)cub[k].x, (float)cub[k].y, (float)cub[k].z};
  const V3 center{(float)params[0], (float)params[1], (float)params[2]};
  const float dims[3] = {(float)params[3], (float)params[4], (float)params[5]};
  const double qd[4] = {(double)(float)params[6], (double)(float)params[7], (double)(float)params[8], (double)(float)params[9]};
  double Rd[3][3];
  quat_right_matrix(qd, Rd);
  M3 R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R.m[i][j] = (float)Rd[i][j];
  // makePlanesFromCuboid (Main.hs:1850-1885): six axis planes of the origin-centred box, rotated, then translated
  std::vector<Plane> planes;
  for (int axis = 0; axis < 3; ++axis)
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      Plane p;
      p.id = h->gen_id();
      PlaneEq e = mk_plane_eq(along((float)sgn, axis), dims[axis] / 2);
      e = translate_eq(center, rotate_eq_around({0, 0, 0}, R, e));
      p.eq = e;
      std::vector<V3> on;
      for (const V3& c : cpts)
        if (std::fabs(signed_distance(e, c)) < 1e-4f) on.push_back(c);
      if (on.size() != 4) return h->fail(HSH_ERR_STATE, "cuboid face does not hold exactly 4 corners within 1e-4");
      // c1, then the others by distance from c1: nearest, FARTHEST, middle -> a closed polygon order
      const V3 c1 = on[0];
      std::vector<V3> rest(on.begin() + 1, on.end());
      std::stable_sort(rest.begin(), rest.end(), [&](const V3& a, const V3& b) { return norm(a - c1) < norm(b - c1); });
      p.bounds = {c1, rest[0], rest[2], rest[1]};
      planes.push_back(std::move(p));
    }
  std::vector<uint32_t> old_ids;
  for (const Plane& p : r->planes) old_ids.push_back(p.id);
  for (int k = 0; k < 8; ++k) r->corners[k].second = cpts[k];
  r->planes = std::move(planes);
  // connections that named a replaced plane are dropped (Main.hs:1843-1847)
  auto gone = [&](uint32_t pid) { return std::find(old_ids.begin(), old_ids.end(), pid) != old_ids.end(); };
  h->links.erase(std::remove_if(h->links.begin(), h->links.end(), [&](const WallLink& w) { return gone(w.p1) || gone(w.p2); }),
                 h->links.end());
  if (steps) *steps = st;
  if (rmse) *rmse = std::sqrt(err);
  if (params_out) std::memcpy(params_out, params, sizeof(params));
  return HSH_OK;
}

int hsh_connect_walls(hsh_house* h, uint32_t plane1, uint32_t plane2, int relation, float thickness, int* connected) {
  if (!h) return fail0(HSH_ERR_ARG, "null house");
  if (connected) *connected = 0;
  Plane *p1 = nullptr, *p2 = nullptr;
  Room* r1 = h->room_of_plane(plane1, &p1);
  Room* r2 = h->room_of_plane(plane2, &p2);
  if (!r1 || !r2) return h->fail(HSH_ERR_STATE, "the planes are not walls of rooms");
  if (relation != HSH_WALL_OPPOSITE && relation != HSH_WALL_SAME) return h->fail(HSH_ERR_ARG, "bad wall relation");
  auto best_axis = [](V3 n) {  // maximum over (|n.v|, axis) pairs: ties go to the LATER axis
    const float v[3] = {std::fabs(n.x), std::fabs(n.y), std::fabs(n.z)};
    int b = 0;
    for (int a = 1; a < 3; ++a)
      if (v[a] >= v[b]) b = a;
    return b;
  };
  const int a1 = best_axis(p1->eq.n), a2 = best_axis(p2->eq.n);
  if (a1 != a2) return h->fail(HSH_ERR_STATE, "Could not guess axis of wall connection");
  for (const WallLink& w : h->links)
    if ((w.p1 == plane1 && w.p2 == plane2) || (w.p1 == plane2 && w.p2 == plane1)) return HSH_OK;  // already linked
  h->links.insert(h->links.begin(), WallLink{a1, relation, relation == HSH_WALL_OPPOSITE ? thickness : 0.f, plane1, plane2});
  if (connected) *connected = 1;
  return HSH_OK;
}

int hsh_disconnect_walls(hsh_house* h, uint32_t plane1, uint32_t plane2) {
  if (!h) return fail0(HSH_ERR_ARG, "null house");
  h->links.erase(std::remove_if(h->links.begin(), h->links.end(),
                                [&](const WallLink& w) { return (w.p1 == plane1 && w.p2 == plane2) || (w.p1 == plane2 && w.p2 == plane1); }),
                 h->links.end());
  return HSH_OK;
}

int hsh_connected_walls(const hsh_house* h, int* axis, int* relation, float* thickness, uint32_t* plane1, uint32_t* plane2, int cap, int* n) {
  if (!h || !n) return fail0(HSH_ERR_ARG, "hsh_connected_walls: null argument");
  *n = (int)h->links.size();
  if (!axis && !relation && !thickness && !plane1 && !plane2) return HSH_OK;
  if (cap < *n) return h->fail(HSH_ERR_CAPACITY, "hsh_connected_walls: capacity");
  for (int i = 0; i < *n; ++i) {
    const WallLink& w = h->links[i];
    if (axis) axis[i] = w.axis;
    if (relation) relation[i] = w.relation;
    if (thickness) thickness[i] = w.thickness;
    if (plane1) plane1[i] = w.p1;
    if (plane2) plane2[i] = w.p2;
  }
  return HSH_OK;
}

int hsh_optimize_room_positions(hsh_house* h, double rmse_xyz[3]) {
  if (!h) return fail0(HSH_ERR_ARG, "null house");
  struct Link {
    Plane *p1, *p2;
    Room *r1, *r2;
    int axis, relation;
    float thickness;
  };
  std::vector<Link> ls;
  for (const WallLink& w : h->links) {
    Link l{nullptr, nullptr, nullptr, nullptr, w.axis, w.relation, w.thickness};
    l.r1 = h->room_of_plane(w.p1, &l.p1);
    l.r2 = h->room_of_plane(w.p2, &l.p2);
    if (!l.r1 || !l.r2) return h->fail(HSH_ERR_STATE, "a connected wall no longer belongs to a room");
    if (l.r1->corners.empty() || l.r2->corners.empty()) return h->fail(HSH_ERR_STATE, "some room in position optimization has no corners!");
    ls.push_back(l);
  }
  if (rmse_xyz) rmse_xyz[0] = rmse_xyz[1] = rmse_xyz[2] = std::numeric_limits<double>::quiet_NaN();
  for (int axis = 0; axis < 3; ++axis) {
    std::vector<uint32_t> ea, eb;
    std::vector<double> ed;
    const Room* first_room = nullptr;
    for (const Link& l : ls) {
      if (l.axis != axis) continue;
      if (!first_room) first_room = l.r1;
      // roomCenterOffsetFromWalls (Main.hs:2182-2184), binary32
      const V3 w1 = point_mean(l.p1->bounds) - corner_mean(*l.r1);
      const V3 w2 = point_mean(l.p2->bounds) - corner_mean(*l.r2);
      const float o = comp(w1 - w2, axis);
      const float sg = o > 0.f ? 1.f : (o < 0.f ? -1.f : 0.f);
      const float wall = l.relation == HSH_WALL_OPPOSITE ? l.thickness : 0.f;
      ea.push_back(l.r1->id);
      eb.push_back(l.r2->id);
      ed.push_back((double)(o + sg * wall));
    }
    if (!first_room) continue;  // nothing to align along this axis
    // the same (room, room) key listed twice: the LAST listed distance is the one every copy carries
    std::map<std::pair<uint32_t, uint32_t>, double> last;
    for (size_t i = 0; i < ea.size(); ++i) last[{ea[i], eb[i]}] = ed[i];
    std::vector<int> comp_of;
    int n_comp = 0;
    connected_components(ea.data(), eb.data(), (int)ea.size(), comp_of, n_comp);
    // the shift applied to EVERY component is the first linked room's centre (captured before any move) -- Main.hs:2151-2152
    const float first_center = comp(corner_mean(*first_room), axis);
    double worst = 0.0;
    for (int c = 0; c < n_comp; ++c) {
      std::map<std::pair<uint32_t, uint32_t>, double> dist;
      for (size_t i = 0; i < ea.size(); ++i)
        if (comp_of[i] == c) dist[{ea[i], eb[i]}] = last[{ea[i], eb[i]}];
      Placement pl;
      if (!lstsq_distances(dist, pl)) continue;  // "WARNING: optimizeRoomPositions singularity error"
      worst = std::max(worst, pl.rmse);
      std::map<uint32_t, float> target;  // Map ID Float: applied in ascending room id
      for (size_t k = 0; k < pl.nodes.size(); ++k) target[pl.nodes[k]] = (float)pl.pos[k] + first_center;
      for (auto& kv : target) {
        Room* r = h->room(kv.first);
        const float old = comp(corner_mean(*r), axis);
        translate_room(*r, along(kv.second - old, axis));
      }
    }
    if (rmse_xyz) rmse_xyz[axis] = worst;
  }
  return HSH_OK;
}

int hsh_room_projection(const hsh_house* h, uint32_t room, float m[16]) {
  const Room* r = h ? h->room(room) : nullptr;
  if (!r || !m) return h ? h->fail(HSH_ERR_ARG, "no such room") : fail0(HSH_ERR_ARG, "null house");
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) m[4 * i + j] = r->proj.m[j][i];
  return HSH_OK;
}

int hsh_room_projection_string(const hsh_house* h, uint32_t room, int xf_format, char* buf, size_t cap) {
  float m[16];
  const int rc = hsh_room_projection(h, room, m);
  if (rc) return rc;
  std::string s;
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) {
      s += show_float(m[4 * i + j]);
      if (xf_format)
        s += j < 3 ? " " : "\n";
      else if (i * 4 + j < 15)
        s += ",";
    }
  }
  if (!buf || cap < s.size() + 1) return h->fail(HSH_ERR_CAPACITY, "hsh_room_projection_string: capacity");
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return HSH_OK;
}

int hsh_export_all_room_xf_files(const hsh_house* h, const char* dir) {
  if (!h || !dir) return fail0(HSH_ERR_ARG, "hsh_export_all_room_xf_files: null argument");
  ::mkdir(dir, 0777);  // createDirectoryIfMissing False
  for (auto& kv : h->rooms) {
    char buf[1024];
    const int rc = hsh_room_projection_string(h, kv.first, 1, buf, sizeof buf);
    if (rc) return rc;
    std::string base = base_of_parent(kv.second.name);
    if (base.empty()) base = "room" + std::to_string(kv.first);
    const std::string path = std::string(dir) + "/" + base + ".xf";
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) return h->fail(HSH_ERR_IO, "cannot write " + path);
    std::fputs(buf, f);
    std::fclose(f);
  }
  return HSH_OK;
}

// ---- house-less numerics ---------------------------------------------------------------------------------------

int hsh_plane_corner(const float eq_nd[12], float corner[3], int* found) {
  if (!eq_nd || !corner || !found) return fail0(HSH_ERR_ARG, "hsh_plane_corner: null argument");
  PlaneEq e[3];
  for (int i = 0; i < 3; ++i) e[i] = {{eq_nd[4 * i], eq_nd[4 * i + 1], eq_nd[4 * i + 2]}, eq_nd[4 * i + 3]};
  V3 c;
  *found = plane_corner(e[0], e[1], e[2], c) ? 1 : 0;
  if (*found) corner[0] = c.x, corner[1] = c.y, corner[2] = c.z;
  return HSH_OK;
}

int hsh_fit_plane(const float* xyz, int n, float eq_nd[4]) {
  if (!xyz || !eq_nd) return fail0(HSH_ERR_ARG, "hsh_fit_plane: null argument");
  if (n < 3) return fail0(HSH_ERR_ARG, "fitPlane: " + std::to_string(n) + " points given, need at least 3");
  std::vector<V3> ps(n);
  for (int i = 0; i < n; ++i) ps[i] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  const V3 m = point_mean(ps);
  double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (const V3& p : ps) {
    const V3 q = p - m;  // binary32 subtraction, then widened (toDoubleVec . (&- m))
    const double d[3] = {q.x, q.y, q.z};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) S[i][j] += d[i] * d[j];
  }
  double w[3], V[3][3];
  eig_sym3(S, w, V);
  V3 nrm{(float)V[0][2], (float)V[1][2], (float)V[2][2]};  // eigenvector of the SMALLEST eigenvalue (last column)
  // LAPACK leaves the eigenvector's sign unspecified; fix it: the largest component is positive
  const float ax = std::fabs(nrm.x), ay = std::fabs(nrm.y), az = std::fabs(nrm.z);
  const float lead = (ax >= ay && ax >= az) ? nrm.x : (ay >= az ? nrm.y : nrm.z);
  if (lead < 0.f) nrm = {-nrm.x, -nrm.y, -nrm.z};
  nrm = normalized(nrm);
  eq_nd[0] = nrm.x, eq_nd[1] = nrm.y, eq_nd[2] = nrm.z;
  eq_nd[3] = dot(nrm, m);
  return HSH_OK;
}

int hsh_rotation_between(const float n1[3], const float n2[3], float rot_right[9]) {
  if (!n1 || !n2 || !rot_right) return fail0(HSH_ERR_ARG, "hsh_rotation_between: null argument");
  M3 R;
  std::string err;
  if (!rotation_between(normalized({n1[0], n1[1], n1[2]}), normalized({n2[0], n2[1], n2[2]}), R, err)) return fail0(HSH_ERR_SINGULAR, err);
  std::memcpy(rot_right, R.m, sizeof(R.m));
  return HSH_OK;
}

int hsh_cuboid_from_params(const double params[10], double corners[24]) {
  if (!params || !corners) return fail0(HSH_ERR_ARG, "hsh_cuboid_from_params: null argument");
  D3 c[8];
  cuboid_from_params(params, c);
  for (int k = 0; k < 8; ++k) corners[3 * k] = c[k].x, corners[3 * k + 1] = c[k].y, corners[3 * k + 2] = c[k].z;
  return HSH_OK;
}

int hsh_guess_dims(const double corners[24], double abc[3]) {
  if (!corners || !abc) return fail0(HSH_ERR_ARG, "hsh_guess_dims: null argument");
  D3 p[8];
  for (int k = 0; k < 8; ++k) p[k] = {corners[3 * k], corners[3 * k + 1], corners[3 * k + 2]};
  guess_dims(p, abc);
  return HSH_OK;
}

int hsh_errfun(const double corners[24], const double params[10], int closest, double* err) {
  if (!corners || !params || !err) return fail0(HSH_ERR_ARG, "hsh_errfun: null argument");
  D3 p[8];
  for (int k = 0; k < 8; ++k) p[k] = {corners[3 * k], corners[3 * k + 1], corners[3 * k + 2]};
  *err = closest ? errfun_closest(p, 8, params) : errfun_ordered(p, params);
  return HSH_OK;
}

int hsh_fit_cuboid(const double corners[24], int mode, int arg_order, double params[10], int* steps, double* err) {
  if (!corners || !params) return fail0(HSH_ERR_ARG, "hsh_fit_cuboid: null argument");
  if (arg_order != HSH_FIT_AS_NAMED && arg_order != HSH_FIT_AS_PASSED) return fail0(HSH_ERR_ARG, "hsh_fit_cuboid: bad arg_order");
  D3 p[8];
  for (int k = 0; k < 8; ++k) p[k] = {corners[3 * k], corners[3 * k + 1], corners[3 * k + 2]};
  switch (mode) {
    case HSH_FIT_FROM_CENTER_FIRST: fit_from_center_first(p, arg_order, params, steps, err); break;
    case HSH_FIT_FROM_CENTER: fit_from_center(p, arg_order, params, steps, err); break;
    case HSH_FIT_ORDERED: fit_ordered(p, arg_order, params, steps, err); break;
    default: return fail0(HSH_ERR_ARG, "hsh_fit_cuboid: bad mode");
  }
  return HSH_OK;
}

int hsh_nm_minimize(hsh_objective f, void* user, int n, const double* start, const double* steps, double eps, int maxit, double* x_out,
                    double* f_out, int* iterations) {
  if (!f || n < 1 || !start || !steps || !x_out) return fail0(HSH_ERR_ARG, "hsh_nm_minimize: bad arguments");
  return nm_minimize(f, user, n, start, steps, eps, maxit, x_out, f_out, iterations);
}

int hsh_lstsq_distances(const uint32_t* a, const uint32_t* b, const double* d, int m, uint32_t* nodes, double* pos, int cap, int* n_nodes,
                        double* rmse) {
  if (!a || !b || !d || m < 1 || !n_nodes) return fail0(HSH_ERR_ARG, "hsh_lstsq_distances: bad arguments");
  std::map<std::pair<uint32_t, uint32_t>, double> dist;
  for (int i = 0; i < m; ++i) dist[{a[i], b[i]}] = d[i];
  Placement pl;
  if (!lstsq_distances(dist, pl)) return fail0(HSH_ERR_SINGULAR, "lstSqDistances: singular system");
  *n_nodes = (int)pl.nodes.size();
  if (cap < *n_nodes) return fail0(HSH_ERR_CAPACITY, "hsh_lstsq_distances: capacity");
  for (int i = 0; i < *n_nodes; ++i) {
    if (nodes) nodes[i] = pl.nodes[i];
    if (pos) pos[i] = pl.pos[i];
  }
  if (rmse) *rmse = pl.rmse;
  return HSH_OK;
}

int hsh_group_connected_components(const uint32_t* a, const uint32_t* b, int m, int* comp_out, int* n_comp) {
  if (m < 0 || (m > 0 && (!a || !b || !comp_out)) || !n_comp) return fail0(HSH_ERR_ARG, "hsh_group_connected_components: bad arguments");
  std::vector<int> c;
  int n = 0;
  if (m > 0) connected_components(a, b, m, c, n);
  for (int i = 0; i < m; ++i) comp_out[i] = c[i];
  *n_comp = n;
  return HSH_OK;
}

int hsh_show_float(float v, char* buf, size_t cap) {
  const std::string s = show_float(v);
  if (!buf || cap < s.size() + 1) return fail0(HSH_ERR_CAPACITY, "hsh_show_float: capacity");
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return HSH_OK;
}

int hsh_read_pcd_xyz(const char* path, float* xyz, size_t cap_points, size_t* n_points) {
  if (!path || !n_points) return fail0(HSH_ERR_ARG, "hsh_read_pcd_xyz: null argument");
  std::vector<V3> pts;
  std::string err;
  if (!read_pcd_xyz(path, pts, err)) return fail0(HSH_ERR_IO, err);
  *n_points = pts.size();
  if (!xyz) return HSH_OK;
  if (cap_points < pts.size()) return fail0(HSH_ERR_CAPACITY, "hsh_read_pcd_xyz: capacity");
  if (!pts.empty()) std::memcpy(xyz, pts.data(), sizeof(V3) * pts.size());
  return HSH_OK;
}

int hsh_read_planes_txt(const char* path, float* eq_nd, int cap, int* n) {
  if (!path || !n) return fail0(HSH_ERR_ARG, "hsh_read_planes_txt: null argument");
  std::vector<PlaneEq> eqs;
  std::string err;
  if (!read_planes_txt(path, eqs, err)) return fail0(HSH_ERR_IO, err);
  *n = (int)eqs.size();
  if (!eq_nd) return HSH_OK;
  if (cap < *n) return fail0(HSH_ERR_CAPACITY, "hsh_read_planes_txt: capacity");
  for (int i = 0; i < *n; ++i) eq_nd[4 * i] = eqs[i].n.x, eq_nd[4 * i + 1] = eqs[i].n.y, eq_nd[4 * i + 2] = eqs[i].n.z, eq_nd[4 * i + 3] = eqs[i].d;
  return HSH_OK;
}

int hsh_write_ply_points(const char* path, const float* xyz, size_t n) {
  if (!path || (n > 0 && !xyz)) return fail0(HSH_ERR_ARG, "hsh_write_ply_points: null argument");
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail0(HSH_ERR_IO, std::string("cannot write ") + path);
  std::fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\nend_header\n", n);
  const bool ok = n == 0 || std::fwrite(xyz, 12, n, f) == n;
  std::fclose(f);
  return ok ? HSH_OK : fail0(HSH_ERR_IO, std::string("short write to ") + path);
}

int hsh_read_ply_points(const char* path, float* xyz, size_t cap_points, size_t* n_points) {
  if (!path || !n_points) return fail0(HSH_ERR_ARG, "hsh_read_ply_points: null argument");
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail0(HSH_ERR_IO, std::string("cannot open ") + path);
  char line[256];
  size_t n = 0;
  bool binary = false, header_done = false, in_vertex = false;
  int n_props = 0;
  while (std::fgets(line, sizeof line, f)) {
    if (!std::strncmp(line, "format binary_little_endian", 27)) binary = true;
    if (!std::strncmp(line, "element ", 8)) {
      in_vertex = !std::strncmp(line, "element vertex ", 15);
      if (in_vertex) n = std::strtoull(line + 15, nullptr, 10);
    }
    if (in_vertex && !std::strncmp(line, "property float", 14)) ++n_props;
    if (!std::strncmp(line, "end_header", 10)) {
      header_done = true;
      break;
    }
  }
  if (!header_done || !binary || n_props < 3) {
    std::fclose(f);
    return fail0(HSH_ERR_IO, std::string("unsupported PLY (need binary_little_endian float vertices): ") + path);
  }
  *n_points = n;
  if (!xyz) {
    std::fclose(f);
    return HSH_OK;
  }
  if (cap_points < n) {
    std::fclose(f);
    return fail0(HSH_ERR_CAPACITY, "hsh_read_ply_points: capacity");
  }
  std::vector<float> rec(n_props);
  for (size_t i = 0; i < n; ++i) {
    if (std::fread(rec.data(), 4, n_props, f) != (size_t)n_props) {
      std::fclose(f);
      return fail0(HSH_ERR_IO, std::string("truncated PLY ") + path);
    }
    xyz[3 * i] = rec[0], xyz[3 * i + 1] = rec[1], xyz[3 * i + 2] = rec[2];
  }
  std::fclose(f);
  return HSH_OK;
}

}  // extern "C"
