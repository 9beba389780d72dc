/* hshouse.h -- C ABI of the host-side room-stitching chain (BASELINE.json configs[0]).
 *
 * This is NOT part of the GPU hot path.  It restates, in C++ behind a plain C ABI, the CPU chain HouseScan runs
 * on KinFu's products: load room directory -> orient -> plane corners -> two-stage Nelder-Mead cuboid fit ->
 * wall connections -> connected components -> per-axis least squares -> one 4x4 per room (.xf / CSV).
 * Every entry point names the reference function it mirrors (file:line under housescan/).
 *
 * Conventions kept from the reference:
 *   - PlaneEq (n, d) means n.x = d with unit n (Main.hs:1354-1361); PCL's planes.txt form is ax+by+cz+d=0, so d
 *     is negated on load (Main.hs:1383-1385).
 *   - room geometry is IEEE binary32 ("Float"); the cuboid fit and the least squares are binary64 ("Double").
 *   - matrices handed across this ABI are ROW-MAJOR, LEFT-multiplicative (p' = M p), the form
 *     roomProjectionToString exports (Main.hs:2271-2284); internally the projection is accumulated in the
 *     reference's right-multiplicative form and transposed on export.
 *   - IDs are uint32 handed out by one counter per house (genID, Main.hs:355-357).
 *
 * Third-party algorithms restated because the dependency is absent from /root/reference (parity UNPINNED):
 *   - GSL multimin `nmsimplex2` (gsl >= 1.12, multimin/simplex2.c) as driven by hmatrix-gsl's `minimize`;
 *   - LAPACK dgesv / dgels / dsyev as reached through hmatrix (`safeLinearSolve`, `safeLinearSolveLS`, `eigSH`);
 *     the reference's own wrapper module HmatrixUtils is missing from the repository (Main.hs:65).
 *
 * All functions return 0 on success or a negative code; the text is in hsh_last_error().  Thread-safety: one
 * house per thread; distinct houses are independent.
 */
#ifndef HSHOUSE_H
#define HSHOUSE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hsh_house hsh_house;

enum { HSH_OK = 0, HSH_ERR_ARG = -1, HSH_ERR_IO = -2, HSH_ERR_STATE = -3, HSH_ERR_SINGULAR = -4, HSH_ERR_CAPACITY = -5 };
enum { HSH_AXIS_X = 0, HSH_AXIS_Y = 1, HSH_AXIS_Z = 2 };
enum { HSH_WALL_OPPOSITE = 0, HSH_WALL_SAME = 1 };             /* WallRelation, Main.hs:339-342 */
/* how the named arguments of the reference's `minimize` calls are interpreted (FitCuboidBFGS.hs:184, :201, :233):
 *   AS_NAMED: `initial` is the start point and `initialSearchBox` the step sizes (what the names say);
 *   AS_PASSED: positional order of hmatrix-gsl's `minimize method eps maxit sizes f start` (recalled signature):
 *              the value named `initial` is used as the step sizes and `initialSearchBox` as the start. */
enum { HSH_FIT_AS_NAMED = 0, HSH_FIT_AS_PASSED = 1 };
enum { HSH_FIT_FROM_CENTER_FIRST = 0, HSH_FIT_FROM_CENTER = 1, HSH_FIT_ORDERED = 2 };

hsh_house* hsh_create(void);
void hsh_destroy(hsh_house* h);
const char* hsh_last_error(const hsh_house* h); /* h may be NULL: error of the last house-less call on this thread */

/* ---- rooms ---------------------------------------------------------------------------------------------- */
/* loadRoom (Main.hs:1738-1762): <dir>/cloud_downsampled.pcd, planes.txt, cloud_plane_hull<k>.pcd; plane normals
 * are made inward facing; roomProj = identity. */
int hsh_load_room(hsh_house* h, const char* dir, uint32_t* room_id);
/* the same from memory: planes_abcd in PCL file form (ax+by+cz+d=0), hull k = hull_xyz[hull_offsets[k]..[k+1]) */
int hsh_add_room(hsh_house* h, const char* name, const float* cloud_xyz, size_t n_points, const float* planes_abcd,
                 int n_planes, const float* hull_xyz, const int* hull_offsets, uint32_t* room_id);
int hsh_room_ids(const hsh_house* h, uint32_t* ids, int cap, int* n);
int hsh_room_planes(const hsh_house* h, uint32_t room, uint32_t* plane_ids, float* eq_nd /* 4 per plane: n, d */, int cap, int* n);
int hsh_plane_bounds(const hsh_house* h, uint32_t plane, float* xyz, int cap_points, int* n);
int hsh_room_corners(const hsh_house* h, uint32_t room, int suggested, uint32_t* ids, float* xyz, int cap, int* n);
int hsh_room_cloud(const hsh_house* h, uint32_t room, float* xyz, size_t cap_points, size_t* n);
int hsh_room_means(const hsh_house* h, uint32_t room, float cloud_mean[3], float corner_mean[3]); /* roomMean :1612, cornerMean :2177 */
int hsh_set_room_corners(hsh_house* h, uint32_t room, const float* xyz, int n);   /* zipGenIDs + roomCorners := */
int hsh_accept_corner_suggestion(hsh_house* h, uint32_t room, uint32_t suggestion_id); /* Main.hs:1541-1545 */

/* ---- rigid edits (each also accumulates roomProj) ------------------------------------------------------------- */
int hsh_translate_room(hsh_house* h, uint32_t room, const float off[3]);          /* translateRoom :1700-1709 */
int hsh_rotate_room(hsh_house* h, uint32_t room, const float rot_right[9]);       /* rotateRoom :1677-1678; row-vector matrix */
int hsh_rotate_kinfu_room(hsh_house* h, uint32_t room);                           /* :1733-1735, 180 deg about X */
int hsh_room_auto_align_axis(hsh_house* h, uint32_t room, const float axis[3]);   /* roomAutoAlignAxis :1894-1905 */
int hsh_auto_align_floor(hsh_house* h, uint32_t room);                            /* autoAlignFloor :1909-1910 */
int hsh_remove_ceiling(hsh_house* h, uint32_t room);                              /* removeCeiling :2643-2665 */

/* ---- corners and cuboid --------------------------------------------------------------------------------------- */
/* suggestPoints (Main.hs:1522-1538).  adopted = 1 when the room had no corners and exactly 8 were suggested. */
int hsh_suggest_points(hsh_house* h, uint32_t room, float cutoff_factor /* 1.2, Main.hs:1084 */, int* n_suggested, int* adopted);
/* fitCuboidToRoom (Main.hs:1814-1847): needs exactly 8 corners; replaces corners and planes by the cuboid's. */
int hsh_fit_cuboid_to_room(hsh_house* h, uint32_t room, int arg_order, int* steps, double* rmse, double params[10]);

/* ---- wall connections and placement ---------------------------------------------------------------------------- */
int hsh_connect_walls(hsh_house* h, uint32_t plane1, uint32_t plane2, int relation, float thickness, int* connected); /* :2019-2052 */
int hsh_disconnect_walls(hsh_house* h, uint32_t plane1, uint32_t plane2);         /* :2055-2071 */
int hsh_connected_walls(const hsh_house* h, int* axis, int* relation, float* thickness, uint32_t* plane1, uint32_t* plane2, int cap, int* n);
int hsh_optimize_room_positions(hsh_house* h, double rmse_xyz[3] /* worst component RMSE per axis, NaN if untouched */); /* :2074-2162 */

/* ---- export ---------------------------------------------------------------------------------------------------- */
int hsh_room_projection(const hsh_house* h, uint32_t room, float m[16]);          /* transpose(fromProjective roomProj) */
int hsh_room_projection_string(const hsh_house* h, uint32_t room, int xf_format, char* buf, size_t cap); /* :2271-2302 */
int hsh_export_all_room_xf_files(const hsh_house* h, const char* dir);            /* :2316-2325, <dir>/<room>.xf */

/* ---- the numerics on their own (no house) ---------------------------------------------------------------------- */
int hsh_plane_corner(const float eq_nd[12], float corner[3], int* found);         /* planeCorner :1413-1430 */
int hsh_fit_plane(const float* xyz, int n, float eq_nd[4]);                        /* fitPlane :1436-1450 */
int hsh_rotation_between(const float n1[3], const float n2[3], float rot_right[9]); /* rotationBetweenPlaneEqs :1553-1560 */
int hsh_cuboid_from_params(const double params[10], double corners[24]);          /* FitCuboidBFGS.hs:98-112 */
int hsh_guess_dims(const double corners[24], double abc[3]);                       /* FitCuboidBFGS.hs:247-252 */
int hsh_errfun(const double corners[24], const double params[10], int closest, double* err); /* :51-65, :68-76 */
int hsh_fit_cuboid(const double corners[24], int mode, int arg_order, double params[10], int* steps, double* err); /* :172-233 */
/* GSL nmsimplex2 as driven by hmatrix-gsl `minimize`: stops when size < eps or after maxit iterations */
typedef double (*hsh_objective)(const double* x, int n, void* user);
int hsh_nm_minimize(hsh_objective f, void* user, int n, const double* start, const double* steps, double eps, int maxit,
                    double* x_out, double* f_out, int* iterations);
/* lstSqDistances (TranslationOptimizer.hs:36-72): edges (a_i, b_i, d_i); duplicate (a,b) keys: the last one wins;
 * nodes come back in index order (node 0 = first node of the smallest key, fixed at 0). */
int hsh_lstsq_distances(const uint32_t* a, const uint32_t* b, const double* d, int m, uint32_t* nodes, double* pos, int cap,
                        int* n_nodes, double* rmse);
/* groupConnectedComponents (GroupConnectedComponents.hs:16-54): comp[i] = component of edge i, numbered by first appearance */
int hsh_group_connected_components(const uint32_t* a, const uint32_t* b, int m, int* comp, int* n_comp);
int hsh_show_float(float v, char* buf, size_t cap);                               /* Haskell `show :: Float -> String` */

/* ---- files ------------------------------------------------------------------------------------------------------ */
int hsh_read_pcd_xyz(const char* path, float* xyz, size_t cap_points, size_t* n_points); /* ascii or binary, float x y z */
int hsh_read_planes_txt(const char* path, float* eq_nd, int cap, int* n);          /* planeEqsFromFile :1379-1389 */
int hsh_write_ply_points(const char* path, const float* xyz, size_t n);            /* binary_little_endian vertices */
int hsh_read_ply_points(const char* path, float* xyz, size_t cap_points, size_t* n_points);

#ifdef __cplusplus
}
#endif
#endif
