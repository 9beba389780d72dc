/*
 * kinfu_oracle.h -- CPU restatement of the KinectFusion hot path (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED: the algorithm this file restates lives in an un-vendored, un-pinned third-party
 * repository (github.com/nh2/pcl, branch niklas-experiments, module gpu/kinfu; named only by URL at
 * /root/reference/README.md:13-14).  It is absent from /root/reference, so there are no golden vectors
 * to pin against.  This oracle follows SURVEY.md Appendix A (A.1-A.6) and Newcombe et al., ISMAR 2011,
 * and is pinned instead against (i) the analytic ground truth of the synthetic scene and (ii) an
 * independent numpy restatement (tests/np_twin.py).  See DESIGN.md "Oracle".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or load this file.
 * The product library (housescan_amd/csrc) never includes it.
 *
 * Boundary shapes that ARE in the reference and are honoured here:
 *   depth frame  : row-major uint16, i = y*w + x, 0 = invalid   (housescan/HoniHelper.hs:20,34-36,45-46;
 *                                                                housescan/Main.hs:1297-1300)
 *   point clouds : packed float32 xyz, 12 B/point               (housescan/Main.hs:120,641)
 *   4x4 poses    : row-major, left-multiplicative p' = M p      (housescan/Main.hs:2271-2302)
 */
#ifndef KINFU_ORACLE_H
#define KINFU_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_DIVISOR 32767
#define ORA_MAX_WEIGHT 128
#define ORA_LEVELS 3
#define ORA_KEY_NONE 0x7fffffff

typedef struct {
  int vol[3];          /* X, Y, Z voxels                                        (A.1) */
  float size[3];       /* metres                                                (A.1) */
  float trunc;         /* requested truncation distance; clamped by ora_tau()   (A.1) */
  int W, H;            /* depth image                                           */
  float fx, fy, cx, cy;
  int icp_iters[3];    /* level 0 (finest) .. 2                                 (A.1) */
  float dist_thresh;   /* 0.10 m                                                (A.1) */
  float angle_thresh;  /* sin(20 deg)                                           (A.1) */
  float move_thresh;   /* integration gate, 0 => always                         (A.2) */
  float init_R[9];     /* row-major cam->world                                  */
  float init_t[3];
} ora_config;

void ora_default_config(ora_config* c, int vol_n);
float ora_tau(const ora_config* c);

/* ---- stage functions (stateless; caller owns all arrays) ---- */
void ora_scale_depth(const uint16_t* depth, int W, int H, float fx, float fy, float cx, float cy, float* out);

/* volume = int16 pairs (tsdf, weight), x fastest; `vol` holds planes [zs0, zs0+nzs) of a vol[2]-plane volume.
 * Returns the number of voxels rewritten (V_upd of SURVEY.md 8(d)). */
uint64_t ora_integrate(int16_t* vol, const int dims[3], const float size[3], float tau, int zs0, int nzs,
                       const float* scaled, int W, int H, float fx, float fy, float cx, float cy,
                       const float R[9], const float t[3]);

void ora_bilateral(const uint16_t* src, int W, int H, uint16_t* dst);
void ora_pyrdown(const uint16_t* src, int W, int H, uint16_t* dst /* W/2 x H/2 */);
void ora_vmap(const uint16_t* depth, int W, int H, float fx, float fy, float cx, float cy, float* vmap /*3*H*W*/);
void ora_nmap(const float* vmap, int W, int H, float* nmap);
void ora_transform_maps(const float* vsrc, const float* nsrc, int W, int H, const float R[9], const float t[3],
                        float* vdst, float* ndst);
void ora_resize_vmap(const float* src, int W, int H, float* dst /* W/2 x H/2 */);
void ora_resize_nmap(const float* src, int W, int H, float* dst);

/* 27 sums: upper triangle of J^T J (21) interleaved with J^T r (6), order (0,0)..(0,6),(1,1)..(1,6),...,(5,5),(5,6).
 * Rows [row0,row1) of the current level only (row-sharded ICP); pass 0,H for all. Returns #valid pixels. */
uint64_t ora_icp_accumulate(const float* vcur, const float* ncur, const float* vprev_g, const float* nprev_g,
                            int W, int H, float fx, float fy, float cx, float cy,
                            const float R[9], const float t[3], const float Rprev[9], const float tprev[3],
                            float dist_thresh, float angle_thresh, int row0, int row1, double out27[27]);
/* returns 1 if solved, 0 if the system is singular / NaN (tracking lost) */
int ora_icp_solve(const double in27[27], float x6[6]);
void ora_pose_update(float R[9], float t[3], const float x6[6]);
void ora_sincos(double x, double* s, double* c);

/* Raycast.  `vol` holds planes [zs0, zs0+nzs); march steps are OWNED by the slab whose owned range
 * [zo0, zo1) contains the z-plane of the step's far sample (single device: zs0=0,nzs=Z,zo0=0,zo1=Z).
 * keys[i] = (step<<1)|type for the terminating event found by this slab (type 0 = surface hit, 1 = abort),
 * ORA_KEY_NONE otherwise.  vmap/nmap get NaN wherever this slab did not find a hit. */
void ora_raycast(const int16_t* vol, const int dims[3], const float size[3], float tau, int zs0, int nzs,
                 int zo0, int zo1, int W, int H, float fx, float fy, float cx, float cy,
                 const float R[9], const float t[3], float* vmap, float* nmap, int32_t* keys,
                 uint64_t* n_steps /* optional: total march steps that touched memory */);

/* TSDF zero-crossing cloud extraction (A.7); returns number of points written (<= cap). */
size_t ora_extract_cloud(const int16_t* vol, const int dims[3], const float size[3], float* xyz, size_t cap);
size_t ora_extract_mesh(const int16_t* vol, const int dims[3], const float size[3], float* tri, size_t cap);

/* ---- whole tracker (A.2) ---- */
typedef struct ora_tracker ora_tracker;
ora_tracker* ora_tracker_create(const ora_config* c);
void ora_tracker_destroy(ora_tracker* k);
void ora_tracker_reset(ora_tracker* k);
/* returns 1 if the frame was tracked, 0 for the first frame or when tracking was lost (volume reset) */
int ora_tracker_process(ora_tracker* k, const uint16_t* depth, float pose16[16]);
const int16_t* ora_tracker_volume(const ora_tracker* k);
const float* ora_tracker_model_vmap(const ora_tracker* k, int level);
const float* ora_tracker_model_nmap(const ora_tracker* k, int level);
void ora_tracker_stage_seconds(const ora_tracker* k, double out4[4]); /* preprocess, icp, integrate, raycast (cumulative) */
uint64_t ora_tracker_last_vupd(const ora_tracker* k);

#ifdef __cplusplus
}
#endif
#endif
