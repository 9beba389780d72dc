/*
 * hskinfu.h -- C ABI of the MI355X-native KinectFusion core for HouseScan.
 *
 * The reference program (nh2/housescan) holds no KinectFusion code and no FFI; it exchanges FILES with an
 * external PCL KinFu fork (/root/reference/README.md:13-14).  This header defines the seam the north_star
 * asks for: a thin C ABI the Haskell host loop would bind with `foreign import ccall`.  Each entry point
 * cites the reference interface whose data shape it honours or whose role it replaces:
 *
 *   depth frames in   : `takeDepthSnapshot :: IO (Either String (Vector Word16, (Int, Int)))`
 *                       housescan/HoniHelper.hs:20-36 -- row-major uint16, i = y*w + x (Main.hs:1297-1300),
 *                       0 = invalid (Main.hs:1297); errors are values, never exceptions (HoniHelper.hs:39-42).
 *   clouds out        : `Cloud { cloudPoints :: Vector Vec3 }` packed float32 xyz, 12 B/point
 *                       housescan/Main.hs:117-121, :641, :792; consumed by addPointCloud Main.hs:806.
 *   poses / transforms: 16 floats row-major, LEFT-multiplicative (p' = M p) -- the form HouseScan exports
 *                       in roomProjectionToString / roomProjectionToXfFormat, Main.hs:2271-2302.
 *   products on disk  : cloud_downsampled.pcd, cloud_bin.pcd (Main.hs:1740, :1334-1345, :2437).
 *
 * Conventions: every call returns HSK_OK (0) or a negative error code; the message is available from
 * hsk_last_error (maps to Haskell `Left String`).  Tracking loss is NOT an error: *tracked = 0 and the
 * volume is reset.  A context is not re-entrant; distinct contexts are independent and
 * may be driven from different OS threads.  The library never retains caller pointers past the call.
 * No torch / C++ types cross this boundary.
 */
#ifndef HSKINFU_H
#define HSKINFU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSK_OK 0
#define HSK_ERR_ARG (-1)
#define HSK_ERR_HIP (-2)
#define HSK_ERR_STATE (-3)
#define HSK_ERR_NOGPU (-4)
#define HSK_ERR_TIMEOUT (-5) /* a pipelined frame did not report within HSK_FRAME_TIMEOUT_S seconds (environment, default 20) */

#define HSK_LEVELS 3
#define HSK_KEY_NONE 0x7fffffff

typedef struct hsk_ctx hsk_ctx; /* opaque; one per volume / room */

typedef struct {
  int vol_x, vol_y, vol_z;      /* voxels, e.g. 256 / 512 / 1024; vol_x and vol_y multiples of 8        */
  float vol_size_m[3];          /* metric extent, default 3 x 3 x 3                                    */
  float trunc_dist_m;           /* default 0.03; clamped to >= 2.1 * max cell                           */
  int width, height;            /* depth image, default 640 x 480 (the shape HoniHelper.hs:34-36 returns) */
  float fx, fy, cx, cy;         /* default 525, 525, 319.5, 239.5                                       */
  int icp_iters[HSK_LEVELS];    /* level 0 (finest) .. 2, default {10, 5, 4}; run coarsest first        */
  float icp_dist_thresh_m;      /* 0.10                                                                 */
  float icp_angle_thresh_sin;   /* sin(20 deg)                                                          */
  float integrate_move_thresh;  /* 0 => integrate every frame                                           */
  float init_pose[16];          /* row-major cam->world; default R = I, t = (1.5, 1.5, -0.3)            */
  int device_id;                /* HIP device ordinal                                                   */
  /* z-slab sharding (multi-GPU): this context stores planes [own_z0 - halo, own_z1 + halo) clipped to the
   * volume and OWNS raycast steps whose far sample lies in [own_z0, own_z1).  Single device: 0, vol_z, 0. */
  int own_z0, own_z1, halo;
  int use_graph;                /* 1 = replay synchronous frames from one hipGraph; 0 (default) = eager;
                                   2 = the main-stream chain of PIPELINED frames (hsk_submit_frame*) from one hipGraph per
                                   image-buffer set: 23 launches become one -- for hosts that scan several rooms at once */
} hsk_config;

/* identity of the sources this library was built from (first 16 hex digits of their sha256; "+exp" appended when it
 * was built with other than the default compiler flags) */
const char* hsk_build_id(void);

/* fills *c with the defaults above for an n^3 volume */
void hsk_default_config(hsk_config* c, int n);

int hsk_create(const hsk_config* c, hsk_ctx** out);
void hsk_destroy(hsk_ctx* k);
int hsk_reset(hsk_ctx* k);
const char* hsk_last_error(const hsk_ctx* k); /* k may be NULL: last create error */

/* Whole tracker step.  `depth` is caller-owned row-major uint16 millimetres (HoniHelper.hs:20; index
 * convention Main.hs:1298-1300), read-only, may be freed on return.  pose_out: row-major,
 * left-multiplicative cam->world (Main.hs:2278-2284). */
int hsk_process_frame(hsk_ctx* k, const uint16_t* depth, int w, int h, float pose_out[16], int* tracked);
/* Same, the depth frame already resident in device memory (HBM) */
int hsk_process_frame_dev(hsk_ctx* k, const void* depth_dev, int w, int h, float pose_out[16], int* tracked);

/* Asynchronous form of the tracker step (throughput): hsk_submit_frame_dev enqueues a frame and returns at once,
 * hsk_wait_frame returns the pose of the OLDEST submitted frame (FIFO).  Up to HSK_MAX_IN_FLIGHT frames may be
 * outstanding, so the GPU runs frame k+1 while the host reads frame k's pose.  After a tracking loss the frames
 * already in flight are dropped (tracked = 0) and the volume is reset before the next submission.  The depth copy
 * and preprocessing of a submitted frame run on a second stream, overlapped with the previous frame; depth_dev must
 * therefore stay valid until hsk_wait_frame has returned that frame.  Its CONTENTS may still be in the making on a
 * stream the context has adopted through hsk_set_stream (an upload or a conversion kernel enqueued there): the second
 * stream is ordered behind everything enqueued on the adopted stream at the time of the call.  Work on any other
 * stream (and any work when the context runs on its own stream) must have completed before the call.
 * WHAT "WAITED" MEANS: hsk_wait_frame returns as soon as the frame's POSE and verdict are final -- when its ICP has ended.
 * The frame's integrate and raycast may still be running on hsk_stream() at that point.  Every call of this library is
 * ordered behind them on that stream, so callers that only use the library see nothing of it; a caller that takes
 * "waited" for "the GPU is idle" or "the volume is current" -- to stop a clock, or to touch the volume or the model maps
 * from another stream -- must call hsk_synchronize() first.  A frame that does not report within HSK_FRAME_TIMEOUT_S
 * seconds (environment, default 20) makes hsk_wait_frame -- and the synchronous hsk_process_frame[_dev], which go through
 * it -- return HSK_ERR_TIMEOUT. */
#define HSK_MAX_IN_FLIGHT 3
int hsk_submit_frame_dev(hsk_ctx* k, const void* depth_dev, int w, int h);
int hsk_submit_frame(hsk_ctx* k, const uint16_t* depth, int w, int h); /* host frame; copied before the call returns */
int hsk_wait_frame(hsk_ctx* k, float pose_out[16], int* tracked);

/* Stage-level entry points: exist so parity tests and rocprof can isolate each kernel. */
int hsk_integrate(hsk_ctx* k, const uint16_t* depth, int w, int h, const float pose[16]);
int hsk_raycast(hsk_ctx* k, const float pose[16], float* vmap /* 3*h*w SoA */, float* nmap, int32_t* keys /* may be NULL */);
int hsk_preprocess(hsk_ctx* k, const uint16_t* depth, int w, int h); /* bilateral, pyramid, vertex/normal maps */
/* 27 sums for pose estimate `pose_est` against the stored model maps and previous pose, rows [row0,row1) */
int hsk_icp_accumulate(hsk_ctx* k, int level, const float pose_est[16], int row0, int row1, double out27[27]);
int hsk_icp_solve(const double in27[27], float x6[6], int* ok);       /* host mirror of the device solve */
int hsk_count_updates(hsk_ctx* k, const uint16_t* depth, int w, int h, const float pose[16], uint64_t* n_upd);

int hsk_download_tsdf(hsk_ctx* k, int16_t* tsdf_weight_pairs /* 2 * X*Y*stored_planes, x fastest */);
int hsk_upload_tsdf(hsk_ctx* k, const int16_t* tsdf_weight_pairs);
/* The weights of deep free space are kept in side tables (one byte per 16 voxels, one per 2048) and written back into the
 * volume when something is about to read them: hsk_download_tsdf does it itself.  (The products -- hsk_extract_cloud,
 * hsk_extract_mesh[_cubes] -- ask of a weight only whether it is zero, which no deferred weight is, and do not.)  This call
 * does only that write-back (enqueued; no host synchronisation) -- it changes nothing any call returns, and exists so
 * that the deferred work can be timed on its own (bench.py: readout_ms). */
int hsk_flush_weights(hsk_ctx* k);
int hsk_stored_planes(const hsk_ctx* k, int* z0, int* nz);
int hsk_get_pose(hsk_ctx* k, float pose[16]);
int hsk_set_pose(hsk_ctx* k, const float pose[16]);
/* kind: 0 current vertex, 1 current normal, 2 model vertex, 3 model normal; out = 3*(h>>level)*(w>>level) floats */
int hsk_download_map(hsk_ctx* k, int kind, int level, float* out);
int hsk_upload_map(hsk_ctx* k, int kind, int level, const float* in);
int hsk_download_depth_level(hsk_ctx* k, int level, uint16_t* out); /* filtered pyramid */
int hsk_download_scaled_depth(hsk_ctx* k, float* out);

/* Everything a read-out allocates on its first use -- the pinned staging pair (2 x 32 MiB, or a plane of the volume if that is
 * more), the row tables of the count pass, the marching-cubes table, a product buffer of `product_bytes` (0: 48 MiB, a scan's
 * cloud and cubes mesh at 512^3) -- made NOW.  A host that shows a cloud from its GL thread (addPointCloud, Main.hs:806)
 * calls this once from the worker thread that created the context: the first hsk_extract_* then costs what every later
 * one does (it was 7 ms against 0.7).  Optional; idempotent; the buffers still grow on demand. */
int hsk_prepare_readout(hsk_ctx* k, size_t product_bytes);
/* TSDF zero-crossing cloud: packed float32 xyz (the layout of Cloud.cloudPoints, Main.hs:120) in voxel order. */
int hsk_extract_cloud(hsk_ctx* k, float* xyz, size_t cap_points, size_t* n_points);
/* TSDF zero level set as a triangle soup, 9 floats per triangle, marching tetrahedra (6 Kuhn tetrahedra per cube),
 * deterministic voxel order, normals towards free space.  Edge vertices shared by neighbouring cubes are bit-identical,
 * so hsk_write_ply_mesh can weld them by exact comparison.  Two-call protocol like hsk_extract_cloud. */
int hsk_extract_mesh(hsk_ctx* k, float* tri_xyz, size_t cap_triangles, size_t* n_triangles);
/* the same level set by MARCHING CUBES, the form the .ply of upstream's KinFu export has (/root/reference/README.md:16-17):
 * about half the triangles.  PCL's 256-case table is not in the reference; this one is generated (segments between cut
 * edges face by face, ambiguous faces cut one inside corner off each, loops fanned from their lowest edge) and has the
 * classic table's counts (820 triangles, at most 5 per cube); same validity rule, vertices and order as hsk_extract_mesh */
int hsk_extract_mesh_cubes(hsk_ctx* k, float* tri_xyz, size_t cap_triangles, size_t* n_triangles);

/* Multi-GPU (z-slab) building blocks; device pointers so that the host's collective (RCCL through
 * torch.distributed) can run on them without a host round trip.  All work is enqueued on hsk_stream(). */
int hsk_mgpu_frame_begin(hsk_ctx* k, const void* depth_dev, int w, int h); /* preprocess + (frame 0) transform */
/* optional: enqueue the copy + preprocessing of the NEXT frame on a second stream, under the current frame's work; the
 * next hsk_mgpu_frame_begin with the same pointer picks it up.  depth_dev: device memory or pinned host memory holding
 * a COMPLETE frame (host-synchronised: this call orders against no stream); its contents must not change until that
 * hsk_mgpu_frame_begin has been enqueued */
int hsk_mgpu_prefetch(hsk_ctx* k, const void* depth_dev, int w, int h);
/* frame_begin + icp_replicated + integrate + raycast_local in one call; everything after the preprocessing is replayed
 * from a hipGraph (config use_graph).  keys_dev: int32[h*w], the SAME buffer on every call */
int hsk_mgpu_frame_front(hsk_ctx* k, const void* depth_dev, int w, int h, void* keys_dev);
int hsk_mgpu_icp_accumulate(hsk_ctx* k, int level, int row0, int row1, void* sums27_dev /* double[27] */);
int hsk_mgpu_icp_update(hsk_ctx* k, const void* sums27_dev);                /* solve + pose update on device */
int hsk_mgpu_icp_replicated(hsk_ctx* k); /* the whole 19-iteration ICP on this rank's (composited) maps, fused kernels */
int hsk_mgpu_integrate(hsk_ctx* k);
int hsk_mgpu_raycast_local(hsk_ctx* k, void* keys_dev /* int32[h*w] */);    /* slab-local march */
int hsk_mgpu_raycast_resolve(hsk_ctx* k, const void* keys_min_dev, void* maps_bits_dev /* int32[6*h*w] */);
/* direct exchange (no collective): where this slab won the pixel (keys_min == its own key, a hit), the bit patterns of its
 * vertex / normal go straight into each of the n_dest (<= 16) composite buffers dest_bits[d] (int32[6*h*w], memory this
 * context's device can write: its own, a peer's with access enabled, or an IPC-mapped one); nothing is written elsewhere */
int hsk_mgpu_raycast_push(hsk_ctx* k, const void* keys_min_dev, void* const* dest_bits, int n_dest);
int hsk_mgpu_frame_end(hsk_ctx* k, const void* keys_min_dev, const void* maps_bits_dev, float pose_out[16], int* tracked);
int hsk_mgpu_frame_index(const hsk_ctx* k);
/* pipelined form of the frame end: queues the pose read-back instead of waiting; collect with hsk_wait_frame (in order,
 * at most HSK_MAX_IN_FLIGHT outstanding).  hsk_mgpu_restart_pending() == 1: the next frame (re)starts the scan and must
 * end with the synchronous hsk_mgpu_frame_end (frame 0, or a pipelined frame lost tracking). */
int hsk_mgpu_frame_end_async(hsk_ctx* k, const void* keys_min_dev, const void* maps_bits_dev);
int hsk_mgpu_restart_pending(const hsk_ctx* k);

/* ---- ONE volume sharded as z-slabs over several GPUs, behind one frame call (SURVEY.md 8(b) n_devices / device_ids,
 * 8(e); BASELINE.json configs[3]).  The host keeps feeding whole depth frames (HoniHelper.hs:20, the loop that would
 * replace Main.hs:1285-1290); slab s of n owns planes [s Z / n, (s + 1) Z / n) plus a redundantly integrated halo, and
 * the per-frame exchanges -- MIN of the raycast's step keys, SUM of the winning vertex / normal bit patterns, optionally
 * the 27 ICP sums of every iteration -- run inside the library: a kernel between slabs that share a device,
 * ncclAllReduce (RCCL over xGMI, loaded at run time) between devices.  Results are bit-identical to a single context.
 * A group is not re-entrant; it is driven from one host thread. */
typedef struct hsk_group hsk_group;
#define HSK_GROUP_FORCE_RCCL 1     /* run the collectives through RCCL even when the group has a single device / rank */
#define HSK_GROUP_ICP_ALLREDUCE 2  /* row-shard the ICP over the slabs and all-reduce its 27 sums every iteration
                                      (default: every slab runs the whole ICP on the composited maps, no collective) */
#define HSK_GROUP_DIRECT 4         /* the per-frame composites as a ONE-HOP exchange over peer-mapped memory instead of
                                      two all-reduces: every slab stores its step keys into a slot of every device's
                                      gather buffer, each device takes the MIN locally, the winner of a pixel stores its
                                      vertex / normal bits into every device's composite (24 B per won pixel and peer
                                      instead of a 7.4 MB all-reduce); stream wait / write-value operations on a shared
                                      flag page order the steps -- no RCCL, no spinning kernel.  Single process: peer
                                      access between the devices; rank form: hipIpc memory handles and a POSIX shared-
                                      memory page named after comm_id (one node).  Not with HSK_GROUP_ICP_ALLREDUCE. */
#define HSK_GROUP_PROFILE 8        /* HSK_GROUP_DIRECT: HIP events round the slab work and the exchange of every frame
                                      (hsk_group_exchange_ms) */
/* single process: slab s lives on device_ids[s] (a device may be named several times); the distinct devices form one
 * communicator.  c->device_id, own_z0, own_z1, halo and use_graph are set by the library. */
int hsk_group_create(const hsk_config* c, int n_slabs, const int* device_ids, int flags, hsk_group** out);
/* one process per GPU (c->device_id): rank r of `world` owns slab r.  comm_id: the 128 bytes hsk_group_unique_id() gave
 * ONE of the ranks, handed to all of them by the host's own means (ignored when world == 1) */
int hsk_group_unique_id(void* id128);
int hsk_group_create_rank(const hsk_config* c, int rank, int world, const void* comm_id, int flags, hsk_group** out);
void hsk_group_destroy(hsk_group* g);
const char* hsk_group_last_error(const hsk_group* g); /* g may be NULL: last create error */
int hsk_group_reset(hsk_group* g);
/* the tracker step of hsk_process_frame, on the sharded volume */
int hsk_group_process_frame(hsk_group* g, const uint16_t* depth, int w, int h, float pose_out[16], int* tracked);
/* pipelined form (as hsk_submit_frame / hsk_wait_frame): the frame and its collectives are enqueued, the pose collected
 * later, in order, at most HSK_MAX_IN_FLIGHT outstanding; the first frame of a (re)started scan completes at submission.
 * After a tracking loss the frames already in flight behind the lost one are dropped (tracked = 0); the submission
 * that follows restarts the scan, its result is handed out behind theirs.
 * A failure in the MIDDLE of a frame (some slabs or devices have taken it, others not) poisons the group: every later
 * call returns HSK_ERR_STATE until hsk_group_reset succeeds (single process) or the group is destroyed (several ranks,
 * RCCL or direct form: the peers may be left in a collective or waiting for a flag).  A peer rank that dies or hangs
 * shows as HSK_ERR_TIMEOUT from hsk_group_wait_frame on every other rank after HSK_FRAME_TIMEOUT_S seconds (environment
 * variable, default 20) -- which poisons the group.  Poisoning lets this rank's own queues drain (direct form: the
 * flags its streams wait for are raised from the host; RCCL form: ncclCommAbort), so hsk_group_destroy returns.
 * STATUS: groups of several slabs on ONE device -- RCCL form, direct form, and the direct form between 2, 3 and 8 OS
 * processes sharing the device -- are tested bit-exact against a single context; groups over more than one DEVICE
 * (ncclCommInitAll / ncclCommInitRank with world > 1, the direct form over xGMI peer mappings) have never run on
 * hardware -- no multi-GPU box was available (tests/test_gpu_multi_device.py runs all three forms when one is, and
 * `bench.py --gpus N` checks every form it times against a single context: "matches_single_gpu"). */
int hsk_group_submit_frame(hsk_group* g, const uint16_t* depth, int w, int h);
/* depth_dev[d]: the frame in the memory of the d-th distinct device of this process (creation order), complete and
 * valid until the frame has been waited for */
int hsk_group_submit_frame_dev(hsk_group* g, const void* const* depth_dev, int w, int h);
int hsk_group_wait_frame(hsk_group* g, float pose_out[16], int* tracked);
/* HSK_GROUP_PROFILE: summed over n_frames tracked frames, on this process's first device: the exchange (waits for the peers
 * included) and the slab's own work before it (ICP + integrate + slab-local raycast) */
int hsk_group_exchange_ms(hsk_group* g, double* sum_ms, double* front_sum_ms, unsigned long long* n_frames);
int hsk_group_n_slabs(const hsk_group* g);            /* slabs held by this process */
/* how many ranks / devices the group's exchange really spans: ncclCommCount of its communicator (RCCL forms), the ranks
 * attached to the shared flag page (direct form between processes), the distinct devices of a single-process group */
int hsk_group_ranks_seen(hsk_group* g, int* n_ranks);
hsk_ctx* hsk_group_slab(hsk_group* g, int i);         /* for hsk_download_map, hsk_extract_cloud, ... on one slab */
/* the planes this process owns, at their place in a full 2 * X * Y * Z array (other planes are left untouched) */
int hsk_group_download_tsdf(hsk_group* g, int16_t* full_tsdf_weight_pairs);

/* streams / profiling */
void* hsk_stream(hsk_ctx* k);                 /* hipStream_t the context launches on */
int hsk_set_stream(hsk_ctx* k, void* stream); /* adopt the caller's hipStream_t (e.g. torch's current stream) */
int hsk_synchronize(hsk_ctx* k);
#define HSK_STAGE_PRE 0
#define HSK_STAGE_ICP 1
#define HSK_STAGE_INTEGRATE 2
#define HSK_STAGE_RAYCAST 3
#define HSK_NSTAGES 4
int hsk_set_profiling(hsk_ctx* k, int on);    /* 1: record HIP events around each stage of process_frame; 2: also at every ICP level */
int hsk_stage_ms(hsk_ctx* k, double sum_ms[HSK_NSTAGES], uint64_t* n_frames, int reset);
/* while profiling at level 2: time of the ICP iterations of each pyramid level, summed over the same frames as hsk_stage_ms (read it
 * before resetting that); index = level, 0 = finest.  Divide by frames x icp_iters[level] for the time of an iteration. */
int hsk_icp_level_ms(hsk_ctx* k, double sum_ms[HSK_LEVELS]);
/* lane-blocks (4 x 1 x 4 voxels) the last integrate's classification pass could not settle and handed to its per-voxel
 * pass (a measure of the classification's slack: bench.py reports it beside V_upd); synchronises the context's stream */
/* host microseconds the pipelined submissions (hsk_submit_frame[_dev]) have spent so far, by phase: [0] the copy of a host frame
   into the pinned staging ring, [1] enqueueing the upload and the preprocessing on the second stream, [2] waiting for that
   preprocessing, [3] enqueueing the frame's main-stream chain; n_submissions: how many.  reset != 0: counted afresh from now. */
int hsk_submit_host_us(hsk_ctx* k, double sum_us[4], uint64_t* n_submissions, int reset);
int hsk_integrate_queue_entries(hsk_ctx* k, uint64_t* n_entries);
/* ... and the lane-blocks of the last integrate's LIGHT class: free space with holes in the depth image under it (each voxel is
   rewritten with F = 1 or left alone according to whether its pixel has depth); not counted by hsk_integrate_queue_entries */
int hsk_integrate_light_entries(hsk_ctx* k, uint64_t* n_entries);
/* the coarse level of the last integrate (one verdict per wave-chunk of 16 x 16 voxels x the pass-A chunk of planes):
 * counts[0] = chunks pass A had to classify lane-block by lane-block ("mixed"), [1] = chunks settled as a whole (outside the
 * frustum, wholly occluded, or wholly free space with the observation recorded in the chunk's byte), [2] = wholly free
 * chunks whose byte could not take it (pass A works them), [3] = chunks currently marked quiet; synchronises the stream */
int hsk_integrate_coarse_counts(hsk_ctx* k, uint64_t counts[4]);
int hsk_bilateral_tables(float ws[169], float wc[512]);
/* Exhaustive self-test, on the GPU itself, of the exact-arithmetic shortcuts the kernels use for the specification's
 * correctly rounded 1/x, sqrt(x) and a/n (hardware approximation + one fused correction step; hsk_dev.h): every binary32
 * value, every (a, n) pair of the domain.  counts: [0] 1/x values checked, [1] wrong, [2] wrong without the correction
 * (shows that the comparison bites), [3..5] the same for sqrt, [6] a/n pairs checked, [7] wrong.  About 0.3 s. */
int hsk_selftest_exact_ops(int device_id, uint64_t counts[8]);

/* Deterministic synthetic depth stream; host-only, no GPU needed. */
int hsk_synth_pose(int frame, float pose[16]);
int hsk_synth_render(const float pose[16], int w, int h, float fx, float fy, float cx, float cy, uint16_t* depth);
/* closed box rooms with furniture for the room-stitching configurations (BASELINE configs[0], [4]); variant 0..3 */
int hsk_synth_room_extents(int variant, float extents[6] /* x0 x1 y0 y1 z0 z1 */);
int hsk_synth_room_pose(int variant, int frame, int n_frames, float pose[16]); /* three turns from near the centre: level, up, down */
int hsk_synth_room_render(int variant, const float pose[16], int w, int h, float fx, float fy, float cx, float cy, uint16_t* depth);
/* The same scenes as a structured-light sensor returns them (housescan/HoniHelper.hs:20-36: a real takeDepthSnapshot frame
   has its invalid pixels in contiguous regions): no return from grazing rays (|n.d| < 0.15), a 3-5 px shadow band on the
   far side of every depth discontinuity, nothing beyond range_cut_m (<= 0: 3.5 m), nothing from absorbing furniture when
   `absorbing` is set, and sigma_mm x (z / 1 m)^2 of Gaussian noise keyed by (seed, pixel).  scene < 0: the open scene of
   hsk_synth_render; 0..3: closed room `scene`.  hole_fraction (may be NULL): the share of pixels without depth. */
int hsk_synth_render_sensor(int scene, const float pose[16], int w, int h, float fx, float fy, float cx, float cy, uint64_t seed,
                            float sigma_mm, float range_cut_m, int absorbing, uint16_t* depth, double* hole_fraction);

/* Products on the file seam (Main.hs:1740, :1320-1345): binary PCD with float32 x y z */
int hsk_write_pcd_xyz(const char* path, const float* xyz, size_t n_points);
/* binary little-endian .ply mesh (the file plyxform / pcl tools take, README.md:16-17): vertices welded by exact
 * coordinates, faces as uchar-count + int indices; zero-area triangles are dropped.  Returns counts when non-NULL. */
int hsk_write_ply_mesh(const char* path, const float* tri_xyz, size_t n_triangles, size_t* n_vertices_out, size_t* n_faces_out);
/* the same welding without a file: indices[3 * n_triangles] into vertices (cap_vertices x 3); degenerate triangles keep index triples with repeats */
int hsk_weld_triangles(const float* tri_xyz, size_t n_triangles, float* vertices, size_t cap_vertices, size_t* n_vertices, int32_t* indices);
int hsk_voxel_downsample(const float* xyz, size_t n, float leaf_m, float* out, size_t cap, size_t* n_out);
/* Plane products loadRoom reads beside the cloud (Main.hs:1392-1404): planes.txt lines "a b c d" in PCL form
 * ax+by+cz+d=0 (planeEqsFromFile, Main.hs:1379-1389) and cloud_plane_hull<k>.pcd polygons (Main.hs:1395-1400).
 * Deterministic RANSAC + PCA refit; labels[i] = plane index of point i or -1. */
int hsk_detect_planes(const float* xyz, size_t n, float dist_thresh_m, float min_fraction, int max_planes, int iterations,
                      float* planes_abcd /* 4 * max_planes */, int* labels /* n, may be NULL */, int* n_planes);
int hsk_plane_hull(const float* xyz, size_t n, const int* labels, int plane, const float abcd[4], float* hull_xyz,
                   size_t cap, size_t* n_hull);
int hsk_write_planes_txt(const char* path, const float* planes_abcd, int n_planes);

/* Room placements coming back from HouseScan: row-major left-multiplicative 4x4, as the 4-line .xf file
 * (roomProjectionToXfFormat, Main.hs:2289-2302) or the one-line CSV (roomProjectionToString, Main.hs:2271-2284). */
int hsk_write_xf(const char* path, const float m[16]);
int hsk_read_xf(const char* path, float m[16]);              /* accepts both layouts */
int hsk_transform_cloud(const float* xyz, size_t n, const float m[16], float* out /* may alias xyz */);

/* Recorded depth streams ("HSKD" raw container; frames in the layout of takeDepthSnapshot, HoniHelper.hs:20-36). */
typedef struct hsk_depth_stream hsk_depth_stream;
hsk_depth_stream* hsk_stream_create(const char* path, int w, int h, float fx, float fy, float cx, float cy);
hsk_depth_stream* hsk_stream_open(const char* path, int* w, int* h, int* n_frames, float intr[4]);
int hsk_stream_write(hsk_depth_stream* s, const uint16_t* depth);
int hsk_stream_read(hsk_depth_stream* s, int index, uint16_t* depth);
int hsk_stream_close(hsk_depth_stream* s);
int hsk_stream_info(const hsk_depth_stream* s, int* w, int* h, int* n_frames, float intr[4]);
/* The recorded-stream frame feed (BASELINE configs[2]: "scan from recorded stream"; the loop that replaces
 * Main.hs:1285-1290 around takeDepthSnapshot, HoniHelper.hs:20-36): frames [first, first + count) of `s` through the
 * tracker, pipelined -- frame i + 1 is read from the file and uploaded while frame i is on the GPU.  poses_out: 16 floats
 * per frame (row-major cam->world), tracked_out: one int per frame (either may be NULL).  The context must have no frame
 * in flight; results are exactly those of hsk_process_frame called with the same frames -- also behind a lost frame:
 * frame i + 1, in flight when frame i reports tracking lost, is dropped on the device, then read again and resubmitted
 * as the first frame of the restarted scan (as the reference-shaped host loop would feed it). */
int hsk_track_stream(hsk_ctx* k, hsk_depth_stream* s, int first, int count, float* poses_out, int* tracked_out);

#ifdef __cplusplus
}
#endif
#endif
