// hskinfu_api.hip -- the C ABI (include/hskinfu.h) over the gfx950 kernels: context, device memory layout,
// the tracker state machine of SURVEY.md A.2 and its hipGraph replay, stage-level entry points, z-slab
// building blocks.  Host-side restatement of the role the external PCL KinFu app plays for HouseScan
// (/root/reference/README.md:13-14); the depth-frame type is HoniHelper.hs:20's (Vector Word16,(w,h)).
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <sched.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <pthread.h>
#include <thread>
#include <vector>

#include "../../include/hskinfu.h"
#include "hsk_dev.h"
#include "hsk_launch.h"

#include "build/build_id.h"
extern "C" const char* hsk_build_id(void) { return HSK_BUILD_ID; }

static thread_local std::string g_create_err;

// image-space buffers of one frame; two sets so that the asynchronous path can preprocess frame k+1 on a second
// stream while frame k is still being tracked / fused / raycast
struct ImgBufs {
  uint16_t* d_raw = nullptr;
  uint16_t* d_dep[HSK_NLEVELS] = {};
  float* d_scaled = nullptr;
  float* d_vcur[HSK_NLEVELS] = {};
  float* d_ncur[HSK_NLEVELS] = {};
  float* d_tmax = nullptr;  // tile tables of the scaled depth (see launch_tile_max)
};

struct hsk_ctx {
  hsk_config cfg;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  VolParams vp;
  ImgLevel lv[HSK_NLEVELS];
  float init_R[9], init_t[3];
  // device memory (all sized once at create; nothing is allocated on the frame path)
  void* d_vol = nullptr;
  size_t vol_bytes = 0;
  ImgBufs ib[2];
  int cur = 0;  // set the enqueue_* helpers work on (0 everywhere except inside the overlapped async submission)
  ImgBufs& B() { return ib[cur]; }
  float* d_vmod[HSK_NLEVELS] = {};
  float* d_nmod[HSK_NLEVELS] = {};
  TrackState* d_st = nullptr;
  TrackState* h_st = nullptr;  // pinned
  double* d_partials = nullptr;
  double* d_partials2 = nullptr;  // ping-pong partner of d_partials (fused ICP iterations)
  void* d_icp_pose = nullptr;     // two IcpPose slots
  double* d_sums = nullptr;
  float h_ws[169] = {};  // bilateral spatial weights (host copy: passed to the kernel by value)
  float* d_wc = nullptr;
  int* d_keys = nullptr;
  unsigned* d_flags = nullptr;       // bitfield, one bit per brick: ever held a negative TSDF
  unsigned char* d_uni = nullptr;    // lane-block summaries (integrate.hip: hsk_uniform_code), one byte per 4x1x4 voxels
  size_t uni_bytes = 0;
  bool weights_pending = false;      // an integrate has been enqueued since the summaries' weights were last written back
  size_t flags_bytes = 0;
  unsigned* d_queue = nullptr;       // integrate pass A -> pass B: count (4 words) + uncertain lane-block ids
  CubeTable* d_cube_tab = nullptr;   // marching-cubes table (hsk_extract_mesh_cubes; filled on first use)
  int2* d_zint = nullptr;            // per lane column: stored-plane range inside the padded frustum
  uint16_t* h_stage = nullptr;  // pinned staging for the incoming depth frame (HSK_MAX_IN_FLIGHT + 1 frames, used in turn)
  unsigned stage_turn = 0;
  unsigned long long* d_counter = nullptr;
  unsigned* d_rowcnt = nullptr;
  unsigned long long* d_rowoff = nullptr;
  // read-out (round 5): what the volume looked like when a product was last counted (a size query followed by the fill
  // finds the rows' counts and offsets in place), a grow-only device buffer for the product, and two pinned staging
  // buffers through which products and the volume reach the caller's pageable memory (lazily allocated)
  uint64_t vol_epoch = 1;       // counted up by everything that changes the volume
  int ro_kind = 0;              // 1 cloud, 2 tetrahedra mesh, 3 cubes mesh: whose counts d_rowcnt / d_rowoff hold
  uint64_t ro_epoch = 0;
  unsigned long long ro_total = 0;
  void* d_out = nullptr;
  size_t out_bytes = 0;
  void* h_pin[2] = {nullptr, nullptr};
  size_t pin_bytes = 0;
  hipEvent_t ev_pin[2] = {};
  int frame = 0;
  std::string err;
  // asynchronous submission ring (hsk_submit_frame_dev / hsk_wait_frame)
  TrackState* h_ring = nullptr;  // pinned, HSK_MAX_IN_FLIGHT + 1 slots
  int* h_slot_fifo = nullptr;    // pinned: ring slot of each pipelined frame, read by the frame's last kernel (RingOut)
  unsigned* d_ring_seq = nullptr;  // device: pipelined frames that have reported
  unsigned ring_seq = 0;         // host mirror: pipelined frames submitted
  unsigned ring_expect[HSK_MAX_IN_FLIGHT + 1] = {};  // mark the frame in each slot will write
  unsigned set_expect[2] = {0, 0};                     // ... and the one that last used each image buffer set
  int set_slot[2] = {-1, -1};
  TrackState* d_ring_view = nullptr;  // device-side addresses of h_ring / h_slot_fifo
  int* d_fifo_view = nullptr;
  hipEvent_t ring_ev[HSK_MAX_IN_FLIGHT + 1] = {};
  int ring_kind[HSK_MAX_IN_FLIGHT + 1] = {};  // 0 tracked-frame candidate, 1 first frame (already complete)
  int ring_head = 0, ring_count = 0;
  bool pending_reset = false;
  // overlapped preprocessing: stream, per-set events (preprocess done / set free again), per-set graphs of the rest
  hipStream_t pstream = nullptr;
  hipEvent_t ev_pre[2] = {}, ev_free[2] = {};
  hipEvent_t ev_src = nullptr;  // orders the second stream behind the caller's (adopted) stream before a depth copy

  bool set_used[2] = {false, false};
  int async_set = 1;
  const void* pf_ptr = nullptr;   // hsk_mgpu_prefetch: depth pointer whose preprocessing is already enqueued ...
  int pf_set = -1;                // ... into this buffer set (on pstream, ev_pre[pf_set] recorded)
  int mgpu_set = 0;               // buffer set of the slab frame in progress
  hipGraph_t sgraph[2] = {};       // slab frame front (ICP + integrate + local raycast) per buffer set
  hipGraphExec_t sgexec[2] = {};
  void* sgraph_keys = nullptr;     // the keys buffer baked into those graphs
  // hipGraph of the steady-state frame
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  bool graph_ready = false;
  // use_graph = 2: the main-stream chain of a PIPELINED frame (19 ICP launches + 3 integrate + raycast) as one graph per
  // image-buffer set -- the host's cost of a frame is then one launch where it was 23 (what limits several rooms on one GPU)
  hipGraph_t pgraph[2] = {};
  hipGraphExec_t pgexec[2] = {};
  // profiling
  bool prof = false;
  bool prof_levels = false;  // profiling level 2: also an event at every ICP level (they cost about 4 us each)
  hipEvent_t ev[HSK_NSTAGES + 1] = {};
  hipEvent_t ev_icp[HSK_NLEVELS + 1] = {};  // profiling: start of each ICP level (coarsest first) and the end of the last
  double icp_level_ms[HSK_NLEVELS] = {};    // ... summed per level, index = level (0 = finest)
  double stage_ms[HSK_NSTAGES] = {};
  // host time of the pipelined submissions, by phase (hsk_submit_host_us): staging copy, copy + preprocessing enqueue, the
  // wait for the preprocessing, the main-stream chain's enqueue; and the submissions counted
  double submit_us[4] = {};
  unsigned long long submit_n = 0;
  uint64_t prof_frames = 0;
};

#define HIPCHK(k, call)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      char buf_[512];                                                                          \
      snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (k)->err = buf_;                                                                         \
      return HSK_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

static int fail(hsk_ctx* k, int code, const char* msg) {
  if (k) k->err = msg;
  return code;
}

static void pose16_to_rt(const float m[16], float R[9], float t[3]) {
  for (int i = 0; i < 3; ++i) {
    R[i * 3] = m[i * 4];
    R[i * 3 + 1] = m[i * 4 + 1];
    R[i * 3 + 2] = m[i * 4 + 2];
    t[i] = m[i * 4 + 3];
  }
}
static void rt_to_pose16(const float R[9], const float t[3], float m[16]) {
  for (int i = 0; i < 3; ++i) {
    m[i * 4] = R[i * 3];
    m[i * 4 + 1] = R[i * 3 + 1];
    m[i * 4 + 2] = R[i * 3 + 2];
    m[i * 4 + 3] = t[i];
  }
  m[12] = m[13] = m[14] = 0.0f;
  m[15] = 1.0f;
}

extern "C" void hsk_default_config(hsk_config* c, int n) {
  memset(c, 0, sizeof(*c));
  c->vol_x = c->vol_y = c->vol_z = n;
  c->vol_size_m[0] = c->vol_size_m[1] = c->vol_size_m[2] = 3.0f;
  c->trunc_dist_m = 0.03f;
  c->width = 640;
  c->height = 480;
  c->fx = c->fy = 525.0f;
  c->cx = 319.5f;
  c->cy = 239.5f;
  c->icp_iters[0] = 10;
  c->icp_iters[1] = 5;
  c->icp_iters[2] = 4;
  c->icp_dist_thresh_m = 0.10f;
  c->icp_angle_thresh_sin = 0.3420201433256687f;
  c->integrate_move_thresh = 0.0f;
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float t[3] = {c->vol_size_m[0] / 2.0f, c->vol_size_m[1] / 2.0f, c->vol_size_m[2] / 2.0f - 1.2f * c->vol_size_m[2] / 2.0f};
  rt_to_pose16(R, t, c->init_pose);
  c->device_id = 0;
  c->own_z0 = 0;
  c->own_z1 = n;
  c->halo = 0;
  c->use_graph = 0;
}

extern "C" int hsk_bilateral_tables(float ws[169], float wc[512]) {
  const float sig_s = 4.5f, sig_c = 30.0f;
  const float s2 = 0.5f / (sig_s * sig_s);
  const float c2 = 0.5f / (sig_c * sig_c);
  for (int dy = -6; dy <= 6; ++dy)
    for (int dx = -6; dx <= 6; ++dx) {
      const float arg = (float)(dx * dx + dy * dy) * s2;
      ws[(dy + 6) * 13 + (dx + 6)] = (float)std::exp(-(double)arg);
    }
  for (int k = 0; k < 512; ++k) {
    const float arg = (float)(k * k) * c2;
    wc[k] = (float)std::exp(-(double)arg);
  }
  return HSK_OK;
}

static void free_all(hsk_ctx* k) {
  if (!k) return;
  (void)hipSetDevice(k->cfg.device_id);
  if (k->gexec) (void)hipGraphExecDestroy(k->gexec);
  if (k->graph) (void)hipGraphDestroy(k->graph);
  for (auto& g : k->pgexec)
    if (g) (void)hipGraphExecDestroy(g);
  for (auto& g : k->pgraph)
    if (g) (void)hipGraphDestroy(g);
  for (auto& g : k->sgexec)
    if (g) (void)hipGraphExecDestroy(g);
  for (auto& g : k->sgraph)
    if (g) (void)hipGraphDestroy(g);
  for (auto& e : k->ev_pre)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : k->ev_free)
    if (e) (void)hipEventDestroy(e);
  if (k->ev_src) (void)hipEventDestroy(k->ev_src);
  if (k->pstream) (void)hipStreamDestroy(k->pstream);
  auto F = [](void* p) {
    if (p) (void)hipFree(p);
  };
  F(k->d_vol);
  for (auto& b : k->ib) {
    F(b.d_raw);
    F(b.d_scaled);
    F(b.d_tmax);
    for (int l = 0; l < HSK_NLEVELS; ++l) {
      F(b.d_dep[l]);
      F(b.d_vcur[l]);
      F(b.d_ncur[l]);
    }
  }
  for (int l = 0; l < HSK_NLEVELS; ++l) {
    F(k->d_vmod[l]);
    F(k->d_nmod[l]);
  }
  F(k->d_st);
  F(k->d_partials);
  F(k->d_partials2);
  F(k->d_icp_pose);
  F(k->d_sums);
  F(k->d_wc);
  F(k->d_keys);
  F(k->d_flags);
  F(k->d_uni);
  F(k->d_zint);
  F(k->d_queue);
  F(k->d_cube_tab);
  F(k->d_counter);
  F(k->d_rowcnt);
  F(k->d_rowoff);
  F(k->d_out);
  for (auto& p : k->h_pin)
    if (p) (void)hipHostFree(p);
  for (auto& e : k->ev_pin)
    if (e) (void)hipEventDestroy(e);
  if (k->h_st) (void)hipHostFree(k->h_st);
  if (k->h_stage) (void)hipHostFree(k->h_stage);
  if (k->h_ring) (void)hipHostFree(k->h_ring);
  if (k->h_slot_fifo) (void)hipHostFree(k->h_slot_fifo);
  if (k->d_ring_seq) (void)hipFree(k->d_ring_seq);
  for (auto& e : k->ring_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : k->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : k->ev_icp)
    if (e) (void)hipEventDestroy(e);
  if (k->own_stream && k->stream) (void)hipStreamDestroy(k->stream);
}

// The weights of deep free space live in the lane-block summaries until somebody reads the volume: bring the volume's
// copies up to date (k_summaries<true>), once -- a second read-out with no integrate in between finds them current.
static void flush_weights(hsk_ctx* k) {
  if (!k->weights_pending) return;
  launch_materialize(k->stream, k->d_vol, k->vp, k->d_uni);
  k->weights_pending = false;
}

static int upload_state(hsk_ctx* k) {
  HIPCHK(k, hipMemcpyAsync(k->d_st, k->h_st, sizeof(TrackState), hipMemcpyHostToDevice, k->stream));
  HIPCHK(k, hipStreamSynchronize(k->stream));
  return HSK_OK;
}
static int download_state(hsk_ctx* k) {
  HIPCHK(k, hipMemcpyAsync(k->h_st, k->d_st, sizeof(TrackState), hipMemcpyDeviceToHost, k->stream));
  HIPCHK(k, hipStreamSynchronize(k->stream));
  return HSK_OK;
}

static int do_reset(hsk_ctx* k) {
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  HIPCHK(k, hipMemsetAsync(k->d_vol, 0, k->vol_bytes, k->stream));
  HIPCHK(k, hipMemsetAsync(k->d_flags, 0, k->flags_bytes, k->stream));
  HIPCHK(k, hipMemsetAsync(k->d_uni, 1, uniform_lane_bytes(k->vp), k->stream));  // 1: "all 16 voxels never observed"
  // ... and the coarse level behind them: 0, "nothing pending, nothing known" (a never-observed block is not quiet)
  HIPCHK(k, hipMemsetAsync(k->d_uni + uniform_lane_bytes(k->vp), 0, k->uni_bytes - uniform_lane_bytes(k->vp), k->stream));
  memset(k->h_st, 0, sizeof(TrackState));
  memcpy(k->h_st->R, k->init_R, sizeof(k->init_R));
  memcpy(k->h_st->t, k->init_t, sizeof(k->init_t));
  memcpy(k->h_st->Rp, k->init_R, sizeof(k->init_R));
  memcpy(k->h_st->tp, k->init_t, sizeof(k->init_t));
  k->frame = 0;
  k->vol_epoch += 1;
  k->pending_reset = false;
  return upload_state(k);
}

extern "C" int hsk_create(const hsk_config* c, hsk_ctx** out) {
  if (!c || !out) {
    g_create_err = "hsk_create: null argument";
    return HSK_ERR_ARG;
  }
  *out = nullptr;
  if (c->vol_x <= 0 || c->vol_y <= 0 || c->vol_z <= 0 || (c->vol_x % 8) != 0 || (c->vol_y % 8) != 0 || c->width <= 0 || c->height <= 0 ||
      (c->width % 4) != 0 || (c->height % 4) != 0 || c->own_z0 < 0 || c->own_z1 > c->vol_z || c->own_z0 >= c->own_z1 ||
      c->halo < 0) {
    g_create_err = "hsk_create: invalid configuration (vol_x, vol_y must be multiples of 8, image dims of 4; 0 <= own_z0 < own_z1 <= vol_z)";
    return HSK_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_err = "hsk_create: no HIP device available (this library has no CPU fallback)";
    return HSK_ERR_NOGPU;
  }
  if (c->device_id < 0 || c->device_id >= ndev) {
    g_create_err = "hsk_create: device_id out of range";
    return HSK_ERR_ARG;
  }
  hsk_ctx* k = new hsk_ctx();
  k->cfg = *c;
  auto bail = [&](int code) {
    g_create_err = k->err;
    free_all(k);
    delete k;
    return code;
  };
#define CK(call)                                   \
  do {                                             \
    int r_ = [&]() -> int {                        \
      HIPCHK(k, call);                             \
      return HSK_OK;                               \
    }();                                           \
    if (r_ != HSK_OK) return bail(r_);             \
  } while (0)
  CK(hipSetDevice(c->device_id));
  CK(hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking));
  k->own_stream = true;
  // geometry
  VolParams& vp = k->vp;
  vp.X = c->vol_x;
  vp.Y = c->vol_y;
  vp.Z = c->vol_z;
  vp.zo0 = c->own_z0;
  vp.zo1 = c->own_z1;
  vp.zs0 = c->own_z0 - c->halo < 0 ? 0 : c->own_z0 - c->halo;
  const int zs1 = c->own_z1 + c->halo > c->vol_z ? c->vol_z : c->own_z1 + c->halo;
  vp.nzs = zs1 - vp.zs0;
  for (int i = 0; i < 3; ++i) vp.size[i] = c->vol_size_m[i];
  vp.cell[0] = vp.size[0] / (float)vp.X;
  vp.cell[1] = vp.size[1] / (float)vp.Y;
  vp.cell[2] = vp.size[2] / (float)vp.Z;
  for (int a = 0; a < 3; ++a) vp.icell[a] = 1.0 / (double)vp.cell[a];
  vp.stream_nt = 0;  // decided below, once the stored plane count is known
  vp.zchunk = 8;  // (decided below as well)
  float m = vp.cell[0] > vp.cell[1] ? vp.cell[0] : vp.cell[1];
  m = m > vp.cell[2] ? m : vp.cell[2];
  const float lo = 2.1f * m;
  vp.tau = c->trunc_dist_m > lo ? c->trunc_dist_m : lo;
  vp.tau_inv = 1.0f / vp.tau;
  for (int l = 0; l < HSK_NLEVELS; ++l) {
    const float s = (float)(1 << l);
    k->lv[l].W = c->width >> l;
    k->lv[l].H = c->height >> l;
    k->lv[l].in = Intr{c->fx / s, c->fy / s, c->cx / s, c->cy / s};
  }
  pose16_to_rt(c->init_pose, k->init_R, k->init_t);
  // memory
  k->vol_bytes = hsk_vol_words(vp) * 4;  // (stored planes padded to whole 64-B blocks of 4 planes: hsk_dev.h, hsk_vox_index)
  vp.stream_nt = k->vol_bytes > ((size_t)1 << 30) ? 1 : 0;  // > 1 GiB: four times the Infinity Cache and more
  vp.zchunk = hsk_pass_a_zchunk(vp.X, vp.Y, vp.nzs);
  CK(hipMalloc(&k->d_vol, k->vol_bytes));
  const size_t P0 = (size_t)c->width * c->height;
  for (auto& b : k->ib) {
    CK(hipMalloc((void**)&b.d_raw, P0 * 2));
    CK(hipMalloc((void**)&b.d_scaled, P0 * 4));
    CK(hipMalloc((void**)&b.d_tmax, tile_table_bytes(c->width, c->height)));
    CK(hipMemset(b.d_tmax, 0, tile_table_bytes(c->width, c->height)));  // (the validity mask's pad words are read, never written: they are zero)
    for (int l = 0; l < HSK_NLEVELS; ++l) {
      const size_t P = (size_t)k->lv[l].W * k->lv[l].H;
      CK(hipMalloc((void**)&b.d_dep[l], P * 2));
      CK(hipMalloc((void**)&b.d_vcur[l], P * 12));
      CK(hipMalloc((void**)&b.d_ncur[l], P * 12));
    }
  }
  for (int l = 0; l < HSK_NLEVELS; ++l) {
    const size_t P = (size_t)k->lv[l].W * k->lv[l].H;
    CK(hipMalloc((void**)&k->d_vmod[l], P * 12));
    CK(hipMalloc((void**)&k->d_nmod[l], P * 12));
  }
  {
    // The second stream (upload + preprocessing of the NEXT frame) at the LOWEST priority (round 6).  Not for the priority: streams
    // of different priorities come from different pools of hardware queues, so this stream can never be mapped onto the
    // hardware queue of the context's main stream -- HIP spreads a process's streams over four queues in turn, and with one
    // more stream alive in the process (an idle second context, the host application's own) the two streams of a context met
    // on one queue: the preprocessing then waited behind the previous frame's whole chain, 4440 -> 3730 frames/s
    // (tools/host_frames_probe.py --idle-ctx).  The main stream at the HIGHEST priority as well was measured too: four rooms
    // on one GPU fall from 7200 to 4700-5100 frames/s in all (profiles/r06/rooms_notes.md: more overlap, longer chains).
    int lo = 0, hi = 0;
    bool made = false;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
      made = hipStreamCreateWithPriority(&k->pstream, hipStreamNonBlocking, lo) == hipSuccess;
    if (!made) {  // (no priorities here: a plain stream, as before)
      (void)hipGetLastError();
      CK(hipStreamCreateWithFlags(&k->pstream, hipStreamNonBlocking));
    }
  }
  for (auto& e : k->ev_pre) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : k->ev_free) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&k->ev_src, hipEventDisableTiming));

  CK(hipMalloc((void**)&k->d_st, sizeof(TrackState)));
  CK(hipHostMalloc((void**)&k->h_st, sizeof(TrackState), hipHostMallocDefault));
  CK(hipHostMalloc((void**)&k->h_stage, P0 * 2 * (HSK_MAX_IN_FLIGHT + 1), hipHostMallocDefault));
  CK(hipHostMalloc((void**)&k->h_ring, sizeof(TrackState) * (HSK_MAX_IN_FLIGHT + 1), hipHostMallocDefault));
  CK(hipHostMalloc((void**)&k->h_slot_fifo, sizeof(int) * HSK_RING_FIFO, hipHostMallocDefault));
  memset(k->h_slot_fifo, 0, sizeof(int) * HSK_RING_FIFO);
  CK(hipHostGetDevicePointer((void**)&k->d_ring_view, k->h_ring, 0));
  CK(hipHostGetDevicePointer((void**)&k->d_fifo_view, k->h_slot_fifo, 0));
  CK(hipMalloc((void**)&k->d_ring_seq, sizeof(unsigned)));
  CK(hipMemset(k->d_ring_seq, 0, sizeof(unsigned)));
  for (auto& e : k->ring_ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const int nb0 = icp_num_blocks(c->width, c->height);
  CK(hipMalloc((void**)&k->d_partials, (size_t)nb0 * 27 * sizeof(double)));
  CK(hipMalloc((void**)&k->d_partials2, (size_t)nb0 * 27 * sizeof(double)));
  CK(hipMalloc(&k->d_icp_pose, icp_pose_bytes()));
  CK(hipMemset(k->d_icp_pose, 0, icp_pose_bytes()));  // accumulator slot 0 must be empty before the first tracked frame
  CK(hipMalloc((void**)&k->d_sums, 27 * sizeof(double)));
  CK(hipMalloc((void**)&k->d_wc, 512 * 4));
  CK(hipMalloc((void**)&k->d_keys, P0 * 4));
  vp.bshift = 3;
  // bitfield <= 4 KiB (bricks of ~94 mm at every volume size): its LDS copy then never limits how many raycast blocks a
  // CU holds, and the march was measured insensitive to the brick size between 47 and 94 mm
  while (hsk_flag_words(vp) > HSK_FLAG_WORDS_MAX && vp.bshift < 6 && ((vp.X >> (vp.bshift + 1)) << (vp.bshift + 1)) == vp.X &&
         ((vp.Y >> (vp.bshift + 1)) << (vp.bshift + 1)) == vp.Y)
    ++vp.bshift;
  // integrate's queue entries are 28-bit lane-block ids (4 x-voxels x 4 planes) with 4 plane bits on top
  if ((size_t)((vp.nzs + 3) / 4) * vp.Y * (vp.X / 4) >= ((size_t)1 << 28)) {
    k->err = "hsk_create: volume too large for the 28-bit lane-block ids of integrate (more than 2^32 stored voxels)";
    return bail(HSK_ERR_ARG);
  }
  k->flags_bytes = (size_t)hsk_flag_words_total(vp) * 4;  // brick bits + super-brick bits
  // k_raycast stages the whole bitfield in dynamic LDS; a launch may ask for at most 64 KiB of it.  Volumes whose
  // dimensions stop the brick edge from growing (e.g. 1000^3: 125 = 5^3 bricks of 8) would be created fine and then
  // fail at the first raycast: refuse them here, with the reason.
  if (k->flags_bytes > 64u * 1024u) {
    k->err = "hsk_create: the brick bitfield of this volume does not fit the raycast's LDS (64 KiB): choose vol_x, vol_y divisible by a larger power of two";
    return bail(HSK_ERR_ARG);
  }
  CK(hipMalloc((void**)&k->d_flags, k->flags_bytes));
  k->uni_bytes = uniform_bytes(vp);
  CK(hipMalloc((void**)&k->d_uni, k->uni_bytes));
  // integrate queues: 256 counters on their own 256-B lines + 256 queues, each sized for its share of pass A's tiles
  CK(hipMalloc((void**)&k->d_queue, integrate_queue_words(vp) * sizeof(unsigned)));
  CK(hipMalloc((void**)&k->d_zint, integrate_zint_entries(vp) * sizeof(int2)));
  CK(hipMalloc((void**)&k->d_counter, 16));
  {
    float ws[169], wc[512];
    hsk_bilateral_tables(ws, wc);
    memcpy(k->h_ws, ws, sizeof(ws));
    CK(hipMemcpy(k->d_wc, wc, sizeof(wc), hipMemcpyHostToDevice));
  }
  for (auto& e : k->ev) CK(hipEventCreate(&e));
  for (auto& e : k->ev_icp) CK(hipEventCreate(&e));
#undef CK
  int r = do_reset(k);
  if (r != HSK_OK) return bail(r);
  *out = k;
  return HSK_OK;
}

static int wait_slot(hsk_ctx* k, int slot, bool pose_only = false);
extern "C" int hsk_wait_frame(hsk_ctx* k, float pose_out[16], int* tracked);

extern "C" void hsk_destroy(hsk_ctx* k) {
  if (!k) return;
  if (k->pstream) (void)hipStreamSynchronize(k->pstream);
  if (k->stream) (void)hipStreamSynchronize(k->stream);  // frames in flight write into the pinned ring until they end
  free_all(k);
  delete k;
}

extern "C" int hsk_reset(hsk_ctx* k) {
  if (!k) return HSK_ERR_ARG;
  for (int i = 0; i < k->ring_count; ++i) {  // frames still in flight are dropped
    const int sl = (k->ring_head + i) % (HSK_MAX_IN_FLIGHT + 1);
    const int r = wait_slot(k, sl);
    if (r != HSK_OK) return r;
    if (k->ring_kind[sl] == 0) k->ring_kind[sl] = 2;
  }
  return do_reset(k);
}

extern "C" const char* hsk_last_error(const hsk_ctx* k) { return k ? k->err.c_str() : g_create_err.c_str(); }

extern "C" void* hsk_stream(hsk_ctx* k) { return k ? (void*)k->stream : nullptr; }
extern "C" int hsk_set_stream(hsk_ctx* k, void* stream) {
  if (!k) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  if (k->stream) HIPCHK(k, hipStreamSynchronize(k->stream));
  if (k->own_stream && k->stream) (void)hipStreamDestroy(k->stream);
  k->stream = (hipStream_t)stream;
  k->own_stream = false;
  return HSK_OK;
}
extern "C" int hsk_synchronize(hsk_ctx* k) {
  if (!k) return HSK_ERR_ARG;
  HIPCHK(k, hipStreamSynchronize(k->stream));
  return HSK_OK;
}

// ------------------------------------------------------------------------------------------------------
// frame building blocks (enqueue only; no host synchronisation)
// ------------------------------------------------------------------------------------------------------
// three launches (round 5; they were nine): the bilateral filter with scaleDepth and the raw tile tables of all three
// granularities; the derived tile tables; pyrDown x 2 with the vertex / normal maps of the three levels
static void enqueue_preprocess(hsk_ctx* k, hipStream_t s) {
  launch_bilateral_scale(s, k->B().d_raw, k->lv[0].W, k->lv[0].H, k->lv[0].in, k->h_ws, k->d_wc, k->B().d_dep[0], k->B().d_scaled,
                         k->B().d_tmax);
  launch_tile_tables(s, k->lv[0].W, k->lv[0].H, k->B().d_tmax);
  launch_pyramid_maps(s, k->B().d_dep, k->lv, k->B().d_vcur, k->B().d_ncur);
}

// fin: leave the frame's last solve to the integrate that the caller enqueues next with the same descriptor (its first
// kernel does it in its prologue: one launch less); only where that integrate always follows
static void enqueue_icp(hsk_ctx* k, IcpFinal* fin = nullptr) {
  if (fin) *fin = IcpFinal{nullptr, nullptr, 0};
  launch_icp_fused(k->stream, k->B().d_vcur, k->B().d_ncur, k->d_vmod, k->d_nmod, k->lv, k->cfg.icp_iters, k->d_st,
                   k->cfg.icp_dist_thresh_m, k->cfg.icp_angle_thresh_sin, k->d_icp_pose, k->d_partials, k->d_partials2, fin,
                   (k->prof && k->prof_levels && !fin) ? k->ev_icp : nullptr);
}

// report_early: a pipelined frame -- the integrate's first kernel, which ends the frame's ICP, tells the host the pose at
// once (pose_mark of the frame's ring slot); the raycast's report, an integrate later, then says that the frame's inputs
// are consumed (ring_mark)
static void enqueue_integrate(hsk_ctx* k, const IcpFinal* fin = nullptr, bool report_early = false) {
  k->weights_pending = true;
  k->vol_epoch += 1;
  const RingOut ring = {k->d_ring_view, k->d_fifo_view, k->d_ring_seq};
  launch_integrate(k->stream, k->d_vol, k->B().d_scaled, k->d_st, k->vp, k->lv[0].W, k->lv[0].H, k->lv[0].in, false,
                   k->d_counter, k->d_flags, k->B().d_tmax, k->d_zint, k->d_queue, (fin && fin->slots) ? fin : nullptr,
                   k->d_uni, report_early ? &ring : nullptr);
}

static void enqueue_raycast_and_resize(hsk_ctx* k, int* keys, bool report = false) {
  hipStream_t s = k->stream;
  // report: the raycast (the frame's last reader of the tracker state) writes it into the pinned ring slot the host
  // assigned to this frame -- a copy node behind the graph cost ~20 us of idle GPU per frame around it
  const RingOut ring = {k->d_ring_view, k->d_fifo_view, k->d_ring_seq};
  const RingOut* rp = report ? &ring : nullptr;
  if (raycast_can_fuse_pyramid(k->vp, k->lv[0].W, k->lv[0].H)) {
    // the raycast writes levels 1 and 2 of the model maps from its own tiles: no second launch, no re-read
    const MapPyramid pyr = {k->d_vmod[1], k->d_nmod[1], k->d_vmod[2], k->d_nmod[2]};
    launch_raycast(s, k->d_vol, k->d_st, k->vp, k->lv[0].W, k->lv[0].H, k->lv[0].in, k->d_vmod[0], k->d_nmod[0], keys, k->d_flags,
                   &pyr, rp);
    return;
  }
  launch_raycast(s, k->d_vol, k->d_st, k->vp, k->lv[0].W, k->lv[0].H, k->lv[0].in, k->d_vmod[0], k->d_nmod[0], keys, k->d_flags,
                 nullptr, rp);
  launch_resize_maps2(s, k->d_vmod[0], k->d_nmod[0], k->lv[0].W, k->lv[0].H, k->d_vmod[1], k->d_nmod[1], k->d_vmod[2],
                      k->d_nmod[2], k->d_st);
}

// Integration gate (this build's specification of it: DESIGN.md section 4): the frame integrates
// iff (angle(R^T Rp) + |t - tp|) / 2 >= thr.  Host side, only when the threshold is positive; the pose pair comes from
// the tracker state the frame has just downloaded.  trace(R^T Rp) is taken column by column: column i of R against
// column i of Rp, the three column products added left to right.
static bool gate_passes(const TrackState* st, float thr) {
  if (!(thr > 0.0f)) return true;
  const float* a = st->R;
  const float* b = st->Rp;
  const float c0 = (a[0] * b[0] + a[3] * b[3]) + a[6] * b[6];
  const float c1 = (a[1] * b[1] + a[4] * b[4]) + a[7] * b[7];
  const float c2 = (a[2] * b[2] + a[5] * b[5]) + a[8] * b[8];
  float cs = (((c0 + c1) + c2) - 1.0f) / 2.0f;
  if (cs > 1.0f) cs = 1.0f;
  if (cs < -1.0f) cs = -1.0f;
  const float turn = acosf(cs);
  const float ex = st->t[0] - st->tp[0], ey = st->t[1] - st->tp[1], ez = st->t[2] - st->tp[2];
  const float shift = sqrtf((ex * ex + ey * ey) + ez * ez);
  return (turn + shift) / 2.0f >= thr;
}

// the steady-state frame: everything between the depth copy and the pose read-back
static void enqueue_tracked_frame(hsk_ctx* k, bool with_events) {
  hipStream_t s = k->stream;
  if (with_events) (void)hipEventRecord(k->ev[0], s);
  enqueue_preprocess(k, k->stream);
  if (with_events) (void)hipEventRecord(k->ev[1], s);
  enqueue_icp(k);
  if (with_events) (void)hipEventRecord(k->ev[2], s);
  enqueue_integrate(k);
  if (with_events) (void)hipEventRecord(k->ev[3], s);
  enqueue_raycast_and_resize(k, nullptr);
  if (with_events) (void)hipEventRecord(k->ev[4], s);
}

// everything after the preprocessing of a tracked frame, on the main stream, for the buffer set k->cur
static void enqueue_tracked_rest(hsk_ctx* k) {
  IcpFinal fin;
  enqueue_icp(k, &fin);  // its first iteration also starts the frame (previous pose <- pose, lost flag)
  enqueue_integrate(k, &fin, true);  // ... and its last solve happens in the first kernel of the integrate, which reports the pose
  enqueue_raycast_and_resize(k, nullptr, true);  // pipelined frames report their state through the ring
}

static int frame_common(hsk_ctx* k, float pose_out[16], int* tracked) {
  // depth is already in d_raw (enqueued on the stream)
  hipStream_t s = k->stream;
  const bool gated = k->cfg.integrate_move_thresh > 0.0f;
  if (k->frame == 0) {
    enqueue_preprocess(k, k->stream);
    enqueue_integrate(k);
    for (int l = 0; l < HSK_NLEVELS; ++l)
      launch_transform_maps(s, k->B().d_vcur[l], k->B().d_ncur[l], k->lv[l].W * k->lv[l].H, k->d_st, k->d_vmod[l], k->d_nmod[l]);
    int r = download_state(k);
    if (r != HSK_OK) return r;
    HIPCHK(k, hipGetLastError());
    k->frame = 1;
    if (pose_out) rt_to_pose16(k->h_st->R, k->h_st->t, pose_out);
    if (tracked) *tracked = 0;
    return HSK_OK;
  }
  if (gated) {
    // host decides whether to integrate: one extra synchronisation, only in this non-default mode
    enqueue_preprocess(k, k->stream);
    enqueue_icp(k);
    int r = download_state(k);
    if (r != HSK_OK) return r;
    if (!k->h_st->lost) {
      if (gate_passes(k->h_st, k->cfg.integrate_move_thresh)) enqueue_integrate(k);
      enqueue_raycast_and_resize(k, nullptr);
    }
  } else if (k->prof) {
    enqueue_tracked_frame(k, true);
  } else if (k->cfg.use_graph == 1) {
    if (!k->graph_ready) {
      HIPCHK(k, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      enqueue_tracked_frame(k, false);
      HIPCHK(k, hipStreamEndCapture(s, &k->graph));
      HIPCHK(k, hipGraphInstantiate(&k->gexec, k->graph, nullptr, nullptr, 0));
      k->graph_ready = true;
    }
    HIPCHK(k, hipGraphLaunch(k->gexec, s));
    k->weights_pending = true;  // (the replayed graph integrates without passing through enqueue_integrate)
    k->vol_epoch += 1;
  } else {
    enqueue_tracked_frame(k, false);
  }
  int r = download_state(k);
  if (r != HSK_OK) return r;
  HIPCHK(k, hipGetLastError());
  if (k->prof && !gated) {
    for (int i = 0; i < HSK_NSTAGES; ++i) {
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, k->ev[i], k->ev[i + 1]) == hipSuccess) k->stage_ms[i] += ms;
    }
    for (int i = 0; i < HSK_NLEVELS && k->prof_levels; ++i) {  // event i starts level HSK_NLEVELS - 1 - i
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, k->ev_icp[i], k->ev_icp[i + 1]) == hipSuccess) k->icp_level_ms[HSK_NLEVELS - 1 - i] += ms;
    }
    k->prof_frames += 1;
  }
  if (k->h_st->lost) {
    r = do_reset(k);
    if (r != HSK_OK) return r;
    if (pose_out) rt_to_pose16(k->h_st->R, k->h_st->t, pose_out);
    if (tracked) *tracked = 0;
    return HSK_OK;
  }
  k->frame += 1;
  if (pose_out) rt_to_pose16(k->h_st->R, k->h_st->t, pose_out);
  if (tracked) *tracked = 1;
  return HSK_OK;
}

// the whole-frame entry points always work on buffer set 0 (their hipGraphs are captured with it); the slab entry
// points may have left the context on the other set or with a prefetch pending
static void leave_slab_bookkeeping(hsk_ctx* k) {
  k->cur = 0;
  k->mgpu_set = 0;
  k->pf_ptr = nullptr;
  k->pf_set = -1;
}

static int check_dims(hsk_ctx* k, const void* depth, int w, int h) {
  if (!k) return HSK_ERR_ARG;
  if (!depth) return fail(k, HSK_ERR_ARG, "depth pointer is null");
  if (w != k->cfg.width || h != k->cfg.height) return fail(k, HSK_ERR_ARG, "depth frame size does not match the context");
  return HSK_OK;
}

static int stage_depth_host(hsk_ctx* k, const uint16_t* depth) {
  const size_t bytes = (size_t)k->cfg.width * k->cfg.height * 2;
  memcpy(k->h_stage, depth, bytes);  // caller's buffer may be freed on return (HoniHelper's Vector is only pinned in unsafeWith)
  HIPCHK(k, hipMemcpyAsync(k->B().d_raw, k->h_stage, bytes, hipMemcpyHostToDevice, k->stream));
  return HSK_OK;
}

// The synchronous calls of a steady-state frame go through the pipelined machinery (submit + wait: eager launches, the
// preprocessing on the second stream, the pose reported through the pinned ring), which leaves less idle GPU around the
// frame than a hipGraph launch followed by a copy and a stream synchronisation (2206 -> 2450 frames/s).  use_graph = 1
// keeps that older form; the first frame of a scan, the gated and the profiled modes always take frame_common.
static bool sync_via_ring(const hsk_ctx* k) {
  return k->cfg.use_graph != 1 && k->frame > 0 && !k->pending_reset && !(k->cfg.integrate_move_thresh > 0.0f) && !k->prof;
}
static int submit_frame(hsk_ctx* k, const void* src, hipMemcpyKind kind, int w, int h);

extern "C" int hsk_process_frame(hsk_ctx* k, const uint16_t* depth, int w, int h, float pose_out[16], int* tracked) {
  int r = check_dims(k, depth, w, h);
  if (r != HSK_OK) return r;
  if (k->ring_count > 0) return fail(k, HSK_ERR_STATE, "frames are in flight: collect them with hsk_wait_frame first");
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  if (sync_via_ring(k)) {
    memcpy(k->h_stage, depth, (size_t)w * h * 2);  // the caller's buffer may be gone when this returns
    r = submit_frame(k, k->h_stage, hipMemcpyHostToDevice, w, h);
    if (r != HSK_OK) return r;
    return hsk_wait_frame(k, pose_out, tracked);
  }
  leave_slab_bookkeeping(k);
  r = stage_depth_host(k, depth);
  if (r != HSK_OK) return r;
  return frame_common(k, pose_out, tracked);
}

extern "C" int hsk_process_frame_dev(hsk_ctx* k, const void* depth_dev, int w, int h, float pose_out[16], int* tracked) {
  int r = check_dims(k, depth_dev, w, h);
  if (r != HSK_OK) return r;
  if (k->ring_count > 0) return fail(k, HSK_ERR_STATE, "frames are in flight: collect them with hsk_wait_frame first");
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  if (sync_via_ring(k)) {
    r = submit_frame(k, depth_dev, hipMemcpyDeviceToDevice, w, h);
    if (r != HSK_OK) return r;
    return hsk_wait_frame(k, pose_out, tracked);
  }
  leave_slab_bookkeeping(k);
  HIPCHK(k, hipMemcpyAsync(k->B().d_raw, depth_dev, (size_t)w * h * 2, hipMemcpyDeviceToDevice, k->stream));
  return frame_common(k, pose_out, tracked);
}


// ------------------------------------------------------------------------------------------------------
// asynchronous submission
// ------------------------------------------------------------------------------------------------------
// Completion of the frame parked in a ring slot.  Pipelined single-device frames (kind 0 with a mark) are announced by
// the device itself: the raycast stores the frame's sequence number into the pinned slot after the state words, and
// the host polls it -- event records on the main stream cost ~16 us of idle GPU per frame (2334 -> 2426 frames/s).
// how long a pipelined frame may take to report before the wait gives up (HSK_FRAME_TIMEOUT_S, seconds; default 20): a
// frame takes milliseconds, so this only ever ends a wait behind a peer rank of a group that has died or hung
static double frame_timeout_s() {
  static const double t = []() {
    const char* e = getenv("HSK_FRAME_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 20.0;
  }();
  return t;
}
// pose_only: the caller wants the frame's pose and verdict (final once its ICP has ended: pose_mark); otherwise that the
// frame has consumed its inputs and the stream has passed its integrate (ring_mark: buffer-set reuse, resets)
static int wait_slot(hsk_ctx* k, int slot, bool pose_only) {
  if (k->ring_expect[slot] == 0u) {
    HIPCHK(k, hipEventSynchronize(k->ring_ev[slot]));
    return HSK_OK;
  }
  volatile TrackState* w = (volatile TrackState*)&k->h_ring[slot];
  const auto t0 = std::chrono::steady_clock::now();
  // a frame takes well under 2 ms: spin that long before giving the core away (a yield can cost a whole time slice when
  // other threads of the process are runnable -- measured: 2620 -> 1450 frames/s with host frames under torch's threads)
  volatile unsigned* mark = pose_only ? &w->pose_mark : &w->ring_mark;
  for (unsigned long spin = 0; *mark != k->ring_expect[slot]; ++spin) {
    __builtin_ia32_pause();
    if ((spin & 4095u) == 4095u) {
      const auto dt = std::chrono::steady_clock::now() - t0;
      if (std::chrono::duration<double>(dt).count() > frame_timeout_s())
        return fail(k, HSK_ERR_TIMEOUT, "a pipelined frame did not report within HSK_FRAME_TIMEOUT_S (default 20 s)");
      if (dt > std::chrono::milliseconds(2)) sched_yield();
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return HSK_OK;
}

// Tracking was lost with frames still in flight: their results stay in the ring (they report tracked = 0), the reset
// itself happens here, before new work is enqueued.
static int reset_behind_lost_frame(hsk_ctx* k) {
  for (int i = 0; i < k->ring_count; ++i) {
    const int sl = (k->ring_head + i) % (HSK_MAX_IN_FLIGHT + 1);
    const int r = wait_slot(k, sl);
    if (r != HSK_OK) return r;
    if (k->ring_kind[sl] == 0) k->ring_kind[sl] = 2;  // dropped on the device (need_reset was set)
  }
  return do_reset(k);
}

extern "C" int hsk_submit_frame_dev(hsk_ctx* k, const void* depth_dev, int w, int h) {
  return submit_frame(k, depth_dev, hipMemcpyDeviceToDevice, w, h);
}

// the same for a frame in HOST memory (what takeDepthSnapshot hands over): copied into one of HSK_MAX_IN_FLIGHT + 1
// pinned staging frames before this returns, so the caller's buffer is free again; the upload overlaps the frame in flight
extern "C" int hsk_submit_frame(hsk_ctx* k, const uint16_t* depth, int w, int h) {
  int r = check_dims(k, depth, w, h);
  if (r != HSK_OK) return r;
  if (k->ring_count >= HSK_MAX_IN_FLIGHT) return fail(k, HSK_ERR_STATE, "too many frames in flight: call hsk_wait_frame first");
  const size_t px = (size_t)w * h;
  uint16_t* stage = k->h_stage + (size_t)(k->stage_turn % (HSK_MAX_IN_FLIGHT + 1)) * px;
  k->stage_turn += 1;
  const auto t0 = std::chrono::steady_clock::now();
  memcpy(stage, depth, px * 2);
  k->submit_us[0] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return submit_frame(k, stage, hipMemcpyHostToDevice, w, h);
}

// src: device memory (kind D2D) or the context's pinned staging buffer (kind H2D)
static int submit_frame(hsk_ctx* k, const void* depth_dev, hipMemcpyKind kind, int w, int h) {
  int r = check_dims(k, depth_dev, w, h);
  if (r != HSK_OK) return r;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  if (k->ring_count >= HSK_MAX_IN_FLIGHT) return fail(k, HSK_ERR_STATE, "too many frames in flight: call hsk_wait_frame first");
  if (k->ring_count == 0) leave_slab_bookkeeping(k);
  hipStream_t s = k->stream;
  const bool sync_path = k->frame == 0 || k->pending_reset || k->cfg.integrate_move_thresh > 0.0f || k->prof;
  if (sync_path) {
    // first frame of a (re)started scan, gated or profiled mode: run it synchronously and park the result
    if (k->pending_reset) {
      r = reset_behind_lost_frame(k);
      if (r != HSK_OK) return r;
    }
    HIPCHK(k, hipMemcpyAsync(k->B().d_raw, depth_dev, (size_t)w * h * 2, kind, s));
    float pose[16];
    int tracked = 0;
    r = frame_common(k, pose, &tracked);
    if (r != HSK_OK) return r;
    const int slot = (k->ring_head + k->ring_count) % (HSK_MAX_IN_FLIGHT + 1);
    k->h_ring[slot] = *k->h_st;
    k->h_ring[slot].lost = tracked ? 0 : 1;
    k->ring_kind[slot] = 1;
    k->ring_expect[slot] = 0u;
    HIPCHK(k, hipEventRecord(k->ring_ev[slot], s));
    k->ring_count += 1;
    return HSK_OK;
  }
  // Overlapped path: depth copy + preprocessing of THIS frame go to the second stream and the other buffer set, so
  // they run while the previous frame is still in its ICP / integrate / raycast on the main stream.
  const int set = k->async_set;
  k->async_set ^= 1;
  k->cur = set;
  // the slot this frame reports into: published to the device through the pinned fifo before anything is launched
  const int slot = (k->ring_head + k->ring_count) % (HSK_MAX_IN_FLIGHT + 1);
  k->h_slot_fifo[k->ring_seq % HSK_RING_FIFO] = slot;
  k->ring_seq += 1;
  k->ring_expect[slot] = k->ring_seq | 0x80000000u;  // never 0 (0 = "this slot completes through its event")
  ((volatile TrackState*)&k->h_ring[slot])->ring_mark = 0u;
  ((volatile TrackState*)&k->h_ring[slot])->pose_mark = 0u;
  hipError_t e = hipSuccess;
  // no events on the main stream: the host itself sees, in the pinned ring, that the previous user of this buffer set has
  // finished (it normally has: its pose was collected before this call)
  if (k->set_used[set] && k->set_slot[set] >= 0 && k->ring_expect[k->set_slot[set]] == k->set_expect[set]) {
    const int r2 = wait_slot(k, k->set_slot[set]);
    if (r2 != HSK_OK) return r2;
  }
  k->set_expect[set] = k->ring_seq | 0x80000000u;
  k->set_slot[set] = slot;
  // A device frame may still be in the making on the context's stream (an upload or a conversion kernel the caller
  // enqueued on the stream it handed over with hsk_set_stream): the second stream is ordered behind it.  The record
  // costs the main queue nothing, the wait sits on the second stream only.  (Host frames come from the context's own
  // pinned staging ring, already complete.)
  if (e == hipSuccess && kind == hipMemcpyDeviceToDevice && !k->own_stream) {
    e = hipEventRecord(k->ev_src, k->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(k->pstream, k->ev_src, 0);
  }
  const auto tp0 = std::chrono::steady_clock::now();
  if (e == hipSuccess) e = hipMemcpyAsync(k->B().d_raw, depth_dev, (size_t)w * h * 2, kind, k->pstream);
  if (e == hipSuccess) {
    enqueue_preprocess(k, k->pstream);
    e = hipEventRecord(k->ev_pre[set], k->pstream);
  }
  const auto tp1 = std::chrono::steady_clock::now();
  // The frame's own work needs the preprocessing.  A hipStreamWaitEvent on the main stream would be the obvious form,
  // but a cross-stream wait costs ~19 us of stalled queue at every frame boundary even when the event fired long ago
  // (2330 -> 2440 frames/s without it).  The host waits instead: it is a frame ahead of the GPU, the preprocessing takes
  // ~60 us, and kernels enqueued after the host has seen it complete need no device-side dependency.
  if (e == hipSuccess) e = hipEventSynchronize(k->ev_pre[set]);
  const auto tp2 = std::chrono::steady_clock::now();
  if (e == hipSuccess) {
    // Eager launches, on purpose: the host runs a frame ahead here, so their launch cost is hidden, while replaying
    // the frame from a hipGraph left ~8 us more idle GPU between consecutive frames (2330 vs 2285 frames/s measured).
    // The graph stays on the synchronous path, where the launch cost is exposed.
    if (k->cfg.use_graph == 2 && k->own_stream) {
      // (the chain's arguments depend on the buffer set only: the kernels read the pose from the tracker state, the ICP's
      // iteration index restarts with every frame, the ring slot comes through the pinned fifo)
      if (!k->pgexec[set]) {
        e = hipStreamBeginCapture(k->stream, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
          enqueue_tracked_rest(k);
          e = hipStreamEndCapture(k->stream, &k->pgraph[set]);
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&k->pgexec[set], k->pgraph[set], nullptr, nullptr, 0);
      } else {
        k->weights_pending = true;  // (what enqueue_integrate notes on the host side)
        k->vol_epoch += 1;
      }
      if (e == hipSuccess) e = hipGraphLaunch(k->pgexec[set], k->stream);
    } else {
      enqueue_tracked_rest(k);
    }
  }
  {
    const auto tp3 = std::chrono::steady_clock::now();
    k->submit_us[1] += std::chrono::duration<double, std::micro>(tp1 - tp0).count();
    k->submit_us[2] += std::chrono::duration<double, std::micro>(tp2 - tp1).count();
    k->submit_us[3] += std::chrono::duration<double, std::micro>(tp3 - tp2).count();
    k->submit_n += 1;
  }
  k->set_used[set] = true;
  k->cur = 0;
  HIPCHK(k, e);
  k->ring_kind[slot] = 0;
  k->ring_count += 1;
  return HSK_OK;
}

extern "C" int hsk_wait_frame(hsk_ctx* k, float pose_out[16], int* tracked) {
  if (!k) return HSK_ERR_ARG;
  if (k->ring_count == 0) return fail(k, HSK_ERR_STATE, "no frame in flight");
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  const int slot = k->ring_head;
  {
    const int rw = wait_slot(k, slot, true);  // (the pose: the frame's volume work may still be running)
    if (rw != HSK_OK) return rw;
  }
  const TrackState& st = k->h_ring[slot];
  k->ring_head = (k->ring_head + 1) % (HSK_MAX_IN_FLIGHT + 1);
  k->ring_count -= 1;
  if (pose_out) rt_to_pose16(st.R, st.t, pose_out);
  if (k->ring_kind[slot] == 1) {  // completed synchronously at submission
    if (tracked) *tracked = st.lost ? 0 : 1;
    return HSK_OK;
  }
  if (k->ring_kind[slot] == 2) {  // was in flight behind a lost frame; the reset has already happened
    if (tracked) *tracked = 0;
    if (pose_out) rt_to_pose16(k->init_R, k->init_t, pose_out);
    return HSK_OK;
  }
  if (st.lost) {
    // the volume is reset lazily: at the next submission, or now if nothing else is in flight
    k->pending_reset = true;
    if (tracked) *tracked = 0;
    if (pose_out) rt_to_pose16(k->init_R, k->init_t, pose_out);
    if (k->ring_count == 0) {
      int r = do_reset(k);
      if (r != HSK_OK) return r;
    }
    return HSK_OK;
  }
  k->frame += 1;
  if (tracked) *tracked = 1;
  return HSK_OK;
}

// The recorded-stream frame feed: what the host loop around takeDepthSnapshot (HoniHelper.hs:20-36, Main.hs:1285-1290)
// becomes when the frames come from a file.  One frame is always in flight ahead of the one being collected, so the
// file read + the copy into the pinned staging ring + the upload of frame i + 1 run under frame i's GPU work.
extern "C" int hsk_track_stream(hsk_ctx* k, hsk_depth_stream* s, int first, int count, float* poses_out, int* tracked_out) {
  if (!k) return HSK_ERR_ARG;
  if (!s) return fail(k, HSK_ERR_ARG, "stream is null");
  int w = 0, h = 0, n = 0;
  if (hsk_stream_info(s, &w, &h, &n, nullptr) != HSK_OK) return fail(k, HSK_ERR_ARG, "not a stream opened for reading");
  if (w != k->cfg.width || h != k->cfg.height) return fail(k, HSK_ERR_ARG, "the stream's frame size does not match the context");
  if (first < 0 || count < 0 || first > n || count > n - first) return fail(k, HSK_ERR_ARG, "frame range outside the stream");
  if (k->ring_count > 0) return fail(k, HSK_ERR_STATE, "frames are in flight: collect them with hsk_wait_frame first");
  uint16_t* buf = (uint16_t*)malloc((size_t)w * h * 2);
  if (!buf) return fail(k, HSK_ERR_STATE, "out of host memory");
  // `next`: the frame to submit next; `collected`: results handed out so far (frames first .. first + collected - 1).
  // A frame that reports tracking LOST (a pipelined frame, kind 0, with tracked = 0) leaves its successor -- already in
  // flight -- dropped on the device: that result is collected and thrown away, and the successor goes in again as the
  // first frame of the restarted scan, which is what hsk_process_frame makes of it.
  int next = 0, collected = 0, r = HSK_OK;
  auto collect = [&](bool* lost_verdict) -> int {
    float pose[16];
    int tr = 0;
    const bool pipelined = k->ring_kind[k->ring_head] == 0;
    const int rc = hsk_wait_frame(k, pose, &tr);
    if (rc != HSK_OK) return rc;
    if (poses_out) memcpy(poses_out + (size_t)16 * collected, pose, sizeof(pose));
    if (tracked_out) tracked_out[collected] = tr;
    collected += 1;
    *lost_verdict = pipelined && !tr;
    return HSK_OK;
  };
  auto after_collect = [&](bool lost) -> int {
    if (!lost) return HSK_OK;
    if (k->ring_count > 0) {  // the frames behind the lost one: dropped on the device; the reset, then their results discarded
      const int rr = reset_behind_lost_frame(k);
      if (rr != HSK_OK) return rr;
    }
    while (k->ring_count > 0) {
      const int rc = hsk_wait_frame(k, nullptr, nullptr);
      if (rc != HSK_OK) return rc;
    }
    next = collected;  // ... and fed again, the first of them restarting the scan
    return HSK_OK;
  };
  while (r == HSK_OK && collected < count) {
    bool lost = false;
    if (next < count) {
      if (hsk_stream_read(s, first + next, buf) != HSK_OK) {
        r = fail(k, HSK_ERR_STATE, "reading a frame from the stream failed");
        break;
      }
      r = hsk_submit_frame(k, buf, w, h);  // copies the frame before it returns
      next += 1;
      if (r == HSK_OK && k->ring_count > 1) {
        r = collect(&lost);
        if (r == HSK_OK) r = after_collect(lost);
      }
    } else {
      r = collect(&lost);
      if (r == HSK_OK) r = after_collect(lost);
    }
  }
  free(buf);
  return r;
}

// ------------------------------------------------------------------------------------------------------
// stage-level entry points
// ------------------------------------------------------------------------------------------------------
static int set_pose_internal(hsk_ctx* k, const float pose[16]) {
  int r = download_state(k);
  if (r != HSK_OK) return r;
  pose16_to_rt(pose, k->h_st->R, k->h_st->t);
  k->h_st->lost = 0;
  return upload_state(k);
}

extern "C" int hsk_get_pose(hsk_ctx* k, float pose[16]) {
  if (!k || !pose) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  int r = download_state(k);
  if (r != HSK_OK) return r;
  rt_to_pose16(k->h_st->R, k->h_st->t, pose);
  return HSK_OK;
}
extern "C" int hsk_set_pose(hsk_ctx* k, const float pose[16]) {
  if (!k || !pose) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  return set_pose_internal(k, pose);
}

extern "C" int hsk_integrate(hsk_ctx* k, const uint16_t* depth, int w, int h, const float pose[16]) {
  int r = check_dims(k, depth, w, h);
  if (r != HSK_OK) return r;
  if (!pose) return fail(k, HSK_ERR_ARG, "pose is null");
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  r = set_pose_internal(k, pose);
  if (r != HSK_OK) return r;
  r = stage_depth_host(k, depth);
  if (r != HSK_OK) return r;
  launch_scale_depth(k->stream, k->B().d_raw, w, h, k->lv[0].in, k->B().d_scaled);
  launch_tile_max(k->stream, k->B().d_scaled, w, h, k->B().d_tmax);
  launch_tile_fine(k->stream, k->B().d_scaled, w, h, k->B().d_tmax);
  enqueue_integrate(k);
  HIPCHK(k, hipStreamSynchronize(k->stream));
  HIPCHK(k, hipGetLastError());
  return HSK_OK;
}

extern "C" int hsk_count_updates(hsk_ctx* k, const uint16_t* depth, int w, int h, const float pose[16], uint64_t* n_upd) {
  int r = check_dims(k, depth, w, h);
  if (r != HSK_OK) return r;
  if (!pose || !n_upd) return fail(k, HSK_ERR_ARG, "null argument");
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  r = set_pose_internal(k, pose);
  if (r != HSK_OK) return r;
  r = stage_depth_host(k, depth);
  if (r != HSK_OK) return r;
  launch_scale_depth(k->stream, k->B().d_raw, w, h, k->lv[0].in, k->B().d_scaled);
  launch_tile_max(k->stream, k->B().d_scaled, w, h, k->B().d_tmax);
  launch_tile_fine(k->stream, k->B().d_scaled, w, h, k->B().d_tmax);
  HIPCHK(k, hipMemsetAsync(k->d_counter, 0, 8, k->stream));
  launch_integrate(k->stream, k->d_vol, k->B().d_scaled, k->d_st, k->vp, w, h, k->lv[0].in, true, k->d_counter, k->d_flags,
                   k->B().d_tmax, k->d_zint, k->d_queue);
  unsigned long long c = 0;
  HIPCHK(k, hipMemcpyAsync(&c, k->d_counter, 8, hipMemcpyDeviceToHost, k->stream));
  HIPCHK(k, hipStreamSynchronize(k->stream));
  *n_upd = c;
  return HSK_OK;
}

extern "C" int hsk_raycast(hsk_ctx* k, const float pose[16], float* vmap, float* nmap, int32_t* keys) {
  if (!k || !pose || !vmap || !nmap) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  int r = set_pose_internal(k, pose);
  if (r != HSK_OK) return r;
  const size_t P = (size_t)k->lv[0].W * k->lv[0].H;
  launch_raycast(k->stream, k->d_vol, k->d_st, k->vp, k->lv[0].W, k->lv[0].H, k->lv[0].in, k->d_vmod[0], k->d_nmod[0],
                 k->d_keys, k->d_flags);
  launch_resize_maps2(k->stream, k->d_vmod[0], k->d_nmod[0], k->lv[0].W, k->lv[0].H, k->d_vmod[1], k->d_nmod[1], k->d_vmod[2],
                      k->d_nmod[2], k->d_st);
  HIPCHK(k, hipMemcpyAsync(vmap, k->d_vmod[0], P * 12, hipMemcpyDeviceToHost, k->stream));
  HIPCHK(k, hipMemcpyAsync(nmap, k->d_nmod[0], P * 12, hipMemcpyDeviceToHost, k->stream));
  if (keys) HIPCHK(k, hipMemcpyAsync(keys, k->d_keys, P * 4, hipMemcpyDeviceToHost, k->stream));
  HIPCHK(k, hipStreamSynchronize(k->stream));
  HIPCHK(k, hipGetLastError());
  return HSK_OK;
}

extern "C" int hsk_preprocess(hsk_ctx* k, const uint16_t* depth, int w, int h) {
  int r = check_dims(k, depth, w, h);
  if (r != HSK_OK) return r;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  r = stage_depth_host(k, depth);
  if (r != HSK_OK) return r;
  enqueue_preprocess(k, k->stream);
  HIPCHK(k, hipStreamSynchronize(k->stream));
  HIPCHK(k, hipGetLastError());
  return HSK_OK;
}

extern "C" int hsk_icp_accumulate(hsk_ctx* k, int level, const float pose_est[16], int row0, int row1, double out27[27]) {
  if (!k || !pose_est || !out27) return HSK_ERR_ARG;
  if (level < 0 || level >= HSK_NLEVELS) return fail(k, HSK_ERR_ARG, "level out of range");
  if (row0 < 0 || row1 > k->lv[level].H || row0 >= row1) return fail(k, HSK_ERR_ARG, "row range out of bounds");
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  // estimate goes to (R,t); the previous pose (Rp,tp) is left as the tracker stored it
  int r = download_state(k);
  if (r != HSK_OK) return r;
  TrackState saved = *k->h_st;
  // the model maps were raycast from the last stored pose: that is the "previous" camera of A.5
  memcpy(k->h_st->Rp, saved.R, sizeof(saved.R));
  memcpy(k->h_st->tp, saved.t, sizeof(saved.t));
  pose16_to_rt(pose_est, k->h_st->R, k->h_st->t);
  k->h_st->lost = 0;
  r = upload_state(k);
  if (r != HSK_OK) return r;
  const int nb = icp_num_blocks(k->lv[level].W, row1 - row0);
  launch_icp_accumulate(k->stream, k->B().d_vcur[level], k->B().d_ncur[level], k->d_vmod[level], k->d_nmod[level],
                        k->lv[level].W, k->lv[level].H, k->lv[level].in, k->d_st, k->cfg.icp_dist_thresh_m,
                        k->cfg.icp_angle_thresh_sin, row0, row1, k->d_partials);
  launch_icp_reduce(k->stream, k->d_partials, nb, k->d_sums);
  HIPCHK(k, hipMemcpyAsync(out27, k->d_sums, 27 * sizeof(double), hipMemcpyDeviceToHost, k->stream));
  HIPCHK(k, hipStreamSynchronize(k->stream));
  *k->h_st = saved;
  return upload_state(k);
}

extern "C" int hsk_icp_solve(const double in27[27], float x6[6], int* ok) {
  if (!in27 || !x6 || !ok) return HSK_ERR_ARG;
  *ok = host_solve6(in27, x6) ? 1 : 0;
  if (!*ok)
    for (int i = 0; i < 6; ++i) x6[i] = 0.0f;
  return HSK_OK;
}

// ---- between device memory and the caller's PAGEABLE host memory (round 5) ------------------------------------------------
// A copy into pageable memory goes through the runtime's own staging at 16-17 GB/s, and hsk_download_tsdf moved 512 MiB
// that way (31 ms; 253 ms at 1024^3), allocating and freeing its device staging inside every call.  Two pinned buffers
// that live with the context: the device fills one (a conversion kernel writing straight into the mapped buffer, or a DMA
// copy) while host threads move the other's content to where the caller wants it.
#define HSK_PIN_BYTES ((size_t)32 << 20)
static int ensure_pinned(hsk_ctx* k) {
  if (k->h_pin[0]) return HSK_OK;
  // (at least one whole plane of the volume: the download and the upload move whole planes -- 4096 x 4096 voxels are 64 MiB)
  const size_t plane = (size_t)k->vp.X * k->vp.Y * 4;
  const size_t want = plane > HSK_PIN_BYTES ? plane : HSK_PIN_BYTES;
  for (int i = 0; i < 2; ++i) {
    hipError_t e = hipHostMalloc(&k->h_pin[i], want, hipHostMallocDefault);
    if (e == hipSuccess && !k->ev_pin[i]) e = hipEventCreateWithFlags(&k->ev_pin[i], hipEventDisableTiming);
    if (e != hipSuccess) {  // (nothing half-made is left behind: the next call tries again)
      for (auto& p : k->h_pin) {
        if (p) (void)hipHostFree(p);
        p = nullptr;
      }
      HIPCHK(k, e);
    }
  }
  k->pin_bytes = want;
  return HSK_OK;
}
// Host copies out of (into) the pinned buffers are shared among a few worker threads that live with the process (started
// on first use, asleep otherwise): a core moves 10-20 GB/s, the PCIe link 55.  A thread per copy cost ~20 us each to start,
// which the pieces of a pipelined copy cannot afford.
namespace {
struct CopyPool {
  std::mutex m;
  std::condition_variable cv_work;
  std::vector<std::thread> workers;
  // a call's slices carry the call's own latch: a read-out waits for ITS slices only, whoever copies them (read-outs of
  // different contexts on different threads -- concurrent rooms, a group's slabs -- used to wait on one global count)
  struct Latch { size_t left = 0; std::condition_variable cv; };
  struct Job { char* dst; const char* src; size_t len; Latch* latch; };
  std::vector<Job> jobs;
  bool stop = false;
  void done(Latch* l) {   // (under m)
    if (--l->left == 0) l->cv.notify_all();
  }
  void worker() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_work.wait(lk, [&] { return stop || !jobs.empty(); });
      if (stop && jobs.empty()) return;
      Job j = jobs.back();
      jobs.pop_back();
      lk.unlock();
      memcpy(j.dst, j.src, j.len);
      lk.lock();
      done(j.latch);
    }
  }
  void run(void* dst, const void* src, size_t bytes) {
    const size_t slice = (size_t)2 << 20;
    if (bytes <= slice) {
      memcpy(dst, src, bytes);
      return;
    }
    Latch latch;
    std::unique_lock<std::mutex> lk(m);
    if (workers.empty()) {
      unsigned n = std::thread::hardware_concurrency();
      n = n == 0 ? 1 : (n > 8 ? 7 : (n > 1 ? n - 1 : 1));
      for (unsigned i = 0; i < n; ++i) workers.emplace_back([this] { worker(); });
    }
    size_t first_len = 0;
    for (size_t off = 0; off < bytes; off += slice) {
      const size_t len = bytes - off < slice ? bytes - off : slice;
      if (off == 0) { first_len = len; continue; }   // the caller copies the first slice itself
      jobs.push_back(Job{(char*)dst + off, (const char*)src + off, len, &latch});
      ++latch.left;
    }
    lk.unlock();
    cv_work.notify_all();
    memcpy(dst, src, first_len);
    lk.lock();
    // (the caller helps with what is left -- its own slices or another call's -- instead of sleeping)
    while (latch.left != 0 && !jobs.empty()) {
      Job j = jobs.back();
      jobs.pop_back();
      lk.unlock();
      memcpy(j.dst, j.src, j.len);
      lk.lock();
      done(j.latch);
    }
    latch.cv.wait(lk, [&] { return latch.left == 0; });
  }
  // fork(): the child inherits `workers` without the threads behind it (joining them is undefined behaviour and hung at
  // exit) and possibly a mutex some other thread held.  The pool is quiesced round the fork and the child starts empty.
  void fork_prepare() { m.lock(); }
  void fork_parent() { m.unlock(); }
  void fork_child() {
    new (&m) std::mutex();
    new (&cv_work) std::condition_variable();
    new (&workers) std::vector<std::thread>();   // (the old vector's thread objects are abandoned, never destroyed)
    new (&jobs) std::vector<Job>();
    stop = false;
  }
  CopyPool();
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : workers) t.join();
  }
};
CopyPool g_copy_pool;
CopyPool::CopyPool() {
  pthread_atfork([] { g_copy_pool.fork_prepare(); }, [] { g_copy_pool.fork_parent(); }, [] { g_copy_pool.fork_child(); });
}
}  // namespace
static void parallel_memcpy(void* dst, const void* src, size_t bytes) { g_copy_pool.run(dst, src, bytes); }
// `bytes` of device memory at src into the caller's dst, in pieces through the pinned pair: the DMA of piece i + 1 runs
// under the host's copy of piece i
static int copy_out(hsk_ctx* k, void* dst, const void* src_dev, size_t bytes) {
  int r = ensure_pinned(k);
  if (r != HSK_OK) return r;
  size_t prev_off = 0, prev_len = 0;
  int i = 0;
  // (pieces of about a quarter of the whole, 2 MiB at least: the DMA of one piece and the host's copy of the one before it
  // overlap only when there are several -- a 30 MB mesh as ONE piece was 0.6 ms of DMA and then 0.75 ms of host copy)
  size_t piece = ((bytes / 4) + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);
  if (piece < ((size_t)1 << 21)) piece = (size_t)1 << 21;
  if (piece > k->pin_bytes) piece = k->pin_bytes;
  for (size_t off = 0; off < bytes; off += piece, ++i) {
    const size_t len = bytes - off < piece ? bytes - off : piece;
    HIPCHK(k, hipMemcpyAsync(k->h_pin[i & 1], (const char*)src_dev + off, len, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipEventRecord(k->ev_pin[i & 1], k->stream));
    if (prev_len) {
      HIPCHK(k, hipEventSynchronize(k->ev_pin[(i & 1) ^ 1]));
      parallel_memcpy((char*)dst + prev_off, k->h_pin[(i & 1) ^ 1], prev_len);
    }
    prev_off = off;
    prev_len = len;
  }
  if (prev_len) {
    HIPCHK(k, hipEventSynchronize(k->ev_pin[(i - 1) & 1]));
    parallel_memcpy((char*)dst + prev_off, k->h_pin[(i - 1) & 1], prev_len);
  }
  return HSK_OK;
}

extern "C" int hsk_stored_planes(const hsk_ctx* k, int* z0, int* nz) {
  if (!k) return HSK_ERR_ARG;
  if (z0) *z0 = k->vp.zs0;
  if (nz) *nz = k->vp.nzs;
  return HSK_OK;
}

extern "C" int hsk_download_tsdf(hsk_ctx* k, int16_t* out) {
  if (!k || !out) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  flush_weights(k);  // the weights of deep free space live in the summaries until read
  // the caller's array is row-major (x fastest, then y, then plane); the volume is stored in 64-B blocks: the conversion
  // kernel writes a batch of planes straight into one of the pinned buffers while the host moves the other's out
  int r = ensure_pinned(k);
  if (r != HSK_OK) return r;
  const size_t plane_bytes = (size_t)k->vp.X * k->vp.Y * 4;
  if (plane_bytes > k->pin_bytes) return fail(k, HSK_ERR_ARG, "hsk_download_tsdf: a plane of this volume exceeds the staging buffer");
  const int batch = (int)(k->pin_bytes / plane_bytes) < k->vp.nzs ? (int)(k->pin_bytes / plane_bytes) : k->vp.nzs;
  int prev_z = 0, prev_n = 0, i = 0;
  for (int zz0 = 0; zz0 < k->vp.nzs; zz0 += batch, ++i) {
    const int nz = k->vp.nzs - zz0 < batch ? k->vp.nzs - zz0 : batch;
    void* pin_dev = nullptr;
    HIPCHK(k, hipHostGetDevicePointer(&pin_dev, k->h_pin[i & 1], 0));
    launch_vol_to_linear(k->stream, k->d_vol, k->vp, zz0, nz, pin_dev);
    HIPCHK(k, hipEventRecord(k->ev_pin[i & 1], k->stream));
    if (prev_n) {
      HIPCHK(k, hipEventSynchronize(k->ev_pin[(i & 1) ^ 1]));
      parallel_memcpy((char*)out + (size_t)prev_z * plane_bytes, k->h_pin[(i & 1) ^ 1], (size_t)prev_n * plane_bytes);
    }
    prev_z = zz0;
    prev_n = nz;
  }
  if (prev_n) {
    HIPCHK(k, hipEventSynchronize(k->ev_pin[(i - 1) & 1]));
    parallel_memcpy((char*)out + (size_t)prev_z * plane_bytes, k->h_pin[(i - 1) & 1], (size_t)prev_n * plane_bytes);
  }
  HIPCHK(k, hipGetLastError());
  return HSK_OK;
}
extern "C" int hsk_flush_weights(hsk_ctx* k) {
  if (!k) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  flush_weights(k);
  HIPCHK(k, hipGetLastError());
  return HSK_OK;
}
extern "C" int hsk_upload_tsdf(hsk_ctx* k, const int16_t* in) {
  if (!k || !in) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  {
    int r = ensure_pinned(k);
    if (r != HSK_OK) return r;
    const size_t plane_bytes = (size_t)k->vp.X * k->vp.Y * 4;
    if (plane_bytes > k->pin_bytes) return fail(k, HSK_ERR_ARG, "hsk_upload_tsdf: a plane of this volume exceeds the staging buffer");
    const int batch = (int)(k->pin_bytes / plane_bytes) < k->vp.nzs ? (int)(k->pin_bytes / plane_bytes) : k->vp.nzs;
    HIPCHK(k, hipMemsetAsync(k->d_vol, 0, k->vol_bytes, k->stream));  // (the padding planes of the last block row)
    int i = 0;
    for (int zz0 = 0; zz0 < k->vp.nzs; zz0 += batch, ++i) {
      const int nz = k->vp.nzs - zz0 < batch ? k->vp.nzs - zz0 : batch;
      if (i >= 2) HIPCHK(k, hipEventSynchronize(k->ev_pin[i & 1]));  // the kernel that read this buffer two batches ago
      parallel_memcpy(k->h_pin[i & 1], (const char*)in + (size_t)zz0 * plane_bytes, (size_t)nz * plane_bytes);
      void* pin_dev = nullptr;
      HIPCHK(k, hipHostGetDevicePointer(&pin_dev, k->h_pin[i & 1], 0));
      launch_vol_from_linear(k->stream, k->d_vol, k->vp, zz0, nz, pin_dev);
      HIPCHK(k, hipEventRecord(k->ev_pin[i & 1], k->stream));
    }
  }
  k->vol_epoch += 1;
  HIPCHK(k, hipMemsetAsync(k->d_flags, 0, k->flags_bytes, k->stream));
  launch_rebuild_flags(k->stream, k->d_vol, k->vp, k->d_flags);
  launch_rebuild_uniform(k->stream, k->d_vol, k->vp, k->d_uni);
  HIPCHK(k, hipStreamSynchronize(k->stream));
  return HSK_OK;
}

static float* map_ptr(hsk_ctx* k, int kind, int level) {
  switch (kind) {
    case 0: return k->B().d_vcur[level];
    case 1: return k->B().d_ncur[level];
    case 2: return k->d_vmod[level];
    case 3: return k->d_nmod[level];
  }
  return nullptr;
}
extern "C" int hsk_download_map(hsk_ctx* k, int kind, int level, float* out) {
  if (!k || !out || level < 0 || level >= HSK_NLEVELS || kind < 0 || kind > 3) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  const size_t P = (size_t)k->lv[level].W * k->lv[level].H;
  HIPCHK(k, hipMemcpyAsync(out, map_ptr(k, kind, level), P * 12, hipMemcpyDeviceToHost, k->stream));
  HIPCHK(k, hipStreamSynchronize(k->stream));
  return HSK_OK;
}
extern "C" int hsk_upload_map(hsk_ctx* k, int kind, int level, const float* in) {
  if (!k || !in || level < 0 || level >= HSK_NLEVELS || kind < 0 || kind > 3) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  const size_t P = (size_t)k->lv[level].W * k->lv[level].H;
  HIPCHK(k, hipMemcpyAsync(map_ptr(k, kind, level), in, P * 12, hipMemcpyHostToDevice, k->stream));
  HIPCHK(k, hipStreamSynchronize(k->stream));
  return HSK_OK;
}
extern "C" int hsk_download_depth_level(hsk_ctx* k, int level, uint16_t* out) {
  if (!k || !out || level < 0 || level >= HSK_NLEVELS) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  const size_t P = (size_t)k->lv[level].W * k->lv[level].H;
  HIPCHK(k, hipMemcpyAsync(out, k->B().d_dep[level], P * 2, hipMemcpyDeviceToHost, k->stream));
  HIPCHK(k, hipStreamSynchronize(k->stream));
  return HSK_OK;
}
extern "C" int hsk_download_scaled_depth(hsk_ctx* k, float* out) {
  if (!k || !out) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  HIPCHK(k, hipMemcpyAsync(out, k->B().d_scaled, (size_t)k->lv[0].W * k->lv[0].H * 4, hipMemcpyDeviceToHost, k->stream));
  HIPCHK(k, hipStreamSynchronize(k->stream));
  return HSK_OK;
}

static int ensure_cube_table(hsk_ctx* k) {
  if (k->d_cube_tab) return HSK_OK;
  CubeTable ct;
  if (hsk_build_cube_table(&ct) != HSK_MC_MAXT) return fail(k, HSK_ERR_STATE, "marching-cubes table: a case with more triangles than the table holds");
  HIPCHK(k, hipMalloc((void**)&k->d_cube_tab, sizeof(CubeTable)));
  HIPCHK(k, hipMemcpy(k->d_cube_tab, &ct, sizeof(CubeTable), hipMemcpyHostToDevice));
  return HSK_OK;
}
static int ensure_row_tables(hsk_ctx* k) {
  if (k->d_rowcnt) return HSK_OK;
  const int nrows = k->vp.Y * (k->vp.zo1 - k->vp.zo0);  // (>= the mesh rows: one pair of buffers for every product)
  HIPCHK(k, hipMalloc((void**)&k->d_rowcnt, (size_t)nrows * 4));
  HIPCHK(k, hipMalloc((void**)&k->d_rowoff, hsk_scan_scratch_entries(nrows) * 8));
  return HSK_OK;
}
static int ensure_product_bytes(hsk_ctx* k, size_t want) {
  if (k->out_bytes >= want) return HSK_OK;
  if (k->d_out) (void)hipFree(k->d_out);
  k->d_out = nullptr;
  k->out_bytes = 0;
  HIPCHK(k, hipMalloc(&k->d_out, want));
  k->out_bytes = want;
  return HSK_OK;
}

// A product of the volume (cloud, mesh): counted row by row, the rows' offsets scanned, then written in voxel order.  The
// callers' protocol is a size query (null buffer) followed by the fill: the second call finds the counts and offsets of
// the first in place when nothing has touched the volume in between (ro_kind / ro_epoch) -- the count sweep ran twice
// per product before.  The product is written into a device buffer that only ever grows and reaches the caller through
// the pinned pair (copy_out).
template <class Count, class Fill>
static int extract_product(hsk_ctx* k, int kind, size_t elem_bytes, float* out, size_t cap, size_t* n_out, Count count, Fill fill) {
  {
    const int r = ensure_row_tables(k);
    if (r != HSK_OK) return r;
  }
  // (NO flush of the deferred weights here, round 5: the products ask of a weight only whether it is zero, and a weight the
  // summaries hold ahead of the volume's copy is never that -- a block leaves "never observed" with a store of (+1, 1),
  // and every deferred state has all 16 weights >= 1 in the volume itself; the TSDF values are always current.  Only
  // hsk_download_tsdf, which hands the weights out, brings them up to date.  A host that shows a cloud after every
  // frame pays for the cloud, not for rewriting the frustum's free space.)
  if (!(k->ro_kind == kind && k->ro_epoch == k->vol_epoch)) {
    k->ro_kind = 0;
    count();
    unsigned long long total = 0;
    HIPCHK(k, hipMemcpyAsync(&total, k->d_counter, 8, hipMemcpyDeviceToHost, k->stream));
    HIPCHK(k, hipStreamSynchronize(k->stream));
    k->ro_kind = kind;
    k->ro_epoch = k->vol_epoch;
    k->ro_total = total;
  }
  *n_out = (size_t)k->ro_total;
  if (!out || cap == 0 || k->ro_total == 0) return HSK_OK;
  const size_t nw = k->ro_total < cap ? (size_t)k->ro_total : cap;
  if (k->out_bytes < nw * elem_bytes) {
    const int r = ensure_product_bytes(k, nw * elem_bytes + (nw * elem_bytes >> 2));  // (a quarter more: a scan grows from call to call)
    if (r != HSK_OK) return r;
  }
  fill((float*)k->d_out, nw);
  return copy_out(k, out, k->d_out, nw * elem_bytes);
}

extern "C" int hsk_prepare_readout(hsk_ctx* k, size_t product_bytes) {
  if (!k) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  int r = ensure_pinned(k);
  if (r == HSK_OK) r = ensure_row_tables(k);
  if (r == HSK_OK) r = ensure_cube_table(k);
  if (r == HSK_OK) r = ensure_product_bytes(k, product_bytes ? product_bytes : (size_t)48 << 20);
  if (r == HSK_OK) HIPCHK(k, (hipError_t)extract_warm());  // (the read-out kernels' code object: 0.7 ms of a process's first product)
  return r;
}

extern "C" int hsk_extract_cloud(hsk_ctx* k, float* xyz, size_t cap_points, size_t* n_points) {
  if (!k || !n_points) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  return extract_product(
      k, 1, 12, xyz, cap_points, n_points,
      [&]() { launch_extract(k->stream, k->d_vol, k->vp, k->d_rowcnt, k->d_rowoff, k->d_counter, nullptr, 0, 0, k->d_flags); },
      [&](float* d, size_t nw) { launch_extract(k->stream, k->d_vol, k->vp, k->d_rowcnt, k->d_rowoff, k->d_counter, d, nw, 1, k->d_flags); });
}

// Triangle soup (9 floats per triangle) of the TSDF zero level set, marching tetrahedra, voxel order.
extern "C" int hsk_extract_mesh(hsk_ctx* k, float* tri_xyz, size_t cap_triangles, size_t* n_triangles) {
  if (!k || !n_triangles) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  TetTable tt;
  hsk_build_tet_table(&tt);
  return extract_product(
      k, 2, 36, tri_xyz, cap_triangles, n_triangles,
      [&]() { launch_extract_mesh(k->stream, k->d_vol, k->vp, tt, k->d_rowcnt, k->d_rowoff, k->d_counter, nullptr, 0, 0, k->d_flags); },
      [&](float* d, size_t nw) { launch_extract_mesh(k->stream, k->d_vol, k->vp, tt, k->d_rowcnt, k->d_rowoff, k->d_counter, d, nw, 1, k->d_flags); });
}

// The same level set by MARCHING CUBES (the form upstream's .ply export has, README.md:16-17): about half the triangles
// of the tetrahedra form.  Table generated by hsk_build_cube_table (PCL's own is not in the reference).
extern "C" int hsk_extract_mesh_cubes(hsk_ctx* k, float* tri_xyz, size_t cap_triangles, size_t* n_triangles) {
  if (!k || !n_triangles) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  {
    const int r = ensure_cube_table(k);
    if (r != HSK_OK) return r;
  }
  return extract_product(
      k, 3, 36, tri_xyz, cap_triangles, n_triangles,
      [&]() { launch_extract_mesh_mc(k->stream, k->d_vol, k->vp, k->d_cube_tab, k->d_rowcnt, k->d_rowoff, k->d_counter, nullptr, 0, 0, k->d_flags); },
      [&](float* d, size_t nw) { launch_extract_mesh_mc(k->stream, k->d_vol, k->vp, k->d_cube_tab, k->d_rowcnt, k->d_rowoff, k->d_counter, d, nw, 1, k->d_flags); });
}

// ------------------------------------------------------------------------------------------------------
// profiling
// ------------------------------------------------------------------------------------------------------
extern "C" int hsk_set_profiling(hsk_ctx* k, int on) {
  if (!k) return HSK_ERR_ARG;
  k->prof = on != 0;
  k->prof_levels = on == 2;
  return HSK_OK;
}
extern "C" int hsk_stage_ms(hsk_ctx* k, double sum_ms[HSK_NSTAGES], uint64_t* n_frames, int reset) {
  if (!k) return HSK_ERR_ARG;
  if (sum_ms)
    for (int i = 0; i < HSK_NSTAGES; ++i) sum_ms[i] = k->stage_ms[i];
  if (n_frames) *n_frames = k->prof_frames;
  if (reset) {
    for (int i = 0; i < HSK_NSTAGES; ++i) k->stage_ms[i] = 0.0;
    for (int i = 0; i < HSK_NLEVELS; ++i) k->icp_level_ms[i] = 0.0;
    k->prof_frames = 0;
  }
  return HSK_OK;
}
// lane-blocks (4 x 1 x 4 voxels) the last integrate's classification pass handed to its per-voxel pass
extern "C" int hsk_integrate_queue_entries(hsk_ctx* k, uint64_t* n_entries) {
  if (!k || !n_entries) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  const size_t words = integrate_queue_counter_words();
  unsigned* h = (unsigned*)malloc(words * 4);
  if (!h) return fail(k, HSK_ERR_STATE, "out of host memory");
  hipError_t e = hipMemcpyAsync(h, k->d_queue, words * 4, hipMemcpyDeviceToHost, k->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(k->stream);
  const uint64_t n = e == hipSuccess ? integrate_queue_entries(h) : 0;
  free(h);
  HIPCHK(k, e);
  *n_entries = n;
  return HSK_OK;
}
// ... and of those, the lane-blocks of the LIGHT class (free space with holes in the depth image under it: hsk_integrate_queue_entries
// counts the per-voxel class only)
extern "C" int hsk_integrate_light_entries(hsk_ctx* k, uint64_t* n_entries) {
  if (!k || !n_entries) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  const size_t words = integrate_queue_counter_words();
  unsigned* h = (unsigned*)malloc(words * 4);
  if (!h) return fail(k, HSK_ERR_STATE, "out of host memory");
  hipError_t e = hipMemcpyAsync(h, k->d_queue, words * 4, hipMemcpyDeviceToHost, k->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(k->stream);
  const uint64_t n = e == hipSuccess ? integrate_queue_light_entries(h) : 0;
  free(h);
  HIPCHK(k, e);
  *n_entries = n;
  return HSK_OK;
}
extern "C" int hsk_integrate_coarse_counts(hsk_ctx* k, uint64_t counts[4]) {
  if (!k || !counts) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  const size_t n = integrate_chunk_count(k->vp);
  unsigned char* h = (unsigned char*)malloc(2 * n);
  if (!h) return fail(k, HSK_ERR_STATE, "out of host memory");
  hipError_t e = hipMemcpyAsync(h, (const char*)k->d_zint + integrate_cflag_offset_bytes(k->vp), n, hipMemcpyDeviceToHost, k->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h + n, k->d_uni + uniform_lane_bytes(k->vp), n, hipMemcpyDeviceToHost, k->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(k->stream);
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  for (size_t i = 0; i < n && e == hipSuccess; ++i) {
    if (h[i] < 3) counts[h[i]] += 1;
    if (h[n + i] != 0) counts[3] += 1;
  }
  free(h);
  HIPCHK(k, e);
  return HSK_OK;
}
// host microseconds the pipelined submissions have spent, by phase, and how many there were (reset != 0: counted from now on)
extern "C" int hsk_submit_host_us(hsk_ctx* k, double sum_us[4], uint64_t* n_submissions, int reset) {
  if (!k) return HSK_ERR_ARG;
  if (sum_us)
    for (int i = 0; i < 4; ++i) sum_us[i] = k->submit_us[i];
  if (n_submissions) *n_submissions = k->submit_n;
  if (reset) {
    for (auto& v : k->submit_us) v = 0.0;
    k->submit_n = 0;
  }
  return HSK_OK;
}
extern "C" int hsk_icp_level_ms(hsk_ctx* k, double sum_ms[HSK_LEVELS]) {
  if (!k || !sum_ms) return HSK_ERR_ARG;
  for (int i = 0; i < HSK_NLEVELS; ++i) sum_ms[i] = k->icp_level_ms[i];
  return HSK_OK;
}

// ------------------------------------------------------------------------------------------------------
// multi-GPU (z-slab) building blocks: enqueue-only, the host runs its collective between them
// ------------------------------------------------------------------------------------------------------
extern "C" int hsk_mgpu_frame_index(const hsk_ctx* k) { return k ? k->frame : -1; }

// Optional: enqueue the depth copy and the preprocessing of the NEXT frame on the second stream, into the buffer set
// the current frame does not use, so that they run under the current frame's ICP / integrate / raycast and collectives.
// The following hsk_mgpu_frame_begin with the same pointer picks the result up instead of preprocessing again.
extern "C" int hsk_mgpu_prefetch(hsk_ctx* k, const void* depth_dev, int w, int h) {
  int r = check_dims(k, depth_dev, w, h);
  if (r != HSK_OK) return r;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  const int set = k->mgpu_set ^ 1;
  if (k->set_used[set]) HIPCHK(k, hipStreamWaitEvent(k->pstream, k->ev_free[set], 0));  // its previous frame has finished
  const int keep = k->cur;
  k->cur = set;
  // (No ordering against the context's stream: the point of the prefetch is to run beside the frame that stream is busy
  // with.  The contract -- include/hskinfu.h -- asks for a frame that is complete when this is called; it may live in
  // device memory or in pinned host memory, hence hipMemcpyDefault.)
  hipError_t e = hipMemcpyAsync(k->B().d_raw, depth_dev, (size_t)w * h * 2, hipMemcpyDefault, k->pstream);
  if (e == hipSuccess) {
    enqueue_preprocess(k, k->pstream);
    e = hipEventRecord(k->ev_pre[set], k->pstream);
  }
  k->cur = keep;
  HIPCHK(k, e);
  k->pf_ptr = depth_dev;
  k->pf_set = set;
  return HSK_OK;
}

extern "C" int hsk_mgpu_frame_begin(hsk_ctx* k, const void* depth_dev, int w, int h) {
  int r = check_dims(k, depth_dev, w, h);
  if (r != HSK_OK) return r;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  if (k->pending_reset) {  // a pipelined frame (hsk_mgpu_frame_end_async) lost tracking: restart the scan with this frame
    r = reset_behind_lost_frame(k);
    if (r != HSK_OK) return r;
  }
  if (k->pf_ptr == depth_dev && k->pf_set >= 0) {
    // preprocessed ahead of time by hsk_mgpu_prefetch: the frame works on that buffer set once the event has fired
    k->mgpu_set = k->pf_set;
    k->cur = k->mgpu_set;
    HIPCHK(k, hipStreamWaitEvent(k->stream, k->ev_pre[k->mgpu_set], 0));
  } else {
    k->cur = k->mgpu_set;  // same set as the previous frame: stream order protects it
    HIPCHK(k, hipMemcpyAsync(k->B().d_raw, depth_dev, (size_t)w * h * 2, hipMemcpyDefault, k->stream));
    enqueue_preprocess(k, k->stream);
  }
  k->pf_ptr = nullptr;
  k->pf_set = -1;
  if (k->frame == 0) {
    enqueue_integrate(k);
    for (int l = 0; l < HSK_NLEVELS; ++l)
      launch_transform_maps(k->stream, k->B().d_vcur[l], k->B().d_ncur[l], k->lv[l].W * k->lv[l].H, k->d_st, k->d_vmod[l],
                            k->d_nmod[l]);
  } else {
    launch_begin_frame(k->stream, k->d_st, nullptr);
  }
  return HSK_OK;
}

// hsk_mgpu_frame_begin + hsk_mgpu_icp_replicated + hsk_mgpu_integrate + hsk_mgpu_raycast_local as ONE call.  The part
// after the preprocessing (about 25 launches) is captured into a hipGraph per buffer set on first use: the z-slab host
// then issues five operations per frame (this, two collectives, resolve, frame end) and stops being launch-bound.
// keys_dev must be the same buffer on every call (it is part of the captured graph).
extern "C" int hsk_mgpu_frame_front(hsk_ctx* k, const void* depth_dev, int w, int h, void* keys_dev) {
  if (!keys_dev) return k ? fail(k, HSK_ERR_ARG, "keys buffer is null") : HSK_ERR_ARG;
  int r = hsk_mgpu_frame_begin(k, depth_dev, w, h);
  if (r != HSK_OK) return r;
  if (k->frame == 0) return HSK_OK;  // first frame: integrate + transformed maps only (frame_begin did it)
  const int set = k->cur;
  const size_t P0 = (size_t)k->lv[0].W * k->lv[0].H;
  auto body = [&]() {
    enqueue_icp(k);  // (k_begin_frame ran in frame_begin; the first fused iteration repeats its bookkeeping: idempotent)
    enqueue_integrate(k);
    launch_raycast(k->stream, k->d_vol, k->d_st, k->vp, k->lv[0].W, k->lv[0].H, k->lv[0].in, k->d_vmod[0], k->d_nmod[0], k->d_keys,
                   k->d_flags);
    (void)hipMemcpyAsync(keys_dev, k->d_keys, P0 * 4, hipMemcpyDeviceToDevice, k->stream);
  };
  if (!k->cfg.use_graph || k->stream == nullptr) {  // (the legacy default stream cannot be captured)
    body();
    HIPCHK(k, hipGetLastError());
    return HSK_OK;
  }
  if (k->sgraph_keys != keys_dev) {  // a different keys buffer: the captured graphs are stale
    for (int i = 0; i < 2; ++i) {
      if (k->sgexec[i]) (void)hipGraphExecDestroy(k->sgexec[i]);
      if (k->sgraph[i]) (void)hipGraphDestroy(k->sgraph[i]);
      k->sgexec[i] = nullptr;
      k->sgraph[i] = nullptr;
    }
    k->sgraph_keys = keys_dev;
  }
  if (!k->sgexec[set]) {
    HIPCHK(k, hipStreamBeginCapture(k->stream, hipStreamCaptureModeThreadLocal));
    body();
    HIPCHK(k, hipStreamEndCapture(k->stream, &k->sgraph[set]));
    HIPCHK(k, hipGraphInstantiate(&k->sgexec[set], k->sgraph[set], nullptr, nullptr, 0));
  }
  HIPCHK(k, hipGraphLaunch(k->sgexec[set], k->stream));
  k->weights_pending = true;
  k->vol_epoch += 1;
  return HSK_OK;
}

extern "C" int hsk_mgpu_icp_accumulate(hsk_ctx* k, int level, int row0, int row1, void* sums27_dev) {
  if (!k || !sums27_dev || level < 0 || level >= HSK_NLEVELS) return HSK_ERR_ARG;
  if (row0 < 0 || row1 > k->lv[level].H || row0 > row1) return fail(k, HSK_ERR_ARG, "row range out of bounds");
  if (row0 == row1) {
    HIPCHK(k, hipMemsetAsync(sums27_dev, 0, 27 * sizeof(double), k->stream));
    return HSK_OK;
  }
  const int nb = icp_num_blocks(k->lv[level].W, row1 - row0);
  launch_icp_accumulate(k->stream, k->B().d_vcur[level], k->B().d_ncur[level], k->d_vmod[level], k->d_nmod[level],
                        k->lv[level].W, k->lv[level].H, k->lv[level].in, k->d_st, k->cfg.icp_dist_thresh_m,
                        k->cfg.icp_angle_thresh_sin, row0, row1, k->d_partials);
  launch_icp_reduce(k->stream, k->d_partials, nb, (double*)sums27_dev);
  return HSK_OK;
}

extern "C" int hsk_mgpu_icp_update(hsk_ctx* k, const void* sums27_dev) {
  if (!k || !sums27_dev) return HSK_ERR_ARG;
  launch_icp_update(k->stream, (const double*)sums27_dev, k->d_st);
  return HSK_OK;
}

extern "C" int hsk_mgpu_icp_replicated(hsk_ctx* k) {
  if (!k) return HSK_ERR_ARG;
  if (k->frame == 0) return fail(k, HSK_ERR_STATE, "no model maps yet: the first frame has no ICP");
  enqueue_icp(k);  // the first fused iteration seeds itself from the tracker state (k_begin_frame already ran: idempotent)
  return HSK_OK;
}

extern "C" int hsk_mgpu_integrate(hsk_ctx* k) {
  if (!k) return HSK_ERR_ARG;
  enqueue_integrate(k);
  return HSK_OK;
}

extern "C" int hsk_mgpu_raycast_local(hsk_ctx* k, void* keys_dev) {
  if (!k || !keys_dev) return HSK_ERR_ARG;
  launch_raycast(k->stream, k->d_vol, k->d_st, k->vp, k->lv[0].W, k->lv[0].H, k->lv[0].in, k->d_vmod[0], k->d_nmod[0],
                 k->d_keys, k->d_flags);
  HIPCHK(k, hipMemcpyAsync(keys_dev, k->d_keys, (size_t)k->lv[0].W * k->lv[0].H * 4, hipMemcpyDeviceToDevice, k->stream));
  return HSK_OK;
}

extern "C" int hsk_mgpu_raycast_resolve(hsk_ctx* k, const void* keys_min_dev, void* maps_bits_dev) {
  if (!k || !keys_min_dev || !maps_bits_dev) return HSK_ERR_ARG;
  launch_resolve(k->stream, k->d_keys, (const int*)keys_min_dev, k->d_vmod[0], k->d_nmod[0], (int*)maps_bits_dev,
                 k->lv[0].W * k->lv[0].H);
  return HSK_OK;
}

// direct exchange: this slab's winning pixels straight into every device's composite buffer (dest_bits[d]: int32[6 P],
// device-accessible from this context's device -- local, peer-enabled or IPC-mapped memory)
extern "C" int hsk_mgpu_raycast_push(hsk_ctx* k, const void* keys_min_dev, void* const* dest_bits, int n_dest) {
  if (!k || !keys_min_dev || !dest_bits || n_dest < 1 || n_dest > HSK_PUSH_MAX) return HSK_ERR_ARG;
  PushDests d;
  d.n = n_dest;
  for (int i = 0; i < HSK_PUSH_MAX; ++i) d.p[i] = i < n_dest ? (int*)dest_bits[i] : nullptr;
  launch_resolve_push(k->stream, k->d_keys, (const int*)keys_min_dev, k->d_vmod[0], k->d_nmod[0], d, k->lv[0].W * k->lv[0].H);
  return HSK_OK;
}

// 1 when the next frame starts (or restarts) the scan: frame 0, or tracking was lost by a pipelined frame.  Such a frame
// has no ICP and no composite: hsk_mgpu_frame_begin / hsk_mgpu_frame_front, then the synchronous hsk_mgpu_frame_end.
extern "C" int hsk_mgpu_restart_pending(const hsk_ctx* k) { return k ? (k->frame == 0 || k->pending_reset) : -1; }

// Pipelined end of a tracked slab frame: adopt the composite, rebuild the model pyramid, and queue the pose read-back
// instead of waiting for it -- the host goes on to enqueue the next frame; hsk_wait_frame returns the results in order
// (at most HSK_MAX_IN_FLIGHT outstanding).  Loss semantics are those of hsk_submit_frame_dev: frames already enqueued
// behind a lost one are dropped on the device and report tracked = 0.
extern "C" int hsk_mgpu_frame_end_async(hsk_ctx* k, const void* keys_min_dev, const void* maps_bits_dev) {
  if (!k) return HSK_ERR_ARG;
  if (k->frame == 0 || k->pending_reset) return fail(k, HSK_ERR_STATE, "the first frame of a scan ends with hsk_mgpu_frame_end");
  if (!keys_min_dev || !maps_bits_dev) return fail(k, HSK_ERR_ARG, "composite buffers are null");
  if (k->ring_count >= HSK_MAX_IN_FLIGHT) return fail(k, HSK_ERR_STATE, "too many frames in flight: call hsk_wait_frame first");
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  // the frame's last kernel reports the tracker state into the ring slot assigned here (no copy node, no event record
  // for the pose: see hsk_submit_frame_dev)
  const int slot = (k->ring_head + k->ring_count) % (HSK_MAX_IN_FLIGHT + 1);
  k->h_slot_fifo[k->ring_seq % HSK_RING_FIFO] = slot;
  k->ring_seq += 1;
  k->ring_expect[slot] = k->ring_seq | 0x80000000u;
  ((volatile TrackState*)&k->h_ring[slot])->ring_mark = 0u;
  ((volatile TrackState*)&k->h_ring[slot])->pose_mark = 0u;
  const RingOut ring = {k->d_ring_view, k->d_fifo_view, k->d_ring_seq};
  // (round 6: adopt, the model pyramid and the report in ONE launch -- they were two, with level 0 written and read back between them)
  launch_adopt_pyramid(k->stream, (const int*)keys_min_dev, (const int*)maps_bits_dev, k->lv[0].W, k->lv[0].H, k->d_vmod[0], k->d_nmod[0],
                       k->d_vmod[1], k->d_nmod[1], k->d_vmod[2], k->d_nmod[2], k->d_st, &ring);
  HIPCHK(k, hipEventRecord(k->ev_free[k->cur], k->stream));  // the prefetch of the frame after next waits on it (second stream)
  k->set_used[k->cur] = true;
  k->ring_kind[slot] = 0;
  k->ring_count += 1;
  HIPCHK(k, hipGetLastError());
  return HSK_OK;
}

extern "C" int hsk_mgpu_frame_end(hsk_ctx* k, const void* keys_min_dev, const void* maps_bits_dev, float pose_out[16],
                                  int* tracked) {
  if (!k) return HSK_ERR_ARG;
  HIPCHK(k, hipSetDevice(k->cfg.device_id));
  const bool first = (k->frame == 0);
  if (!first && k->ring_count > 0) return fail(k, HSK_ERR_STATE, "frames are in flight: collect them with hsk_wait_frame first");
  if (!first) {
    if (!keys_min_dev || !maps_bits_dev) return fail(k, HSK_ERR_ARG, "composite buffers are null");
    launch_adopt_pyramid(k->stream, (const int*)keys_min_dev, (const int*)maps_bits_dev, k->lv[0].W, k->lv[0].H, k->d_vmod[0], k->d_nmod[0],
                         k->d_vmod[1], k->d_nmod[1], k->d_vmod[2], k->d_nmod[2], k->d_st, nullptr);
  }
  // the frame's image buffers are free again once everything enqueued so far has run
  HIPCHK(k, hipEventRecord(k->ev_free[k->cur], k->stream));
  k->set_used[k->cur] = true;  // k->cur keeps naming this frame's set: downloads of its images read the right buffers
  int r = download_state(k);
  if (r != HSK_OK) return r;
  HIPCHK(k, hipGetLastError());
  if (first) {
    k->frame = 1;
    if (pose_out) rt_to_pose16(k->h_st->R, k->h_st->t, pose_out);
    if (tracked) *tracked = 0;
    return HSK_OK;
  }
  if (k->h_st->lost) {
    r = do_reset(k);
    if (r != HSK_OK) return r;
    if (pose_out) rt_to_pose16(k->h_st->R, k->h_st->t, pose_out);
    if (tracked) *tracked = 0;
    return HSK_OK;
  }
  k->frame += 1;
  if (pose_out) rt_to_pose16(k->h_st->R, k->h_st->t, pose_out);
  if (tracked) *tracked = 1;
  return HSK_OK;
}
