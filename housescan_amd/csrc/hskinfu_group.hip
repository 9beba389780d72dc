// hskinfu_group.hip -- ONE TSDF volume sharded as z-slabs over several GPUs, behind the C ABI (SURVEY.md 8(b), 8(e);
// BASELINE.json configs[3]).  The caller -- HouseScan's frame loop, Main.hs:1282-1313 -- keeps feeding whole depth frames
// (HoniHelper.hs:20); the slab frame loop and its collectives live here, against RCCL's C API directly:
//
//   per frame   every slab: preprocess + 19 ICP iterations + integrate + slab-local raycast   (hsk_mgpu_frame_front)
//               MIN of the per-pixel step keys over all slabs     -> which slab saw each ray end first
//               the winner contributes the bit patterns of its vertex / normal, SUM over all slabs (exact, keeps NaN)
//               every slab adopts the composite, rebuilds the model pyramid, reports the pose     (hsk_mgpu_frame_end*)
//
// Slabs that share a device are combined by a small kernel; across devices ncclAllReduce does it (one communicator rank
// per distinct device of this process, ncclCommInitAll, or one rank per process, ncclCommInitRank).  RCCL is loaded at
// run time (dlopen): a process that already holds a copy -- torch does -- must get that one, and a single-device group
// needs none.  One stream per device carries its slabs' kernels and its collectives, so stream order is the only
// synchronisation on the frame path.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hskinfu.h"

namespace {

struct Rccl {
  void* so = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// the process-wide RCCL entry points; an error string when the library or a symbol is missing
Rccl* rccl(std::string* err) {
  static Rccl R;
  static std::once_flag once;
  static std::string why;
  // (groups are driven from several host threads -- one per room in BASELINE configs[4] -- so the first calls may race)
  std::call_once(once, [&]() {
    // One RCCL per process, and the one that belongs to the HIP runtime in use: (1) the copy the host names
    // (HSK_RCCL_PATH: the Python mirror sets it to torch's when it shares torch's HIP runtime -- a second copy beside
    // torch's corrupts the heap at exit), (2) a copy the process has loaded already, (3) the system's.
    const char* named = getenv("HSK_RCCL_PATH");
    if (named && *named) R.so = dlopen(named, RTLD_NOW | RTLD_LOCAL);
    for (const char* name : {"librccl.so", "librccl.so.1"})
      if (!R.so) R.so = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!R.so) R.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!R.so) {
      why = std::string("RCCL is not loadable: ") + dlerror();
    } else {
#define SYM(f)                                                              \
  R.f = (decltype(R.f))dlsym(R.so, "nccl" #f);                              \
  if (!R.f && why.empty()) why = "RCCL lacks the symbol nccl" #f;
      SYM(GetUniqueId) SYM(CommInitRank) SYM(CommInitAll) SYM(CommDestroy) SYM(CommAbort) SYM(CommCount) SYM(AllReduce) SYM(GroupStart) SYM(GroupEnd)
      SYM(GetErrorString)
#undef SYM
    }
  });
  if (!why.empty()) {
    if (err) *err = why;
    return nullptr;
  }
  return &R;
}

__global__ void k_min_into(int* __restrict__ acc, const int* __restrict__ x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] = min(acc[i], x[i]);
}
__global__ void k_add_into(int* __restrict__ acc, const int* __restrict__ x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] = acc[i] + x[i];  // bit patterns: at most one slab contributes a non-zero word per pixel
}
__global__ void k_add_into_f64(double* __restrict__ acc, const double* __restrict__ x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] = acc[i] + x[i];  // exact: the ICP products are multiples of 2^-26
}

// direct exchange, step 1: a slab's step keys into its slot of every device's gather buffer (peer-mapped memory)
#define HSK_GROUP_MAX_DEV 16
#define HSK_PUSH_MAX_HOST 16  // hsk_mgpu_raycast_push takes at most this many destinations (hsk_dev.h: HSK_PUSH_MAX)
struct KeyDests {
  int* p[HSK_GROUP_MAX_DEV];
  int n;
};
__global__ void k_push_keys(const int* __restrict__ keys, KeyDests dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int v = keys[i];
  for (int d = 0; d < dst.n; ++d) dst.p[d][i] = v;
}
// ... step 2: MIN over the slots of all slabs (this device's copy of the gather buffer)
__global__ void k_min_slots(const int* __restrict__ slots, int n_slots, int n, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int m = slots[i];
  for (int s = 1; s < n_slots; ++s) m = min(m, slots[(size_t)s * n + i]);
  out[i] = m;
}

// shared page of the direct exchange: flags F1 / F2 [destination device][source device] (frame sequence numbers, written
// by hipStreamWriteValue32 on the source's stream, waited for by hipStreamWaitValue32 on the destination's), and, in the
// rank form, the bootstrap area through which the processes of one node hand each other their hipIpc memory handles
struct DirectPage {
  unsigned f1[HSK_GROUP_MAX_DEV][HSK_GROUP_MAX_DEV];
  unsigned f2[HSK_GROUP_MAX_DEV][HSK_GROUP_MAX_DEV];
  volatile int ready[HSK_GROUP_MAX_DEV], attached[HSK_GROUP_MAX_DEV];
  hipIpcMemHandle_t kg[HSK_GROUP_MAX_DEV], cm[HSK_GROUP_MAX_DEV];
};

struct Slab {
  hsk_ctx* k = nullptr;
  int dev_slot = 0;       // index into Group::devs
  int index = 0;          // global slab number (row shard of the all-reduced ICP)
  int* keys = nullptr;    // int32[P]   this slab's step keys
  int* bits = nullptr;    // int32[6P]  this slab's contribution to the composite
  double* sums = nullptr; // double[27] this slab's share of an ICP iteration (all-reduce mode)
};

struct Dev {
  int id = 0;
  hipStream_t stream = nullptr;
  int* kmin = nullptr;        // MIN of the keys over all slabs (after the collective: over all devices)
  int* bsum = nullptr;        // SUM of the bit patterns
  double* sums = nullptr;     // double[27]
  ncclComm_t comm = nullptr;
  std::vector<int> slabs;     // indices into Group::slabs
  // direct exchange
  int gidx = 0;               // this device's number among ALL devices of the group (rank form: the rank)
  int* kg = nullptr;          // gather buffer: step keys of every slab of the group, int32[n_slabs_total][P]
  int* cm = nullptr;          // composite: vertex / normal bit patterns of the winners, int32[6 P]
  int* kg_of[HSK_GROUP_MAX_DEV] = {};  // every device's gather buffer / composite as THIS device addresses it
  int* cm_of[HSK_GROUP_MAX_DEV] = {};
  // HSK_GROUP_PROFILE: frame start, exchange start / end -- two sets, used by alternate frames, so that a set is read
  // (at the submission two frames later) long after its frame has been collected: reading the previous frame's would
  // make every submission wait for the frame in flight
  hipEvent_t ev_s0[2] = {}, ev_x0[2] = {}, ev_x1[2] = {};
};

}  // namespace

struct hsk_group {
  hsk_config cfg;
  int flags = 0;
  int world = 1, rank0 = 0;  // communicator size and the rank of this process's first device
  int n_slabs_total = 1;
  bool use_rccl = false;
  std::vector<Slab> slabs;
  std::vector<Dev> devs;
  uint16_t* h_stage = nullptr;  // pinned ring for host frames
  unsigned stage_turn = 0;
  // one entry per submitted frame, oldest first.  ready: the result is already here (the first frame of a (re)started
  // scan completes at submission; frames that were in flight behind a lost one are collected when the scan restarts);
  // otherwise hsk_group_wait_frame collects it from the slabs
  struct Pending {
    bool ready = false;
    float pose[16] = {};
    int tracked = 0;
  };
  std::deque<Pending> fifo;
  int in_flight() const { return (int)fifo.size(); }
  // A frame could only be enqueued on some slabs / devices (an error in the middle of group_enqueue), or the slabs
  // disagree: their frame counts differ from here on and, with several ranks, the peers sit in a collective this rank
  // never joined.  Every later call fails until hsk_group_reset succeeds (single process) or the group is destroyed.
  bool poisoned = false;
  std::string poison_why;
  std::string err;
  // direct exchange (HSK_GROUP_DIRECT)
  bool direct = false;
  int n_dev_total = 1;             // devices of the whole group (rank form: world)
  unsigned seq = 0;                // exchanges enqueued so far (the value the flags are raised to)
  DirectPage* page = nullptr;      // host view of the shared page
  DirectPage* page_dev[HSK_GROUP_MAX_DEV] = {};  // ... as each LOCAL device addresses it (index: slot in devs)
  bool page_is_shm = false;
  std::vector<void*> ipc_opened;   // peer buffers opened through hipIpcOpenMemHandle (rank form)
  double exch_ms = 0.0, front_ms = 0.0;  // HSK_GROUP_PROFILE
  unsigned long long exch_frames = 0;
  bool exch_pending[2] = {false, false};
  int ev_set = 0;
};

static thread_local std::string g_group_err;

#define GFAIL(g, code, msg) \
  do {                      \
    (g)->err = (msg);       \
    return (code);          \
  } while (0)
#define GPOISONED(g)                                                                                                           \
  do {                                                                                                                         \
    if ((g)->poisoned) {                                                                                                       \
      (g)->err = "the group is poisoned by an earlier failure in the middle of a frame (" + (g)->poison_why +                 \
                 "): hsk_group_reset or hsk_group_destroy";                                                                   \
      return HSK_ERR_STATE;                                                                                                    \
    }                                                                                                                          \
  } while (0)
#define GHIP(g, call)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      (g)->err = std::string(#call " failed: ") + hipGetErrorString(e_);                               \
      return HSK_ERR_HIP;                                                                              \
    }                                                                                                  \
  } while (0)
#define GSLAB(g, s, call)                                                                              \
  do {                                                                                                 \
    int r_ = (call);                                                                                   \
    if (r_ != HSK_OK) {                                                                                \
      (g)->err = std::string("slab ") + std::to_string((s).index) + ": " + hsk_last_error((s).k);     \
      return r_;                                                                                       \
    }                                                                                                  \
  } while (0)
#define GNCCL(g, R, call)                                                                              \
  do {                                                                                                 \
    ncclResult_t r_ = (call);                                                                          \
    if (r_ != ncclSuccess) {                                                                           \
      (g)->err = std::string(#call " failed: ") + (R)->GetErrorString(r_);                             \
      return HSK_ERR_HIP;                                                                              \
    }                                                                                                  \
  } while (0)

// owned planes of slab i of n: contiguous, cover [0, Z) exactly, differ by at most one plane
static void slab_range(int i, int n, int Z, int* z0, int* z1) {
  const int base = Z / n, rem = Z % n;
  *z0 = i * base + (i < rem ? i : rem);
  *z1 = *z0 + base + (i < rem ? 1 : 0);
}
// planes a slab stores beyond what it owns: an owned raycast step (its FAR sample lies in an owned plane) reads its near
// sample (one step back) and the refined vertex within [t - step, t + 2 step] (deviation D3): two steps either side of
// the far sample's plane at most, +-1 cell for the normal taps, +-1 voxel for the trilinear taps
static int slab_halo(const hsk_config* c) {
  float m = c->vol_size_m[0] / (float)c->vol_x;
  const float cy = c->vol_size_m[1] / (float)c->vol_y, cz = c->vol_size_m[2] / (float)c->vol_z;
  m = m > cy ? m : cy;
  m = m > cz ? m : cz;
  const float tau = c->trunc_dist_m > 2.1f * m ? c->trunc_dist_m : 2.1f * m;
  const float steps = 2.0f * (0.8f * tau) / cz;
  int h = (int)steps;
  if ((float)h < steps) ++h;
  return h + 3;
}

static void group_free(hsk_group* g) {
  if (!g) return;
  // a group that never built a communicator (single device, no FORCE_RCCL) must not load RCCL at teardown either
  bool has_comm = false;
  for (auto& d : g->devs) has_comm = has_comm || d.comm != nullptr;
  Rccl* R = has_comm ? rccl(nullptr) : nullptr;
  for (auto& s : g->slabs) {
    if (s.k) {
      (void)hipSetDevice(g->devs[s.dev_slot].id);
      (void)hsk_synchronize(s.k);
      if (s.keys) (void)hipFree(s.keys);
      if (s.bits) (void)hipFree(s.bits);
      if (s.sums) (void)hipFree(s.sums);
      hsk_destroy(s.k);
    }
  }
  for (void* p : g->ipc_opened) (void)hipIpcCloseMemHandle(p);
  for (auto& d : g->devs) {
    (void)hipSetDevice(d.id);
    if (d.kg) (void)hipFree(d.kg);
    if (d.cm) (void)hipFree(d.cm);
    for (int i = 0; i < 2; ++i) {
      if (d.ev_s0[i]) (void)hipEventDestroy(d.ev_s0[i]);
      if (d.ev_x0[i]) (void)hipEventDestroy(d.ev_x0[i]);
      if (d.ev_x1[i]) (void)hipEventDestroy(d.ev_x1[i]);
    }
    if (d.comm && R) (void)R->CommDestroy(d.comm);
    if (d.kmin) (void)hipFree(d.kmin);
    if (d.bsum) (void)hipFree(d.bsum);
    if (d.sums) (void)hipFree(d.sums);
    if (d.stream) (void)hipStreamDestroy(d.stream);
  }
  if (g->h_stage) (void)hipHostFree(g->h_stage);
  if (g->page) {
    if (g->page_is_shm) {
      (void)hipHostUnregister(g->page);
      (void)munmap(g->page, sizeof(DirectPage));
    } else {
      (void)hipHostFree(g->page);
    }
  }
  delete g;
}

static double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// Direct exchange: buffers, the shared flag page, and every device's view of every other device's buffers.  Single
// process: peer access between the group's devices.  Rank form (comm_id != null): the processes of ONE node meet in a
// POSIX shared-memory page named after comm_id, hand each other hipIpc handles of their two buffers and map them.
static int direct_setup(hsk_group* g, const void* comm_id) {
  const size_t P = (size_t)g->cfg.width * g->cfg.height;
  const bool ranks = comm_id != nullptr && g->world > 1;
  g->n_dev_total = ranks ? g->world : (int)g->devs.size();
  if (g->n_dev_total > HSK_GROUP_MAX_DEV || g->n_dev_total > HSK_PUSH_MAX_HOST) GFAIL(g, HSK_ERR_ARG, "direct exchange: too many devices");
  for (size_t di = 0; di < g->devs.size(); ++di) {
    Dev& d = g->devs[di];
    int can = 0;
    GHIP(g, hipSetDevice(d.id));
    GHIP(g, hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, d.id));
    if (!can) GFAIL(g, HSK_ERR_STATE, "direct exchange: this device has no stream wait-value operations (use the RCCL form)");
    d.gidx = ranks ? g->rank0 : (int)di;
    GHIP(g, hipMalloc((void**)&d.kg, (size_t)g->n_slabs_total * P * 4));
    GHIP(g, hipMalloc((void**)&d.cm, 6 * P * 4));
    GHIP(g, hipMemset(d.kg, 0, (size_t)g->n_slabs_total * P * 4));
    GHIP(g, hipMemset(d.cm, 0, 6 * P * 4));
    GHIP(g, hipDeviceSynchronize());  // (null-stream memsets, non-blocking streams: the buffers are handed out complete)
  }
  if (ranks) {
    char name[64];
    const unsigned char* b = (const unsigned char*)comm_id;
    int o = snprintf(name, sizeof(name), "/hskx_");
    for (int i = 0; i < 16; ++i) o += snprintf(name + o, sizeof(name) - (size_t)o, "%02x", (unsigned)(b[i] ^ b[16 + i] ^ b[32 + i] ^ b[48 + i]));
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) GFAIL(g, HSK_ERR_STATE, "direct exchange: shm_open failed");
    // from here to the point where every rank holds its mapping, a rank that gives up takes the NAME with it (the peers
    // then time out on a page nobody else will ever open, instead of /dev/shm keeping it)
    struct Unlinker {
      const char* n;
      bool armed = true;
      ~Unlinker() {
        if (armed) (void)shm_unlink(n);
      }
    } unlinker{name};
    if (ftruncate(fd, (off_t)sizeof(DirectPage)) != 0) {
      close(fd);
      GFAIL(g, HSK_ERR_STATE, "direct exchange: ftruncate failed");
    }
    void* m = mmap(nullptr, sizeof(DirectPage), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) GFAIL(g, HSK_ERR_STATE, "direct exchange: mmap failed");
    g->page = (DirectPage*)m;
    g->page_is_shm = true;
    GHIP(g, hipHostRegister(m, sizeof(DirectPage), hipHostRegisterMapped | hipHostRegisterPortable));
    Dev& own = g->devs[0];
    GHIP(g, hipHostGetDevicePointer((void**)&g->page_dev[0], m, 0));
    GHIP(g, hipDeviceSynchronize());  // the memsets above: the buffers are handed out complete
    GHIP(g, hipIpcGetMemHandle(&g->page->kg[own.gidx], own.kg));
    GHIP(g, hipIpcGetMemHandle(&g->page->cm[own.gidx], own.cm));
    __sync_synchronize();
    g->page->ready[own.gidx] = 1;
    const double t0 = now_s();
    for (int r = 0; r < g->world; ++r)
      while (!g->page->ready[r]) {
        if (now_s() - t0 > 120.0) GFAIL(g, HSK_ERR_STATE, "direct exchange: a rank did not show up within 120 s");
        usleep(200);
      }
    __sync_synchronize();
    for (int r = 0; r < g->world; ++r) {
      if (r == own.gidx) {
        own.kg_of[r] = own.kg;
        own.cm_of[r] = own.cm;
        continue;
      }
      void *pk = nullptr, *pc = nullptr;
      GHIP(g, hipIpcOpenMemHandle(&pk, g->page->kg[r], hipIpcMemLazyEnablePeerAccess));
      g->ipc_opened.push_back(pk);
      GHIP(g, hipIpcOpenMemHandle(&pc, g->page->cm[r], hipIpcMemLazyEnablePeerAccess));
      g->ipc_opened.push_back(pc);
      own.kg_of[r] = (int*)pk;
      own.cm_of[r] = (int*)pc;
    }
    __sync_synchronize();
    g->page->attached[own.gidx] = 1;
    for (int r = 0; r < g->world; ++r)
      while (!g->page->attached[r]) {
        if (now_s() - t0 > 120.0) GFAIL(g, HSK_ERR_STATE, "direct exchange: a rank did not attach within 120 s");
        usleep(200);
      }
    unlinker.armed = own.gidx == 0;  // everybody holds a mapping: the name can go (rank 0 takes it away)
    return HSK_OK;
  }
  void* m = nullptr;
  GHIP(g, hipHostMalloc(&m, sizeof(DirectPage), hipHostMallocPortable | hipHostMallocMapped));
  memset(m, 0, sizeof(DirectPage));
  g->page = (DirectPage*)m;
  for (size_t a = 0; a < g->devs.size(); ++a) {
    Dev& da = g->devs[a];
    GHIP(g, hipSetDevice(da.id));
    GHIP(g, hipHostGetDevicePointer((void**)&g->page_dev[a], m, 0));
    for (size_t bi = 0; bi < g->devs.size(); ++bi) {
      Dev& db = g->devs[bi];
      if (a != bi) {
        int can = 0;
        GHIP(g, hipDeviceCanAccessPeer(&can, da.id, db.id));
        if (!can) GFAIL(g, HSK_ERR_STATE, "direct exchange: the devices of the group cannot access each other's memory");
        const hipError_t e = hipDeviceEnablePeerAccess(db.id, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) GHIP(g, e);
        (void)hipGetLastError();
      }
      da.kg_of[db.gidx] = db.kg;
      da.cm_of[db.gidx] = db.cm;
    }
  }
  return HSK_OK;
}

// slabs [first, first + n_local) of n_total on device_ids[]; communicator of `world` ranks, this process's devices being
// ranks rank0, rank0 + 1, ... (one per distinct device); comm_id null: single process (ncclCommInitAll)
static int group_build(const hsk_config* c, int n_total, int first, int n_local, const int* device_ids, int world, int rank0,
                       const void* comm_id, int flags, hsk_group** out) {
  if (!c || !out || n_total <= 0 || n_local <= 0 || first < 0 || first + n_local > n_total || !device_ids) {
    g_group_err = "hsk_group_create: invalid argument";
    return HSK_ERR_ARG;
  }
  *out = nullptr;
  if (n_total > c->vol_z) {
    g_group_err = "hsk_group_create: more slabs than planes";
    return HSK_ERR_ARG;
  }
  hsk_group* g = new hsk_group();
  g->cfg = *c;
  g->flags = flags;
  g->n_slabs_total = n_total;
  g->world = world;
  g->rank0 = rank0;
  auto bail = [&](int code) {
    g_group_err = g->err;
    group_free(g);
    return code;
  };
  const size_t P = (size_t)c->width * c->height;
  const int halo = slab_halo(c);
  for (int i = 0; i < n_local; ++i) {
    const int dev = device_ids[i];
    int slot = -1;
    for (size_t d = 0; d < g->devs.size(); ++d)
      if (g->devs[d].id == dev) slot = (int)d;
    if (slot < 0) {
      Dev d;
      d.id = dev;
      g->devs.push_back(d);
      slot = (int)g->devs.size() - 1;
    }
    Slab s;
    s.dev_slot = slot;
    s.index = first + i;
    g->slabs.push_back(s);
    g->devs[slot].slabs.push_back((int)g->slabs.size() - 1);
  }
  for (auto& d : g->devs) {
    if (hipSetDevice(d.id) != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&d.kmin, P * 4) != hipSuccess ||
        hipMalloc((void**)&d.bsum, P * 24) != hipSuccess || hipMalloc((void**)&d.sums, 27 * sizeof(double)) != hipSuccess) {
      g->err = "hsk_group_create: device setup failed (device id out of range, or out of memory)";
      return bail(HSK_ERR_HIP);
    }
    for (int i = 0; i < 2 && (flags & HSK_GROUP_PROFILE); ++i)
      if (hipEventCreate(&d.ev_s0[i]) != hipSuccess || hipEventCreate(&d.ev_x0[i]) != hipSuccess || hipEventCreate(&d.ev_x1[i]) != hipSuccess) {
        g->err = "hsk_group_create: profiling events";
        return bail(HSK_ERR_HIP);
      }
  }
  for (auto& s : g->slabs) {
    hsk_config sc = *c;
    sc.device_id = g->devs[s.dev_slot].id;
    slab_range(s.index, n_total, c->vol_z, &sc.own_z0, &sc.own_z1);
    sc.halo = n_total > 1 ? halo : 0;
    sc.use_graph = 0;
    int r = hsk_create(&sc, &s.k);
    if (r != HSK_OK) {
      g->err = std::string("hsk_group_create: slab ") + std::to_string(s.index) + ": " + hsk_last_error(nullptr);
      return bail(r);
    }
    (void)hipSetDevice(sc.device_id);
    if (hsk_set_stream(s.k, g->devs[s.dev_slot].stream) != HSK_OK || hipMalloc((void**)&s.keys, P * 4) != hipSuccess ||
        hipMalloc((void**)&s.bits, P * 24) != hipSuccess || hipMalloc((void**)&s.sums, 27 * sizeof(double)) != hipSuccess) {
      g->err = "hsk_group_create: slab buffers";
      return bail(HSK_ERR_HIP);
    }
  }
  if (hipHostMalloc((void**)&g->h_stage, P * 2 * (HSK_MAX_IN_FLIGHT + 1), hipHostMallocDefault) != hipSuccess) {
    g->err = "hsk_group_create: pinned staging";
    return bail(HSK_ERR_HIP);
  }
  g->direct = (flags & HSK_GROUP_DIRECT) != 0;
  if (g->direct) {
    if (flags & HSK_GROUP_ICP_ALLREDUCE) {
      g->err = "hsk_group_create: HSK_GROUP_DIRECT carries the composites only; the all-reduced ICP needs the RCCL form";
      return bail(HSK_ERR_ARG);
    }
    const int r = direct_setup(g, comm_id);
    if (r != HSK_OK) return bail(r);
  }
  g->use_rccl = !g->direct && (world > 1 || (flags & HSK_GROUP_FORCE_RCCL));
  if (g->use_rccl) {
    std::string why;
    Rccl* R = rccl(&why);
    if (!R) {
      g->err = why;
      return bail(HSK_ERR_STATE);
    }
    ncclResult_t r = ncclSuccess;
    if (comm_id) {
      if (g->devs.size() != 1) {
        g->err = "hsk_group_create_rank: one device per process";
        return bail(HSK_ERR_ARG);
      }
      ncclUniqueId id;
      memcpy(&id, comm_id, sizeof(id));
      (void)hipSetDevice(g->devs[0].id);
      r = R->CommInitRank(&g->devs[0].comm, world, id, rank0);
    } else {
      std::vector<int> ids;
      std::vector<ncclComm_t> comms(g->devs.size());
      for (auto& d : g->devs) ids.push_back(d.id);
      r = R->CommInitAll(comms.data(), (int)ids.size(), ids.data());
      if (r == ncclSuccess)
        for (size_t d = 0; d < g->devs.size(); ++d) g->devs[d].comm = comms[d];
    }
    if (r != ncclSuccess) {
      g->err = std::string("RCCL communicator: ") + R->GetErrorString(r);
      return bail(HSK_ERR_HIP);
    }
  }
  *out = g;
  return HSK_OK;
}

extern "C" int hsk_group_create(const hsk_config* c, int n_slabs, const int* device_ids, int flags, hsk_group** out) {
  if (!device_ids || n_slabs <= 0) {
    g_group_err = "hsk_group_create: invalid argument";
    return HSK_ERR_ARG;
  }
  int distinct = 0;
  for (int i = 0; i < n_slabs; ++i) {
    bool seen = false;
    for (int j = 0; j < i; ++j) seen = seen || device_ids[j] == device_ids[i];
    distinct += seen ? 0 : 1;
  }
  return group_build(c, n_slabs, 0, n_slabs, device_ids, distinct, 0, nullptr, flags, out);
}

extern "C" int hsk_group_unique_id(void* id128) {
  if (!id128) return HSK_ERR_ARG;
  std::string why;
  Rccl* R = rccl(&why);
  if (R) {
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) {
      g_group_err = "ncclGetUniqueId failed";
      return HSK_ERR_HIP;
    }
    memcpy(id128, &id, sizeof(id));
    return HSK_OK;
  }
  // no RCCL in this process: 128 random bytes name a direct-exchange group (HSK_GROUP_DIRECT) just as well
  unsigned char* out = (unsigned char*)id128;
  FILE* f = fopen("/dev/urandom", "rb");
  const size_t got = f ? fread(out, 1, 128, f) : 0;
  if (f) fclose(f);
  if (got != 128) {
    g_group_err = why + "; and /dev/urandom is not readable";
    return HSK_ERR_STATE;
  }
  return HSK_OK;
}

extern "C" int hsk_group_create_rank(const hsk_config* c, int rank, int world, const void* comm_id, int flags, hsk_group** out) {
  if (!c || rank < 0 || rank >= world || (world > 1 && !comm_id)) {
    g_group_err = "hsk_group_create_rank: invalid argument";
    return HSK_ERR_ARG;
  }
  const int dev = c->device_id;
  return group_build(c, world, rank, 1, &dev, world, rank, comm_id, flags, out);
}

extern "C" void hsk_group_destroy(hsk_group* g) { group_free(g); }
extern "C" const char* hsk_group_last_error(const hsk_group* g) { return g ? g->err.c_str() : g_group_err.c_str(); }
extern "C" int hsk_group_n_slabs(const hsk_group* g) { return g ? (int)g->slabs.size() : -1; }
// how many ranks / devices the exchange of this group really spans: the RCCL communicator's own count, the ranks
// attached to the direct form's shared page, or the distinct devices of a single-process group
extern "C" int hsk_group_ranks_seen(hsk_group* g, int* n) {
  if (!g || !n) return HSK_ERR_ARG;
  *n = (int)g->devs.size();
  if (g->use_rccl && !g->devs.empty() && g->devs[0].comm) {
    Rccl* R = rccl(nullptr);
    int c = 0;
    if (!R) GFAIL(g, HSK_ERR_STATE, "RCCL is not loaded");
    GNCCL(g, R, R->CommCount(g->devs[0].comm, &c));
    *n = c;
  } else if (g->direct && g->page_is_shm && g->page) {
    int c = 0;
    for (int r = 0; r < HSK_GROUP_MAX_DEV; ++r) c += g->page->attached[r] ? 1 : 0;
    *n = c;
  }
  return HSK_OK;
}
extern "C" hsk_ctx* hsk_group_slab(hsk_group* g, int i) { return (g && i >= 0 && i < (int)g->slabs.size()) ? g->slabs[i].k : nullptr; }

// all-reduce `buf` (per device) in place over the communicator; one call per device inside an RCCL group
static int group_allreduce(hsk_group* g, bool keys, ncclDataType_t type, ncclRedOp_t op, size_t count, int which) {
  (void)keys;
  if (!g->use_rccl) return HSK_OK;
  Rccl* R = rccl(nullptr);
  GNCCL(g, R, R->GroupStart());
  for (auto& d : g->devs) {
    void* buf = which == 0 ? (void*)d.kmin : (which == 1 ? (void*)d.bsum : (void*)d.sums);
    ncclResult_t r = R->AllReduce(buf, buf, count, type, op, d.comm, d.stream);
    if (r != ncclSuccess) {
      (void)R->GroupEnd();
      g->err = std::string("ncclAllReduce failed: ") + R->GetErrorString(r);
      return HSK_ERR_HIP;
    }
  }
  GNCCL(g, R, R->GroupEnd());
  return HSK_OK;
}

// the 19 ICP iterations with the 27 sums all-reduced every iteration (the north_star's form): slab i takes image rows
// [i H / n, (i + 1) H / n) of the level, devices add their slabs' shares, RCCL adds the devices'
static int group_icp_allreduce(hsk_group* g) {
  for (int level = HSK_LEVELS - 1; level >= 0; --level) {
    const int h = g->cfg.height >> level;
    for (int it = 0; it < g->cfg.icp_iters[level]; ++it) {
      for (auto& d : g->devs) {
        GHIP(g, hipSetDevice(d.id));
        bool first = true;
        for (int si : d.slabs) {
          Slab& s = g->slabs[si];
          const int n = g->n_slabs_total;
          const int r0 = (int)((long)s.index * h / n), r1 = (int)((long)(s.index + 1) * h / n);
          GSLAB(g, s, hsk_mgpu_icp_accumulate(s.k, level, r0, r1, first ? (void*)d.sums : (void*)s.sums));
          if (!first) hipLaunchKernelGGL(k_add_into_f64, dim3(1), dim3(32), 0, d.stream, d.sums, s.sums, 27);
          first = false;
        }
      }
      int r = group_allreduce(g, false, ncclFloat64, ncclSum, 27, 2);
      if (r != HSK_OK) return r;
      for (auto& d : g->devs) {
        GHIP(g, hipSetDevice(d.id));
        for (int si : d.slabs) GSLAB(g, g->slabs[si], hsk_mgpu_icp_update(g->slabs[si].k, d.sums));
      }
    }
  }
  return HSK_OK;
}

// HSK_GROUP_PROFILE: the times held by event set `set` of the first local device into the sums
static void group_read_profile(hsk_group* g, int set) {
  if (!g->exch_pending[set] || g->devs.empty()) return;
  Dev& d = g->devs[0];
  float a = 0.0f, b = 0.0f;
  if (hipEventSynchronize(d.ev_x1[set]) == hipSuccess && hipEventElapsedTime(&a, d.ev_s0[set], d.ev_x0[set]) == hipSuccess &&
      hipEventElapsedTime(&b, d.ev_x0[set], d.ev_x1[set]) == hipSuccess) {
    g->front_ms += a;
    g->exch_ms += b;
    g->exch_frames += 1;
  }
  g->exch_pending[set] = false;
}

// the oldest frame the slabs still hold -> e (every slab must report the same pose and verdict)
static int group_collect(hsk_group* g, hsk_group::Pending& e) {
  float pose[16], first[16];
  int tr = 0, tr0 = 0;
  for (size_t i = 0; i < g->slabs.size(); ++i) {
    Slab& s = g->slabs[i];
    GHIP(g, hipSetDevice(g->devs[s.dev_slot].id));
    GSLAB(g, s, hsk_wait_frame(s.k, pose, &tr));
    if (i == 0) {
      memcpy(first, pose, sizeof(pose));
      tr0 = tr;
    } else if (tr != tr0 || memcmp(first, pose, sizeof(pose)) != 0) {
      GFAIL(g, HSK_ERR_STATE, "slabs disagree on the pose: the composite is not the same on every slab");
    }
  }
  memcpy(e.pose, first, sizeof(first));
  e.tracked = tr0;
  e.ready = true;
  return HSK_OK;
}

// enqueue one whole frame on every device: depth upload, frame front of every slab, the two composites; frame end
// queued (async) or, for the first frame of a scan, taken synchronously.  depth_dev: per device, or null = from host.
// *started: set once slab state may have changed (a failure after that leaves the slabs out of step: the caller poisons)
static int group_enqueue_body(hsk_group* g, const uint16_t* depth_host, const void* const* depth_dev, int w, int h, bool* started) {
  const size_t P = (size_t)w * h;
  const int P_i = (int)P;
  const uint16_t* src = nullptr;
  if (depth_host) {
    uint16_t* stage = g->h_stage + (size_t)(g->stage_turn % (HSK_MAX_IN_FLIGHT + 1)) * P;
    g->stage_turn += 1;
    memcpy(stage, depth_host, P * 2);  // the caller's buffer may be freed on return (HoniHelper's Vector is pinned only inside unsafeWith)
    src = stage;
  }
  const bool restart = hsk_mgpu_restart_pending(g->slabs[0].k) == 1;
  const bool icp_ar = (g->flags & HSK_GROUP_ICP_ALLREDUCE) != 0;
  *started = true;
  if (restart) {
    // A pipelined frame lost tracking and the caller has seen it (hsk_group_wait_frame reported tracked = 0).  Frames
    // submitted behind it were dropped on the device; their results are collected now and parked in the FIFO, the
    // synchronous result of the restart goes in behind them -- as hsk_submit_frame does for a single context.
    for (auto& e : g->fifo)
      if (!e.ready) {
        const int r = group_collect(g, e);
        if (r != HSK_OK) return r;
      }
  }
  const bool overlap = g->in_flight() > 0 && !restart;
  for (size_t di = 0; di < g->devs.size(); ++di) {
    Dev& d = g->devs[di];
    GHIP(g, hipSetDevice(d.id));
    if (d.ev_s0[0] && di == 0 && !restart) {
      g->ev_set ^= 1;
      group_read_profile(g, g->ev_set);  // the times of the frame before last: read before its events are reused
      GHIP(g, hipEventRecord(d.ev_s0[g->ev_set], d.stream));
    }
    // the frame as every slab of this device reads it: the caller's device buffer, or the pinned staging slot itself
    // (device-visible).  With a frame still in flight the copy + preprocessing go to each slab's second stream
    // (hsk_mgpu_prefetch), beside the work the device's stream is busy with; the frame front then picks them up.
    const void* frame = src ? (const void*)src : depth_dev[di];
    for (int si : d.slabs) {
      Slab& s = g->slabs[si];
      if (overlap) GSLAB(g, s, hsk_mgpu_prefetch(s.k, frame, w, h));
      if (icp_ar && !restart)
        GSLAB(g, s, hsk_mgpu_frame_begin(s.k, frame, w, h));
      else
        GSLAB(g, s, hsk_mgpu_frame_front(s.k, frame, w, h, s.keys));
    }
  }
  if (restart) {
    // first frame of a (re)started scan: integrate + transformed maps only, finished synchronously
    hsk_group::Pending e;
    for (auto& s : g->slabs) {
      GHIP(g, hipSetDevice(g->devs[s.dev_slot].id));
      GSLAB(g, s, hsk_mgpu_frame_end(s.k, nullptr, nullptr, e.pose, &e.tracked));
    }
    e.ready = true;
    g->fifo.push_back(e);
    return HSK_OK;
  }
  if (icp_ar) {
    int r = group_icp_allreduce(g);
    if (r != HSK_OK) return r;
    for (auto& d : g->devs) {
      GHIP(g, hipSetDevice(d.id));
      for (int si : d.slabs) {
        Slab& s = g->slabs[si];
        GSLAB(g, s, hsk_mgpu_integrate(s.k));
        GSLAB(g, s, hsk_mgpu_raycast_local(s.k, s.keys));
      }
    }
  }
  if (g->direct) {
    // One-hop exchange (SURVEY.md 8(e) "xGMI fit").  Step 1: every slab's keys into its slot of EVERY device's gather
    // buffer, then the flag "device d's keys of exchange seq are in place" raised at every other device.
    g->seq += 1;
    const unsigned seq = g->seq;
    const int nd = g->n_dev_total;
    for (size_t di = 0; di < g->devs.size(); ++di) {
      Dev& d = g->devs[di];
      GHIP(g, hipSetDevice(d.id));
      if (d.ev_x0[0] && di == 0) GHIP(g, hipEventRecord(d.ev_x0[g->ev_set], d.stream));
      for (int si : d.slabs) {
        Slab& s = g->slabs[si];
        KeyDests kd;
        kd.n = nd;
        for (int r = 0; r < HSK_GROUP_MAX_DEV; ++r) kd.p[r] = r < nd ? d.kg_of[r] + (size_t)s.index * P : nullptr;
        hipLaunchKernelGGL(k_push_keys, dim3((P_i + 255) / 256), dim3(256), 0, d.stream, s.keys, kd, P_i);
      }
      for (int r = 0; r < nd; ++r)
        if (r != d.gidx) GHIP(g, hipStreamWriteValue32(d.stream, &g->page_dev[di]->f1[r][d.gidx], seq, 0));
    }
    // Step 2: once every other device's keys have arrived: the MIN over all slots, locally; the winner of a pixel stores
    // its vertex / normal bits into every device's composite; flag "device d's winners are in place".
    for (size_t di = 0; di < g->devs.size(); ++di) {
      Dev& d = g->devs[di];
      GHIP(g, hipSetDevice(d.id));
      for (int r = 0; r < nd; ++r)
        if (r != d.gidx) GHIP(g, hipStreamWaitValue32(d.stream, &g->page_dev[di]->f1[d.gidx][r], seq, hipStreamWaitValueGte, 0xffffffffu));
      hipLaunchKernelGGL(k_min_slots, dim3((P_i + 255) / 256), dim3(256), 0, d.stream, d.kg, g->n_slabs_total, P_i, d.kmin);
      for (int si : d.slabs) GSLAB(g, g->slabs[si], hsk_mgpu_raycast_push(g->slabs[si].k, d.kmin, (void* const*)d.cm_of, nd));
      for (int r = 0; r < nd; ++r)
        if (r != d.gidx) GHIP(g, hipStreamWriteValue32(d.stream, &g->page_dev[di]->f2[r][d.gidx], seq, 0));
    }
    // Step 3: once every other device's winners have arrived: adopt the composite, rebuild the pyramid, report.
    for (size_t di = 0; di < g->devs.size(); ++di) {
      Dev& d = g->devs[di];
      GHIP(g, hipSetDevice(d.id));
      for (int r = 0; r < nd; ++r)
        if (r != d.gidx) GHIP(g, hipStreamWaitValue32(d.stream, &g->page_dev[di]->f2[d.gidx][r], seq, hipStreamWaitValueGte, 0xffffffffu));
      if (d.ev_x1[0] && di == 0) {
        GHIP(g, hipEventRecord(d.ev_x1[g->ev_set], d.stream));
        g->exch_pending[g->ev_set] = true;
      }
      for (int si : d.slabs) GSLAB(g, g->slabs[si], hsk_mgpu_frame_end_async(g->slabs[si].k, d.kmin, d.cm));
    }
    g->fifo.push_back(hsk_group::Pending());
    return HSK_OK;
  }
  // composite 1: the first event along every ray
  if (g->devs[0].ev_x0[0]) {
    GHIP(g, hipSetDevice(g->devs[0].id));
    GHIP(g, hipEventRecord(g->devs[0].ev_x0[g->ev_set], g->devs[0].stream));
  }
  for (auto& d : g->devs) {
    GHIP(g, hipSetDevice(d.id));
    bool first = true;
    for (int si : d.slabs) {
      Slab& s = g->slabs[si];
      if (first)
        GHIP(g, hipMemcpyAsync(d.kmin, s.keys, P * 4, hipMemcpyDeviceToDevice, d.stream));
      else
        hipLaunchKernelGGL(k_min_into, dim3((P_i + 255) / 256), dim3(256), 0, d.stream, d.kmin, s.keys, P_i);
      first = false;
    }
  }
  int r = group_allreduce(g, true, ncclInt32, ncclMin, P, 0);
  if (r != HSK_OK) return r;
  // composite 2: the winner's vertex / normal bit patterns
  for (auto& d : g->devs) {
    GHIP(g, hipSetDevice(d.id));
    bool first = true;
    for (int si : d.slabs) {
      Slab& s = g->slabs[si];
      GSLAB(g, s, hsk_mgpu_raycast_resolve(s.k, d.kmin, first ? (void*)d.bsum : (void*)s.bits));
      if (!first) hipLaunchKernelGGL(k_add_into, dim3((6 * P_i + 255) / 256), dim3(256), 0, d.stream, d.bsum, s.bits, 6 * P_i);
      first = false;
    }
  }
  r = group_allreduce(g, false, ncclInt32, ncclSum, 6 * P, 1);
  if (r != HSK_OK) return r;
  if (g->devs[0].ev_x1[0]) {
    GHIP(g, hipSetDevice(g->devs[0].id));
    GHIP(g, hipEventRecord(g->devs[0].ev_x1[g->ev_set], g->devs[0].stream));
    g->exch_pending[g->ev_set] = true;
  }
  for (auto& d : g->devs) {
    GHIP(g, hipSetDevice(d.id));
    for (int si : d.slabs) GSLAB(g, g->slabs[si], hsk_mgpu_frame_end_async(g->slabs[si].k, d.kmin, d.bsum));
  }
  g->fifo.push_back(hsk_group::Pending());  // state is touched only now, after every slab has taken the frame
  return HSK_OK;
}

// Poisoning also lets this process's own queues DRAIN, so that hsk_group_reset / hsk_group_destroy (stream
// synchronisation, hipFree) return instead of hanging behind work that waits for a peer which will never answer:
//  * direct exchange: every flag this process's streams wait for (its own rows of F1 / F2) is raised to the last
//    enqueued exchange by a host store into the shared page -- the frames in flight finish on stale data, their results
//    are never handed out (every call fails from here on);
//  * RCCL between ranks: the communicator is aborted (ncclCommAbort: the collective kernels poll its abort flag).
static void group_poison(hsk_group* g) {
  g->poisoned = true;
  g->poison_why = g->err;
  if (g->direct && g->page) {
    for (auto& d : g->devs)
      for (int r = 0; r < g->n_dev_total; ++r) {
        __atomic_store_n(&g->page->f1[d.gidx][r], g->seq, __ATOMIC_RELEASE);
        __atomic_store_n(&g->page->f2[d.gidx][r], g->seq, __ATOMIC_RELEASE);
      }
    __sync_synchronize();
  }
  if (g->use_rccl && g->world > 1) {
    Rccl* R = rccl(nullptr);
    for (auto& d : g->devs)
      if (d.comm && R) {
        (void)hipSetDevice(d.id);
        (void)R->CommAbort(d.comm);
        d.comm = nullptr;
      }
  }
}

static int group_enqueue(hsk_group* g, const uint16_t* depth_host, const void* const* depth_dev, int w, int h) {
  GPOISONED(g);
  if (w != g->cfg.width || h != g->cfg.height) GFAIL(g, HSK_ERR_ARG, "depth frame size does not match the group");
  // (the slabs' rings hold the frames not yet collected: ready entries parked in the FIFO do not count against them)
  int uncollected = 0;
  for (auto& e : g->fifo) uncollected += e.ready ? 0 : 1;
  if (uncollected >= HSK_MAX_IN_FLIGHT || g->in_flight() >= 2 * HSK_MAX_IN_FLIGHT)
    GFAIL(g, HSK_ERR_STATE, "too many frames in flight: call hsk_group_wait_frame first");
  bool started = false;
  const int r = group_enqueue_body(g, depth_host, depth_dev, w, h, &started);
  if (r != HSK_OK && started) group_poison(g);
  return r;
}

extern "C" int hsk_group_submit_frame(hsk_group* g, const uint16_t* depth, int w, int h) {
  if (!g || !depth) return HSK_ERR_ARG;
  return group_enqueue(g, depth, nullptr, w, h);
}
extern "C" int hsk_group_submit_frame_dev(hsk_group* g, const void* const* depth_dev, int w, int h) {
  if (!g || !depth_dev) return HSK_ERR_ARG;
  for (size_t d = 0; d < g->devs.size(); ++d)
    if (!depth_dev[d]) GFAIL(g, HSK_ERR_ARG, "a device pointer is null (one per distinct device of the group, in creation order)");
  return group_enqueue(g, nullptr, depth_dev, w, h);
}

extern "C" int hsk_group_wait_frame(hsk_group* g, float pose_out[16], int* tracked) {
  if (!g) return HSK_ERR_ARG;
  GPOISONED(g);
  if (g->fifo.empty()) GFAIL(g, HSK_ERR_STATE, "no frame in flight");
  hsk_group::Pending& e = g->fifo.front();
  if (!e.ready) {
    const int r = group_collect(g, e);
    if (r != HSK_OK) {  // some slabs have handed the frame over, others not
      group_poison(g);
      return r;
    }
  }
  if (pose_out) memcpy(pose_out, e.pose, sizeof(e.pose));
  if (tracked) *tracked = e.tracked;
  g->fifo.pop_front();  // only now: every slab has reported
  return HSK_OK;
}

extern "C" int hsk_group_process_frame(hsk_group* g, const uint16_t* depth, int w, int h, float pose_out[16], int* tracked) {
  if (!g || !depth) return HSK_ERR_ARG;
  GPOISONED(g);
  if (g->in_flight() > 0) GFAIL(g, HSK_ERR_STATE, "frames are in flight: collect them with hsk_group_wait_frame first");
  int r = group_enqueue(g, depth, nullptr, w, h);
  if (r != HSK_OK) return r;
  return hsk_group_wait_frame(g, pose_out, tracked);
}

// Drops the frames in flight and starts the scan afresh.  Also the way out of a poisoned single-process group (every
// slab context is reset, whatever frame it was on); a poisoned multi-rank group (RCCL or direct form) stays poisoned --
// its peers may sit in a collective or wait for a flag, and their sequence numbers no longer agree with this rank's --
// and can only be destroyed.
extern "C" int hsk_group_reset(hsk_group* g) {
  if (!g) return HSK_ERR_ARG;
  if (g->poisoned && g->world > 1 && (g->use_rccl || g->direct)) GPOISONED(g);
  for (auto& s : g->slabs) {
    GHIP(g, hipSetDevice(g->devs[s.dev_slot].id));
    GSLAB(g, s, hsk_reset(s.k));  // (collects and drops the slab's frames in flight itself)
  }
  g->fifo.clear();
  g->poisoned = false;
  g->poison_why.clear();
  return HSK_OK;
}

// HSK_GROUP_PROFILE: on the group's first local device, per tracked frame: the slab's own work (ICP + integrate + slab
// raycast: frame start -> exchange start) and the exchange (steps 1..3, waits for the peers included), summed over the
// frames whose times have been read so far (the last submitted frame's are read at the next submission)
extern "C" int hsk_group_exchange_ms(hsk_group* g, double* sum_ms, double* front_sum_ms, unsigned long long* n_frames) {
  if (!g) return HSK_ERR_ARG;
  if (g->fifo.empty() && !g->poisoned && !g->devs.empty() && hipSetDevice(g->devs[0].id) == hipSuccess) {
    group_read_profile(g, 0);  // every frame has been collected: both sets are complete
    group_read_profile(g, 1);
  }
  if (sum_ms) *sum_ms = g->exch_ms;
  if (front_sum_ms) *front_sum_ms = g->front_ms;
  if (n_frames) *n_frames = g->exch_frames;
  return HSK_OK;
}

// the planes this process owns, placed at their z in a full-volume array (planes of other processes are left untouched)
extern "C" int hsk_group_download_tsdf(hsk_group* g, int16_t* full) {
  if (!g || !full) return HSK_ERR_ARG;
  const size_t plane = (size_t)g->cfg.vol_x * g->cfg.vol_y * 2;
  std::vector<int16_t> tmp;
  for (auto& s : g->slabs) {
    GHIP(g, hipSetDevice(g->devs[s.dev_slot].id));
    int z0 = 0, nz = 0, o0 = 0, o1 = 0;
    GSLAB(g, s, hsk_stored_planes(s.k, &z0, &nz));
    slab_range(s.index, g->n_slabs_total, g->cfg.vol_z, &o0, &o1);
    tmp.resize((size_t)nz * plane);
    GSLAB(g, s, hsk_download_tsdf(s.k, tmp.data()));
    memcpy(full + (size_t)o0 * plane, tmp.data() + (size_t)(o0 - z0) * plane, (size_t)(o1 - o0) * plane * sizeof(int16_t));
  }
  return HSK_OK;
}
