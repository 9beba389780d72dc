/* rooms_native.c -- BASELINE configs[4] on ONE GPU, without Python in the process: M rooms, a POSIX thread and an
 * hsk_ctx each, every room fed its own stretch of the synthetic stream through the boundary's own calls
 * (hsk_submit_frame / hsk_wait_frame, host pointers, one frame in flight ahead).  VERDICT r05 item 4: bench.py's
 * concurrent_rooms block drives the rooms from Python threads (GIL hand-offs, 20 frames) and nobody knew whether the GPU
 * or the host's launch path limits four rooms.  This is the same workload from C threads for >= 200 frames each: aggregate
 * frames/s, per-room frames/s, host time inside submit / wait per room; under `rocprofv3 --kernel-trace` its kernel
 * timeline says how far the rooms' kernels overlap (tools/rooms_overlap.py).
 *
 *   gcc -O2 -std=c11 -pthread -Iinclude tools/rooms_native.c -Lhousescan_amd -lhskinfu -Wl,-rpath,$PWD/housescan_amd \
 *       -Wl,-rpath-link,/opt/rocm/lib -o rooms_native
 *   ./rooms_native ROOMS [N=512] [FRAMES=240] [room|open] [ahead=1] [dev|host] [use_graph=0]
 * "room": every room is one of the closed rooms of synth.cpp scanned from inside (hsk_synth_room_*), "open": SURVEY.md
 * 8(d)'s stream, room r starting 25 frames after room r - 1.   dev: the frames are uploaded once and fed by device pointer
 * (hsk_submit_frame_dev; needs the HIP runtime for hipMalloc / hipMemcpy, resolved with dlsym so that this file stays C). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "hskinfu.h"

typedef struct {
  int room, n, frames, inside, ahead, dev;
  hsk_ctx* k;
  uint16_t* depth;  /* frames x w x h, host */
  void* depth_dev;  /* the same on the device (dev mode) */
  int w, h, lost, rc, v0;
  hsk_config cfg;
  int (*upload)(void*, const void*, size_t, int);
  double t_submit, t_wait, t_total;
  pthread_barrier_t* gate;
} room_t;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int submit(room_t* r, int f) {
  const size_t px = (size_t)r->w * r->h;
  if (r->dev) return hsk_submit_frame_dev(r->k, (const char*)r->depth_dev + (size_t)f * px * 2, r->w, r->h);
  return hsk_submit_frame(r->k, r->depth + (size_t)f * px, r->w, r->h);
}

static void* run_room(void* arg) {
  room_t* r = (room_t*)arg;
  float pose[16];
  int tracked = 0;
  const size_t px = (size_t)r->w * r->h;
  /* the room's frames, rendered by its own thread (the rooms render side by side: seconds instead of half a minute) */
  for (int f = 0; f < r->frames; ++f) {
    float gt[16];
    if (r->inside) {
      hsk_synth_room_pose((r->room + r->v0) & 3, f, 720, gt);
      hsk_synth_room_render((r->room + r->v0) & 3, gt, r->w, r->h, r->cfg.fx, r->cfg.fy, r->cfg.cx, r->cfg.cy, r->depth + (size_t)f * px);
    } else {
      hsk_synth_pose(25 * (r->room + r->v0) + f, gt);
      hsk_synth_render(gt, r->w, r->h, r->cfg.fx, r->cfg.fy, r->cfg.cx, r->cfg.cy, r->depth + (size_t)f * px);
    }
  }
  if (r->dev && r->upload && r->upload(r->depth_dev, r->depth, (size_t)r->frames * px * 2, 1 /* hipMemcpyHostToDevice */) != 0) r->rc = HSK_ERR_STATE;
  /* frame 0 and five warm-up frames, synchronously */
  for (int f = 0; f < 6 && r->rc == HSK_OK; ++f) r->rc = hsk_process_frame(r->k, r->depth + (size_t)f * px, r->w, r->h, pose, &tracked);
  hsk_synchronize(r->k);
  if (getenv("ROOMS_VERBOSE")) fprintf(stderr, "room %d: warm-up done (rc %d)\n", r->room, r->rc);
  hsk_submit_host_us(r->k, NULL, NULL, 1);
  pthread_barrier_wait(r->gate);
  const double t0 = now_s();
  int sub = 6, got = 6;
  while (got < r->frames && r->rc == HSK_OK) {
    if (getenv("ROOMS_VERBOSE") && (got % 8) == 0) fprintf(stderr, "room %d: at frame %d\n", r->room, got);
    while (sub < r->frames && sub - got <= r->ahead && r->rc == HSK_OK) {
      const double a = now_s();
      r->rc = submit(r, sub++);
      r->t_submit += now_s() - a;
    }
    if (r->rc != HSK_OK) break;
    const double a = now_s();
    r->rc = hsk_wait_frame(r->k, pose, &tracked);
    r->t_wait += now_s() - a;
    r->lost += !tracked;
    ++got;
  }
  if (getenv("ROOMS_VERBOSE")) fprintf(stderr, "room %d: %d frames collected (rc %d)\n", r->room, got, r->rc);
  hsk_synchronize(r->k); /* the last wait returns with the pose: that frame's integrate and raycast are part of the work */
  if (getenv("ROOMS_VERBOSE")) fprintf(stderr, "room %d: synchronized\n", r->room);
  r->t_total = now_s() - t0;
  return NULL;
}

int main(int argc, char** argv) {
  const int rooms = argc > 1 ? atoi(argv[1]) : 4, n = argc > 2 ? atoi(argv[2]) : 512, frames = argc > 3 ? atoi(argv[3]) : 240;
  const int inside = argc > 4 ? strcmp(argv[4], "open") != 0 : 1, ahead = argc > 5 ? atoi(argv[5]) : 1;
  const int dev = argc > 6 && strcmp(argv[6], "dev") == 0, use_graph = argc > 7 ? atoi(argv[7]) : 0;
  if (rooms < 1 || rooms > 16 || frames < 8) return 2;
  int (*p_malloc)(void**, size_t) = NULL;
  int (*p_memcpy)(void*, const void*, size_t, int) = NULL;
  if (dev) {
    void* hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (hip) {
      p_malloc = (int (*)(void**, size_t))dlsym(hip, "hipMalloc");
      p_memcpy = (int (*)(void*, const void*, size_t, int))dlsym(hip, "hipMemcpy");
    }
    if (!p_malloc || !p_memcpy) {
      fprintf(stderr, "dev mode: the HIP runtime could not be resolved\n");
      return 1;
    }
  }
  const int v0 = getenv("ROOMS_FIRST_VARIANT") ? atoi(getenv("ROOMS_FIRST_VARIANT")) : 0;
  room_t R[16];
  pthread_barrier_t gate;
  pthread_barrier_init(&gate, NULL, (unsigned)rooms + 1);
  memset(R, 0, sizeof(R));
  for (int r = 0; r < rooms; ++r) {
    hsk_config cfg;
    hsk_default_config(&cfg, n);
    float gt[16];
    if (inside)
      hsk_synth_room_pose((r + v0) & 3, 0, 720, gt);
    else
      hsk_synth_pose(25 * (r + v0), gt);
    memcpy(cfg.init_pose, gt, sizeof(gt));
    cfg.use_graph = use_graph;
    R[r].room = r; R[r].n = n; R[r].frames = frames; R[r].inside = inside; R[r].ahead = ahead; R[r].dev = dev;
    R[r].w = cfg.width; R[r].h = cfg.height; R[r].gate = &gate;
    const size_t px = (size_t)cfg.width * cfg.height;
    R[r].depth = (uint16_t*)malloc((size_t)frames * px * 2);
    R[r].cfg = cfg;
    R[r].v0 = v0;
    if (hsk_create(&cfg, &R[r].k) != HSK_OK) {
      fprintf(stderr, "hsk_create: %s\n", hsk_last_error(NULL));
      return 1;
    }
    if (dev) {
      if (p_malloc(&R[r].depth_dev, (size_t)frames * px * 2) != 0) {
        fprintf(stderr, "dev mode: allocation failed\n");
        return 1;
      }
      R[r].upload = p_memcpy;
    }
  }
  if (getenv("ROOMS_START_FILE")) { /* several PROCESSES (tools/rooms_procs.sh) start their timed regions together */
    struct timespec nap = {0, 2000000};
    FILE* f;
    while ((f = fopen(getenv("ROOMS_START_FILE"), "r")) == NULL) nanosleep(&nap, NULL);
    fclose(f);
  }
  pthread_t th[16];
  for (int r = 0; r < rooms; ++r) pthread_create(&th[r], NULL, run_room, &R[r]);
  pthread_barrier_wait(&gate);
  const double t0 = now_s();
  for (int r = 0; r < rooms; ++r) pthread_join(th[r], NULL);
  const double dt = now_s() - t0;
  int lost = 0, bad = 0;
  for (int r = 0; r < rooms; ++r) {
    lost += R[r].lost;
    bad += R[r].rc != HSK_OK;
    if (R[r].rc != HSK_OK) fprintf(stderr, "room %d: %s\n", r, hsk_last_error(R[r].k));
  }
  const int timed = frames - 6;
  printf("{\"rooms\": %d, \"volume\": %d, \"stream\": \"%s\", \"frames_per_room\": %d, \"ahead\": %d, \"frames\": \"%s\", \"use_graph\": %d, \"frames_per_s_in_all\": %.1f, \"per_room\": %.1f, "
         "\"lost_frames\": %d, \"host_us_per_frame_in_submit\": %.1f, \"host_us_per_frame_in_wait\": %.1f, \"build\": \"%s\"}\n",
         rooms, n, inside ? "room scan (camera inside)" : "open scene of 8(d)", timed, ahead, dev ? "device pointers" : "host pointers", use_graph, rooms * timed / dt, timed / dt, lost,
         1e6 * R[0].t_submit / timed, 1e6 * R[0].t_wait / timed, hsk_build_id());
  {
    double us[4];
    uint64_t ns = 0;
    hsk_submit_host_us(R[0].k, us, &ns, 0);
    if (ns)
      printf("   room 0, host us per submission: staging copy %.1f, upload + preprocessing enqueue %.1f, wait for the preprocessing %.1f, main chain enqueue %.1f\n",
             us[0] / ns, us[1] / ns, us[2] / ns, us[3] / ns);
  }
  fflush(stdout);
  if (getenv("ROOMS_NO_TEARDOWN")) _exit(bad ? 1 : 0);
  for (int r = 0; r < rooms; ++r) {
    hsk_destroy(R[r].k);
    free(R[r].depth);
  }
  return bad ? 1 : 0;
}
