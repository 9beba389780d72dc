/* abi_harness.c -- the caller Haskell's FFI would be: a plain C program that links libhskinfu (no dlopen, no Python,
 * no torch in the process) and drives it the way the frame loop replacing Main.hs:1285-1290 would: whole depth frames
 * in (the shape of HoniHelper.hs:20), poses out.  It runs the same frames through one context and through a group of
 * two z-slabs, and prints every pose as hex words; tests/test_gpu_group.py compares them with the Python-driven runs.
 *
 *   gcc -std=c99 -Iinclude tests/abi_harness.c -Lhousescan_amd -lhskinfu -Wl,-rpath,$PWD/housescan_amd -o abi_harness
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hskinfu.h"

static void print_pose(const char* tag, int frame, const float pose[16], int tracked) {
  unsigned w[16];
  memcpy(w, pose, sizeof(w));
  printf("%s %d %d", tag, frame, tracked);
  for (int i = 0; i < 12; ++i) printf(" %08x", w[i]);
  printf("\n");
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 64, frames = argc > 2 ? atoi(argv[2]) : 4;
  hsk_config cfg;
  hsk_default_config(&cfg, n);
  uint16_t* depth = (uint16_t*)malloc((size_t)cfg.width * cfg.height * 2);
  float gt[16], pose[16];
  int tracked = 0;

  hsk_ctx* k = NULL;
  if (hsk_create(&cfg, &k) != HSK_OK) {
    fprintf(stderr, "hsk_create: %s\n", hsk_last_error(NULL));
    return 1;
  }
  for (int f = 0; f < frames; ++f) {
    hsk_synth_pose(f, gt);
    hsk_synth_render(gt, cfg.width, cfg.height, cfg.fx, cfg.fy, cfg.cx, cfg.cy, depth);
    if (hsk_process_frame(k, depth, cfg.width, cfg.height, pose, &tracked) != HSK_OK) {
      fprintf(stderr, "hsk_process_frame: %s\n", hsk_last_error(k));
      return 1;
    }
    print_pose("single", f, pose, tracked);
  }
  size_t n_pts = 0;
  if (hsk_extract_cloud(k, NULL, 0, &n_pts) != HSK_OK) return 1;
  printf("cloud %zu\n", n_pts);
  hsk_destroy(k);

  const int devs[2] = {0, 0}; /* two slabs, composited inside the library; on a multi-GPU node: {0, 1} */
  hsk_group* g = NULL;
  if (hsk_group_create(&cfg, 2, devs, 0, &g) != HSK_OK) {
    fprintf(stderr, "hsk_group_create: %s\n", hsk_group_last_error(NULL));
    return 1;
  }
  for (int f = 0; f < frames; ++f) {
    hsk_synth_pose(f, gt);
    hsk_synth_render(gt, cfg.width, cfg.height, cfg.fx, cfg.fy, cfg.cx, cfg.cy, depth);
    if (hsk_group_process_frame(g, depth, cfg.width, cfg.height, pose, &tracked) != HSK_OK) {
      fprintf(stderr, "hsk_group_process_frame: %s\n", hsk_group_last_error(g));
      return 1;
    }
    print_pose("group", f, pose, tracked);
  }
  hsk_group_destroy(g);
  free(depth);
  printf("done\n");
  return 0;
}
