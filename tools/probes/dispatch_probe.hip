// dispatch_probe.hip -- what does a workgroup cost that leaves at once?  Half of pass A's launch (8 k of 16 k workgroups at
// 512^3, 65 k of 131 k at 1024^3) lies outside the view frustum and leaves after one scalar load; this measures the floor
// under that: launches of pass A's shape (256 threads as 64 x 4, the 3-D grids of 512^3 and 1024^3) whose workgroups
//   (a) all leave after one scalar load and a compare (the dead half's path),
//   (b) leave likewise, the kernel built with 16 preloaded kernarg dwords and a 200-byte argument block as pass A is,
// and, for scale, (c) the same number of workgroups as 64-thread and as 1024-thread workgroups.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 dispatch_probe.hip -o /tmp/dispatch_probe && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>

struct Big {
  float f[40];
};

__global__ __launch_bounds__(256) void k_leave(const int2* __restrict__ wgz, int zchunk, unsigned gdx, int* __restrict__ sink) {
  const unsigned long long w2 = *(const unsigned long long*)(wgz + (blockIdx.y * gdx + blockIdx.x));
  const int lo = (int)(unsigned)w2, hi = (int)(unsigned)(w2 >> 32);
  const int zbeg = blockIdx.z * zchunk;
  if ((zbeg > hi) | (zbeg + zchunk - 1 < lo)) return;
  sink[threadIdx.x] = zbeg;  // (never: every range is empty)
}

__global__ __launch_bounds__(256) void k_leave_big(const int2* __restrict__ wgz, int zchunk, unsigned gdx, unsigned a, double* p0,
                                                   unsigned char* p1, const float2* p2, int b, int c, int d, int e, unsigned f,
                                                   Big big, int* __restrict__ sink) {
  const unsigned long long w2 = *(const unsigned long long*)(wgz + (blockIdx.y * gdx + blockIdx.x));
  const int lo = (int)(unsigned)w2, hi = (int)(unsigned)(w2 >> 32);
  const int zbeg = blockIdx.z * zchunk;
  if ((zbeg > hi) | (zbeg + zchunk - 1 < lo)) return;
  sink[threadIdx.x] = zbeg + (int)a + b + c + d + e + (int)f + (int)big.f[threadIdx.x % 40] + (int)(size_t)p0 + (int)(size_t)p1 + (int)(size_t)p2;
}

template <int T>
__global__ __launch_bounds__(T) void k_leave_1d(const int2* __restrict__ wgz, int* __restrict__ sink) {
  const unsigned long long w2 = *(const unsigned long long*)(wgz + (blockIdx.x & 255u));
  if ((int)(unsigned)w2 > (int)(unsigned)(w2 >> 32)) return;
  sink[threadIdx.x] = 1;
}

int main() {
  int2* d_wgz;
  int* d_sink;
  hipMalloc(&d_wgz, 4096 * sizeof(int2));
  hipMalloc(&d_sink, 4096);
  int2 h[4096];
  for (auto& v : h) v = make_int2(0x7fffffff, -0x7fffffff);
  hipMemcpy(d_wgz, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  Big big = {};
  struct Shape {
    const char* what;
    dim3 grid;
    int zchunk;
  } shapes[] = {{"512^3, 8-plane chunks (pass A today)", dim3(8, 32, 64), 8},
                {"512^3, 16-plane chunks", dim3(8, 32, 32), 16},
                {"1024^3, 16-plane chunks (pass A today)", dim3(16, 64, 64), 16},
                {"1024^3, 8-plane chunks", dim3(16, 64, 128), 8}};
  const int reps = 50;
  for (const Shape& s : shapes) {
    const long wgs = (long)s.grid.x * s.grid.y * s.grid.z;
    for (int form = 0; form < 2; ++form) {
      float best = 1e30f, sum = 0.0f;
      for (int rep = 0; rep < reps + 5; ++rep) {
        hipEventRecord(e0, 0);
        if (form == 0)
          hipLaunchKernelGGL(k_leave, s.grid, dim3(64, 4), 0, 0, d_wgz, s.zchunk, s.grid.x, d_sink);
        else
          hipLaunchKernelGGL(k_leave_big, s.grid, dim3(64, 4), 0, 0, d_wgz, s.zchunk, s.grid.x, 1u, nullptr, nullptr, nullptr, 1, 2, 3,
                             4, 5u, big, d_sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 5) {
          best = ms < best ? ms : best;
          sum += ms;
        }
      }
      printf("%-42s %s: %7ld workgroups leave at once: %.1f us a launch (fastest %.1f) = %.2f ns a workgroup\n", s.what,
             form ? "(b) 200-B arguments" : "(a) small arguments ", wgs, sum / reps * 1e3f, best * 1e3f, sum / reps * 1e6f / wgs);
    }
  }
  for (int t = 0; t < 3; ++t) {
    const long wgs = 16384;
    float sum = 0.0f;
    for (int rep = 0; rep < reps + 5; ++rep) {
      hipEventRecord(e0, 0);
      if (t == 0) hipLaunchKernelGGL(k_leave_1d<64>, dim3(wgs), dim3(64), 0, 0, d_wgz, d_sink);
      if (t == 1) hipLaunchKernelGGL(k_leave_1d<256>, dim3(wgs), dim3(256), 0, 0, d_wgz, d_sink);
      if (t == 2) hipLaunchKernelGGL(k_leave_1d<1024>, dim3(wgs), dim3(1024), 0, 0, d_wgz, d_sink);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 5) sum += ms;
    }
    printf("(c) %ld workgroups of %4d threads, 1-D grid: %.1f us a launch\n", wgs, t == 0 ? 64 : (t == 1 ? 256 : 1024), sum / reps * 1e3f);
  }
  return 0;
}
