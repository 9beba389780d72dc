// grid_sync_probe.hip -- what does a grid-wide barrier cost inside one launch, against the ~1.35 us of a kernel boundary?
// (a) cooperative_groups::grid_group::sync() of a cooperative launch; (b) a hand-written barrier: one device-scope atomic
// per block on a counter, then a polled device-scope load.  240 blocks x 256 threads (the ICP's fine level), N barriers
// per launch, HIP events around the launch.
//   hipcc -O3 --offload-arch=gfx950 grid_sync_probe.hip -o /tmp/grid_sync_probe && /tmp/grid_sync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_coop(int n, unsigned* sink) {
  cg::grid_group g = cg::this_grid();
  unsigned acc = 0;
  for (int i = 0; i < n; ++i) {
    g.sync();
    acc += i;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *sink = acc;
}

__global__ void k_manual(int n, unsigned* counter, unsigned* sink) {
  unsigned acc = 0;
  for (int i = 0; i < n; ++i) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(i + 1) * gridDim.x;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {}
    }
    __syncthreads();
    acc += i;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *sink = acc;
}

__global__ void k_empty(unsigned* sink) { if (threadIdx.x == 0 && blockIdx.x == 0) *sink = 1; }

int main() {
  unsigned *d_sink, *d_counter;
  hipMalloc(&d_sink, 4);
  hipMalloc(&d_counter, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 240, threads = 256;
  for (int n : {100, 1000}) {
    float ms = 0;
    void* args[] = {(void*)&n, (void*)&d_sink};
    hipLaunchCooperativeKernel((void*)k_coop, dim3(blocks), dim3(threads), args, 0, 0);  // warm
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipError_t e = hipLaunchCooperativeKernel((void*)k_coop, dim3(blocks), dim3(threads), args, 0, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("cooperative grid.sync: n=%d  %.3f us per barrier (launch %s)\n", n, 1e3 * ms / n, hipGetErrorString(e));
    hipMemset(d_counter, 0, 4);
    hipLaunchKernelGGL(k_manual, dim3(blocks), dim3(threads), 0, 0, n, d_counter, d_sink);
    hipDeviceSynchronize();
    hipMemset(d_counter, 0, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_manual, dim3(blocks), dim3(threads), 0, 0, n, d_counter, d_sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("manual atomic barrier: n=%d  %.3f us per barrier\n", n, 1e3 * ms / n);
  }
  {
    float ms = 0;
    const int n = 1000;
    hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, 0, d_sink);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, 0, d_sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("back-to-back empty kernels (240 x 256): %.3f us per launch\n", 1e3 * ms / n);
  }
  return 0;
}
