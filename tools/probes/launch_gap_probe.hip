// launch_gap_probe.hip -- the gap between two DEPENDENT launches of an ICP-shaped kernel (the ICP is 19 such
// launches; 1.25 us of every iteration is the boundary).  A kernel of 240 workgroups x 256 threads reads 32 shards x 27 doubles that
// the previous launch added to (f64 atomics), does ~1 us of arithmetic, adds to the other slot, and stamps s_memrealtime (100 MHz)
// at its first instruction and after its last atomic.  gap = min over blocks of start(i + 1) - max over blocks of end(i).
// Variants: (a) eager launches on one stream; (b) the 19 launches as ONE hipGraph; (c) accumulators in fine-grained device memory
// (hipDeviceMallocFinegrained), (d) in uncached memory (hipDeviceMallocUncached); each eager and as a graph.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/launch_gap_probe.hip -o launch_gap_probe && ./launch_gap_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define NB 240
#define NIT 19
__global__ __launch_bounds__(256) void iter(double* slots, int it, unsigned long long* stamps, float* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const double* in = slots + (size_t)(it & 1) * 32 * 32;
  double* out = slots + (size_t)((it + 1) & 1) * 32 * 32;
  double s = 0.0;
  if (threadIdx.x < 64) {   // the previous iteration's sums: 32 shards x 27, as the ICP's prologue reads them
    for (int k = threadIdx.x; k < 32 * 27; k += 64) s += __hip_atomic_load(&in[(k / 27) * 32 + k % 27], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  float a = (float)s + threadIdx.x;
  for (int i = 0; i < 400; ++i) a = a * 1.0001f + 0.5f;   // ~1 us of dependent arithmetic: the pixel phase
  if (threadIdx.x < 27) atomicAdd(&out[(blockIdx.x & 31) * 32 + threadIdx.x], (double)(a > 1e30f ? 1.0 : 0.5));
  if (a == 12345.f) sink[0] = a;
  __threadfence();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[((size_t)it * NB + blockIdx.x) * 2] = t0;
    stamps[((size_t)it * NB + blockIdx.x) * 2 + 1] = t1;
  }
}
static void report(const char* name, unsigned long long* d_st, int reps_done) {
  std::vector<unsigned long long> st((size_t)NIT * NB * 2);
  hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  double gaps = 0, lens = 0;
  for (int it = 0; it + 1 < NIT; ++it) {
    unsigned long long end = 0, start = ~0ull, s0 = ~0ull;
    for (int b = 0; b < NB; ++b) {
      end = std::max(end, st[((size_t)it * NB + b) * 2 + 1]);
      s0 = std::min(s0, st[((size_t)it * NB + b) * 2]);
      start = std::min(start, st[((size_t)(it + 1) * NB + b) * 2]);
    }
    gaps += (double)(start - end) * 0.01;
    lens += (double)(end - s0) * 0.01;
  }
  printf("%-58s gap %.2f us   kernel %.2f us   (last of %d repetitions)\n", name, gaps / (NIT - 1), lens / (NIT - 1), reps_done);
}
int main() {
  unsigned long long* d_st; float* d_sink;
  hipMalloc(&d_st, (size_t)NIT * NB * 2 * 8); hipMalloc(&d_sink, 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  struct Mem { const char* name; unsigned flag; bool plain; } mems[] = {{"coarse-grained (hipMalloc)", 0, true}, {"fine-grained (hipDeviceMallocFinegrained)", hipDeviceMallocFinegrained, false},
                                                                {"uncached (hipDeviceMallocUncached)", hipDeviceMallocUncached, false}};
  for (auto& m : mems) {
    double* slots = nullptr;
    hipError_t e = m.plain ? hipMalloc((void**)&slots, 2 * 32 * 32 * 8) : hipExtMallocWithFlags((void**)&slots, 2 * 32 * 32 * 8, m.flag);
    if (e != hipSuccess) { printf("%s: allocation failed: %s\n", m.name, hipGetErrorString(e)); continue; }
    hipMemset(slots, 0, 2 * 32 * 32 * 8);
    char name[160];
    const int reps = 50;
    for (int r = 0; r < reps; ++r) {
      for (int it = 0; it < NIT; ++it) hipLaunchKernelGGL(iter, dim3(NB), dim3(256), 0, s, slots, it, d_st, d_sink);
      hipStreamSynchronize(s);
    }
    snprintf(name, sizeof(name), "eager, %s", m.name);
    report(name, d_st, reps);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int it = 0; it < NIT; ++it) hipLaunchKernelGGL(iter, dim3(NB), dim3(256), 0, s, slots, it, d_st, d_sink);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < reps; ++r) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
    snprintf(name, sizeof(name), "ONE hipGraph of 19 kernel nodes, %s", m.name);
    report(name, d_st, reps);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    hipFree(slots);
  }
  return 0;
}
