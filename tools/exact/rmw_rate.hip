// Read-modify-write streaming rate of the GPU it runs on, for the shapes the integrate stage uses, over a buffer the
// size of a 512^3 volume (512 MiB: twice the Infinity Cache) and of a 1024^3 one (4 GiB).  Every 16-B vector is loaded,
// each of its words incremented, and stored.  Bytes counted: read + written.
//   linear    : a wave's 64 lanes take 1 KiB contiguous; grid-stride; U vectors per lane in flight
//   pieces    : pass A's footprint -- a wave takes 16 lanes x 16 B = 256 B in each of 4 rows (row stride 2 KiB at 512^3),
//               in 8 planes (plane stride 1 MiB), i.e. 32 pieces of 256 B; blocks of 4 waves, x fastest
//   read-only / write-only linear sweeps for reference
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int U, int MODE>  // MODE 0: RMW, 1: read only, 2: write only
__global__ __launch_bounds__(256) void k_linear(uint4* __restrict__ v, size_t n, unsigned* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
    uint4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (MODE != 2 && i + u * stride < n) q[u] = v[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + u * stride >= n) continue;
      if (MODE == 2) q[u] = make_uint4((unsigned)i, 1u, 2u, 3u);
      q[u].x += 0x10000u; q[u].y += 0x10000u; q[u].z += 0x10000u; q[u].w += 0x10000u;
      if (MODE == 1) acc += q[u].x ^ q[u].w; else v[i + u * stride] = q[u];
    }
  }
  if (MODE == 1 && acc == 0x12345u) sink[0] = acc;
}

// volume X x Y x Z of 4-B voxels; block (64, 4): lane -> 4 voxels in x (16 lanes = 64 voxels), 4 rows; 4 waves = 16 rows
template <int PLANES>
__global__ __launch_bounds__(256) void k_pieces(uint4* __restrict__ v, int X, int Y, int Z) {
  const int lane = threadIdx.x;
  const int x0 = (blockIdx.x * 16 + (lane & 15)) * 4;
  const int y = (blockIdx.y * 4 + threadIdx.y) * 4 + (lane >> 4);
  const size_t plane_vec = (size_t)X * Y / 4;
  const size_t idx0 = ((size_t)y * X + x0) / 4;
  const int z0 = blockIdx.z * PLANES;
  uint4 q[PLANES];
#pragma unroll
  for (int u = 0; u < PLANES; ++u) q[u] = v[idx0 + (size_t)(z0 + u) * plane_vec];
#pragma unroll
  for (int u = 0; u < PLANES; ++u) {
    q[u].x += 0x10000u; q[u].y += 0x10000u; q[u].z += 0x10000u; q[u].w += 0x10000u;
    v[idx0 + (size_t)(z0 + u) * plane_vec] = q[u];
  }
}
// the same with a wave taking 1 KiB of ONE row (256 voxels) in PLANES planes
template <int PLANES>
__global__ __launch_bounds__(256) void k_rows(uint4* __restrict__ v, int X, int Y, int Z) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int x0 = (blockIdx.x * 64 + lane) * 4;
  const int y = blockIdx.y * 4 + w;
  const size_t plane_vec = (size_t)X * Y / 4;
  const size_t idx0 = ((size_t)y * X + x0) / 4;
  const int z0 = blockIdx.z * PLANES;
  uint4 q[PLANES];
#pragma unroll
  for (int u = 0; u < PLANES; ++u) q[u] = v[idx0 + (size_t)(z0 + u) * plane_vec];
#pragma unroll
  for (int u = 0; u < PLANES; ++u) {
    q[u].x += 0x10000u; q[u].y += 0x10000u; q[u].z += 0x10000u; q[u].w += 0x10000u;
    v[idx0 + (size_t)(z0 + u) * plane_vec] = q[u];
  }
}

template <typename F>
static double time_us(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0 / reps;
}

int main(int argc, char** argv) {
  for (int n : {512, 1024}) {
    const size_t bytes = (size_t)n * n * n * 4, nv = bytes / 16;
    uint4* d; unsigned* sink;
    CK(hipMalloc((void**)&d, bytes)); CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(d, 0, bytes));
    const int reps = n == 512 ? 20 : 5;
    auto rep = [&](const char* name, double us, double factor) {
      printf("%4d^3  %-44s %8.1f us  %6.2f TB/s\n", n, name, us, factor * bytes / us / 1e6);
    };
    rep("linear RMW, 1 vector in flight, 8 blocks/CU", time_us([&] { hipLaunchKernelGGL((k_linear<1, 0>), dim3(2048), dim3(256), 0, 0, d, nv, sink); }, reps), 2);
    rep("linear RMW, 4 vectors in flight, 8 blocks/CU", time_us([&] { hipLaunchKernelGGL((k_linear<4, 0>), dim3(2048), dim3(256), 0, 0, d, nv, sink); }, reps), 2);
    rep("linear RMW, 8 vectors in flight, 8 blocks/CU", time_us([&] { hipLaunchKernelGGL((k_linear<8, 0>), dim3(2048), dim3(256), 0, 0, d, nv, sink); }, reps), 2);
    rep("linear RMW, 4 in flight, one block per 4 KiB x 4", time_us([&] { hipLaunchKernelGGL((k_linear<4, 0>), dim3((unsigned)(nv / 1024)), dim3(256), 0, 0, d, nv, sink); }, reps), 2);
    rep("linear read only, 4 in flight", time_us([&] { hipLaunchKernelGGL((k_linear<4, 1>), dim3(2048), dim3(256), 0, 0, d, nv, sink); }, reps), 1);
    rep("linear write only, 4 in flight", time_us([&] { hipLaunchKernelGGL((k_linear<4, 2>), dim3(2048), dim3(256), 0, 0, d, nv, sink); }, reps), 1);
    rep("pass-A pieces (256 B x 4 rows) x 8 planes", time_us([&] { hipLaunchKernelGGL((k_pieces<8>), dim3(n / 64, n / 16, n / 8), dim3(64, 4), 0, 0, d, n, n, n); }, reps), 2);
    rep("pass-A pieces (256 B x 4 rows) x 4 planes", time_us([&] { hipLaunchKernelGGL((k_pieces<4>), dim3(n / 64, n / 16, n / 4), dim3(64, 4), 0, 0, d, n, n, n); }, reps), 2);
    rep("rows (1 KiB of one row) x 8 planes", time_us([&] { hipLaunchKernelGGL((k_rows<8>), dim3(n / 256, n / 4, n / 8), dim3(256), 0, 0, d, n, n, n); }, reps), 2);
    rep("rows (1 KiB of one row) x 4 planes", time_us([&] { hipLaunchKernelGGL((k_rows<4>), dim3(n / 256, n / 4, n / 4), dim3(256), 0, 0, d, n, n, n); }, reps), 2);
    rep("rows (1 KiB of one row) x 1 plane", time_us([&] { hipLaunchKernelGGL((k_rows<1>), dim3(n / 256, n / 4, n), dim3(256), 0, 0, d, n, n, n); }, reps), 2);
    CK(hipMemcpyAsync(d, d, 16, hipMemcpyDeviceToDevice, 0));
    CK(hipDeviceSynchronize());
    CK(hipFree(d)); CK(hipFree(sink));
  }
  return 0;
}
