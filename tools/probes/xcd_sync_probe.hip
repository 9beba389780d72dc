// xcd_sync_probe.hip -- VERDICT r03 item 4: what does a barrier cost between the workgroups of ONE XCD (32 CUs, one L2),
// against the ~1.4 us of a kernel boundary and the 9.9 us of a chip-wide polled barrier (grid_sync_probe.hip)?
// Workgroups go round the eight XCDs in turn, so those with blockIdx.x % 8 == k sit on one XCD (checked: every
// participant records its HW_REG_XCC_ID).  32 participants x 256 threads arrive on a counter and poll it:
//   (a) L2-local: the arrive and the poll are atomics WITHOUT device scope (workgroup scope: no sc1 bit) -- a read-modify-
//       write always executes in the L2 of the issuing XCD, which all 32 CUs share, so nothing leaves the XCD;
//   (b) the same at device scope (memory side), same 32 workgroups;
//   (c) with the iteration's payload: 27 f64 atomic adds per workgroup into L2 before the arrive, and a read of the 27
//       sums after the barrier (the shape of an ICP iteration's hand-over).
//   hipcc -O3 --offload-arch=gfx950 xcd_sync_probe.hip -o /tmp/xcd_sync_probe && /tmp/xcd_sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE, bool PAYLOAD>
__global__ void k_xcd(int n, int xcd, unsigned* counter, double* sums, unsigned* xcc_of, double* sink) {
  if ((int)(blockIdx.x & 7u) != xcd) return;
  const unsigned part = blockIdx.x >> 3, n_part = gridDim.x >> 3;
  if (threadIdx.x == 0) xcc_of[part] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_XCC_ID, 4 bits
  double acc = 0.0;
  __shared__ bool return_flag;
  if (threadIdx.x == 0) return_flag = false;
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    double* slot = sums + (size_t)(i % 3) * 32;
    if (PAYLOAD && threadIdx.x < 27) __hip_atomic_fetch_add(slot + threadIdx.x, 1.0, __ATOMIC_RELAXED, SCOPE);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, SCOPE);
      const unsigned want = (unsigned)(i + 1) * n_part;
      // (bounded: participants that do NOT share an L2 would never see each other's arrivals -- give up after ~50 ms)
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_fetch_add(counter, 0u, __ATOMIC_RELAXED, SCOPE) < want)
        if (wall_clock64() - t0 > 5000000ull) {
          __hip_atomic_exchange(counter + 32, 1u, __ATOMIC_RELAXED, SCOPE);
          break;
        }
    }
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(counter + 32, 0u, __ATOMIC_RELAXED, SCOPE) != 0u) return_flag = true;
    __syncthreads();
    if (return_flag) return;  // (a wait that timed out: every workgroup leaves as soon as it notices)
    if (PAYLOAD) {
      if (threadIdx.x < 27) acc += __hip_atomic_fetch_add(slot + threadIdx.x, 0.0, __ATOMIC_RELAXED, SCOPE);
      // (the slot the iteration AFTER NEXT adds into is cleared now: nobody touches it before the next barrier -- the ICP's
      // three rotating slots: add / read / clear)
      if (part == 0 && threadIdx.x < 27) __hip_atomic_exchange(sums + (size_t)((i + 2) % 3) * 32 + threadIdx.x, 0.0, __ATOMIC_RELAXED, SCOPE);
    }
  }
  if (threadIdx.x < 27 && part == 0) sink[threadIdx.x] = acc;
}

template <int SCOPE, bool PAYLOAD>
static void run(const char* what, int n, int xcd, unsigned* d_counter, double* d_sums, unsigned* d_xcc, double* d_sink, hipEvent_t e0, hipEvent_t e1) {
  const int parts = 32, blocks = parts * 8, threads = 256;
  for (int rep = 0; rep < 2; ++rep) {  // (first: warm)
    hipMemset(d_counter, 0, 256);
    hipMemset(d_sums, 0, 3 * 32 * 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_xcd<SCOPE, PAYLOAD>), dim3(blocks), dim3(threads), 0, 0, n, xcd, d_counter, d_sums, d_xcc, d_sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> x(parts);
  hipMemcpy(x.data(), d_xcc, parts * 4, hipMemcpyDeviceToHost);
  bool same = true;
  for (int i = 1; i < parts; ++i) same = same && x[i] == x[0];
  double s[27];
  hipMemcpy(s, d_sink, sizeof(s), hipMemcpyDeviceToHost);
  unsigned gave_up = 0;
  hipMemcpy(&gave_up, d_counter + 32, 4, hipMemcpyDeviceToHost);
  if (gave_up) printf("   !! a wait timed out: the participants did not see each other's arrivals\n");
  printf("%-58s n=%d  %.3f us per barrier   (participants on XCC %u, all the same: %s%s)\n", what, n, 1e3 * ms / n, x[0], same ? "yes" : "NO",
         PAYLOAD ? (s[0] == 32.0 * n ? ", sums exact" : ", SUMS WRONG") : "");
}

int main() {
  unsigned *d_counter, *d_xcc;
  double *d_sums, *d_sink;
  hipMalloc(&d_counter, 256);
  hipMalloc(&d_xcc, 64 * 4);
  hipMalloc(&d_sums, 3 * 32 * 8);
  hipMalloc(&d_sink, 27 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int n : {200, 2000}) {
    run<__HIP_MEMORY_SCOPE_WORKGROUP, false>("(a) one XCD, L2-local atomics (no device scope)", n, 0, d_counter, d_sums, d_xcc, d_sink, e0, e1);
    run<__HIP_MEMORY_SCOPE_AGENT, false>("(b) one XCD, device-scope atomics", n, 0, d_counter, d_sums, d_xcc, d_sink, e0, e1);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("(c) one XCD, L2-local, 27 f64 sums added + read back", n, 0, d_counter, d_sums, d_xcc, d_sink, e0, e1);
    run<__HIP_MEMORY_SCOPE_AGENT, true>("(d) one XCD, device scope, 27 f64 sums added + read back", n, 0, d_counter, d_sums, d_xcc, d_sink, e0, e1);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, false>("(a') the same on XCD 5", n, 5, d_counter, d_sums, d_xcc, d_sink, e0, e1);
  }
  return 0;
}
