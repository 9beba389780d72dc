// VALU issue rate on the GPU it runs on: wave64 instructions per cycle per SIMD for plain f32, packed f32 and
// transcendental instructions (8 waves per SIMD, independent chains, no memory).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  for (int i = 0; i < N_ITER; ++i) {
    if (KIND == 0) {  // 8 independent v_fma_f32
      a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f); a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
      a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f); a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
    } else if (KIND == 1) {  // 4 independent v_pk_fma_f32 (8 FMAs)
      p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c); p2 = __builtin_elementwise_fma(p2, m, c); p3 = __builtin_elementwise_fma(p3, m, c);
    } else if (KIND == 2) {  // 8 independent v_rcp_f32
      a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
      a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
    } else {  // 8 independent v_add_u32-ish integer ops
      int b0 = __float_as_int(a0), b1 = __float_as_int(a1), b2 = __float_as_int(a2), b3 = __float_as_int(a3);
      b0 = b0 * 3 + 1; b1 = b1 * 3 + 1; b2 = b2 * 3 + 1; b3 = b3 * 3 + 1;
      a0 = __int_as_float(b0 & 0x3fffffff); a1 = __int_as_float(b1 & 0x3fffffff); a2 = __int_as_float(b2 & 0x3fffffff); a3 = __int_as_float(b3 & 0x3fffffff);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int KIND>
static void run(const char* name, int per_iter, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = (double)blocks * 4 * N_ITER * per_iter;
  printf("%-28s %8.1f us  %.3f wave-instr / ns chip-wide  = %.2f cycles per wave-instr per SIMD at 2.4 GHz\n", name, ms * 1e3,
         wave_instr / (ms * 1e6), 1024.0 * 2.4 / (wave_instr / (ms * 1e6)));
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_fma_f32", 8, d);
  run<1>("v_pk_fma_f32", 4, d);
  run<2>("v_rcp_f32", 8, d);
  run<3>("int mul+add+and (3 ops x4)", 12, d);
  return 0;
}
