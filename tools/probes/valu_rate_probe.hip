// valu_rate_probe.hip -- issue rates of the vector instructions the ICP's exact accumulation could be built from: one wave
// per SIMD (as the ICP runs), N independent chains of each instruction, cycles per instruction from s_memtime.
//   hipcc -O3 --offload-arch=gfx950 valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 64
template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, double seed, long long iseed) {
  double a[8];
  long long b[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = iseed + i + threadIdx.x; }
  const double m = seed * 0.5;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < 256; ++r) {
#pragma unroll
    for (int q = 0; q < REP / 8; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) a[i] = a[i] + m;                                  // v_add_f64
        if (KIND == 1) a[i] = __builtin_fma(a[i], m, m);                 // v_fma_f64
        if (KIND == 2) a[i] = __builtin_rint(a[i] * m);                  // v_mul_f64 + v_rndne_f64
        if (KIND == 3) b[i] = b[i] + (long long)__double_as_longlong(m); // 64-bit integer add
        if (KIND == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(b[i]) : "v"(iseed));
        if (KIND == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (KIND == 6) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
        if (KIND == 7) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
        if (KIND == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; long long sb = 0;
  for (int i = 0; i < 8; ++i) { s += a[i]; sb += b[i]; }
  out[blockIdx.x * 256 + threadIdx.x] = s + (double)sb;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 8);
  const char* names[] = {"a += m (f64 add, compiler)", "fma_f64 (compiler)", "rint(a * m) (mul + rndne)", "64-bit integer add (compiler)",
                         "v_lshl_add_u64 (asm)", "v_add_f64 (asm)", "v_fma_f64 (asm)", "v_rndne_f64 (asm)", "v_mul_f64 (asm)"};
  for (int kind = 0; kind < 9; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (kind) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(240), dim3(256), 0, 0, out, cyc, 1.000001, 3ll); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(240), dim3(256), 0, 0, out, cyc, 1.000001, 3ll); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(240), dim3(256), 0, 0, out, cyc, 1.000001, 3ll); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(240), dim3(256), 0, 0, out, cyc, 1.000001, 3ll); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(240), dim3(256), 0, 0, out, cyc, 1.000001, 3ll); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(240), dim3(256), 0, 0, out, cyc, 1.000001, 3ll); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(240), dim3(256), 0, 0, out, cyc, 1.000001, 3ll); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(240), dim3(256), 0, 0, out, cyc, 1.000001, 3ll); break;
        case 8: hipLaunchKernelGGL(k<8>, dim3(240), dim3(256), 0, 0, out, cyc, 1.000001, 3ll); break;
      }
      hipDeviceSynchronize();
    }
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on gfx9?  report raw ticks per instruction; the ratios are what matters
    printf("%-34s %8llu ticks for %d instructions a lane-wave: %.4f ticks each\n", names[kind], c, 256 * REP, (double)c / (256.0 * REP));
  }
  return 0;
}
