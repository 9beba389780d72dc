// Exhaustive check (all 2^32 binary32 bit patterns) on the GPU it runs on:
//   1. v_rcp_f32 + one FMA Newton step  ==  the correctly rounded 1.0f / x         (normal x, normal result)
//   2. v_sqrt_f32 + one FMA correction  ==  the correctly rounded sqrtf(x)         (normal x)
// The kernels of integrate pass B rely on both (hsk_rcp_exact / hsk_sqrt_exact in hsk_dev.h).  Prints the number
// of values checked, how often the BARE instructions are wrong (shows that the comparison bites) and the mismatches
// of the refined forms.  Exit code 1 when a refined form is ever wrong.
// Build: hipcc --offload-arch=gfx950 -O2 -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -o rcp_sqrt_check rcp_sqrt_check.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float rcp_refined(float x) {
  const float r0 = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r0, 1.0f);
  return __builtin_fmaf(r0, e, r0);
}
__device__ __forceinline__ float sqrt_refined(float x) {
  const float s0 = __builtin_amdgcn_sqrtf(x);
#ifdef SQRT_H_RSQ
  const float h = 0.5f * __builtin_amdgcn_rsqf(x);   // 1 / (2 sqrt x), approximately
#else
  const float h = 0.5f * __builtin_amdgcn_rcpf(s0);  // 1 / (2 s0), approximately
#endif
  const float d = __builtin_fmaf(-s0, s0, x);        // exact residual (s0 is within 1 ulp of the root)
  return __builtin_fmaf(d, h, s0);
}

// tally: [0] rcp values checked, [1] bare v_rcp_f32 wrong, [2] refined rcp wrong,
//        [3] sqrt values checked, [4] bare v_sqrt_f32 wrong, [5] refined sqrt wrong
__global__ void k_check(unsigned long long* tally, unsigned* first_bad, unsigned* by_exp) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;  // 2^28 threads, 16 patterns each
  unsigned long long t[6] = {0, 0, 0, 0, 0, 0};
  for (uint32_t rep = 0; rep < 16; ++rep) {
    const uint32_t bits = tid + (rep << 28);
    const float x = __uint_as_float(bits);
    const unsigned ex = (bits >> 23) & 255u;
    if (ex == 0 || ex == 255) continue;  // zero, denormal, inf, NaN: not in the domain
    {
      const float want = 1.0f / x;  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
      const unsigned wex = (__float_as_uint(want) >> 23) & 255u;
      if (wex != 0 && wex != 255) {
        t[0] += 1;
        t[1] += __float_as_uint(__builtin_amdgcn_rcpf(x)) != __float_as_uint(want);
        if (__float_as_uint(rcp_refined(x)) != __float_as_uint(want)) {
          t[2] += 1;
          atomicMin(&first_bad[0], bits);
        }
      }
    }
    if (!(bits >> 31)) {
      const float want = sqrtf(x);
      t[3] += 1;
      t[4] += __float_as_uint(__builtin_amdgcn_sqrtf(x)) != __float_as_uint(want);
      if (__float_as_uint(sqrt_refined(x)) != __float_as_uint(want)) {
        t[5] += 1;
        atomicMin(&first_bad[1], bits);
        atomicAdd(&by_exp[ex], 1u);
        if (ex == 126 || ex == 128 || ex == 125) printf("sqrt miss: x = 0x%08x  got 0x%08x  want 0x%08x  bare 0x%08x\n", bits, __float_as_uint(sqrt_refined(x)), __float_as_uint(want), __float_as_uint(__builtin_amdgcn_sqrtf(x)));
      }
    }
  }
  for (int q = 0; q < 6; ++q) {
    unsigned long long v = t[q];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&tally[q], v);
  }
}

int main() {
  unsigned long long *d_t, h_t[6];
  unsigned *d_f, h_f[2] = {0xffffffffu, 0xffffffffu};
  if (hipMalloc(&d_t, sizeof(h_t)) != hipSuccess || hipMalloc(&d_f, sizeof(h_f)) != hipSuccess) return 2;
  (void)hipMemset(d_t, 0, sizeof(h_t));
  (void)hipMemcpy(d_f, h_f, sizeof(h_f), hipMemcpyHostToDevice);
  unsigned *d_e, h_e[256];
  if (hipMalloc(&d_e, sizeof(h_e)) != hipSuccess) return 2;
  (void)hipMemset(d_e, 0, sizeof(h_e));
  hipLaunchKernelGGL(k_check, dim3(1u << 20), dim3(256), 0, 0, d_t, d_f, d_e);
  const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
  if (e1 != hipSuccess || e2 != hipSuccess) {
    printf("launch %s / sync %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    return 2;
  }
  (void)hipMemcpy(h_t, d_t, sizeof(h_t), hipMemcpyDeviceToHost);
  (void)hipMemcpy(h_f, d_f, sizeof(h_f), hipMemcpyDeviceToHost);
  printf("rcp : %llu values checked, bare v_rcp_f32 wrong on %llu, refined wrong on %llu (first 0x%08x)\n", h_t[0], h_t[1], h_t[2], h_f[0]);
  printf("sqrt: %llu values checked, bare v_sqrt_f32 wrong on %llu, refined wrong on %llu (first 0x%08x)\n", h_t[3], h_t[4], h_t[5], h_f[1]);
  (void)hipMemcpy(h_e, d_e, sizeof(h_e), hipMemcpyDeviceToHost);
  for (int e = 0; e < 256; ++e)
    if (h_e[e]) printf("  sqrt: biased exponent %d: %u refined results wrong\n", e, h_e[e]);
  return (h_t[2] || h_t[5] || !h_t[0] || !h_t[3]) ? 1 : 0;
}
