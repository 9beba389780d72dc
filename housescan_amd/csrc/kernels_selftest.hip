// kernels_selftest.hip -- exhaustive self-test of the exact-arithmetic shortcuts the kernels rely on (hsk_dev.h:
// hsk_rcp_exact, hsk_sqrt_exact, hsk_div_small_exact).  The shortcuts are correct on gfx950 because this test says so
// on the hardware itself, over their whole domains, not because of a rounding-error argument.
#pragma clang fp contract(off)
#include "../../include/hskinfu.h"
#include "hsk_dev.h"

// counts: [0] 1/x values checked        [1] hsk_rcp_exact wrong        [2] bare v_rcp_f32 wrong (shows the test bites)
//         [3] sqrt values checked       [4] hsk_sqrt_exact wrong       [5] bare v_sqrt_f32 wrong
//         [6] a/n pairs checked         [7] hsk_div_small_exact wrong
__global__ __launch_bounds__(256) void k_selftest_unary(unsigned long long* __restrict__ counts) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;  // 2^28 threads, 16 bit patterns each
  unsigned long long t[6] = {0, 0, 0, 0, 0, 0};
  for (uint32_t rep = 0; rep < 16; ++rep) {
    const uint32_t bits = tid + (rep << 28);
    const float x = __uint_as_float(bits);
    const unsigned ex = (bits >> 23) & 255u;
    if (ex == 0 || ex == 255) continue;  // zero, denormal, inf, NaN
    const float inv = 1.0f / x;          // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
    const unsigned iex = (__float_as_uint(inv) >> 23) & 255u;
    if (iex != 0 && iex != 255) {
      t[0] += 1;
      t[1] += __float_as_uint(hsk_rcp_exact(x)) != __float_as_uint(inv);
      t[2] += __float_as_uint(__builtin_amdgcn_rcpf(x)) != __float_as_uint(inv);
    }
    if (!(bits >> 31) && ex >= 25) {     // x >= 2^-102
      const float root = sqrtf(x);
      t[3] += 1;
      t[4] += __float_as_uint(hsk_sqrt_exact(x)) != __float_as_uint(root);
      t[5] += __float_as_uint(__builtin_amdgcn_sqrtf(x)) != __float_as_uint(root);
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    unsigned long long v = t[q];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counts[q], v);
  }
}

// every a with 2^-100 <= |a| < 512 (and a = +0) against every n = 1 .. 129: blockIdx.y = n - 1
__global__ __launch_bounds__(256) void k_selftest_div(unsigned long long* __restrict__ counts) {
  const float n = (float)(blockIdx.y + 1);
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;  // 2^24 threads, 256 bit patterns each
  unsigned long long checked = 0, wrong = 0;
  for (uint32_t rep = 0; rep < 256; ++rep) {
    const uint32_t bits = tid + (rep << 24);
    const unsigned ex = (bits >> 23) & 255u;
    const bool zero = bits == 0u;  // (-0 gives +0 where a/n gives -0: the quotient is only ever converted to an integer)
    if (!zero && (ex < 27u || ex > 135u)) continue;
    const float a = __uint_as_float(bits);
    const float want = a / n;
    checked += 1;
    wrong += __float_as_uint(hsk_div_small_exact(a, n)) != __float_as_uint(want);
  }
  for (int o = 32; o > 0; o >>= 1) {
    checked += __shfl_down(checked, o, 64);
    wrong += __shfl_down(wrong, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (checked) atomicAdd(&counts[6], checked);
    if (wrong) atomicAdd(&counts[7], wrong);
  }
}

extern "C" int hsk_selftest_exact_ops(int device_id, uint64_t counts[8]) {
  if (!counts) return HSK_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return HSK_ERR_NOGPU;
  if (device_id < 0 || device_id >= ndev) return HSK_ERR_ARG;
  if (hipSetDevice(device_id) != hipSuccess) return HSK_ERR_HIP;
  unsigned long long* d = nullptr;
  if (hipMalloc((void**)&d, 8 * sizeof(unsigned long long)) != hipSuccess) return HSK_ERR_HIP;
  hipError_t e = hipMemset(d, 0, 8 * sizeof(unsigned long long));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_selftest_unary, dim3(1u << 20), dim3(256), 0, 0, d);
    hipLaunchKernelGGL(k_selftest_div, dim3(1u << 16, 129), dim3(256), 0, 0, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  unsigned long long h[8] = {};
  if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return HSK_ERR_HIP;
  for (int i = 0; i < 8; ++i) counts[i] = h[i];
  return HSK_OK;
}
