// exchange.hip -- composite helpers of the z-slab exchange: which slab won a pixel, and its maps.
#pragma clang fp contract(off)
#include "hsk_dev.h"
#include "hsk_launch.h"

// ------------------------------------------------------------------------------------------------------
// multi-GPU composite helpers: after the MIN all-reduce of the step keys, a slab keeps its
// maps only where it won; the bit patterns are then SUM-all-reduced as int32 (exact, keeps NaN and -0).
// ------------------------------------------------------------------------------------------------------
__global__ void k_resolve(const int* __restrict__ keys_local, const int* __restrict__ keys_min,
                          const float* __restrict__ vmap, const float* __restrict__ nmap, int* __restrict__ bits, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int kl = keys_local[i], km = keys_min[i];
  const bool mine = (kl == km) && (km != HSK_KEY_NONE_I) && ((km & 1) == 0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    bits[c * P + i] = mine ? __float_as_int(vmap[c * P + i]) : 0;
    bits[(3 + c) * P + i] = mine ? __float_as_int(nmap[c * P + i]) : 0;
  }
}
// (k_adopt -- the composite into the model maps of level 0 -- lives in kernels_image.hip since round 6, fused with the model
// pyramid and the pose report: k_adopt_pyramid)
// Direct exchange (hskinfu_group's one-hop form, SURVEY.md 8(e) "xGMI fit"): the slab that won a pixel stores the bit
// patterns of its vertex / normal straight into EVERY device's composite buffer (its own included) -- peer-mapped
// memory, one hop over xGMI -- and nothing where it lost: a pixel has at most one winner among all slabs (a march step is
// owned by exactly one slab), so the writers never collide, and k_adopt reads the composite only where the MIN key says
// "hit".  Replaces the 7.4 MB all-reduce(SUM) by 24 B per won pixel and peer.
__global__ void k_resolve_push(const int* __restrict__ keys_local, const int* __restrict__ keys_min,
                               const float* __restrict__ vmap, const float* __restrict__ nmap, PushDests dst, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int kl = keys_local[i], km = keys_min[i];
  if (!((kl == km) && (km != HSK_KEY_NONE_I) && ((km & 1) == 0))) return;
  int w[6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    w[c] = __float_as_int(vmap[c * P + i]);
    w[3 + c] = __float_as_int(nmap[c * P + i]);
  }
  for (int d = 0; d < dst.n; ++d) {
    int* __restrict__ b = dst.p[d];
#pragma unroll
    for (int c = 0; c < 6; ++c) b[c * P + i] = w[c];
  }
}
void launch_resolve_push(hipStream_t s, const int* keys_local, const int* keys_min, const float* vmap, const float* nmap,
                         const PushDests& dst, int P) {
  hipLaunchKernelGGL(k_resolve_push, dim3((P + 255) / 256), dim3(256), 0, s, keys_local, keys_min, vmap, nmap, dst, P);
}
void launch_resolve(hipStream_t s, const int* keys_local, const int* keys_min, const float* vmap, const float* nmap,
                    int* bits, int P) {
  hipLaunchKernelGGL(k_resolve, dim3((P + 255) / 256), dim3(256), 0, s, keys_local, keys_min, vmap, nmap, bits, P);
}

