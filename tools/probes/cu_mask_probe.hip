// cu_mask_probe.hip -- which XCDs does a CU-masked stream run on?  (hipExtStreamCreateWithCUMask; VERDICT r05 item 4: "CU masks giving
// each room's ICP its own XCDs".)  For a few mask patterns: a kernel of 2048 workgroups records HW_REG_XCC_ID and the CU it ran on; the
// histogram over XCDs is printed.  Also the time of a fixed amount of work on the masked stream (is a quarter of the chip a quarter as fast?).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cu_mask_probe.hip -o cu_mask_probe && ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void where(unsigned* xcc, unsigned* hwid, float* sink, int spin) {
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID, 4 bits
    hwid[blockIdx.x] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));   // HW_REG_HW_ID
  }
  if (a == 12345.678f) sink[0] = a;
}
int main() {
  const int nb = 2048;
  unsigned *d_x, *d_h;
  float* d_s;
  hipMalloc(&d_x, nb * 4); hipMalloc(&d_h, nb * 4); hipMalloc(&d_s, 4);
  struct Pat { const char* name; std::vector<uint32_t> m; };
  std::vector<Pat> pats;
  pats.push_back({"all 256 bits", std::vector<uint32_t>(8, 0xffffffffu)});
  pats.push_back({"first 64 bits (words 0-1)", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}});
  pats.push_back({"bits with (bit % 8) in {0,1}", std::vector<uint32_t>(8, 0x03030303u)});
  pats.push_back({"bits with (bit % 8) == 0", std::vector<uint32_t>(8, 0x01010101u)});
  pats.push_back({"bits with (bit % 4) == 0", std::vector<uint32_t>(8, 0x11111111u)});
  pats.push_back({"last 64 bits (words 6-7)", {0, 0, 0, 0, 0, 0, 0xffffffffu, 0xffffffffu}});
  for (auto& p : pats) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)p.m.size(), p.m.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", p.name, hipGetErrorString(e)); continue; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(where, dim3(nb), dim3(256), 0, s, d_x, d_h, d_s, 2000);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(where, dim3(nb), dim3(256), 0, s, d_x, d_h, d_s, 20000);
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> x(nb), h(nb);
    hipMemcpy(x.data(), d_x, nb * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h.data(), d_h, nb * 4, hipMemcpyDeviceToHost);
    int hist[16] = {0};
    std::vector<char> seen(16 * 4096, 0);
    int distinct = 0;
    for (int i = 0; i < nb; ++i) {
      hist[x[i] & 15]++;
      const unsigned cu = (h[i] >> 8) & 0xfff;   // (cu_id, sh_id, se_id fields of HW_ID: enough to tell CUs apart within an XCD)
      if (!seen[(x[i] & 15) * 4096 + cu]) { seen[(x[i] & 15) * 4096 + cu] = 1; ++distinct; }
    }
    printf("%-32s: blocks per XCD", p.name);
    for (int i = 0; i < 8; ++i) printf(" %4d", hist[i]);
    printf("   distinct (xcd, cu) %3d   10 launches of fixed work: %.3f ms\n", distinct, ms);
    hipStreamDestroy(s);
  }
  return 0;
}
