// Probe for the direct (no-RCCL) slab exchange: do stream memory operations and cross-process peer mappings behave on this
// box the way hskinfu_group's direct mode needs?   build: hipcc --offload-arch=gfx950 -o /tmp/svp stream_value_probe.hip -lrt
//   svp local           one process: wait / write value on pinned host memory and on device memory, two streams
//   svp ipc A|B <name>  two processes on device 0: A allocates a device buffer + a POSIX shm flag page, B maps both,
//                       writes the buffer from a kernel and raises the flag from its stream; A's stream waits for it
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void k_fill(int* p, int n, int v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void k_sum(const int* p, int n, unsigned long long* out) { unsigned long long s = 0; for (int i = threadIdx.x; i < n; i += blockDim.x) s += (unsigned)p[i]; atomicAdd(out, s); }

struct Shared { volatile unsigned flag_b_done, flag_a_seen; volatile int a_ready, b_attached; hipIpcMemHandle_t handle; };

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  CK(hipSetDevice(0));
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("CanUseStreamWaitValue = %d\n", can);
  if (!strcmp(argv[1], "local")) {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    for (int kind = 0; kind < 3; ++kind) {
      unsigned* flag = nullptr;
      const char* what = kind == 0 ? "pinned host (hipHostMalloc)" : (kind == 1 ? "device (hipMalloc)" : "signal memory (hipExtMallocWithFlags)");
      hipError_t e = kind == 0 ? hipHostMalloc((void**)&flag, 64, hipHostMallocPortable) : (kind == 1 ? hipMalloc((void**)&flag, 64) : hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
      if (e != hipSuccess) { printf("%s: alloc failed: %s\n", what, hipGetErrorString(e)); continue; }
      CK(hipMemset(flag, 0, 8));
      CK(hipDeviceSynchronize());
      int* buf; unsigned long long* out;
      CK(hipMalloc((void**)&buf, 4 << 20)); CK(hipMalloc((void**)&out, 8)); CK(hipMemset(out, 0, 8));
      e = hipStreamWaitValue32(sa, flag, 7, hipStreamWaitValueGte, 0xffffffffu);
      if (e != hipSuccess) { printf("%s: hipStreamWaitValue32 -> %s\n", what, hipGetErrorString(e)); continue; }
      hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, sa, buf, 1 << 20, out);
      usleep(20000);
      const bool early = hipStreamQuery(sa) == hipSuccess;
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, sb, buf, 1 << 20, 1);
      e = hipStreamWriteValue32(sb, flag, 7, 0);
      if (e != hipSuccess) { printf("%s: hipStreamWriteValue32 -> %s\n", what, hipGetErrorString(e)); continue; }
      CK(hipStreamSynchronize(sb));
      CK(hipStreamSynchronize(sa));
      unsigned long long h = 0; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
      unsigned long long want = 0; for (unsigned i = 0; i < (1u << 20); ++i) want += 1u + i;
      printf("%s: waiter ran early=%d, sum %s\n", what, (int)early, h == want ? "correct (saw the writer's data)" : "WRONG");
    }
    return 0;
  }
  if (!strcmp(argv[1], "ipc") && argc >= 4) {
    const bool A = argv[2][0] == 'A';
    int fd = shm_open(argv[3], O_CREAT | O_RDWR, 0600);
    if (fd < 0) { perror("shm_open"); return 2; }
    if (ftruncate(fd, 4096) != 0) { perror("ftruncate"); return 2; }
    Shared* S = (Shared*)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    CK(hipHostRegister((void*)S, 4096, hipHostRegisterMapped | hipHostRegisterPortable));
    unsigned* dflag = nullptr;
    CK(hipHostGetDevicePointer((void**)&dflag, (void*)&S->flag_b_done, 0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 1 << 20;
    if (A) {
      int* buf; CK(hipMalloc((void**)&buf, n * 4)); CK(hipMemset(buf, 0, n * 4)); CK(hipDeviceSynchronize());
      CK(hipIpcGetMemHandle((hipIpcMemHandle_t*)&S->handle, buf));
      __sync_synchronize(); S->a_ready = 1;
      unsigned long long* out; CK(hipMalloc((void**)&out, 8)); CK(hipMemset(out, 0, 8));
      hipError_t e = hipStreamWaitValue32(s, dflag, 1, hipStreamWaitValueGte, 0xffffffffu);
      printf("A: hipStreamWaitValue32 on the registered shm page -> %s\n", hipGetErrorString(e));
      if (e != hipSuccess) return 3;
      hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, s, buf, n, out);
      CK(hipStreamSynchronize(s));
      unsigned long long h = 0; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
      unsigned long long want = 0; for (unsigned i = 0; i < (unsigned)n; ++i) want += 5u + i;
      printf("A: after B's flag: sum %s\n", h == want ? "correct (saw B's peer writes)" : "WRONG");
      S->flag_a_seen = 1;
      shm_unlink(argv[3]);
      return h == want ? 0 : 4;
    } else {
      while (!S->a_ready) usleep(1000);
      int* peer = nullptr;
      hipError_t e = hipIpcOpenMemHandle((void**)&peer, *(hipIpcMemHandle_t*)&S->handle, hipIpcMemLazyEnablePeerAccess);
      printf("B: hipIpcOpenMemHandle -> %s\n", hipGetErrorString(e));
      if (e != hipSuccess) return 3;
      usleep(200000);  // let A's stream reach its wait
      hipLaunchKernelGGL(k_fill, dim3(n / 256), dim3(256), 0, s, peer, n, 5);
      e = hipStreamWriteValue32(s, dflag, 1, 0);
      printf("B: hipStreamWriteValue32 on the registered shm page -> %s\n", hipGetErrorString(e));
      CK(hipStreamSynchronize(s));
      for (int i = 0; i < 5000 && !S->flag_a_seen; ++i) usleep(1000);
      printf("B: A %s\n", S->flag_a_seen ? "finished" : "did NOT finish within 5 s");
      CK(hipIpcCloseMemHandle(peer));
      return 0;
    }
  }
  return 1;
}
