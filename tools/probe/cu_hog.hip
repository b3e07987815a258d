// Probe (not product code): a kernel that HOLDS CUs the way an RCCL channel kernel does during a collective -- one 512-thread
// workgroup with 128 KB of LDS per CU (so nothing of this library's persistent kernels fits next to it), spinning on the clock
// for `us` microseconds.  `blocks` = 8 r puts r of them on every XCD (block b lands on XCD b % 8).  tools/probe/comm_cus_ab.py
// runs a training step under it to price PVRL_COMPUTE_CUS (csrc/common.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(512) void cu_hog_kernel(long ticks, int* sink) {
  extern __shared__ char lds[];
  const long t0 = wall_clock64();
  lds[threadIdx.x] = (char)threadIdx.x;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (sink && lds[(threadIdx.x + 1) & 511] == 77 && ticks < 0) *sink = 1;
}

extern "C" int pvrl_probe_cu_hog(int blocks, double us, int* sink, void* stream) {
  if (blocks <= 0) return 0;
  hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const long ticks = (long)(us * 100.0);      // wall_clock64: 100 MHz constant clock on gfx9
  hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(512), 128 * 1024, (hipStream_t)stream, ticks, sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
