// Diagnostic only (never linked into libupnerf_hip.so): leave a chosen bit pattern in the LDS of every CU, so that a kernel
// that reads LDS it has not written shows up as a change of its results (tools/uninit_probe.py --lds).
#include <hip/hip_runtime.h>
#include <stdint.h>

extern __shared__ uint32_t lds[];

__global__ __launch_bounds__(256) void poison_lds_kernel(uint32_t pattern, int words, uint32_t* sink) {
  for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern;
  __syncthreads();
  uint32_t x = 0;
  for (int i = threadIdx.x; i < words; i += 256 * 64) x ^= lds[i];
  if (x == 0x12345678u && sink) sink[0] = x;  // keeps the stores alive
}

extern "C" int poison_lds(uint32_t pattern, void* stream) {
  const int bytes = 160 * 1024;
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
    set = true;
  }
  hipLaunchKernelGGL(poison_lds_kernel, dim3(256 * 16), dim3(256), bytes, (hipStream_t)stream, pattern, bytes / 4, (uint32_t*)nullptr);
  return (int)hipGetLastError();
}
