// How fast can a CU pull L2-resident bytes into LDS with LDS-DMA (global_load_lds_dwordx4), against ordinary 16-byte loads into
// registers?  One workgroup per CU, NW waves, every wave loops over a 256 KiB window (L2 / L1 resident after the first pass).
//   hipcc --offload-arch=gfx950 -O3 tools/repro/ldsdma_rate.hip -o /tmp/ldsdma_rate && /tmp/ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: LDS-DMA, 1: register loads, 2: DMA + ds_read of what arrived, 4: ds_read_b128 only (mode 3, loads and LDS reads interleaved through one dependent accumulator, measured a latency chain and was dropped)
__global__ __launch_bounds__(512) void pull(const char* __restrict__ src, int iters, int window, unsigned long long* cyc, float* sink, char* __restrict__ big) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = src + (size_t)(blockIdx.x % 8) * window;  // 8 windows: the workgroups of an XCD share one
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int off = ((it * 8 + u) * nw + wave) * 1024 % window;
      if (MODE == 6 || MODE == 7 || MODE == 8) {  // 1 KiB stores: 6 nt / 7 plain, streaming through a big buffer; 8 nt into the window
        char* dstp = MODE == 8 ? (char*)base + off
                               : big + ((size_t)blockIdx.x * (size_t)iters * 8 * nw + (size_t)(it * 8 + u) * nw + wave) * 1024;
        const f32x4 v = {1.f, 2.f, 3.f, (float)it};
        if (MODE == 7) *(f32x4*)(dstp + lane * 16) = v;
        else __builtin_nontemporal_store(v, (f32x4*)(dstp + lane * 16));
      } else if (MODE == 5) {  // buffer form: SGPR resource + 32-bit lane offset
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, window, 0x00020000);
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + lane * 16, 0, 0);
        acc += __builtin_bit_cast(f32x4, v);
      } else if (MODE == 1 || MODE == 3) {
        const f32x4 v = *(const f32x4*)(base + off + lane * 16);
        acc += v;
        if (MODE == 3) acc += *(const volatile f32x4*)(lds + ((wave * 8 + u) * 1024 + lane * 16 + it * 16) % 65536);
      } else if (MODE == 4) {
        acc += *(const volatile f32x4*)(lds + ((wave * 8 + u) * 1024 + lane * 16 + it * 16) % 65536);
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds + (wave * 8 + u) * 1024), 16, 0, 0);
      }
    }
    if (MODE == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += *(const f32x4*)(lds + (wave * 8 + u) * 1024 + lane * 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[0] = acc[0];
  if (lds[threadIdx.x] == 77) sink[1] = 1.f;
}

int main() {
  const int window = 256 << 10, iters = 2000;
  char* big;
  hipMalloc(&big, (size_t)256 * iters * 8 * 8 * 1024);  // 32 GiB at most: one fresh KiB per store instruction
  char* src;
  hipMalloc(&src, 8 * window);
  hipMemset(src, 1, 8 * window);
  unsigned long long* cyc;
  float* sink;
  hipMalloc(&cyc, 256 * 8);
  hipMalloc(&sink, 8);
  for (int nw : {1, 2, 4, 8}) {
    for (int mode : {0, 1, 2, 4, 5, 6, 7, 8}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float ms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(pull<0>, dim3(256), dim3(64 * nw), 65536, 0, src, iters, window, cyc, sink, big);
        if (mode == 1) hipLaunchKernelGGL(pull<1>, dim3(256), dim3(64 * nw), 65536, 0, src, iters, window, cyc, sink, big);
        if (mode == 2) hipLaunchKernelGGL(pull<2>, dim3(256), dim3(64 * nw), 65536, 0, src, iters, window, cyc, sink, big);
        if (mode == 3) hipLaunchKernelGGL(pull<3>, dim3(256), dim3(64 * nw), 65536, 0, src, iters, window, cyc, sink, big);
        if (mode == 4) hipLaunchKernelGGL(pull<4>, dim3(256), dim3(64 * nw), 65536, 0, src, iters, window, cyc, sink, big);
        if (mode == 6) hipLaunchKernelGGL(pull<6>, dim3(256), dim3(64 * nw), 65536, 0, src, iters, window, cyc, sink, big);
        if (mode == 7) hipLaunchKernelGGL(pull<7>, dim3(256), dim3(64 * nw), 65536, 0, src, iters, window, cyc, sink, big);
        if (mode == 8) hipLaunchKernelGGL(pull<8>, dim3(256), dim3(64 * nw), 65536, 0, src, iters, window, cyc, sink, big);
        if (mode == 5) hipLaunchKernelGGL(pull<5>, dim3(256), dim3(64 * nw), 65536, 0, src, iters, window, cyc, sink, big);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<unsigned long long> h(256);
      hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      double avg = 0;
      for (auto c : h) avg += (double)c;
      avg /= 256;
      // (what s_memtime counts is calibrated against the launch time measured with HIP events)
      const double bytes = (double)iters * 8 * nw * 1024;
      printf("waves %d  mode %s: launch %.3f ms (%.0f ticks: %.2f GHz tick rate), %.1f B per tick and CU, %.2f TB/s over 256 CUs = %.1f B/clk/CU at 2.4 GHz\n",
             nw, mode == 0 ? "LDS-DMA        " : mode == 1 ? "register loads " : mode == 2 ? "LDS-DMA+ds_read" : mode == 3 ? "reg loads+ds_rd" : mode == 4 ? "ds_read only   " : mode == 5 ? "buffer loads   " : mode == 6 ? "nt stores HBM  " : mode == 7 ? "plain stores HBM" : "nt stores in L2", ms, avg, avg / (ms * 1e6), bytes / avg,
             bytes * 256 / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 2.4e9);
    }
  }
  return 0;
}
