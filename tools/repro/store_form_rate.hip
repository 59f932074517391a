// Does the ADDRESS FORM of a 1 KiB wave store change what a CU's store path sustains?  (round 6)
// tools/repro/ldsdma_rate.hip found 1 KiB stores into an L2-resident window saturating at ~22 B/clk/CU from two waves on, against
// 43-48 for loads.  One reading: a store moves 4 data dwords + 2 address dwords per lane to the texture-address unit; if that
// transfer is the limit, fewer address dwords per lane raise the rate.  Modes (all: 16 B per lane, 1 KiB contiguous per wave
// instruction, nt, into an L2-resident 256 KiB window per XCD):
//   0  global_store_dwordx4  v[addr 64-bit]                          (what csrc/field16.hip issues: "V, V, off nt")
//   1  global_store_dwordx4  v_off32, s[base]                         (saddr form: one address dword per lane)
//   2  buffer_store_dwordx4  v_off32, s[rsrc]  offen                  (raw buffer, one address dword per lane)
//   3  buffer_store_dwordx4  off, s[rsrc], s_off  + ADD_TID_ENABLE    (no address VGPR at all: lane * 16 added by the descriptor)
//   4  mode 0 as 2 x global_store_dwordx2 per 1 KiB                   (same bytes in twice the instructions)
//   5  mode 3 with every wave also issuing one 32x32x16 f16 MFMA per store (does matrix work hide under the store issue?)
//   hipcc --offload-arch=gfx950 -O3 tools/repro/store_form_rate.hip -o /tmp/store_form_rate && /tmp/store_form_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void push(char* __restrict__ dst, int iters, int window, unsigned long long* cyc, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  char* base = dst + (size_t)(blockIdx.x % 8) * window;
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  f32x16 acc = {0};
  const h8 ha = {1, 1, 1, 1, 1, 1, 1, 1};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, window, 0x00020000);
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)base, 16, window, 0x00800000);  // stride 16, ADD_TID_ENABLE (bit 23); DATA_FORMAT must be 0: with add_tid its bits extend the stride
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int off = __builtin_amdgcn_readfirstlane(((it * 8 + u) * nw + wave) * 1024 % window);
      if (MODE == 0) {
        __builtin_nontemporal_store(v, (f32x4*)(base + off + lane * 16));
      } else if (MODE == 1) {
        const int vo = off + lane * 16;
        asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(vo), "v"(v), "s"(base) : "memory");
      } else if (MODE == 2) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rs, off + lane * 16, 0, 2 /* nt */);
      } else if (MODE == 3 || MODE == 5) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rt, 0, off, 2);
        if (MODE == 5) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, ha, acc, 0, 0, 0);
      } else if (MODE == 4) {
        const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
        __builtin_nontemporal_store(a, (f32x2*)(base + off + lane * 8));
        __builtin_nontemporal_store(b, (f32x2*)(base + off + 512 + lane * 8));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc[0] == 12345.f) sink[0] = acc[0];
}

int main() {
  const int window = 256 << 10, iters = 2000;
  char* dst;
  hipMalloc(&dst, 8 * window + (256 << 20));  // (slack behind the windows: a mis-built descriptor then corrupts nothing and the check below reports it)
  hipMemset(dst, 0, 8 * window);
  unsigned long long* cyc;
  float* sink;
  hipMalloc(&cyc, 256 * 8);
  hipMalloc(&sink, 8);
  const char* names[] = {"global 64-bit vaddr", "global saddr+voff32", "buffer offen       ", "buffer add_tid     ", "2 x dwordx2 vaddr  ", "add_tid + 1 MFMA   "};
  for (int nw : {1, 2, 4, 8}) {
    for (int mode = 0; mode < 6; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float ms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(push<0>, dim3(256), dim3(64 * nw), 0, 0, dst, iters, window, cyc, sink);
        if (mode == 1) hipLaunchKernelGGL(push<1>, dim3(256), dim3(64 * nw), 0, 0, dst, iters, window, cyc, sink);
        if (mode == 2) hipLaunchKernelGGL(push<2>, dim3(256), dim3(64 * nw), 0, 0, dst, iters, window, cyc, sink);
        if (mode == 3) hipLaunchKernelGGL(push<3>, dim3(256), dim3(64 * nw), 0, 0, dst, iters, window, cyc, sink);
        if (mode == 4) hipLaunchKernelGGL(push<4>, dim3(256), dim3(64 * nw), 0, 0, dst, iters, window, cyc, sink);
        if (mode == 5) hipLaunchKernelGGL(push<5>, dim3(256), dim3(64 * nw), 0, 0, dst, iters, window, cyc, sink);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double bytes = (double)iters * 8 * nw * 1024;
      printf("waves %d  %s: launch %.3f ms, %.2f TB/s over 256 CUs = %.1f B/clk/CU at 2.4 GHz (%.0f cycles per 1 KiB and CU)\n", nw, names[mode], ms,
             bytes * 256 / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 2.4e9, 1024.0 / (bytes / (ms * 1e-3) / 2.4e9));
    }
  }
  // correctness of the add_tid form: lane l of the last store wrote {1, 2, 3, l} at window offset ... + 16 l
  std::vector<float> h(window / 4);
  hipMemcpy(h.data(), dst, window, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < window / 16; ++i) bad += !(h[4 * i] == 1.f && h[4 * i + 1] == 2.f && h[4 * i + 2] == 3.f && h[4 * i + 3] == (float)(i % 64));
  printf("window check after the last mode: %d of %d 16-byte pieces unexpected\n", bad, window / 16);
  return 0;
}
