// Round 6 probe of VERDICT r5 item 1 / DESIGN.md 9.1: the f16x3 trunk as ONE 128-sample, eight-wave workgroup per CU whose weight
// slabs reach both 64-sample halves through an LDS ring filled by LDS-DMA -- against the shipped structure (two independent
// 64-sample, four-wave workgroups per CU, weight fragments L2 / L1 -> registers).  Trunk only: L identical 256 x 256 layers, the
// shipped epilogue (bias, ReLU, sign bits, tile maximum through LDS, exponent, hi / lo plane write, two barriers), fp32 activation
// stores of every layer (1 KB per sample and layer, whole lines from the planes).  No encoding, no heads, no parity claim: what is
// measured is cycles per trunk layer and the launch time of the same work in four structures:
//   mode 0  shipped structure: 64 rows x 4 waves, mma16_lds (weights two k-blocks ahead in registers), stores behind the K loop
//   mode 1  shipped structure, stores inside the next K loop, one 1 KiB piece per k-block
//   mode 2  pair: 128 rows x 8 waves, 3 x 8 KiB LDS ring (unit = one plane of one k-block, all eight waves consume it), one
//           s_barrier per unit, weight fragments LDS -> registers one unit ahead; stores behind the K loop
//   mode 3  pair + stores inside the K loop
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iupnerf_amd/csrc tools/repro/pair_trunk_probe.hip -o tools/repro/pair_trunk_probe
#include "common16.cuh"
#include <cstdio>
#include <vector>

namespace {

constexpr int W = 256, HALF = 64;
__device__ unsigned long long probe_acc[8];

#define PSTAMP(i)                                               \
  do {                                                          \
    const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    t_acc[i] += _t - t_prev;                                    \
    t_prev = _t;                                                \
  } while (0)

// one 1 KiB piece (four rows of a 64-row half) of the planes -> fp32 rows in global memory (field16.hip:PlaneStore)
__device__ __forceinline__ void store_piece(const char* Ph, const char* Pl, float* __restrict__ dst, float un, int htid, int it) {
  int t = htid;
  asm volatile("" : "+v"(t));
  const int idx = t + it * 256, row = idx >> 6, g = idx & 63;
  const int o = poff<W>(row, 4 * g);
  const u32x2_t h = *(const u32x2_t*)(Ph + o), l = *(const u32x2_t*)(Pl + o);
  const f32x4 v = {mix16<0>(h[0], un, mix16<0>(l[0], un, 0.f)), mix16<1>(h[0], un, mix16<1>(l[0], un, 0.f)),
                   mix16<0>(h[1], un, mix16<0>(l[1], un, 0.f)), mix16<1>(h[1], un, mix16<1>(l[1], un, 0.f))};
  __builtin_nontemporal_store(v, (f32x4*)&dst[(size_t)row * W + 4 * g]);
}
struct LoopStore {
  const char *Ph, *Pl;
  float* dst;
  float un;
  int htid;
  __device__ __forceinline__ void operator()(int it) const { store_piece(Ph, Pl, dst, un, htid, it); }
};
// the same with the LDS words of piece t read one k-block ahead of their conversion and store (mode 12): the store's LDS round trip
// passes under the previous k-block's MFMAs instead of standing in front of this one's
struct LoopStorePipe {
  const char *Ph, *Pl;
  float* dst;
  float un;
  int htid;
  u32x2_t h, l;
  __device__ __forceinline__ void read(int it) {
    int t = htid;
    asm volatile("" : "+v"(t));
    const int idx = t + it * 256, row = idx >> 6, g = idx & 63;
    const int o = poff<W>(row, 4 * g);
    h = *(const u32x2_t*)(Ph + o);
    l = *(const u32x2_t*)(Pl + o);
  }
  __device__ __forceinline__ void flush(int it) const {
    int t = htid;
    asm volatile("" : "+v"(t));
    const int idx = t + it * 256, row = idx >> 6, g = idx & 63;
    const f32x4 v = {mix16<0>(h[0], un, mix16<0>(l[0], un, 0.f)), mix16<1>(h[0], un, mix16<1>(l[0], un, 0.f)),
                     mix16<0>(h[1], un, mix16<0>(l[1], un, 0.f)), mix16<1>(h[1], un, mix16<1>(l[1], un, 0.f))};
    __builtin_nontemporal_store(v, (f32x4*)&dst[(size_t)row * W + 4 * g]);
  }
  __device__ __forceinline__ void operator()(int it) {
    if (it > 0) flush(it - 1);
    read(it);
  }
};

// the shipped epilogue of a trunk layer for one 64-row half (4 waves): returns the new exponent; planes rewritten
template <int NBAR>
__device__ __forceinline__ int epilogue(f32x16 (&acc)[2][2], char* Ph, char* Pl, const float* bias_s, float* smax, int ecur, int wel,
                                        int wv, int lane, unsigned long long* hmask_dst, unsigned long long (&t_acc)[8],
                                        unsigned long long& t_prev) {
  const int hh = lane >> 5, n0 = 64 * wv;
  f32x4 bl[2][4];
  load_cols(bl, bias_s, n0, hh);
  const unsigned long long bits = acc_fma_relu_pack(acc, ldexpf(1.0f, -(ecur + wel)), bl);
  const float wm = acc_absmax(acc);
  if (lane == 0) smax[wv] = wm;
  __builtin_nontemporal_store(bits, hmask_dst);
  PSTAMP(2);
  __syncthreads();
  PSTAMP(3);
  const float mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  const int e = scale_exp(mx);
  acc_to_planes<2, W>(acc, Ph, Pl, 0, n0, 0, e, lane);
  PSTAMP(4);
  __syncthreads();
  PSTAMP(5);
  return e;
}

__device__ __forceinline__ void fill_planes(char* Ph, char* Pl, int htid, int seed) {
  for (int i = htid; i < HALF * W / 4; i += 256) {
    const int row = i >> 6, g = i & 63;
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (float)(((row * 131 + (4 * g + c) * 71 + seed * 13) % 97) - 30) * (1.0f / 64.0f);
    h4 hi, lo;
    split_quad<2>(ldexpf(fmaxf(v[0], 0.f), 13), ldexpf(fmaxf(v[1], 0.f), 13), ldexpf(fmaxf(v[2], 0.f), 13), ldexpf(fmaxf(v[3], 0.f), 13), hi, lo);
    const int o = poff<W>(row, 4 * g);
    *(h4*)(Ph + o) = hi;
    *(h4*)(Pl + o) = lo;
  }
}

// ---- modes 0 / 1: the shipped structure ------------------------------------------------------------------------------------------
template <bool INLOOP, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void trunk_single(const char* __restrict__ P16, const float* __restrict__ bias, const int* __restrict__ wexp,
                                                       float* __restrict__ h, unsigned long long* __restrict__ hmask, int M, int L) {
  __shared__ __attribute__((aligned(16))) char planes[2 * HALF * W * 2];
  __shared__ float smax[4];
  __shared__ __attribute__((aligned(16))) float bias_s[8 * W];
  char *Ph = planes, *Pl = planes + HALF * W * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * HALF;
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
  for (int l = 0; l < L && l < 8; ++l) bias_s[l * W + tid] = bias[l * W + tid];
  fill_planes(Ph, Pl, tid, blockIdx.x);
  __syncthreads();
  int ecur = 13;
  for (int l = 0; l < L; ++l) {
    PSTAMP(0);
    f32x16 acc[2][2];
    acc_zero(acc);
    const int wel = __builtin_amdgcn_readfirstlane(wexp[l & 7]);
    const char* Wl = P16 + (size_t)(l & 7) * (W * W * 4);
    float* dst = h + ((size_t)((l + L - 1) % L) * M + m0) * W;  // h_{l-1} (layer 0 stores the filled planes as "h_{L-1}")
    if (INLOOP && PIPE) {
      LoopStorePipe ps{Ph, Pl, dst, ldexpf(1.0f, -ecur), tid};
      mma16_lds<2, W, W / 16, 2>(acc, Ph, Pl, 0, 0, Wl, W / 16, 64 * wave, 0, lane, ps);
      ps.flush(15);
    } else if (INLOOP) {
      LoopStore ps{Ph, Pl, dst, ldexpf(1.0f, -ecur), tid};
      mma16_lds<2, W, W / 16, 2>(acc, Ph, Pl, 0, 0, Wl, W / 16, 64 * wave, 0, lane, ps);
    } else {
      mma16_lds<2, W, W / 16, 2>(acc, Ph, Pl, 0, 0, Wl, W / 16, 64 * wave, 0, lane);
    }
    PSTAMP(1);
    if (!INLOOP) {
#pragma unroll 4
      for (int it = 0; it < 16; ++it) store_piece(Ph, Pl, dst, ldexpf(1.0f, -ecur), tid, it);
    }
    ecur = epilogue<0>(acc, Ph, Pl, bias_s + (l & 7) * W, smax, ecur, wel, wave, lane,
                       hmask + ((size_t)l * gridDim.x + blockIdx.x) * 256 + tid, t_acc, t_prev);
  }
  if (lane == 0 && (blockIdx.x & 15) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&probe_acc[i], t_acc[i]);
}

// ---- modes 2 / 3: the pair ---------------------------------------------------------------------------------------------------------
// LDS: [half 0: hi, lo][half 1: hi, lo] planes 128 KiB | ring 3 x 8 KiB | bias row 1 KiB | maxima
constexpr int RING_SLOT = 8192, PAIR_LDS = 4 * HALF * W * 2 + 3 * RING_SLOT + W * 4 + 64;

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
}

// EXP (elimination switches, wrong synchronisation by construction -- timing only): 1 = no barrier at the odd units, 2 = no DMA (the
// ring is read as it stands), 4 = no barrier at any unit, 8 = every wave at a raised priority while it issues its MFMAs
template <bool INLOOP, int EXP = 0>
__global__ __launch_bounds__(512, 2) void trunk_pair(const char* __restrict__ P16, const float* __restrict__ bias, const int* __restrict__ wexp,
                                                     float* __restrict__ h, unsigned long long* __restrict__ hmask, int M, int L) {
  __shared__ __attribute__((aligned(16))) char lds[PAIR_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, wv = wave & 3, htid = tid & 255;
  char* Ph = lds + half * (2 * HALF * W * 2);
  char* Pl = Ph + HALF * W * 2;
  char* ring = lds + 4 * HALF * W * 2;
  float* bias_s = (float*)(ring + 3 * RING_SLOT);
  float* smax = bias_s + W + 4 * half;
  const int tile64 = 2 * blockIdx.x + half, ntile64 = 2 * gridDim.x;
  const int m0 = tile64 * HALF;
  const int li = lane & 31, lh = lane >> 5;
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
  fill_planes(Ph, Pl, htid, tile64);
  // the weight stream: unit g = 32 l + 2 t + p (layer, k-block, plane) = the eight 1 KiB fragments [ntile 0..7] of that plane and
  // k-block; wave w copies fragment w; slot g % 3.  Past the last unit the DMA re-reads the last one (the waits count instructions).
  const int G = 32 * L;
  int s0 = 0;  // slot of the first unit of the current layer: (32 l) % 3
  auto slot_of = [&](int k) {  // slot of unit 32 l + k, k a compile-time offset: no run-time division in the loop
    int s = s0 + (k % 3);
    return s >= 3 ? s - 3 : s;
  };
  auto dma = [&](int g, int k) {
    const int gc = g < G ? g : G - 1;
    const char* src = P16 + (size_t)((gc >> 5) & 7) * (W * W * 4) + (size_t)wave * 32768 + (size_t)(gc & 31) * 1024 + lane * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(ring + slot_of(k) * RING_SLOT + wave * 1024), 16, 0, 0);
  };
  h8 wr[2][2], xh[2][2], xl[2][2];  // [set][nt] one plane of one k-block; [set][mt]
  auto ldw = [&](int k) {           // this wave's two n-tiles of unit 32 l + k
    const char* s = ring + slot_of(k) * RING_SLOT + (2 * wv) * 1024 + lane * 16;
    wr[k & 1][0] = *(const h8*)s;
    wr[k & 1][1] = *(const h8*)(s + 1024);
  };
  auto ldx = [&](int t) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int o = poff<W>(32 * mt + li, 16 * t + 8 * lh);
      xh[t & 1][mt] = *(const h8*)(Ph + o);
      xl[t & 1][mt] = *(const h8*)(Pl + o);
    }
  };
  dma(0, 0);
  dma(1, 1);
  dma(2, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ldw(0);
  int ecur = 13;
  for (int l = 0; l < L; ++l) {
    PSTAMP(0);
    f32x16 acc[2][2];
    acc_zero(acc);
    const int wel = __builtin_amdgcn_readfirstlane(wexp[l & 7]);
    const float bv = bias[(l & 7) * W + htid];  // (retired by the vmcnt(0) of the first unit)
    float* dst = h + ((size_t)((l + L - 1) % L) * M + m0) * W;
    const float un = ldexpf(1.0f, -ecur);
    ldx(0);
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int g = 32 * l + u, t = u >> 1;
      // top of unit u: my part of unit g + 1 has landed, my reads of unit g are done; after the barrier everybody's have
      if (u == 0) wait_vm<0>();
      else if (!(EXP & 2)) wait_vm<INLOOP ? 2 : 1>();
      else if (INLOOP) wait_vm<1>();
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      if (!(EXP & 4) && !((EXP & 1) && (u & 1))) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(EXP & 2)) dma(g + 3, u + 3);   // into the slot unit g has just left
      if (INLOOP && (u & 1) == 0) store_piece(Ph, Pl, dst, un, htid, t);
      if (u == 0 && half == 0) bias_s[htid] = bv;  // (read behind this layer's K loop)
      ldw(u + 1);                          // (the first unit of the next layer at u = 31; past the end: a slot nobody fills again)
      if ((u & 1) == 0 && t + 1 < 16) ldx(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (EXP & 8) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[g & 1][nt], xh[t & 1][mt], acc[mt][nt], 0, 0, 0);
          if ((u & 1) == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[g & 1][nt], xl[t & 1][mt], acc[mt][nt], 0, 0, 0);
        }
      if (EXP & 8) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    PSTAMP(1);
    if (!INLOOP) {
#pragma unroll 4
      for (int it = 0; it < 16; ++it) store_piece(Ph, Pl, dst, un, htid, it);
    }
    ecur = epilogue<0>(acc, Ph, Pl, bias_s, smax, ecur, wel, wv, lane, hmask + ((size_t)l * ntile64 + tile64) * 256 + htid, t_acc, t_prev);
    s0 = s0 + 2 >= 3 ? s0 - 1 : s0 + 2;  // 32 % 3 == 2
  }
  if (lane == 0 && (blockIdx.x & 7) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&probe_acc[i], t_acc[i]);
}

}  // namespace

int main(int argc, char** argv) {
  const int R = 4096, S = argc > 1 ? atoi(argv[1]) : 192, L = 8, M = R * S;
  std::vector<_Float16> w((size_t)8 * W * W * 2);
  for (size_t i = 0; i < w.size(); ++i) w[i] = (_Float16)((float)((int)((i * 2654435761u) >> 20 & 1023) - 512) * ((i / 512) % 2 ? 0.004f : 1.0f));
  std::vector<float> b(8 * W);
  for (int i = 0; i < 8 * W; ++i) b[i] = 0.01f * (float)((i * 37) % 41 - 20);
  std::vector<int> we(8, 14);
  char* P16;
  float *bias, *h;
  int* wexp;
  unsigned long long* hm;
  (void)hipMalloc(&P16, w.size() * 2);
  (void)hipMalloc(&bias, b.size() * 4);
  (void)hipMalloc(&wexp, 32);
  (void)hipMalloc(&h, (size_t)L * M * W * 4);
  (void)hipMalloc(&hm, (size_t)L * (M / 64) * 256 * 8);
  (void)hipMemcpy(P16, w.data(), w.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(bias, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(wexp, we.data(), 32, hipMemcpyHostToDevice);
  const char* names[] = {"shipped structure, stores behind the K loop", "shipped structure, stores inside the K loop",
                         "pair + LDS ring, stores behind the K loop  ", "pair + LDS ring, stores inside the K loop  ",
                         "pair, stores behind; NO barrier at odd units", "pair, stores inside; NO barrier at odd units",
                         "pair, stores behind; NO DMA                 ", "pair, stores inside; NO DMA                 ",
                         "pair, stores behind; NO unit barriers       ", "pair, stores inside; NO unit barriers       ",
                         "pair, stores behind; MFMAs at raised priority", "pair, stores inside; MFMAs at raised priority",
                         "shipped structure, stores inside the K loop, LDS words one k-block ahead"};
  const int NMODE = 13;
  printf("trunk probe: %d samples, %d layers of 256 x 256 (f16x3), fp32 activation stores of every layer\n", M, L);
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < NMODE; ++mode) {
      unsigned long long z[8] = {0};
      (void)hipMemcpyToSymbol(HIP_SYMBOL(probe_acc), z, sizeof(z));
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, 0);
      if (mode == 0) hipLaunchKernelGGL(trunk_single<false>, dim3(M / 64), dim3(256), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 1) hipLaunchKernelGGL(trunk_single<true>, dim3(M / 64), dim3(256), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 2) hipLaunchKernelGGL(trunk_pair<false>, dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 3) hipLaunchKernelGGL(trunk_pair<true>, dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 4) hipLaunchKernelGGL((trunk_pair<false, 1>), dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 5) hipLaunchKernelGGL((trunk_pair<true, 1>), dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 6) hipLaunchKernelGGL((trunk_pair<false, 2>), dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 7) hipLaunchKernelGGL((trunk_pair<true, 2>), dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 8) hipLaunchKernelGGL((trunk_pair<false, 4>), dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 9) hipLaunchKernelGGL((trunk_pair<true, 4>), dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 10) hipLaunchKernelGGL((trunk_pair<false, 8>), dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 12) hipLaunchKernelGGL((trunk_single<true, true>), dim3(M / 64), dim3(256), 0, 0, P16, bias, wexp, h, hm, M, L);
      if (mode == 11) hipLaunchKernelGGL((trunk_pair<true, 8>), dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      const hipError_t err = hipGetLastError();
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long a[8];
      (void)hipMemcpyFromSymbol(a, HIP_SYMBOL(probe_acc), sizeof(a));
      const double waves = ((mode < 2 || mode == 12) ? 4.0 * (M / 64 / 16) : 8.0 * (M / 128 / 8)) * (double)L;  // reporting waves x layers
      if (rep == 1)
        printf("mode %2d  %s: %.3f ms%s  per wave and layer (ticks of s_memtime): top %.0f  K loop %.0f  epilogue+stores %.0f  barrier %.0f  plane write %.0f  barrier %.0f  sum %.0f\n",
               mode, names[mode], ms, err == hipSuccess ? "" : " [LAUNCH ERROR]", a[0] / waves, a[1] / waves, a[2] / waves, a[3] / waves, a[4] / waves, a[5] / waves,
               (a[0] + a[1] + a[2] + a[3] + a[4] + a[5]) / waves);
    }
  // the four structures compute the same thing up to summation order: compare h of the last run (mode 3) with mode 0
  hipLaunchKernelGGL(trunk_pair<true>, dim3(M / 128), dim3(512), 0, 0, P16, bias, wexp, h, hm, M, L);
  (void)hipDeviceSynchronize();
  std::vector<float> h3((size_t)2 * 64 * W), h0((size_t)2 * 64 * W);
  (void)hipMemcpy(h3.data(), h + (size_t)3 * M * W, h3.size() * 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(trunk_single<false>, dim3(M / 64), dim3(256), 0, 0, P16, bias, wexp, h, hm, M, L);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h0.data(), h + (size_t)3 * M * W, h0.size() * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (size_t i = 0; i < h0.size(); ++i) {
    num = fmax(num, fabs((double)h3[i] - h0[i]));
    den = fmax(den, fabs((double)h0[i]));
  }
  printf("h_3 of the first 128 samples, pair + in-loop stores vs shipped structure: max |diff| %.3g of max %.3g\n", num, den);
  return 0;
}
