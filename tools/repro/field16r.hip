// Register-resident forward pass of the fused NeRF field on the f16 matrix cores: the activations of a sample never leave
// the registers of the wave that owns it, the WEIGHTS are what streams through LDS -- once per workgroup, shared by its
// waves (BASELINE.json north_star: "LDS staging of MLP weights").
//
// Why (measured on csrc/field16.hip, stamps build, per wave and trunk layer): with the 64-sample tile in LDS and the
// weights streamed L2 -> registers by every wave for its own 64 output columns, a CU pulls 256 KB (f16x3) per 64 samples
// and layer through its L1; the K loop then runs at the ~32 B/clk a CU gets from L2 (16.5k cycles against 6.1k of MFMA
// work), whatever the prefetch depth.  Here a workgroup of 4 waves owns 128 samples (32 per wave) and the 256 KB of a layer
// enter the CU ONCE for all of them: half the bytes per sample, through LDS-DMA (no registers, no VALU).
//
// Reference behaviour: models/nerf.py:80-124 (NeRF.forward) + 126-147 (positional_encoding), as csrc/field16.hip.
// How the layers chain without LDS (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand"): the
// contraction is issued transposed, D^T[n][m] = sum_k W[n][k] X[m][k], so a lane holds ITS sample (column m = lane & 31) and,
// in the 16 registers of a 32 x 32 result, rows n = 8(r/4) + 4(lane/32) + r%4 of the feature tile.  Converted to fp16 in
// place, registers 0..7 / 8..15 ARE the B-operand fragments of k-blocks 2j / 2j+1 of the next layer, with the k order
// inside a block permuted to 8(j/4) + 4h + j%4 -- upnerf_frag16(perm = 1) writes the weights in that order.
//
// Exponents.  f16x3 carries value * 2^e as fp16 hi + lo; field16.hip picks e from the tile's exact maximum, which needs every
// output of the layer before any can be converted.  Here a layer's outputs are converted tile by tile as they complete (so the
// conversion overlaps the next tile's MFMAs), with an exponent known BEFORE the layer runs: |W x + b|_inf <= wnorm *
// |x|_inf + |b|_inf with wnorm = max_n sum_k |W[n][k]| from upnerf_frag16 and |x|_inf the wave's exact input maximum.  The
// bound is loose by the usual gap between the 1-norm bound and the attained maximum (a few bits): hi + lo still carries 22
// bits for everything within 2^-13 of the layer's maximum and an ABSOLUTE error of 2^-35 of that maximum below -- the same
// fp32-level accuracy (parity tests unchanged).
//
// Weight stream.  One slab = one 32-feature output tile of one matrix = (K/16) k-blocks x NP planes x 1 KiB, contiguous in
// the fragment buffer; the four waves DMA a quarter of its 1 KiB chunks each (global_load_lds_dwordx4) into a ring of three
// LDS slots, two slabs ahead of the MFMAs; per slab ONE counted s_waitcnt vmcnt + ONE s_barrier (no vmcnt(0) in the loop).
#include "common16.cuh"

#define R_THREADS 256
#define R_WAVES 4
#define R_TILE 128                                   // samples per workgroup, 32 per wave
#define R_NSLOT 3
#define R_MAXKB 21                                   // k-blocks of the widest matrix (rgb head: 256 + 80)
#define R_PE_LD 68                                   // floats per row of the encoding exchange scratch
// vectors staged in LDS (floats): trunk biases [8][256], final bias, rgb1 / cand1 / cand2 biases, w_sigma, w_rgb2 [3][128],
// w_csigma, then per-vector maxima |b|_inf [16]
#define R_V_BE 2048
#define R_V_BR1 2304
#define R_V_BC1 2432
#define R_V_BC2 2560
#define R_V_WSIG 2688
#define R_V_WR2 2944
#define R_V_WCSIG 3328
#define R_V_BMAX 3456
#define R_V_TOTAL 3472

// Diagnostic build only (-DUPNERF_STAMPS): shader-clock sums per phase of the slab loop, read with upnerf_stamps_read_r:
// 0 wait for the slab's DMA, 1 barrier, 2 DMA issue, 3 MFMA loop, 4 epilogue,
// 5 layer tail (masks, maxima, operand copy), 6 prologue (vectors, encoding), 7 heads outside the slab loop
#ifdef UPNERF_STAMPS
__device__ unsigned long long upnerf_stamp_acc_r[8];
#define RS_DECL unsigned long long _rs[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long _rp = __builtin_amdgcn_s_memtime()
#define RS(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); _rs[i] += _t - _rp; _rp = _t; } while (0)
#define RS_FLUSH do { if (lane == 0 && (blockIdx.x & 15) == 0) for (int _i = 0; _i < 8; ++_i) atomicAdd(&upnerf_stamp_acc_r[_i], _rs[_i]); } while (0)
#else
#define RS_DECL
#define RS(i)
#define RS_FLUSH
#endif

namespace {

template <int NP>
struct RCfg {
  static constexpr int SLOT = R_MAXKB * NP * 1024;                  // bytes per ring slot
  static constexpr int RING = R_NSLOT * SLOT;
  static constexpr int LDS = RING + R_V_TOTAL * 4;
};

__device__ __forceinline__ float pow2r(int n) { return ldexpf(1.0f, n); }

__device__ __forceinline__ float wave_max_r(float m) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  return m;
}

// s_waitcnt vmcnt(n), n wave-uniform (values above 31 wait for 31); expcnt / lgkmcnt untouched.  gfx9 encoding:
// vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
__device__ __forceinline__ void wait_vmcnt(int n) {
#define R_W(N) case N: __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14)); break;
  switch (n) {
    R_W(0) R_W(1) R_W(2) R_W(3) R_W(4) R_W(5) R_W(6) R_W(7) R_W(8) R_W(9) R_W(10) R_W(11) R_W(12) R_W(13) R_W(14) R_W(15)
    R_W(16) R_W(17) R_W(18) R_W(19) R_W(20) R_W(21) R_W(22) R_W(23) R_W(24) R_W(25) R_W(26) R_W(27) R_W(28) R_W(29) R_W(30)
    default: __builtin_amdgcn_s_waitcnt(0x0F70 | 15 | (1 << 14)); break;
  }
#undef R_W
}

// ---- the order in which the slabs of a pass are consumed (and prefetched) -------------------------------------------
// stages: 0..D-1 trunk layers, D final layer, D+1 rgb head layer 1, D+2 candidate head layer 1, D+3 candidate head layer 2
struct Slab {
  int stage, tile;
};
struct SeqCfg {
  int D, skip, last_stage;  // last_stage: D-1 (density only), D (final only), else D+3 with the flags below
  bool rgb, cand;
};
__device__ __forceinline__ int stage_kb(const SeqCfg& c, int st) {  // k-blocks (row length of the matrix / 16)
  if (st < c.D) return st == 0 ? UPNERF_X0 / 16 : (st == c.skip ? (UPNERF_X0 + 256) / 16 : 16);
  if (st == c.D) return 16;
  if (st == c.D + 1) return (256 + UPNERF_AUXK) / 16;
  if (st == c.D + 2) return (256 + UPNERF_CK) / 16;
  return 8;
}
__device__ __forceinline__ int stage_tiles(const SeqCfg& c, int st) { return st <= c.D ? 8 : 4; }
__device__ __forceinline__ bool stage_on(const SeqCfg& c, int st) {
  if (st <= c.D) return st <= c.last_stage;
  if (st > c.last_stage) return false;
  return st == c.D + 1 ? c.rgb : c.cand;
}
__device__ __forceinline__ Slab slab_next(const SeqCfg& c, Slab s) {
  if (s.stage < 0) return s;
  if (++s.tile < stage_tiles(c, s.stage)) return s;
  s.tile = 0;
  do {
    ++s.stage;
  } while (s.stage <= c.D + 3 && !stage_on(c, s.stage));
  if (s.stage > c.D + 3) s.stage = -1;
  return s;
}
__device__ __forceinline__ int stage_off(const upnerf_layout& L, const SeqCfg& c, int st) {  // float offset of the matrix
  if (st < c.D) return L.w[st];
  if (st == c.D) return L.we;
  if (st == c.D + 1) return L.wr1;
  if (st == c.D + 2) return L.wc1;
  return L.wc2;
}

// Staging of a slab: every wave moves a quarter of its 1 KiB chunks, global -> registers (issued one slab-time before the
// data is needed: L2 latency hides under ~1500 cycles of MFMAs) -> LDS ring slot (ds_write_b128).  LDS-DMA
// (global_load_lds) was built first and measured: ~150 cycles of ISSUE time per 1 KiB instruction on a wave that owns its
// SIMD alone (1 226 cycles per slab against 1 536 of MFMAs -- the loader cannot be another wave here, the register file is
// full), a plain load + LDS write costs a fraction of that.
#define R_STAGE 11  // chunks per wave of the largest slab: ceil(21 * 2 / 4)
template <int NP>
struct Stage {
  f32x4 r[R_STAGE];
  int n;  // chunks this wave holds (the same in all waves; indices past the end repeat the last chunk)
};
template <int NP>
__device__ __forceinline__ void slab_load(Stage<NP>& st, const upnerf_layout& L, const SeqCfg& c, Slab s,
                                          const char* __restrict__ P16, int wave, int lane) {
  st.n = 0;
  if (s.stage < 0) return;
  const int kb = stage_kb(c, s.stage), chunks = kb * NP, per = (chunks + R_WAVES - 1) / R_WAVES;
  const char* src = P16 + 4 * (size_t)stage_off(L, c, s.stage) + (size_t)s.tile * kb * 2048 + lane * 16;
  st.n = per;
#pragma unroll
  for (int q = 0; q < R_STAGE; ++q) {
    if (q < per) {
      int ch = wave + R_WAVES * q;
      ch = ch < chunks ? ch : chunks - 1;
      st.r[q] = *(const f32x4*)(src + (size_t)ch * (NP == 2 ? 1024 : 2048));  // f16: hi blocks only
    }
  }
}
template <int NP>
__device__ __forceinline__ void slab_store(const Stage<NP>& st, const SeqCfg& c, Slab s, char* ring, int slot, int wave, int lane) {
  if (s.stage < 0) return;
  const int chunks = stage_kb(c, s.stage) * NP;
  char* dst = ring + slot * RCfg<NP>::SLOT + lane * 16;
#pragma unroll
  for (int q = 0; q < R_STAGE; ++q) {
    if (q < st.n) {
      int ch = wave + R_WAVES * q;
      ch = ch < chunks ? ch : chunks - 1;
      *(f32x4*)(dst + ch * 1024) = st.r[q];
    }
  }
}

// acc (32 features x 32 samples, transposed) = slab k-blocks [0, T1) . o1  +  k-blocks [T1, T1 + T2) . o2
// One wave per SIMD: nothing hides a latency unless the instruction stream does.
//  * the weight fragments come from LDS through a ring R_PF k-blocks ahead of their MFMAs (ds_read latency ~130 cycles
//    against 96 / 32 matrix cycles per k-block);
//  * the MFMAs alternate between two accumulators (summed at the end): none waits for the result of the one issued just
//    before it;
//  * the loop runs in regions of four k-blocks (12 MFMAs); `fill(g)`, called once in region g < 4, is a quarter of the
//    PREVIOUS tile's epilogue (one register quad: bias, ReLU, sign bits, store, fp16 split), and sched_group_barrier lays
//    the region out as MFMA, a few vector instructions, MFMA, ... so that this vector work issues in the shadow of the
//    matrix pipe (an MFMA holds the issue port for 8 of its 32 cycles).
#ifndef R_PF
#define R_PF 2
#endif
#ifndef R_FILL_VALU
#define R_FILL_VALU 7
#endif
template <int NP, int T1, int T2, int N1, int N2, class F>
__device__ __forceinline__ void kloop(f32x16& acc, const char* slot, const h8 (&o1h)[N1], const h8 (&o1l)[N1],
                                      const h8 (&o2h)[N2], const h8 (&o2l)[N2], int lane, F fill) {
  static_assert(T1 <= N1 && T2 <= N2, "operand arrays");
  constexpr int T = T1 + T2, SETS = R_PF + 1, G = (T + 3) / 4;
  const char* p = slot + lane * 16;
  h8 ah[SETS], al[SETS];
  f32x16& acc0 = acc;
  f32x16 acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
#pragma unroll
  for (int t = 0; t < R_PF && t < T; ++t) {
    ah[t] = *(const h8*)(p + (t * NP) * 1024);
    if constexpr (NP == 2) al[t] = *(const h8*)(p + (t * NP + 1) * 1024);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int t = 4 * g; t < 4 * g + 4 && t < T; ++t) {
      if (t + R_PF < T) {
        ah[(t + R_PF) % SETS] = *(const h8*)(p + ((t + R_PF) * NP) * 1024);
        if constexpr (NP == 2) al[(t + R_PF) % SETS] = *(const h8*)(p + ((t + R_PF) * NP + 1) * 1024);
      }
      const h8 xh = t < T1 ? o1h[t < T1 ? t : 0] : o2h[t >= T1 ? t - T1 : 0];
      f32x16& a = (t & 1) ? acc1 : acc0;
      a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t % SETS], xh, a, 0, 0, 0);
      if constexpr (NP == 2) {
        const h8 xl = t < T1 ? o1l[t < T1 ? t : 0] : o2l[t >= T1 ? t - T1 : 0];
        f32x16& b = (t & 1) ? acc0 : acc1;  // the hi*lo terms of block t go to the OTHER accumulator: no back-to-back pairs
        b = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t % SETS], xh, b, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t % SETS], xl, a, 0, 0, 0);
      }
    }
    if (g < 4) fill(g);
    if (G == 1) {  // a single region (K = 64): the whole previous epilogue rides on it
      fill(1);
      fill(2);
      fill(3);
    }
    // layout of the region: the prefetch reads first, then MFMA / vector work interleaved
#pragma unroll
    for (int i = 0; i < 4 * (NP == 2 ? 3 : 1); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // one LDS read
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, R_FILL_VALU, 0);  // vector instructions in its shadow
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
}

// 8 fp32 values (natural units, operand order) -> one B-operand fragment at exponent e
template <int NP>
__device__ __forceinline__ void make_op(const float (&v)[8], int e, h8& hi, h8& lo) {
  h4 a, b, c, d;
  split_quad<NP>(ldexpf(v[0], e), ldexpf(v[1], e), ldexpf(v[2], e), ldexpf(v[3], e), a, c);
  split_quad<NP>(ldexpf(v[4], e), ldexpf(v[5], e), ldexpf(v[6], e), ldexpf(v[7], e), b, d);
  hi = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  if constexpr (NP == 2) lo = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
}

// operand fragment of k-block s of a row-major fp32 side input row (k order of the register chain: 8(j/4) + 4h + j%4)
template <int NP>
__device__ __forceinline__ void row_op(const float* __restrict__ row, int s, int hh, int e, h8& hi, h8& lo) {
  const f32x4 a = *(const f32x4*)(row + 16 * s + 4 * hh), b = *(const f32x4*)(row + 16 * s + 8 + 4 * hh);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  make_op<NP>(v, e, hi, lo);
}

enum { EP_RELU = 1, EP_MASK = 2, EP_CONV = 4 };

// Epilogue of one 32-feature tile j of a layer: v = act(fma(acc, un, bias)); optional sign bits, running maximum, fp32 store
// of the row-major tensor, conversion into the next operand fragments (k-blocks 2j, 2j+1) at exponent eo, and up to three
// dot products with LDS-staged vectors (density / colour outputs).
template <int NP, int FLAGS, int NDOT, int NB>
__device__ __forceinline__ void quad_epilogue(const f32x16& acc, int j, int q, float un, const float* bias_s /*LDS, this layer*/,
                                              unsigned long long& bits, float& vmax, float* __restrict__ dst /*row of this
                                              sample or nullptr*/, int eo, h8 (&nh)[NB], h8 (&nl)[NB],
                                              const float* dotw_s /*LDS [NDOT][ld]*/, int dot_ld, float (&dot)[3], int hh) {
  {
    const int col = 32 * j + 8 * q + 4 * hh;
    const f32x4 b = *(const f32x4*)&bias_s[col];
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = fmaf(acc[4 * q + u], un, b[u]);
      if (FLAGS & EP_RELU) v[u] = fmaxf(v[u], 0.0f);
      if (FLAGS & EP_MASK) bits |= (v[u] > 0.0f) ? (1ull << ((16 * j + 4 * q + u) & 63)) : 0ull;
    }
    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    if (dst) *(f32x4*)&dst[col] = f32x4{v[0], v[1], v[2], v[3]};
    if constexpr (NDOT > 0) {
#pragma unroll
      for (int c = 0; c < NDOT; ++c) {
        const f32x4 w = *(const f32x4*)&dotw_s[c * dot_ld + col];
        dot[c] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
      }
    }
    if constexpr ((FLAGS & EP_CONV) != 0) {
      h4 hi, lo;
      split_quad<NP>(ldexpf(v[0], eo), ldexpf(v[1], eo), ldexpf(v[2], eo), ldexpf(v[3], eo), hi, lo);
      const int blk = 2 * j + (q >> 1);
      if (q & 1) {
        nh[blk] = __builtin_shufflevector(nh[blk], __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 12, 13, 14, 15);
        if constexpr (NP == 2)
          nl[blk] = __builtin_shufflevector(nl[blk], __builtin_shufflevector(lo, lo, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 12, 13, 14, 15);
      } else {
        nh[blk] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3);
        if constexpr (NP == 2) nl[blk] = __builtin_shufflevector(lo, lo, 0, 1, 2, 3, 0, 1, 2, 3);
      }
    }
  }
}
template <int NP, int FLAGS, int NDOT, int NB>
__device__ __forceinline__ void tile_epilogue(const f32x16& acc, int j, float un, const float* bias_s, unsigned long long& bits,
                                              float& vmax, float* __restrict__ dst, int eo, h8 (&nh)[NB], h8 (&nl)[NB],
                                              const float* dotw_s, int dot_ld, float (&dot)[3], int hh) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    quad_epilogue<NP, FLAGS, NDOT>(acc, j, q, un, bias_s, bits, vmax, dst, eo, nh, nl, dotw_s, dot_ld, dot, hh);
}

// ------------------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(R_THREADS, 1) void field16r_fwd_kernel(upnerf_layout L, upnerf_field_fwd_args a) {
  constexpr int W = 256, W2 = 128;
  using C = RCfg<NP>;
  __shared__ __attribute__((aligned(16))) char lds[C::LDS];  // ONE object: [ring slots | staged vectors]
  char* ring = lds;
  float* vec_s = (float*)(lds + C::RING);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int S = a.S, M = a.R * a.S, D = L.D;
  const int m = blockIdx.x * R_TILE + 32 * wave + li;
  const bool valid = m < M;
  const int mc = valid ? m : M - 1;
  const int ray = mc / S;
  const float* __restrict__ P = a.P;
  const char* __restrict__ P16 = (const char*)a.P16;
  const int* __restrict__ wexp = a.wexp;
  const float* __restrict__ wnorm = a.wnorm;
  RS_DECL;

  SeqCfg sq;
  sq.D = D;
  sq.skip = L.skip;
  sq.rgb = a.use_rgb != 0;
  sq.cand = a.use_cand != 0;
  sq.last_stage = (!a.e && !sq.rgb && !sq.cand) ? D - 1 : ((!sq.rgb && !sq.cand) ? D : D + 3);

  // ---- stage the vectors (ordinary loads: all of them BEFORE the first DMA is in flight; one independent load per thread
  // and vector, no per-element selects -- hipcc turns those into a branch and a full wait per element)
  for (int l = 0; l < 8; ++l) vec_s[256 * l + tid] = l < D ? P[L.b[l < D ? l : 0] + tid] : 0.0f;
  vec_s[R_V_BE + tid] = P[L.be + tid];
  vec_s[R_V_WSIG + tid] = P[L.wsig + tid];
  if (tid < W2) {
    vec_s[R_V_BR1 + tid] = sq.rgb ? P[L.br1 + tid] : 0.0f;
    vec_s[R_V_BC1 + tid] = sq.cand ? P[L.bc1 + tid] : 0.0f;
    vec_s[R_V_BC2 + tid] = sq.cand ? P[L.bc2 + tid] : 0.0f;
    vec_s[R_V_WCSIG + tid] = sq.cand ? P[L.wcsig + tid] : 0.0f;
  }
  for (int idx = tid; idx < 3 * W2; idx += R_THREADS) vec_s[R_V_WR2 + idx] = sq.rgb ? P[L.wr2 + idx] : 0.0f;
  // ---- sample position (rendering.py:251 / 308) and its encoding (nerf.py:126-147): each lane half evaluates 15 of the 30
  // (coordinate, band) pairs of ITS sample once; the halves meet in an LDS scratch (the ring, not yet in use)
  h8 Xh[4], Xl[4];  // encoding operand (layer 0; rebuilt for the skip layer), at the exponent of the layer it enters
  float xm;
  {
    const float zz = a.z[mc];
    float xyz[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) xyz[n] = valid ? mul_then_add(a.rays_o[3 * ray + n], a.rays_d[3 * ray + n], zz) : 0.0f;
    xm = fmaxf(fmaxf(fabsf(xyz[0]), fabsf(xyz[1])), fmaxf(fabsf(xyz[2]), 1.0f));  // |sin|, |cos| <= 1
    float* pe = (float*)ring + (32 * wave + li) * R_PE_LD;
    const float* __restrict__ wkd = a.wk_xyz_dev;
    if (hh == 0) {
      pe[0] = xyz[0];
      pe[1] = xyz[1];
      pe[2] = xyz[2];
      pe[63] = 0.0f;
    }
#pragma unroll 1
    for (int p = 0; p < 15; ++p) {
      const int pp = 15 * hh + p, n = pp / 10, k = pp - 10 * n;
      const float xv = n == 0 ? xyz[0] : (n == 1 ? xyz[1] : xyz[2]);
      float sv, cv;
      sincos_f32_via_f64(xv * ldexpf(PI_F, k), sv, cv);
      const float wk = wkd ? wkd[k] : a.wk_xyz[k];
      pe[3 + 20 * n + k] = sv * wk;
      pe[3 + 20 * n + 10 + k] = cv * wk;
    }
  }
  __syncthreads();
  const float x0max = wave_max_r(xm);
  {
    // this lane's 32 encoding features in operand order (feature 16 s + 8 (j/4) + 4 hh + j%4): operand of layer 0 and the
    // row-major x0 tensor (skip layer, backward pass, weight gradients)
    const float* pe = (const float*)ring + (32 * wave + li) * R_PE_LD;
    const int e0 = scale_exp(x0max);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float x0v[8];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 v = *(const f32x4*)&pe[16 * s + 8 * g + 4 * hh];
#pragma unroll
        for (int u = 0; u < 4; ++u) x0v[4 * g + u] = v[u];
        if (valid) *(f32x4*)&a.x0[(size_t)m * UPNERF_X0 + 16 * s + 8 * g + 4 * hh] = v;
      }
      make_op<NP>(x0v, e0, Xh[s], Xl[s]);
    }
  }
  // per-vector maxima |b|_inf (bounds of the layer outputs), one wave per group of vectors
  if (wave == 0) {
    for (int l = 0; l < 12; ++l) {
      // 0..7 trunk, 8 final, 9 rgb1, 10 cand1, 11 cand2
      const int base = l < 8 ? 256 * l : (l == 8 ? R_V_BE : (l == 9 ? R_V_BR1 : (l == 10 ? R_V_BC1 : R_V_BC2)));
      const int n = l <= 8 ? 256 : 128;
      float mx = 0.0f;
      for (int i = lane; i < n; i += 64) mx = fmaxf(mx, fabsf(vec_s[base + i]));
      mx = wave_max_r(mx);
      if (lane == 0) vec_s[R_V_BMAX + l] = mx;
    }
  }
  if (a.amax && lane == 0) atomicMax((unsigned int*)(a.amax + D + 4), __float_as_uint(x0max));
  // largest magnitude of this wave's per-ray side inputs (they enter the heads at the exponent of e)
  float sidemax = 0.0f;
  if (sq.rgb) {
    const float* __restrict__ row = a.aux + (size_t)ray * UPNERF_AUXK;
    for (int i = 4 * hh; i < UPNERF_AUXK; i += 8) {
      const f32x4 v = *(const f32x4*)&row[i];
      sidemax = fmaxf(sidemax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
  }
  if (sq.cand) {
    const float* __restrict__ row = a.c_rows + (size_t)ray * UPNERF_CK;
    for (int i = 4 * hh; i < UPNERF_CK; i += 8) {
      const f32x4 v = *(const f32x4*)&row[i];
      sidemax = fmaxf(sidemax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
  }
  sidemax = wave_max_r(sidemax);
  __syncthreads();  // scratch reads done (the ring is free for the weight stream), vector maxima visible

  const float bsig = P[L.bsig];
  float br2[3] = {0.f, 0.f, 0.f}, bcsig = 0.0f;
  if (sq.rgb) {
#pragma unroll
    for (int c = 0; c < 3; ++c) br2[c] = P[L.br2 + c];
  }
  if (sq.cand) bcsig = P[L.bcsig];

  // ---- weight stream: the slab being consumed sits in ring slot `slot`, the next one is already in LDS, the one after it
  // sits in this wave's staging registers (its loads were issued one slab ago) and moves to LDS at the top of the slab
  Stage<NP> stg;
  Slab pre = {0, 0};
  slab_load<NP>(stg, L, sq, pre, P16, wave, lane);
  slab_store<NP>(stg, sq, pre, ring, 0, wave, lane);
  pre = slab_next(sq, pre);
  slab_load<NP>(stg, L, sq, pre, P16, wave, lane);
  slab_store<NP>(stg, sq, pre, ring, 1, wave, lane);
  pre = slab_next(sq, pre);
  slab_load<NP>(stg, L, sq, pre, P16, wave, lane);  // slab 2 stays in registers until the first slab_begin
  RS(6);
  int slot = 0;
  auto slab_begin = [&]() -> const char* {
    RS(4);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes of the previous top have landed
    __builtin_amdgcn_s_barrier();        // ... in every wave, and every wave is done reading the slot refilled below
    __builtin_amdgcn_sched_barrier(0);
    RS(1);
    int pslot = slot + 2;
    pslot = pslot >= R_NSLOT ? pslot - R_NSLOT : pslot;
    slab_store<NP>(stg, sq, pre, ring, pslot, wave, lane);
    pre = slab_next(sq, pre);
    slab_load<NP>(stg, L, sq, pre, P16, wave, lane);
    RS(2);
    return ring + slot * C::SLOT;
  };
  auto slab_end = [&](int) { slot = slot + 1 >= R_NSLOT ? 0 : slot + 1; };

  // ReLU sign bits go out in the layout field16.hip's backward kernel reads (one 64-bit word per thread of ITS 64-sample,
  // 4-wave tiling: wave w' there owns feature tiles 2w', 2w'+1 of rows {i, 32 + i}): this lane holds, for its row, the
  // 32-bit half (row group mt = this wave & 1) of the words of all four of those waves.
  const int nt64 = (M + 63) >> 6, t64 = m >> 6, mt64 = wave & 1;
  const bool mask_on = a.hmask != nullptr && t64 < nt64;
  unsigned int* __restrict__ hm32 = (unsigned int*)a.hmask;
  unsigned short* __restrict__ hm16 = (unsigned short*)a.hmask;
  auto mask_store = [&](int l, unsigned long long b0, unsigned long long b1) {
    if (!mask_on) return;
    const size_t base = ((size_t)l * nt64 + t64) * 256 + lane;  // word index of (tile, wave 0, lane)
    hm32[(base + 0) * 2 + mt64] = (unsigned int)b0;
    hm32[(base + 64) * 2 + mt64] = (unsigned int)(b0 >> 32);
    hm32[(base + 128) * 2 + mt64] = (unsigned int)b1;
    hm32[(base + 192) * 2 + mt64] = (unsigned int)(b1 >> 32);
  };

  h8 Bh[16], Bl[16];    // operand of the running layer (previous layer's outputs)
  h8 Nh[16], Nl[16];    // operand of the next layer, filled tile by tile
  float dot3[3] = {0.f, 0.f, 0.f};
  int e_in = scale_exp(x0max);  // exponent of the operand the running layer reads
  float amax_in = x0max;        // its exact largest magnitude in this wave

  // ---- trunk (nerf.py:84-87)
#pragma unroll 1
  for (int l = 0; l < D; ++l) {
    const bool is_skip = l == L.skip;
    if (is_skip) {
      amax_in = fmaxf(amax_in, x0max);
      // the encoding re-enters at THIS layer's operand exponent (e_in covers x0max: chosen below, one layer earlier); its
      // row was stored by this lane at the start of the kernel and has never been in this CU's L1
      const float* __restrict__ xrow = a.x0 + (size_t)mc * UPNERF_X0;
#pragma unroll
      for (int s = 0; s < 4; ++s) row_op<NP>(xrow, s, hh, e_in, Xh[s], Xl[s]);
    }
    const float un = pow2r(-(e_in + wexp[l]));
    float bound = wnorm[l] * amax_in + vec_s[R_V_BMAX + l];
    if (l + 1 == L.skip) bound = fmaxf(bound, x0max);
    const int eo = scale_exp(bound);
    const float* bias_s = vec_s + 256 * l;
    float* __restrict__ hrow = (a.h && valid) ? a.h + ((size_t)l * M + m) * W : nullptr;
    unsigned long long bits0 = 0ull, bits1 = 0ull;
    float vmax = 0.0f;
    f32x16 accA, accB;
    // tile j runs its MFMAs while tile j-1 (in the other accumulator) goes through its epilogue
#define R_TRUNK_TILE(J, ACC, PREV)                                                                                          \
    {                                                                                                                       \
      const char* p = slab_begin();                                                                                         \
      auto fill = [&](int q) {                                                                                              \
        if (J > 0)                                                                                                          \
          quad_epilogue<NP, EP_RELU | EP_MASK | EP_CONV, 0>(PREV, J - 1, q, un, bias_s, (J - 1) < 4 ? bits0 : bits1, vmax,  \
                                                            hrow, eo, Nh, Nl, nullptr, 0, dot3, hh);                        \
      };                                                                                                                    \
      if (l == 0) kloop<NP, 4, 0>(ACC, p, Xh, Xl, Bh, Bl, lane, fill);                                                      \
      else if (is_skip) kloop<NP, 4, 16>(ACC, p, Xh, Xl, Bh, Bl, lane, fill);                                               \
      else kloop<NP, 0, 16>(ACC, p, Xh, Xl, Bh, Bl, lane, fill);                                                            \
      RS(3);                                                                                                                \
      slab_end((J > 0 && hrow) ? 4 : 0);                                                                                    \
    }
    R_TRUNK_TILE(0, accA, accB)
    R_TRUNK_TILE(1, accB, accA)
    R_TRUNK_TILE(2, accA, accB)
    R_TRUNK_TILE(3, accB, accA)
    R_TRUNK_TILE(4, accA, accB)
    R_TRUNK_TILE(5, accB, accA)
    R_TRUNK_TILE(6, accA, accB)
    R_TRUNK_TILE(7, accB, accA)
#undef R_TRUNK_TILE
    tile_epilogue<NP, EP_RELU | EP_MASK | EP_CONV, 0>(accB, 7, un, bias_s, bits1, vmax, hrow, eo, Nh, Nl, nullptr, 0, dot3, hh);
    RS(4);
    mask_store(l, bits0, bits1);
    amax_in = wave_max_r(vmax);
    if (a.amax && lane == 0) atomicMax((unsigned int*)(a.amax + l), __float_as_uint(amax_in));
    e_in = eo;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      Bh[s] = Nh[s];
      if constexpr (NP == 2) Bl[s] = Nl[s];
    }
    RS(5);
  }

  // ---- shared density head (nerf.py:89): softplus(w . h + b), from the operand fragments (hi + lo = h to 2^-22)
  {
    float sdot = 0.0f;
    const float* wsig_s = vec_s + R_V_WSIG;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const f32x4 w0 = *(const f32x4*)&wsig_s[16 * s + 4 * hh], w1 = *(const f32x4*)&wsig_s[16 * s + 8 + 4 * hh];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v0 = (float)Bh[s][u], v1 = (float)Bh[s][4 + u];
        if constexpr (NP == 2) {
          v0 += (float)Bl[s][u];
          v1 += (float)Bl[s][4 + u];
        }
        sdot += v0 * w0[u] + v1 * w1[u];
      }
    }
    sdot += __shfl_xor(sdot, 32);
    if (hh == 0 && valid) a.sigma_s[m] = softplus_f(sdot * pow2r(-e_in) + bsig);
  }
  RS(7);
  if (sq.last_stage < D) {
    RS_FLUSH;
    return;  // density-only pass (nerf.py:90-91 `sigma_only`)
  }

  // ---- xyz_encoding_final (nerf.py:93), no activation; its operand form E feeds both heads
  float amax_e;
  int e_E;
  {
    const float un = pow2r(-(e_in + wexp[8]));
    const float bound = fmaxf(wnorm[D] * amax_in + vec_s[R_V_BMAX + 8], sidemax);  // the heads add per-ray rows at E's exponent
    e_E = scale_exp(bound);
    float* __restrict__ erow = (a.e && valid) ? a.e + (size_t)m * W : nullptr;
    unsigned long long nobits = 0ull;
    float vmax = 0.0f;
    f32x16 accA, accB;
#define R_FINAL_TILE(J, ACC, PREV)                                                                                  \
    {                                                                                                               \
      const char* p = slab_begin();                                                                                 \
      auto fill = [&](int q) {                                                                                      \
        if (J > 0)                                                                                                  \
          quad_epilogue<NP, EP_CONV, 0>(PREV, J - 1, q, un, vec_s + R_V_BE, nobits, vmax, erow, e_E, Nh, Nl,        \
                                        nullptr, 0, dot3, hh);                                                      \
      };                                                                                                            \
      kloop<NP, 0, 16>(ACC, p, Xh, Xl, Bh, Bl, lane, fill);                                                         \
      RS(3);                                                                                                        \
      slab_end((J > 0 && erow) ? 4 : 0);                                                                            \
    }
    R_FINAL_TILE(0, accA, accB)
    R_FINAL_TILE(1, accB, accA)
    R_FINAL_TILE(2, accA, accB)
    R_FINAL_TILE(3, accB, accA)
    R_FINAL_TILE(4, accA, accB)
    R_FINAL_TILE(5, accB, accA)
    R_FINAL_TILE(6, accA, accB)
    R_FINAL_TILE(7, accB, accA)
#undef R_FINAL_TILE
    tile_epilogue<NP, EP_CONV, 0>(accB, 7, un, vec_s + R_V_BE, nobits, vmax, erow, e_E, Nh, Nl, nullptr, 0, dot3, hh);
    amax_e = wave_max_r(vmax);
    if (a.amax && lane == 0) atomicMax((unsigned int*)(a.amax + D), __float_as_uint(amax_e));
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      Bh[s] = Nh[s];
      if constexpr (NP == 2) Bl[s] = Nl[s];
    }
  }
  if (sq.last_stage <= D) {
    RS_FLUSH;
    return;
  }

  // ---- colour head (folded first layer, nerf.py:95 + 102-109; rgb_share_layer.2 + sigmoid, nerf.py:56-61)
  if (sq.rgb) {
    h8 Ah[5], Al[5];  // [PE(dir) | appearance | 0] of this sample's ray as operand k-blocks
    const float* __restrict__ row = a.aux + (size_t)ray * UPNERF_AUXK;
#pragma unroll
    for (int s = 0; s < 5; ++s) row_op<NP>(row, s, hh, e_E, Ah[s], Al[s]);
    const float un = pow2r(-(e_E + wexp[11]));
    float* __restrict__ rrow = (a.r1 && valid) ? a.r1 + (size_t)m * W2 : nullptr;
    unsigned long long nobits = 0ull;
    float vmax = 0.0f;
    dot3[0] = dot3[1] = dot3[2] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const char* p = slab_begin();
      f32x16 acc;
      kloop<NP, 16, 5>(acc, p, Bh, Bl, Ah, Al, lane, [](int) {});
      tile_epilogue<NP, EP_RELU, 3>(acc, j, un, vec_s + R_V_BR1, nobits, vmax, rrow, 0, Nh, Nl, vec_s + R_V_WR2, W2, dot3, hh);
      slab_end(rrow ? 4 : 0);
    }
    const float mx = wave_max_r(vmax);
    if (a.amax && lane == 0) atomicMax((unsigned int*)(a.amax + D + 3), __float_as_uint(mx));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = dot3[c] + __shfl_xor(dot3[c], 32);
      if (hh == 0 && valid) a.rgb[(size_t)m * 3 + c] = sigmoid_f(d + br2[c]);
    }
  }
  // ---- candidate head (nerf.py:97-100)
  if (sq.cand) {
    h8 Ch[1], Cl[1];
    row_op<NP>(a.c_rows + (size_t)ray * UPNERF_CK, 0, hh, e_E, Ch[0], Cl[0]);
    float amax_g1;
    int e_G;
    {
      const float un = pow2r(-(e_E + wexp[9]));
      e_G = scale_exp(wnorm[D + 1] * fmaxf(amax_e, sidemax) + vec_s[R_V_BMAX + 10]);
      float* __restrict__ grow = (a.g1 && valid) ? a.g1 + (size_t)m * W2 : nullptr;
      unsigned long long bits = 0ull;
      float vmax = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const char* p = slab_begin();
        f32x16 acc;
          kloop<NP, 16, 1>(acc, p, Bh, Bl, Ch, Cl, lane, [](int) {});
        tile_epilogue<NP, EP_RELU | EP_MASK | EP_CONV, 0>(acc, j, un, vec_s + R_V_BC1, bits, vmax, grow, e_G, Nh, Nl, nullptr, 0,
                                                          dot3, hh);
        slab_end(grow ? 4 : 0);
      }
      if (mask_on) {  // candidate head: field16.hip's wave w owns feature tile w, bits [16 mt, 16 mt + 16) of its word
        const size_t base = ((size_t)D * nt64 + t64) * 256 + lane;
#pragma unroll
        for (int w = 0; w < 4; ++w) hm16[(base + 64 * w) * 4 + mt64] = (unsigned short)(bits >> (16 * w));
      }
      amax_g1 = wave_max_r(vmax);
      if (a.amax && lane == 0) atomicMax((unsigned int*)(a.amax + D + 1), __float_as_uint(amax_g1));
    }
    {
      const float un = pow2r(-(e_G + wexp[10]));
      float* __restrict__ grow = (a.g2 && valid) ? a.g2 + (size_t)m * W2 : nullptr;
      unsigned long long nobits = 0ull;
      float vmax = 0.0f;
      dot3[0] = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const char* p = slab_begin();
        f32x16 acc;
          kloop<NP, 8, 0>(acc, p, Nh, Nl, Bh, Bl, lane, [](int) {});
        tile_epilogue<NP, EP_RELU, 1>(acc, j, un, vec_s + R_V_BC2, nobits, vmax, grow, 0, Bh, Bl, vec_s + R_V_WCSIG, W2, dot3, hh);
        slab_end(grow ? 4 : 0);
      }
      const float d = dot3[0] + __shfl_xor(dot3[0], 32);
      if (hh == 0 && valid) a.sigma_c[m] = softplus_f(d + bcsig);
    }
  }
  RS(7);
  RS_FLUSH;
}

}  // namespace

// Entry point of the register-resident forward pass; the caller (upnerf_field_fwd_f16x3, field16.hip) has validated the
// arguments.  Needs a->P16 written by upnerf_frag16 with perm_fwd = 1 and a->wnorm from the same call.
int upnerf_field16r_fwd_launch(const upnerf_layout* L, const upnerf_field_fwd_args* a, void* stream) {
  const long long M = (long long)a->R * a->S;
  const int grid = (int)((M + R_TILE - 1) / R_TILE);
  if (a->planes == 1)
    hipLaunchKernelGGL((field16r_fwd_kernel<1>), dim3(grid), dim3(R_THREADS), 0, (hipStream_t)stream, *L, *a);
  else
    hipLaunchKernelGGL((field16r_fwd_kernel<2>), dim3(grid), dim3(R_THREADS), 0, (hipStream_t)stream, *L, *a);
  return (int)hipGetLastError();
}

#ifdef UPNERF_STAMPS
extern "C" int upnerf_stamps_read_r(unsigned long long* out8, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(upnerf_stamp_acc_r), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(upnerf_stamp_acc_r), z, sizeof(z)));
  }
  return 0;
}
#endif
