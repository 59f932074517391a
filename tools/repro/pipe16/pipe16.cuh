// Software-pipelined trunk stages of the 128-sample field kernels (csrc/field16.hip, TILE = 128, eight waves).
//
// Why.  A workgroup of eight waves owns 128 samples; wave w computes ONE 32-column n-tile of a 256-wide layer for all 128
// rows, so every weight fragment is fetched from L2 once per 128 rows (half the L2 -> CU stream of the 64-sample tiling).
// Run in lockstep, such a workgroup leaves the matrix pipe idle through every epilogue (bias, ReLU, sign bits, fp16 hi/lo
// split, LDS writes: ~6.5 vector instructions per accumulator element) and through two barriers per layer.  Rows do not mix
// in a layer (only columns do), so the tile is cut into two ROW halves, A = rows 0..63 and B = rows 64..127, and B trails
// A by PL_LAG = 8 of the 16 k-blocks:
//
//     phase 1 (8 k-steps)   MFMA  A(0..7)              | vector: epilogue of B, previous stage -> planes B     barrier
//     phase 2 (8 k-steps)   MFMA  A(8..15), B(0..7)    | vector: stores of the previous stage's activations     barrier
//     phase 3 (8 k-steps)   MFMA  B(8..15)             | vector: epilogue of A, this stage     -> planes A     barrier
//
// The matrix pipe has work in every phase; the epilogue of one half runs in the shadow of the other half's MFMAs.  A
// weight fragment loaded for A(t) is kept in registers until B(t) has used it eight steps later (a ring of at most ten
// live 1 KiB fragment pairs per wave, statically indexed: every phase is fully unrolled).  Each half carries its own
// power-of-two exponent (they never meet in a contraction).
//
// Exponents from a bound.  The exponent of a half's next planes is needed before its epilogue starts, so it comes from
// bound = max|acc| * 2^-(e_in + e_w) + max|bias| >= max|relu(acc * 2^-(e_in+e_w) + bias)|, reduced over the eight waves
// through LDS at the barrier that already separates the phases.  The bound is mapped into [2^14, 2^15): at most one bit
// above the exact maximum's [2^13, 2^14) of the lockstep kernels and within fp16 range (65504).
//
// Activation stores ride in the NEXT stage's K loop.  The fp16 operand fragments a wave reads for its MFMAs ARE the
// previous stage's output (hi + lo = 22 mantissa bits of it), and every wave reads all of them; wave w converts and stores
// the fragments of k-blocks 2w and 2w+1 ((hi + lo) * 2^-e, two v_fma_mix_f32 per element, 16-byte stores, one 128-byte
// line per row and wave).  No accumulator copy is kept, no store burst sits between the K loops, and the bytes leave the
// CU spread over the whole stage.
#pragma once
#include "common16.cuh"

#define PL_LAG 8

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int NP>
struct XFrag {
  h8 h[2], l[2];
};

// LDS writes of this wave have landed, then the workgroup barrier; global stores stay in flight (no vmcnt wait)
__device__ __forceinline__ void pl_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// maximum of non-negative floats over the wave, as a wave-uniform value (DPP butterflies inside the 16-lane rows, then
// one v_readlane per row: no LDS traffic, no dependent permute latency)
__device__ __forceinline__ float wave_max_nn(float m) {
  int v = __builtin_bit_cast(int, m);  // non-negative floats order like their bit patterns
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false));  // row_ror:8
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false));  // row_ror:4
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false));  // row_ror:2
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false));  // row_ror:1
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return __builtin_bit_cast(float, max(max(a, b), max(c, d)));
}

// exponent that brings a positive bound into [2^14, 2^15); 0 for zero
__device__ __forceinline__ int bound_exp(float bound) {
  if (!(bound > 0.0f)) return 0;
  int ex;
  (void)frexpf(bound, &ex);  // bound = m * 2^ex, m in [0.5, 1)
  const int e = 15 - ex;
  return e > 100 ? 100 : (e < -100 ? -100 : e);
}

// (fp16 half HALF of `pair`) * s + c in one instruction
template <int HALF>
__device__ __forceinline__ float mix16(unsigned int pair, float s, float c) {
  float r;
  if constexpr (HALF == 0)
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(s), "v"(c));
  else
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(s), "v"(c));
  return r;
}

// operand fragments of one row half for k-block t: rows rbyte/512 + 32*mt, k = 16t + 8hh .. +8
template <int NP, int W>
__device__ __forceinline__ void pl_ldx(XFrag<NP>& x, const char* Ph, const char* Pl, int rbyte, int rsw, int t, int hh) {
  static_assert(W == 256, "plane rows are 512 bytes");
  const int ch = ((2 * t + hh) ^ rsw) << 4;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int o = rbyte + mt * (32 * W * 2) + ch;
    x.h[mt] = *(const h8*)(Ph + o);
    if constexpr (NP == 2) x.l[mt] = *(const h8*)(Pl + o);
  }
}

// weight fragment pair of k-block t of this wave's n-tile (wp: fragment-ordered matrix at the n-tile's first k-block + lane*16)
template <int NP>
__device__ __forceinline__ void pl_ldw(h8& wh, h8& wl, const char* __restrict__ wp, int t) {
  wh = *(const h8*)(wp + (size_t)t * 2048);
  if constexpr (NP == 2) wl = *(const h8*)(wp + (size_t)t * 2048 + 1024);
}

// acc[2H], acc[2H+1] += W-fragment . X-fragments (transposed product as in common16.cuh:mma16_step)
template <int NP, int H>
__device__ __forceinline__ void pl_mma(f32x16 (&acc)[4], const XFrag<NP>& x, const h8& wh, const h8& wl) {
  acc[2 * H] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, x.h[0], acc[2 * H], 0, 0, 0);
  acc[2 * H + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, x.h[1], acc[2 * H + 1], 0, 0, 0);
  if constexpr (NP == 2) {
    acc[2 * H] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, x.h[0], acc[2 * H], 0, 0, 0);
    acc[2 * H + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, x.h[1], acc[2 * H + 1], 0, 0, 0);
    acc[2 * H] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, x.l[0], acc[2 * H], 0, 0, 0);
    acc[2 * H + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, x.l[1], acc[2 * H + 1], 0, 0, 0);
  }
}

template <int H>
__device__ __forceinline__ void pl_zero(f32x16 (&acc)[4]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[2 * H][r] = 0.0f;
    acc[2 * H + 1][r] = 0.0f;
  }
}

template <int H>
__device__ __forceinline__ void pl_scale(f32x16 (&acc)[4], float s) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[2 * H][r] *= s;
    acc[2 * H + 1][r] *= s;
  }
}

// max |acc| of one row half over the wave (wave-uniform)
template <int H>
__device__ __forceinline__ float pl_absmax(const f32x16 (&acc)[4]) {
  float m = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    m = __builtin_fmaxf(m, __builtin_fmaxf(fabsf(acc[2 * H][r]), fabsf(acc[2 * H][r + 1])));
    m = __builtin_fmaxf(m, __builtin_fmaxf(fabsf(acc[2 * H + 1][r]), fabsf(acc[2 * H + 1][r + 1])));
  }
  return wave_max_nn(m);
}

// One PIECE of the activation stores of a stage: two whole rows (all 256 columns) of one row half per wave, read back
// from the planes: lane j converts columns 4j .. 4j+3, so a store instruction writes one row = 1 KiB contiguous.  (Per-lane
// row pieces -- what the MFMA layouts hand out naturally, 32 bytes in each of 32 rows per instruction -- were measured at
// half the HBM write rate and ~14 B/clk/CU of issue.)  The eight waves share a half's 64 rows: wave w stores rows
// 8 * piece... + w, four pieces of two rows per half and stage.
//   fp32: (hi + lo) * un, two v_fma_mix_f32 per element;  fp16 storage: the hi plane as it stands (lane j: 16-byte chunk
//   j % 32 of row j / 32 -> both rows of the piece in ONE instruction).
// base32 / base16: tensor rows of this tile half (row 0 of the half), or nullptr; nrows: rows of the half inside the tensor.
template <int NP, int W>
__device__ __forceinline__ void pl_store_piece(const char* Ph, const char* Pl, int half_row0, int piece, int lane, int wave,
                                               float* __restrict__ base32, uint16_t* __restrict__ base16, int nrows, float un) {
  if (base32) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 16 * piece + 8 * i + wave, k = 4 * lane;
      const int o = poff<W>(half_row0 + r, k);
      const u32x2_t uh = *(const u32x2_t*)(Ph + o);  // four fp16 values
      f32x4 v;
      if constexpr (NP == 2) {
        const u32x2_t ul = *(const u32x2_t*)(Pl + o);
        v = f32x4{mix16<0>(uh[0], un, mix16<0>(ul[0], un, 0.f)), mix16<1>(uh[0], un, mix16<1>(ul[0], un, 0.f)),
                  mix16<0>(uh[1], un, mix16<0>(ul[1], un, 0.f)), mix16<1>(uh[1], un, mix16<1>(ul[1], un, 0.f))};
      } else {
        v = f32x4{mix16<0>(uh[0], un, 0.f), mix16<1>(uh[0], un, 0.f), mix16<0>(uh[1], un, 0.f), mix16<1>(uh[1], un, 0.f)};
      }
      if (r < nrows) *(f32x4*)(base32 + (size_t)r * W + k) = v;
    }
  } else if (base16) {
    const int r = 16 * piece + 8 * (lane >> 5) + wave, k = 8 * (lane & 31);
    const f32x4 v = *(const f32x4*)(Ph + poff<W>(half_row0 + r, k));
    if (r < nrows) *(f32x4*)(base16 + (size_t)r * W + k) = v;
  }
}

// One register quad of a row half: v = [relu]((acc * s + b) * pe) in PLANE units (pe = 2^e_out, s = 2^-(e_in + e_w): acc * s is
// the natural pre-activation), sign bits, running maximum (plane units), fp16 hi / lo split, one ds_write_b64 per plane.
// c = 4*mt + q picks the quad; H, c, RELU fold to constants after unrolling (static register indices).  The bias row lives in
// LDS (bias_lds: this layer's [256] floats): sixteen registers per wave are worth more than one broadcast read per quad.
template <int NP, int W, int H, bool RELU>
__device__ __forceinline__ void pl_epi_quad(f32x16 (&acc)[4], int c, const float* bias_lds, float s, float pe, char* Ph, char* Pl,
                                            int rbyte, int rsw, int n0, int hh, unsigned int& bits, float& vmax) {
  const int mt = c >> 2, q = c & 3;
  const f32x4 b = *(const f32x4*)&bias_lds[n0 + 8 * q + 4 * hh];  // this lane's four columns of the quad (LDS broadcast read)
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = fmaf(acc[2 * H + mt][4 * q + j], s, b[j]) * pe;
    if constexpr (RELU) {
      v[j] = fmaxf(v[j], 0.0f);
      bits |= (v[j] > 0.0f) ? (1u << (mt * 16 + 4 * q + j)) : 0u;
      vmax = fmaxf(vmax, v[j]);
    } else {
      vmax = fmaxf(vmax, fabsf(v[j]));
    }
  }
  // the sign bits and the maximum are folded in HERE: left to itself hipcc keeps all 32 values of a half alive and computes
  // both after the last quad (32 registers, ~150 serial instructions in front of the barrier)
  asm volatile("" : "+v"(bits), "+v"(vmax));
  h4 hi, lo;
  split_quad<NP>(v[0], v[1], v[2], v[3], hi, lo);
  const int o = rbyte + mt * (32 * W * 2) + ((((n0 >> 3) + q) ^ rsw) << 4) + 8 * hh;
  *(h4*)(Ph + o) = hi;
  if constexpr (NP == 2) *(h4*)(Pl + o) = lo;
}

// Backward counterpart: v = (sign bit of the forward activation ? acc * sp : 0) in PLANE units (sp = 2^(e_out - e_in - e_w): the
// natural pre-activation gradient acc * 2^-(e_in + e_w), exactly rescaled), running maximum of |v|, split, LDS writes.
template <int NP, int W, int H>
__device__ __forceinline__ void pl_epi_bwd_quad(f32x16 (&acc)[4], int c, float sp, unsigned int bits, char* Ph, char* Pl, int rbyte,
                                                int rsw, int n0, int hh, float& vmax) {
  const int mt = c >> 2, q = c & 3;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = ((bits >> (mt * 16 + 4 * q + j)) & 1u) ? acc[2 * H + mt][4 * q + j] * sp : 0.0f;
    vmax = fmaxf(vmax, fabsf(v[j]));
  }
  asm volatile("" : "+v"(vmax));  // folded in here, not after the last quad (see pl_epi_quad)
  h4 hi, lo;
  split_quad<NP>(v[0], v[1], v[2], v[3], hi, lo);
  const int o = rbyte + mt * (32 * W * 2) + ((((n0 >> 3) + q) ^ rsw) << 4) + 8 * hh;
  *(h4*)(Ph + o) = hi;
  if constexpr (NP == 2) *(h4*)(Pl + o) = lo;
}

// The K loop of one 256-deep stage in the three phases of the header.  Functors (all inlined; arguments fold to constants):
//   epiB(c)  one eighth (c = 0..7) of the epilogue of half B of the PREVIOUS stage (phase 1);  epiA(c)  of half A of this
//            stage (phase 3)
//   pieceA(j) / pieceB(j)   one quarter (j = 0..3) of this wave's share of the activation stores of half A / B (the planes
//            of A hold the previous stage's output through phases 1 and 2, those of B through phases 2 and 3)
//   preB()      start of phase 2, before B's first MFMA (accumulators of B are free from here on: zero / pre-load them)
//   endP1() / endP2() / endP3()   publish maxima, barrier, read them back
//
// Weight requests.  CDNA4 retires a wave's loads and stores through ONE in-order counter: a wait for a weight fragment is
// also a wait for every older activation store of the wave.  The ring therefore runs FOUR steps ahead where steps are
// short: phase 3 requests the next stage's fragments 0..3 (the ring has drained by then), phase 1 requests fragment t + 4
// at step t (t = 0..5), phase 2 fragment t + 2 (its steps carry twelve MFMAs per wave).  Never more than ten pairs live.
// wnext: in: this stage's fragments 0..3; out: the next stage's (wp_next; nullptr: none).
#define PL_PRE 4
template <int NP>
struct WPre {
  h8 h[PL_PRE], l[PL_PRE];
};

template <int NP>
__device__ __forceinline__ void pl_preload(WPre<NP>& w, const char* __restrict__ wp) {
#pragma unroll
  for (int t = 0; t < PL_PRE; ++t) pl_ldw<NP>(w.h[t], w.l[t], wp, t);
}

// timing experiments (diagnostic builds only; results are wrong with any of them)
#ifdef PL_EXP_NOLDX
#define PL_LDX(x, rb, t) do { } while (0)
#else
#define PL_LDX(x, rb, t) pl_ldx<NP, W>(x, Ph, Pl, rb, rsw, t, hh)
#endif
#ifdef PL_EXP_NOEPI
#define PL_EPI(e) do { } while (0)
#else
#define PL_EPI(e) e
#endif
#ifdef PL_EXP_NOPIECE
#define PL_PIECE(e) do { } while (0)
#else
#define PL_PIECE(e) e
#endif

template <int NP, int W, class EpiB, class EpiA, class PieceA, class PieceB, class PreB, class End1, class End2, class End3>
__device__ __forceinline__ void pl_stage(f32x16 (&acc)[4], const char* Ph, const char* Pl, int rbyteA, int rsw, int hh,
                                         const char* __restrict__ wp, WPre<NP>& wnext, const char* __restrict__ wp_next,
                                         EpiB&& epiB, EpiA&& epiA, PieceA&& pieceA, PieceB&& pieceB, PreB&& preB, End1&& endP1,
                                         End2&& endP2, End3&& endP3) {
  constexpr int T = W / 16, LAG = PL_LAG, PRE = PL_PRE, PF2 = 2;
  static_assert(T == 16 && LAG == 8, "phase structure");
  // the per-k-block LDS addresses are invariant across the caller's layer loop: hipcc would hoist all of them out of it and
  // spill (sixteen live registers); an opaque redefinition per stage keeps them local to the step that uses them
  asm volatile("" : "+v"(rbyteA), "+v"(rsw));
  const int rbyteB = rbyteA + 64 * W * 2;
  h8 wh[T], wl[T];
  XFrag<NP> xa, xb;
#pragma unroll
  for (int t = 0; t < PRE; ++t) {
    wh[t] = wnext.h[t];
    if constexpr (NP == 2) wl[t] = wnext.l[t];
  }
  pl_ldx<NP, W>(xa, Ph, Pl, rbyteA, rsw, 0, hh);
  // ---- phase 1: A(0..7) | epilogue of B (previous stage) | store pieces of A at steps 2, 6
#pragma unroll
  for (int t = 0; t < LAG; ++t) {
    if (t + PRE < LAG + PF2) pl_ldw<NP>(wh[t + PRE], wl[t + PRE], wp, t + PRE);
    pl_mma<NP, 0>(acc, xa, wh[t], wl[t]);
    PL_LDX(xa, rbyteA, t + 1);
    PL_EPI(epiB(t));
    if (t == 2) PL_PIECE(pieceA(0));
    if (t == 6) PL_PIECE(pieceA(1));
    __builtin_amdgcn_sched_barrier(0);
  }
  endP1();
  // ---- phase 2: A(8..15), B(0..7) | store pieces of A at steps 9, 13, of B at steps 11, 15.  B goes first in a step: its
  // MFMAs retire the oldest fragment pair of the ring before the request for the newest one is issued.
  pl_ldx<NP, W>(xb, Ph, Pl, rbyteB, rsw, 0, hh);
  preB();
#pragma unroll
  for (int t = LAG; t < T; ++t) {
    pl_mma<NP, 1>(acc, xb, wh[t - LAG], wl[t - LAG]);
    PL_LDX(xb, rbyteB, t - LAG + 1);
    if (t + PF2 < T) pl_ldw<NP>(wh[t + PF2], wl[t + PF2], wp, t + PF2);
    pl_mma<NP, 0>(acc, xa, wh[t], wl[t]);
    if (t + 1 < T) PL_LDX(xa, rbyteA, t + 1);
    if (t == 9) PL_PIECE(pieceA(2));
    if (t == 13) PL_PIECE(pieceA(3));
    if (t == 11) PL_PIECE(pieceB(0));
    if (t == 15) PL_PIECE(pieceB(1));
    __builtin_amdgcn_sched_barrier(0);
  }
  endP2();
  // ---- phase 3: B(8..15) | epilogue of A (this stage) | store pieces of B at steps 17, 21 | the next stage's fragments
  // 0..3 at steps 20..23 (short live ranges: the ring has drained)
#pragma unroll
  for (int t = T; t < T + LAG; ++t) {
    if (wp_next && t - T >= LAG - PRE) pl_ldw<NP>(wnext.h[t - T - (LAG - PRE)], wnext.l[t - T - (LAG - PRE)], wp_next, t - T - (LAG - PRE));
    pl_mma<NP, 1>(acc, xb, wh[t - LAG], wl[t - LAG]);
    if (t + 1 < T + LAG) PL_LDX(xb, rbyteB, t - LAG + 1);
    PL_EPI(epiA(t - T));
    if (t == T + 1) PL_PIECE(pieceB(2));
    if (t == T + 5) PL_PIECE(pieceB(3));
    __builtin_amdgcn_sched_barrier(0);
  }
  endP3();
}
