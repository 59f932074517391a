// Building blocks of the field kernels on the f16 matrix cores (csrc/field16.hip), in two arithmetic modes:
//   NP = 2 ("f16x3")  fp32-accurate: a 64-sample tile lives in LDS as TWO fp16 planes (hi, lo) with
//                     value = (hi + lo) * 2^-e; every 32x32x16 block is three v_mfma_f32_32x32x16_f16
//                     (hi*hi + hi*lo + lo*hi; fp16 products are exact in fp32, accumulation is fp32): 96 matrix cycles
//                     instead of the 512 of eight v_mfma_f32_32x32x2_f32.  The split residue (lo*lo and the bits below
//                     hi+lo) is 2^-22 relative -- the order of the fp32 rounding of the accumulation itself.
//   NP = 1 ("f16")    fp16 weights and activations, ONE MFMA per block, fp32 accumulate (BASELINE.json configs[3]); the lo
//                     planes / lo weight blocks are simply never written or read.
// e is a per-tile, per-stage power-of-two exponent chosen so that the tile's largest magnitude lands in [2^13, 2^14)
// (exact scaling, no overflow, full use of the fp16 range).  Weights are re-packed per step into hi/lo form in MFMA
// fragment order with one exponent per matrix (upnerf_frag16).
//
// TRANSPOSED accumulators.  The contraction is issued as  D^T[n][m] = sum_k W[n][k] X[m][k]:  the weight fragment is the
// MFMA's A operand, the activation fragment its B operand (both lane maps are the same: lane l holds row / column l&31,
// k = 8*(l>>5) .. +8, so the operand loads are unchanged).  A lane then holds, for ITS sample row m = row0 + 32*mt +
// (l&31), four CONSECUTIVE output features per register quad:  n = n0 + 32*nt + 8*(r>>2) + 4*(l>>5) + (r&3).
// That turns the epilogue from 64 scalar 2-byte LDS writes + 64 scalar conversions per lane and layer into packed work:
// v_cvt_pk_f16_f32 on register pairs, one v_fma_mix_f32 per residual, ds_write_b64 per quad and plane, 16-byte global
// stores, per-row scalars (one LDS read per lane instead of one per register).
#pragma once
#include "common.cuh"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

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

#define WEXP_SLOTS 16  // exponent table: trunk layers 0..7, 8 = final, 9 = cand1, 10 = cand2, 11 = rgb1, 12 = head^T

// exponent that brings a positive maximum into [2^13, 2^14); 0 for an all-zero tile
__device__ __forceinline__ int scale_exp(float mx) {
  if (!(mx > 0.0f)) return 0;
  int ex;
  (void)frexpf(mx, &ex);  // mx = m * 2^ex, m in [0.5, 1)
  int e = 14 - ex;
  return e > 100 ? 100 : (e < -100 ? -100 : e);
}

__device__ __forceinline__ void split16(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// x - (float)h for the fp16 value in half HALF of a packed pair, in ONE vector instruction (v_fma_mix_f32 reads the fp16
// operand straight from its half register: h * -1.0 + x; exact, h is x rounded to 11 bits)
template <int HALF>
__device__ __forceinline__ float resid16(unsigned int hpair, float x) {
  float r;
  if constexpr (HALF == 0)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
  else
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
  return r;
}

// four fp32 values (already scaled) -> packed fp16 quads hi (and lo = the rounded residuals when NP == 2)
template <int NP>
__device__ __forceinline__ void split_quad(float x0, float x1, float x2, float x3, h4& hi, h4& lo) {
  const h2 a = __builtin_convertvector(f32x2{x0, x1}, h2), b = __builtin_convertvector(f32x2{x2, x3}, h2);  // v_cvt_pk_f16_f32
  hi = __builtin_shufflevector(a, b, 0, 1, 2, 3);
  if constexpr (NP == 2) {
    const unsigned int ua = __builtin_bit_cast(unsigned int, a), ub = __builtin_bit_cast(unsigned int, b);
    const h2 c = __builtin_convertvector(f32x2{resid16<0>(ua, x0), resid16<1>(ua, x1)}, h2);
    const h2 d = __builtin_convertvector(f32x2{resid16<0>(ub, x2), resid16<1>(ub, x3)}, h2);
    lo = __builtin_shufflevector(c, d, 0, 1, 2, 3);
  }
}

// byte offset of element (row, k) in a [64][W] fp16 plane; 16-byte chunks XOR-swizzled by the row so that the 16 rows
// one ds_read_b128 lane group touches fall into 16 different bank quads (512-byte rows: bank = 4 * (chunk % 16)).
template <int W>
__device__ __forceinline__ int poff(int row, int k) {
  static_assert(W == 256, "f16 field kernels are built for W = 256");
  return row * (W * 2) + ((((k >> 3) ^ (row & 15))) << 4) + ((k & 7) << 1);
}

template <int MT, int NT>
__device__ __forceinline__ float acc_absmax(const f32x16 (&acc)[MT][NT]) {
  float m = 0.0f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(fabsf(acc[mt][nt][r]), fabsf(acc[mt][nt][r + 1])));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  return m;
}

// acc[nt x mt] += W-fragment (A operand) . X-fragment (B operand): transposed product, see the header
template <int NP, int MT, int NT>
__device__ __forceinline__ void mma16_step(f32x16 (&acc)[MT][NT], const h8 (&xh)[MT], const h8 (&xl)[MT],
                                           const h8 (&wh)[NT], const h8 (&wl)[NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[nt], xh[mt], acc[mt][nt], 0, 0, 0);
      if constexpr (NP == 2) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[nt], xh[mt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[nt], xl[mt], acc[mt][nt], 0, 0, 0);
      }
    }
}

// acc (transposed) += planes[:, kA0 .. kA0+16*T) . Wf[n][kB0 .. kB0+16*T)^T      (scaled integers-in-fp16)
//   Ph, Pl: LDS planes;  Wf: fragment-ordered hi/lo matrix with Kp16 = Kp/16 k-blocks per 32-column tile:
//   byte ((ntile * Kp16 + t) * 2 + plane) * 1024 + lane * 16.   T = number of 16-deep k-blocks (compile time).
// The loop is LATENCY-bound on the weight stream if only one k-block is in flight: a fragment comes from L2 (~650 cycles,
// measured as the K-loop time per k-block in the f16 mode), while the MFMAs of one k-block take 384 (f16x3) or 128 (f16)
// matrix cycles.  So the weight fragments travel through a RING of AHEAD+1 register sets: block t+AHEAD is requested
// before the MFMAs of block t issue (AHEAD x MFMA time >= L2 latency); the activation fragments (LDS, ~100 cycles) stay
// one block ahead.  One ring revolution is unrolled, so every ring index is a compile-time register name; sched_barrier
// fences keep the requests ABOVE the matrix work (hipcc otherwise sinks them to their first use).
// weight-fragment load.  Experiments (make variant): UPNERF_W_NT = non-temporal (L1-bypassing) loads; UPNERF_EXP_SAMEB = every
// k-block reads the FIRST block of its n-tile (wrong results: the kernel with its weight stream served from L1)
#if defined(UPNERF_W_NT)
#define W_LOAD(p, t) __builtin_nontemporal_load((const h8*)((p) + (size_t)(t) * 2048))
#define W_LOAD_LO(p, t) __builtin_nontemporal_load((const h8*)((p) + (size_t)(t) * 2048 + 1024))
#elif defined(UPNERF_EXP_SAMEB)
#define W_LOAD(p, t) (*(const h8*)((p) + (size_t)((t) & 1) * 2048))
#define W_LOAD_LO(p, t) (*(const h8*)((p) + (size_t)((t) & 1) * 2048 + 1024))
#else
#define W_LOAD(p, t) (*(const h8*)((p) + (size_t)(t) * 2048))
#define W_LOAD_LO(p, t) (*(const h8*)((p) + (size_t)(t) * 2048 + 1024))
#endif
// `piece(t)` (optional) runs once per k-block, in the request section of block t -- behind the weight requests of block
// t + AHEAD and the operand reads of block t + 1, in front of the MFMAs of block t: the place of work that rides under the
// contraction (the previous stage's activation stores, one 1 KiB piece per k-block: field16.hip:PlaneStore).
struct NoPiece {
  __device__ __forceinline__ void operator()(int) const {}
};
template <int NP, int W, int T, int AHEAD, int MT, int NT, class PIECE = NoPiece>
__device__ __forceinline__ void mma16_lds(f32x16 (&acc)[MT][NT], const char* Ph, const char* Pl, int row0, int kA0,
                                          const char* __restrict__ Wf, int Kp16, int n0, int kB0, int lane,
                                          PIECE&& piece = PIECE()) {
  constexpr int SETS = (AHEAD + 1 > T) ? T : AHEAD + 1;  // ring size; the ring runs SETS - 1 blocks ahead
  constexpr int AH = SETS - 1;
#ifdef UPNERF_EXP_SETPRIO
  struct Prio { __device__ Prio() { __builtin_amdgcn_s_setprio(UPNERF_EXP_SETPRIO); } __device__ ~Prio() { __builtin_amdgcn_s_setprio(0); } } prio_;
#endif
#ifndef F16_FORCE_UNROLLED_K
#define F16_FORCE_UNROLLED_K 0
#endif
  if constexpr (F16_FORCE_UNROLLED_K || SETS % 2 != 0 || T % SETS != 0) {
    // ring sizes that do not divide the loop (two blocks ahead = three sets): the whole K loop unrolled, every ring index a
    // compile-time constant.  The lane's row offsets are made opaque first: hipcc would otherwise hoist all T LDS addresses
    // out of the caller's layer loop and spill them.
    int li = lane & 31, lh = lane >> 5;
    asm volatile("" : "+v"(li), "+v"(lh));
#ifdef UPNERF_EXP_KROT
    // experiment (round 6): every workgroup walks the K dimension from its own starting block -- the CUs of an XCD then ask the L2
    // for DIFFERENT fragments at any moment instead of all for the same one (same products, another summation order)
    static_assert((T & (T - 1)) == 0, "rotation: power-of-two trip counts");
    const int rot = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) & (T - 1));
#define KR(t) (((t) + rot) & (T - 1))
#else
#define KR(t) (t)
#endif
    const char* bpu[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bpu[nt] = Wf + ((size_t)((n0 >> 5) + nt) * Kp16 + (kB0 >> 4)) * 2048 + (li + 32 * lh) * 16;
    h8 uwh[SETS][NT], uwl[SETS][NT], uxh[2][MT], uxl[2][MT];
    auto ldw = [&](int t) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        uwh[t % SETS][nt] = W_LOAD(bpu[nt], KR(t));
        if constexpr (NP == 2) uwl[t % SETS][nt] = W_LOAD_LO(bpu[nt], KR(t));
      }
    };
    auto ldx = [&](int t) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int o = poff<W>(row0 + 32 * mt + li, kA0 + 16 * KR(t) + 8 * lh);
        uxh[t & 1][mt] = *(const h8*)(Ph + o);
        if constexpr (NP == 2) uxl[t & 1][mt] = *(const h8*)(Pl + o);
      }
    };
#pragma unroll
    for (int t = 0; t < AH; ++t) ldw(t);
    ldx(0);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + AH < T) ldw(t + AH);
      if (t + 1 < T) ldx(t + 1);
      piece(t);
      __builtin_amdgcn_sched_barrier(0);
      mma16_step<NP>(acc, uxh[t & 1], uxl[t & 1], uwh[t % SETS], uwl[t % SETS]);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef KR
    return;
  }
  static_assert(F16_FORCE_UNROLLED_K || SETS % 2 != 0 || T % SETS != 0 || std::is_same<typename std::remove_reference<PIECE>::type, NoPiece>::value,
                "per-k-block pieces: unrolled K loops only");
  const int i = lane & 31, hh = lane >> 5;
  const char* bp[NT];
  int arow[MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bp[nt] = Wf + ((size_t)((n0 >> 5) + nt) * Kp16 + (kB0 >> 4)) * 2048 + lane * 16;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) arow[mt] = row0 + 32 * mt + i;
  h8 wh[SETS][NT], wl[SETS][NT], xh[2][MT], xl[2][MT];
  auto loadw = [&](h8 (&h)[NT], h8 (&l)[NT], int t) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      h[nt] = W_LOAD(bp[nt], t);
      if constexpr (NP == 2) l[nt] = W_LOAD_LO(bp[nt], t);
    }
  };
  auto loadx = [&](h8 (&h)[MT], h8 (&l)[MT], int t) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int o = poff<W>(arow[mt], kA0 + 16 * t + 8 * hh);
      h[mt] = *(const h8*)(Ph + o);
      if constexpr (NP == 2) l[mt] = *(const h8*)(Pl + o);
    }
  };
#pragma unroll
  for (int t = 0; t < AH; ++t) loadw(wh[t], wl[t], t);
  loadx(xh[0], xl[0], 0);
  // all groups but the last: every request is in range (rolled: the unrolled form lets hipcc hoist all LDS addresses and spill)
#pragma unroll 1
  for (int t0 = 0; t0 < T - SETS; t0 += SETS) {
#pragma unroll
    for (int u = 0; u < SETS; ++u) {
      loadw(wh[(u + AH) % SETS], wl[(u + AH) % SETS], t0 + u + AH);
      loadx(xh[(u + 1) & 1], xl[(u + 1) & 1], t0 + u + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma16_step<NP>(acc, xh[u & 1], xl[u & 1], wh[u], wl[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // last group: requests past the end are dropped at compile time
#pragma unroll
  for (int u = 0; u < SETS; ++u) {
    if (u + AH < SETS) loadw(wh[(u + AH) % SETS], wl[(u + AH) % SETS], T - SETS + u + AH);
    if (u + 1 < SETS) loadx(xh[(u + 1) & 1], xl[(u + 1) & 1], T - SETS + u + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma16_step<NP>(acc, xh[u & 1], xl[u & 1], wh[u], wl[u]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Two contractions that read the SAME plane columns in one K loop (round 6): acc[mt][0] += planes . WfA[n0 ..)^T, acc[mt][1] +=
// planes . WfB[n0 ..)^T -- the first layers of the colour and the candidate head both contract the 256 columns of e, each 128 wide
// (one n-tile per wave): run one after the other, each loop has six MFMAs per k-block to cover its weight fragments' L2 latency
// with (a trunk layer's loop has twelve) and reads the activation fragments again.  Joined, the k-block is a trunk layer's: four
// weight fragments, two activation fragments, twelve MFMAs.  Per accumulator the products and their order are those of mma16_lds
// with NT = 1 (bitwise the same sums).  T k-blocks, ring of AHEAD + 1 fragment sets, fully unrolled (as mma16_lds's unrolled form).
template <int NP, int W, int T, int AHEAD, int MT>
__device__ __forceinline__ void mma16_lds_pair(f32x16 (&acc)[MT][2], const char* Ph, const char* Pl, int row0, int kA0,
                                               const char* __restrict__ WfA, int Kp16A, const char* __restrict__ WfB, int Kp16B, int n0,
                                               int lane) {
  constexpr int SETS = (AHEAD + 1 > T) ? T : AHEAD + 1;
  constexpr int AH = SETS - 1;
  int li = lane & 31, lh = lane >> 5;
  asm volatile("" : "+v"(li), "+v"(lh));
  const char* bpu[2];
  bpu[0] = WfA + ((size_t)(n0 >> 5) * Kp16A) * 2048 + (li + 32 * lh) * 16;
  bpu[1] = WfB + ((size_t)(n0 >> 5) * Kp16B) * 2048 + (li + 32 * lh) * 16;
  h8 uwh[SETS][2], uwl[SETS][2], uxh[2][MT], uxl[2][MT];
  auto ldw = [&](int t) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      uwh[t % SETS][nt] = W_LOAD(bpu[nt], t);
      if constexpr (NP == 2) uwl[t % SETS][nt] = W_LOAD_LO(bpu[nt], t);
    }
  };
  auto ldx = [&](int t) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int o = poff<W>(row0 + 32 * mt + li, kA0 + 16 * t + 8 * lh);
      uxh[t & 1][mt] = *(const h8*)(Ph + o);
      if constexpr (NP == 2) uxl[t & 1][mt] = *(const h8*)(Pl + o);
    }
  };
#pragma unroll
  for (int t = 0; t < AH; ++t) ldw(t);
  ldx(0);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t + AH < T) ldw(t + AH);
    if (t + 1 < T) ldx(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma16_step<NP>(acc, uxh[t & 1], uxl[t & 1], uwh[t % SETS], uwl[t % SETS]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Same with the activation operand converted on the fly from fp32 rows in global memory (short side inputs: encoding for
// the skip connection, per-ray embedding rows): xrow_ptr[mt] points at this lane's row at column 8*(lane>>5); `e` is the
// exponent of the LDS planes the same accumulators are fed from.  K % 16 == 0.
// K is a compile-time constant (64, 80 or 16): every row piece is requested up front and the weight fragments travel one k-block
// ahead, all in straight-line code -- the rolled form (one k-block per trip: requests, `s_waitcnt vmcnt(0)`, MFMAs) paid one L2
// round trip per k-block with the matrix pipe idle, ten of them per tile in the schedule phase with all heads on.
template <int NP, int K, int MT, int NT>
__device__ __forceinline__ void mma16_glb(f32x16 (&acc)[MT][NT], const float* const (&xrow_ptr)[MT], int e,
                                          const char* __restrict__ Wf, int Kp16, int n0, int kB0, int lane) {
  constexpr int T = K >> 4;
  static_assert(K % 16 == 0 && T >= 1 && T <= 5, "short side inputs only");
  f32x4 xr[T][MT][2];
  h8 wh[2][NT], wl[2][NT];
  const char* bp[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bp[nt] = Wf + ((size_t)((n0 >> 5) + nt) * Kp16 + (kB0 >> 4)) * 2048 + lane * 16;
  auto ldw = [&](int t) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      wh[t & 1][nt] = *(const h8*)(bp[nt] + (size_t)t * 2048);
      if constexpr (NP == 2) wl[t & 1][nt] = *(const h8*)(bp[nt] + (size_t)t * 2048 + 1024);
    }
  };
  ldw(0);
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      xr[t][mt][0] = *(const f32x4*)(xrow_ptr[mt] + 16 * t);
      xr[t][mt][1] = *(const f32x4*)(xrow_ptr[mt] + 16 * t + 4);
    }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) ldw(t + 1);
    h8 xh[MT], xl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 x0 = xr[t][mt][0], x1 = xr[t][mt][1];
      h4 a, b, c, d;
      split_quad<NP>(ldexpf(x0[0], e), ldexpf(x0[1], e), ldexpf(x0[2], e), ldexpf(x0[3], e), a, c);
      split_quad<NP>(ldexpf(x1[0], e), ldexpf(x1[1], e), ldexpf(x1[2], e), ldexpf(x1[3], e), b, d);
      xh[mt] = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
      if constexpr (NP == 2) xl[mt] = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    mma16_step<NP>(acc, xh, xl, wh[t & 1], wl[t & 1]);
  }
}

// ---- transposed-accumulator epilogue pieces ---------------------------------------------------------------------------
// this lane's sample row of m-tile mt:  row0 + 32*mt + (lane&31);  its columns of quad (nt, q): see acc_col
__device__ __forceinline__ int acc_col(int n0, int nt, int q, int hh) { return n0 + 32 * nt + 8 * q + 4 * hh; }

// per-column vector (bias, head weight, per-ray gradient row) in this lane's quads: v[nt][q] = src[acc_col .. +4)
template <int NT>
__device__ __forceinline__ void load_cols(f32x4 (&v)[NT][4], const float* __restrict__ src, int n0, int hh) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) v[nt][q] = *(const f32x4*)&src[acc_col(n0, nt, q, hh)];
}

// Write accumulator values (natural units) into the planes at column offset c0 with scale 2^e: one ds_write_b64 per quad
// and plane.
template <int NP, int W, int MT, int NT>
__device__ __forceinline__ void acc_to_planes(const f32x16 (&acc)[MT][NT], char* Ph, char* Pl, int row0, int n0, int c0,
                                              int e, int lane) {
  const int i = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = row0 + 32 * mt + i;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h4 hi, lo;
        split_quad<NP>(ldexpf(acc[mt][nt][4 * q], e), ldexpf(acc[mt][nt][4 * q + 1], e), ldexpf(acc[mt][nt][4 * q + 2], e),
                       ldexpf(acc[mt][nt][4 * q + 3], e), hi, lo);
        const int o = poff<W>(row, c0 + acc_col(n0, nt, q, hh));
        *(h4*)(Ph + o) = hi;
        if constexpr (NP == 2) *(h4*)(Pl + o) = lo;
      }
  }
}

// Accumulators (natural units) -> row-major fp32 global tile, straight from the registers: 16 bytes per lane and quad (the
// two lane halves of a row make 32 contiguous bytes; the four quads of an n-tile complete the row's 128-byte line).
template <int MT, int NT>
__device__ __forceinline__ void acc_store_global(const f32x16 (&acc)[MT][NT], float* __restrict__ dst, int ldg, int m0,
                                                 int M, int row0, int n0, int lane) {
  const int i = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + row0 + 32 * mt + i;
    if (m < M) {
      float* __restrict__ p = dst + (size_t)m * ldg;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4*)&p[acc_col(n0, nt, q, hh)] =
              f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
    }
  }
}

// v = fma(acc, un, bias[col]) (optionally ReLU) ; bias in this lane's quads
template <bool RELU, int MT, int NT>
__device__ __forceinline__ void acc_fma_bias(f32x16 (&acc)[MT][NT], float un, const f32x4 (&bias)[NT][4]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaf(acc[mt][nt][r], un, bias[nt][r >> 2][r & 3]);
        acc[mt][nt][r] = RELU ? fmaxf(v, 0.0f) : v;
      }
}

// the same with one scale per half of the wave's m-tiles (row halves of the tile that carry different exponents; a wave
// whose rows all lie in one half passes the same value twice)
template <bool RELU, int MT, int NT>
__device__ __forceinline__ void acc_fma_bias_h(f32x16 (&acc)[MT][NT], float un_lo, float un_hi, const f32x4 (&bias)[NT][4]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaf(acc[mt][nt][r], (2 * mt < MT) ? un_lo : un_hi, bias[nt][r >> 2][r & 3]);
        acc[mt][nt][r] = RELU ? fmaxf(v, 0.0f) : v;
      }
}

// v = relu(fma(acc, un, bias)) with the sign bits packed per lane: bit e = (mt*NT + nt)*16 + r of the 64-bit word says
// whether this lane's accumulator element e is positive (2 KiB per 64-row tile and layer instead of re-reading the 64 KiB
// activation tile in the backward pass, which uses the same wave tiling for the gradient of that activation).
template <int MT, int NT>
__device__ __forceinline__ unsigned long long acc_fma_relu_pack(f32x16 (&acc)[MT][NT], float un,
                                                                const f32x4 (&bias)[NT][4]) {
  static_assert(MT * NT * 16 <= 64, "mask word is 64 bits");
  unsigned int lo = 0u, hi = 0u;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int e = (mt * NT + nt) * 16 + r;
        const float v = fmaxf(fmaf(acc[mt][nt][r], un, bias[nt][r >> 2][r & 3]), 0.0f);
        acc[mt][nt][r] = v;
        if (e < 32) lo |= (v > 0.0f) ? (1u << e) : 0u;
        else hi |= (v > 0.0f) ? (1u << (e - 32)) : 0u;
      }
  return ((unsigned long long)hi << 32) | lo;
}

template <int MT, int NT>
__device__ __forceinline__ void acc_scale(f32x16 (&acc)[MT][NT], float un) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= un;
}

// fp32 accumulators -> swizzled fp32 LDS scratch [rows][ldw] (16-byte granules, common.cuh:swz4)
template <int MT, int NT>
__device__ __forceinline__ void acc_to_lds_t(const f32x16 (&acc)[MT][NT], float* Hs, int ldw, int row0, int n0, int lane) {
  const int i = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4*)&Hs[swz4(row0 + 32 * mt + i, acc_col(n0, nt, q, hh), ldw)] =
            f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
}

// hi (+ lo) of 8 consecutive plane elements as fp32
template <int NP, int W>
__device__ __forceinline__ void plane_row8(const char* Ph, const char* Pl, int row, int col, float (&out)[8]) {
  const int o = poff<W>(row, col);
  const h8 vh = *(const h8*)(Ph + o);
  if constexpr (NP == 2) {
    const h8 vl = *(const h8*)(Pl + o);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = (float)vh[j] + (float)vl[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = (float)vh[j];
  }
}

// ---- the 1- and 3-wide heads on the matrix pipe (round 5; nerf.py:89, 98, 108).  rowdot16 below costs a wave ~3k cycles per
// output column of a 128-deep head (weights from L2, 64 conversions and 32 FMAs per lane, the plane row read once per column): the
// three heads of a tile were 18k of a forward kernel's 293k cycles for 0.1 % of its MACs.  Here the heads' weights are split into
// scaled fp16 (hi, lo) ONCE per workgroup in the prologue (head_stage: [768] = w_sigma | W_r2 rows 0..2 | w_csigma, one common
// power-of-two exponent) and a wave contracts ITS 16 rows of the tile against them with v_mfma_f32_16x16x32_f16:
//   D[n][m] = sum_k Wh[n][k] X[m][k]  (+ Wl . Xh + Wh . Xl),   n < NOUT <= 16 head outputs, m = 16 rows, K / 32 blocks;
// the weight fragment (A operand: lane l holds row n = l & 15, eight k of group l >> 4) comes from the LDS staging (lanes n >=
// NOUT hold zeros), the activation fragment (B operand: row m = l & 15, the same eight k) from the planes.  Both operands use
// the same lane -> k assignment, so the sum is over all k whatever k the hardware calls them.  Result: lanes 0..15 hold, in
// acc[0 .. NOUT), the outputs of row  row16 + lane.
#define HEAD_STAGE_N 768  // 256 (w_sigma) + 3 x 128 (W_r2) + 128 (w_csigma)
template <int NP, int W, int K, int NOUT>
__device__ __forceinline__ f32x4 head16_mfma(const char* Ph, const char* Pl, int row16, int c0, const _Float16* sh, const _Float16* sl,
                                             int lane) {
  static_assert(K % 32 == 0 && NOUT <= 4, "one 16 x 16 x 32 block per 32 columns; outputs in the first register quad");
  const int n = lane & 15, g = lane >> 4;
  const bool live = n < NOUT;
  const int nn = live ? n : 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int b = 0; b < K / 32; ++b) {
    const int k = 32 * b + 8 * g;
    h8 wh = *(const h8*)(sh + nn * K + k), wl = *(const h8*)(sl + nn * K + k);
    wh = live ? wh : zero;
    wl = live ? wl : zero;
    const h8 xh = *(const h8*)(Ph + poff<W>(row16 + n, c0 + k));
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc, 0, 0, 0);
    if constexpr (NP == 2) {
      const h8 xl = *(const h8*)(Pl + poff<W>(row16 + n, c0 + k));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc, 0, 0, 0);
    }
  }
  return acc;
}

// dot of plane row segment [c0, c0+K) with w[0..K), split over the TPR adjacent threads that share a row; K is a
// compile-time constant and the weight loads are issued before anything consumes them.
template <int NP, int W, int TPR, int K>
__device__ __forceinline__ float rowdot16(const char* Ph, const char* Pl, int row, int part, int c0,
                                          const float* __restrict__ w, float unscale) {
  constexpr int N8 = K / TPR / 8;
  const int kb = part * (K / TPR);
  f32x4 w0[N8], w1[N8];
#pragma unroll
  for (int q = 0; q < N8; ++q) {
    w0[q] = *(const f32x4*)&w[kb + 8 * q];
    w1[q] = *(const f32x4*)&w[kb + 8 * q + 4];
  }
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < N8; ++q) {
    float v[8];
    plane_row8<NP, W>(Ph, Pl, row, c0 + kb + 8 * q, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s += v[j] * w0[q][j];
      s += v[4 + j] * w1[q][j];
    }
  }
#pragma unroll
  for (int d = 1; d < TPR; d <<= 1) s += __shfl_xor(s, d);
  return s * unscale;
}
