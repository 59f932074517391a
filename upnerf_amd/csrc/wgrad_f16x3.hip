// Weight-gradient contraction on the f16 matrix cores with fp32-level accuracy ("f16x3"):
//     dW[n][k] = sum_m A[m][n] B[m][k],   A = 2^-ea (Ah + Al),  B = 2^-eb (Bh + Bl),   Ah, Al, Bh, Bl in fp16
//     dW ~= 2^-(ea+eb) sum_m (Ah Bh + Ah Bl + Al Bh)                      (the Al Bl term is below 2^-22 relative)
// Three v_mfma_f32_32x32x16_f16 per 32x32x16 block replace eight v_mfma_f32_32x32x2_f32: 5.3x fewer matrix cycles,
// which moves this kernel from MFMA-bound to HBM-bound (it streams 2 KB per sample per layer).  Products of fp16 values
// are exact in fp32 and accumulation is fp32, so the only extra error over the fp32 kernel is the 2^-22 split residue.
// ea, eb are power-of-two exponents chosen by the caller from max|A|, max|B| so that the scaled values peak near 2^14
// (fp16 overflows at 65504; values more than 2^17 below the peak keep fewer than 22 bits, their products are then
// negligible against the peak-sized terms of the same sum).
//
// The reduction index m is the ROW index of both row-major operands, i.e. it is strided in memory, while the f16 MFMA
// wants 8 consecutive m per lane: the tiles are staged in LDS row-major (fp32 -> (hi, lo) fp16 on the way in) and read
// back with ds_read_b64_tr_b16, gfx950's transposing LDS read (4 rows x 16 columns per 16-lane group, delivered
// column-major).  LDS image: 256-byte rows of 128 fp16 with the 16-byte-chunk XOR that makes the transposed reads
// conflict-free (cdna_hip_programming.md T10, image (b)).
#include "common.cuh"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// Rows per staged chunk: two 16-deep MFMA steps, or one for the 256x256 block, whose 128 accumulator registers per wave
// leave room for two register sets only at half the chunk size (same bytes in flight, but continuously).
#ifndef FX_CHUNK_BIG
#define FX_CHUNK_BIG 16
#endif

// byte offset of fp16 element (row, col) inside one plane; col % 4 == 0 for 8-byte accesses
template <int FX_CHUNK>
__device__ __forceinline__ int himg(int row, int col) {
  const int panel = col >> 7, c = col & 127;
  return panel * (FX_CHUNK * 256) + 256 * row + 16 * ((c >> 3) ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 2 * (c & 7);
}

__device__ __forceinline__ h4 tr_read(const char* base, int byteoff) {
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(base + byteoff));
  return __builtin_bit_cast(h4, v);
}

// 512 threads = 8 waves per workgroup, one workgroup per CU, two waves per SIMD: the fp32 -> (hi, lo) conversion of one
// wave overlaps the MFMAs of the other, each thread stages half as many rows (two register sets fit: the loads of chunk
// c+2 are in flight while chunk c is contracted), and each wave keeps at most 128 accumulator registers.
// NP = 2: f16x3 (hi/lo split of both operands, three MFMAs per block, fp32-level accuracy); NP = 1: f16 (operands rounded
// to fp16, one MFMA per block, fp32 accumulate: the "f16" field mode of BASELINE.json configs[3]).
#define FX_THREADS 512
#define WG_LOAD(p) NT_LOAD(p)  // operand rows: read once per step (common.cuh: streaming accesses)
#define WG_OPLOAD(dst, p) dst = WG_LOAD((const f32x4*)(p))
// Round 6: the loads of a register set stay where the source puts them -- right behind the barrier, in front of the contraction.  Left
// to itself hipcc SINKS them (they have no consumer until the next staging) to the END of the contraction: the ISA of round 5's build
// had the requests of set 0 behind the staging of set 1 and `s_waitcnt vmcnt(0)` in front of that staging -- one register set in flight
// instead of two, 32 KB per CU instead of 64.  WG_NO_PIN: the old placement, for A/B builds.
// Measured (alternating runs on one box, profiles/r06_ab_wgrad_pin.txt): the fp16-operand kernels (wgrad_f16p_kernel, configs[3]) gain
// 0.9 % on the Trevi step with two resp. four sets really in flight; the f16x3 kernel (three MFMAs per product) LOSES 0.6 % on the
// headline step -- its stream is held by the clock the part keeps under matrix work + memory traffic (profiles/r06_wgrad_planes.txt),
// not by the bytes in flight, and the sunk requests interleave better with its MFMAs.  So: pinned in wgrad_f16p_kernel
// (WG_PIN_LOADS_P), not in wgrad_f16x3_kernel (WG_PIN_LOADS: -DWG_PIN_X3 turns it on, -DWG_NO_PIN turns both off).
#if defined(WG_PIN_X3) && !defined(WG_NO_PIN)
#define WG_PIN_LOADS __builtin_amdgcn_sched_barrier(0)
#else
#define WG_PIN_LOADS
#endif
#ifdef WG_NO_PIN
#define WG_PIN_LOADS_P
#else
#define WG_PIN_LOADS_P __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef UPNERF_EXP_HALFROW
#define HALFROW_OK(T, c4) (!((T) == 256 && (c4) >= UPNERF_EXP_HALFROW))
#else
#define HALFROW_OK(T, c4) true
#endif
// HASV: a 1-wide head that reads the same B rows rides along (upnerf_wgrad_f16x3_chain_v): vsum[k] += v[m] B[m][k] in fp32 on the
// rows as they pass through the staging registers, sum of v beside it; per-split partials to vslabs[split][K + 4].
template <int NP, int MTW, int NTW, bool HASV = false>
__global__ __launch_bounds__(FX_THREADS, 1) void wgrad_f16x3_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                                  const float* __restrict__ B, int ldb,
                                                                  const int* __restrict__ expo_a, const int* __restrict__ expo_b,
                                                                  float* __restrict__ slabs, float* __restrict__ bslabs,
                                                                  int rows_per_split, upnerf_wgrad_pending prev,
                                                                  const float* __restrict__ vrow = nullptr,
                                                                  float* __restrict__ vslabs = nullptr) {
  constexpr int TN = 64 * MTW, TK = 64 * NTW;
  constexpr int FX_CHUNK = (MTW * NTW == 16) ? FX_CHUNK_BIG : 32;
  constexpr int PN = (TN + 127) / 128, PK = (TK + 127) / 128;           // 128-column panels per plane
  constexpr int SZA = PN * FX_CHUNK * 256, SZB = PK * FX_CHUNK * 256;  // bytes per plane
  constexpr int A4 = FX_CHUNK * TN / 4 / FX_THREADS, B4 = FX_CHUNK * TK / 4 / FX_THREADS;
  static_assert(A4 >= 1 && B4 >= 1, "tile too small for 512 threads");
  constexpr int WK = (TK >= 128) ? 4 : 2, WN = 8 / WK;                  // waves along k / along n
  constexpr int MT = (TN / WN >= 32) ? TN / WN / 32 : 1, NT = TK / WK / 32;  // 32x32 tiles per wave
  constexpr bool HALF = TN / WN < 32;                                   // 64x64 block: only 4 of the 8 waves contract
  // [buffer][A hi | A lo | B hi | B lo]
  __shared__ __attribute__((aligned(16))) char lds[2 * (2 * SZA + 2 * SZB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const bool active = !HALF || wave < 4;
  const int wn = HALF ? ((wave & 3) >> 1) : wave / WK, wk = HALF ? (wave & 1) : wave % WK;
  const int n0 = wn * 32 * MT, k0 = wk * 32 * NT;
  const int split = blockIdx.x;
  const int nblk = blockIdx.y * TN, kblk = blockIdx.z * TK;
  const int mbeg = split * rows_per_split;
  const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  // Prologue: the first prev.rblocks workgroups sum the slabs the PREVIOUS weight-gradient launch left (upnerf_wgrad_f16x3_chain)
  if (prev.nsplit > 0) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wg < prev.rblocks) {
      wgrad_reduce_body(wg, tid, prev, (f32x4(*)[64])lds);
      __syncthreads();
    }
  }
  const int ea = expo_a[0], eb = expo_b[0];
  const float sa = ldexpf(1.0f, ea), sb = ldexpf(1.0f, eb);

  f32x16 acc[MT][NT];
  acc_zero(acc);
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};  // this thread's 4 columns of A, summed over its rows (fp32, unscaled)

  // Loads are UNCONDITIONAL (rows past the split's end are clamped to its last row, columns past the matrix to column 0) and the
  // out-of-range pieces are zeroed when the chunk is staged.  With a branch around every load (`in range ? load : 0`) hipcc
  // cannot count the loads of the two register sets apart and waits vmcnt(0) at the top of the loop -- for the set requested
  // one contraction earlier too, i.e. for a full HBM round trip per two chunks (cdna_hip_programming.md, 'Three .s-level traps'
  // (c); round 4's elimination: 0.381 ms with the wait, 0.276 ms of staging + MFMAs without any load).
  f32x4 ra0[A4], rb0[B4], ra1[A4], rb1[B4];
  float rv0[B4], rv1[B4];           // HASV: v of the B rows in flight
  f32x4 vsum = {0.f, 0.f, 0.f, 0.f};  // HASV: this thread's 4 columns of sum_m v[m] B[m][k]
  float vtot = 0.0f;                // HASV: sum of v over this thread's rows (threads of column group 0 only)
  const int mlast = mend > mbeg ? mend - 1 : (mbeg < M ? mbeg : M - 1);
  auto gload = [&](f32x4 (&ra)[A4], f32x4 (&rb)[B4], float (&rv)[B4], int mc) {
#pragma unroll
    for (int q = 0; q < A4; ++q) {
      const int idx = tid + q * FX_THREADS, row = idx / (TN / 4), c4 = idx - row * (TN / 4);
      const int m = mc + row < mlast ? mc + row : mlast;
      const int col = nblk + 4 * c4 < N ? nblk + 4 * c4 : 0;
      WG_OPLOAD(ra[q], &A[(size_t)m * lda + col]);
    }
#pragma unroll
    for (int q = 0; q < B4; ++q) {
      const int idx = tid + q * FX_THREADS, row = idx / (TK / 4), c4 = idx - row * (TK / 4);
      const int m = mc + row < mlast ? mc + row : mlast;
      const int col = kblk + 4 * c4 < K ? kblk + 4 * c4 : 0;
      WG_OPLOAD(rb[q], &B[(size_t)m * ldb + col]);
      if constexpr (HASV) rv[q] = vrow[m];
    }
  };
  auto split_store = [&](char* hi, char* lo, f32x4 v, float s, int row, int col) {
    // packed forms: v_pk_mul_f32, v_cvt_pk_f16_f32, v_pk_fma_f32 -- 6 vector instructions per two elements
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
    const f2 x0 = f2{v[0], v[1]} * s, x1 = f2{v[2], v[3]} * s;
    const hh2 h0 = __builtin_convertvector(x0, hh2), h1 = __builtin_convertvector(x1, hh2);
    const int off = himg<FX_CHUNK>(row, col);
    *(h4*)(hi + off) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
    if constexpr (NP == 2) {
      const hh2 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f2), hh2);
      const hh2 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f2), hh2);
      *(h4*)(lo + off) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
    }
  };
  // mc: first row of the chunk the set holds (rows >= mend and columns past the matrix are staged as zeros)
  auto lstore = [&](const f32x4 (&ra)[A4], const f32x4 (&rb)[B4], const float (&rv)[B4], int buf, int mc) {
    char* base = lds + buf * (2 * SZA + 2 * SZB);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < A4; ++q) {
      const int idx = tid + q * FX_THREADS, row = idx / (TN / 4), c4 = idx - row * (TN / 4);
      const f32x4 v = (mc + row < mend && nblk + 4 * c4 < N && HALFROW_OK(TN, c4)) ? ra[q] : z4;
      bsum += v;
      split_store(base, base + SZA, v, sa, row, 4 * c4);
    }
#pragma unroll
    for (int q = 0; q < B4; ++q) {
      const int idx = tid + q * FX_THREADS, row = idx / (TK / 4), c4 = idx - row * (TK / 4);
      const f32x4 v = (mc + row < mend && kblk + 4 * c4 < K && HALFROW_OK(TK, c4)) ? rb[q] : z4;
      split_store(base + 2 * SZA, base + 2 * SZA + SZB, v, sb, row, 4 * c4);
      if constexpr (HASV) {
        const float vm = mc + row < mend ? rv[q] : 0.0f;
        vsum += v * vm;
        vtot += c4 == 0 ? vm : 0.0f;
      }
    }
  };
  // transposed-read address pieces of this lane (cdna_hip_programming.md T10): group g, row q, column quad p
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int trow = 8 * (g >> 1) + tq, tcol = 16 * (g & 1) + 4 * tp;

  auto contract = [&](int buf) {
    if (!active) return;
    const char* base = lds + buf * (2 * SZA + 2 * SZB);
#pragma unroll
    for (int kk = 0; kk < FX_CHUNK / 16; ++kk) {
      h8 ah[MT], al[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int o0 = himg<FX_CHUNK>(16 * kk + trow, n0 + 32 * mt + tcol), o1 = himg<FX_CHUNK>(16 * kk + trow + 4, n0 + 32 * mt + tcol);
        const h4 x0 = tr_read(base, o0), x1 = tr_read(base, o1);
        ah[mt] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (NP == 2) {
          const h4 y0 = tr_read(base + SZA, o0), y1 = tr_read(base + SZA, o1);
          al[mt] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int o0 = himg<FX_CHUNK>(16 * kk + trow, k0 + 32 * nt + tcol), o1 = himg<FX_CHUNK>(16 * kk + trow + 4, k0 + 32 * nt + tcol);
        const h4 x0 = tr_read(base + 2 * SZA, o0), x1 = tr_read(base + 2 * SZA, o1);
        const h8 bh = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        h8 bl;
        if constexpr (NP == 2) {
          const h4 y0 = tr_read(base + 2 * SZA + SZB, o0), y1 = tr_read(base + 2 * SZA + SZB, o1);
          bl = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
          if constexpr (NP == 2) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
          }
        }
      }
    }
  };
  // Two register sets (loads of chunk c+2 in flight while chunk c is contracted) where the accumulators leave room for
  // them; the 256x256 block (128 accumulator registers per wave) keeps one set.
  constexpr bool TWO_SETS = MT * NT * 16 <= 64 || FX_CHUNK == 16;
  if constexpr (TWO_SETS) {
    // rows beyond mend are staged as zeros, so an odd number of chunks simply contracts one all-zero chunk
    // (WG_EXP_*: timing experiments with wrong results -- what is left of the launch without the loads / the staging (which is
    // also the only consumer of the loads: nothing waits for them any more) / the MFMAs; tools/bench_wgrad.py)
    // Straight-line on purpose.  Round 5 tried to run the two jobs of an interval (contract the staged chunk | stage the next one)
    // in opposite orders in waves 0-3 and 4-7, so that one wave of a SIMD converts while its partner owns the matrix pipe
    // (MI355X_MICROARCH.md 'Two waves per SIMD' item 9): (a) both orders written out under one wave-uniform branch inside the
    // loop: 225 spilled registers; (b) one copy of each job in a two-trip loop: no spill, but hipcc waits vmcnt(0) at every join
    // -- same launch time as before (0.550 against 0.545 ms stand-alone incl. the reduction); (c) a branch around two whole loops:
    // 47 spills; (d) the loop of (b) with the operand loads hidden in inline asm and hand-counted waits: hipcc re-used the loads'
    // destination registers for the contraction's operand reads while the loads were in flight (seen in the ISA; never run).
    gload(ra0, rb0, rv0, mbeg);
    WG_PIN_LOADS;  // (set 0 is requested BEFORE set 1: the wait in front of the first staging then counts four younger loads)
    gload(ra1, rb1, rv1, mbeg + FX_CHUNK);
    WG_PIN_LOADS;
#pragma unroll 1
    for (int mc = mbeg; mc < mend; mc += 2 * FX_CHUNK) {
#ifndef WG_EXP_NOSTAGE
      lstore(ra0, rb0, rv0, 0, mc);
#endif
      __syncthreads();
#ifndef WG_EXP_NOLOAD
      gload(ra0, rb0, rv0, mc + 2 * FX_CHUNK);
      WG_PIN_LOADS;
#endif
#ifndef WG_EXP_NOMMA
      contract(0);
#endif
#ifndef WG_EXP_NOSTAGE
      lstore(ra1, rb1, rv1, 1, mc + FX_CHUNK);
#endif
      __syncthreads();
#ifndef WG_EXP_NOLOAD
      gload(ra1, rb1, rv1, mc + 3 * FX_CHUNK);
      WG_PIN_LOADS;
#endif
#ifndef WG_EXP_NOMMA
      contract(1);
#endif
    }
  } else {
    int buf = 0;
    gload(ra0, rb0, rv0, mbeg);
#pragma unroll 1
    for (int mc = mbeg; mc < mend; mc += FX_CHUNK) {
      lstore(ra0, rb0, rv0, buf, mc);
      __syncthreads();
      gload(ra0, rb0, rv0, mc + FX_CHUNK);
      contract(buf);
      buf ^= 1;
    }
  }
  // partial slab [split][by][bz][TN][TK], unscaled
  const float unscale = ldexpf(1.0f, -(ea + eb));
  const size_t blk = ((size_t)split * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z;
  float* slab = slabs + blk * TN * TK;
  if (active) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
          slab[n * TK + k0 + 32 * nt + li] = acc[mt][nt][r] * unscale;
        }
  }
  if (bslabs && blockIdx.z == 0) {
    // column sums: thread t owns columns 4*(t % (TN/4)) ..+3; the NTHREADS/(TN/4) threads sharing them meet in LDS
    __syncthreads();
    f32x4* red = (f32x4*)lds;
    red[tid] = bsum;
    __syncthreads();
    constexpr int Q = TN / 4, G = FX_THREADS / Q;
    if (tid < Q) {
      f32x4 s = red[tid];
#pragma unroll
      for (int j = 1; j < G; ++j) s += red[tid + j * Q];
      *(f32x4*)&bslabs[((size_t)split * gridDim.y + blockIdx.y) * TN + 4 * tid] = s;
    }
  }
  if constexpr (HASV) {
    // thread t owns columns 4 * (t % (TK/4)) ..+3 of B; the threads sharing them meet in LDS (fixed order); the sums of v live in
    // the threads of column group 0
    static_assert(TN == 256 && TK == 256, "the vector head rides on the 256 x 256 block");
    __syncthreads();
    f32x4* red = (f32x4*)lds;
    float* redt = (float*)(red + FX_THREADS);
    red[tid] = vsum;
    redt[tid] = vtot;
    __syncthreads();
    constexpr int Q = TK / 4, G = FX_THREADS / Q;
    if (tid < Q) {
      f32x4 s = red[tid];
#pragma unroll
      for (int j = 1; j < G; ++j) s += red[tid + j * Q];
      *(f32x4*)&vslabs[(size_t)split * (TK + 4) + 4 * tid] = s;
    }
    if (tid == 0) {
      float t = redt[0];
#pragma unroll
      for (int j = 1; j < G; ++j) t += redt[j * Q];
      vslabs[(size_t)split * (TK + 4) + TK] = t;
    }
  }
}

// ---- 256 x 256 block on FOUR waves, one per SIMD (round 5; f16x3 arithmetic only) ------------------------------------------------
// The eight-wave kernel above runs its chunk as [stage | barrier | operand reads | MFMAs] with all waves in lockstep: the two
// waves of a SIMD share the matrix pipe while both contract and leave it idle while both convert, and nothing covers the LDS
// latency of the operand reads (round 4's elimination: staging + MFMAs alone 0.276 ms, loads + MFMAs alone 0.222, everything
// 0.381; MFMA busy 0.46).  hipcc would not give the two halves different orders (see that kernel's loop), so this kernel removes
// the sharing instead: ONE wave per SIMD owns the pipe and 512 registers -- a 128 x 128 quarter of the block in 256 accumulator
// registers, TWO operand-fragment sets (the reads of chunk c + 1 are issued in front of the MFMAs of chunk c), two register sets
// of fp32 rows in flight -- and its own instruction stream interleaves the conversion of chunk c + 2 with the MFMAs of chunk c
// (an MFMA holds the vector issue for 8 of its 32 cycles).  Four LDS images (chunk c + 2 is written while c and c + 1 are read),
// one barrier per 16-row chunk.  Same arithmetic, same summation order inside a slab as the eight-wave kernel's (a row's products
// are added in row order either way), so the slabs -- and everything downstream -- are bitwise those of that kernel.
#define W4_THREADS 256
template <int NP>
__global__ __launch_bounds__(W4_THREADS, 1) void wgrad_f16x3_w4_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                                     const float* __restrict__ B, int ldb,
                                                                     const int* __restrict__ expo_a, const int* __restrict__ expo_b,
                                                                     float* __restrict__ slabs, float* __restrict__ bslabs,
                                                                     int rows_per_split, upnerf_wgrad_pending prev) {
  static_assert(NP == 2, "the one-MFMA form has its own kernel family (wgrad_f16p_kernel)");
  constexpr int TN = 256, TK = 256, CH = 16, NBUF = 4;
  constexpr int SZ = 2 * CH * 256;          // bytes per plane (two 128-column panels of 16 rows)
  constexpr int BUF = 4 * SZ;               // [A hi | A lo | B hi | B lo]
  constexpr int R4 = CH * TN / 4 / W4_THREADS;  // 16-byte pieces of A (and of B) per thread and chunk: 4
  constexpr int MT = 4, NT = 4;
  __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF];  // 128 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int n0 = (wave >> 1) * 128, k0 = (wave & 1) * 128;
  const int split = blockIdx.x;
  const int mbeg = split * rows_per_split;
  const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  if (prev.nsplit > 0) {  // the first prev.rblocks workgroups sum the slabs of the previous launch of the run
    const int wg = blockIdx.x;
    if (wg < prev.rblocks) {
      wgrad_reduce_body<4>(wg, tid, prev, (f32x4(*)[64])lds);
      __syncthreads();
    }
  }
  const int ea = expo_a[0], eb = expo_b[0];
  const float sa = ldexpf(1.0f, ea), sb = ldexpf(1.0f, eb);
  f32x16 acc[MT][NT];
  acc_zero(acc);
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const int mlast = mend > mbeg ? mend - 1 : (mbeg < M ? mbeg : M - 1);
  // this thread's pieces: row r0 + 4 q of the chunk, columns 4 c4 .. 4 c4 + 3 (64 threads per 256-column row)
  const int r0 = tid >> 6, c4 = tid & 63;
  const bool cola = 4 * c4 < N, colb = 4 * c4 < K;
  const float* __restrict__ pa = A + (cola ? 4 * c4 : 0);
  const float* __restrict__ pb = B + (colb ? 4 * c4 : 0);
  f32x4 ra[2][R4], rb[2][R4];
  auto gload = [&](f32x4 (&xa)[R4], f32x4 (&xb)[R4], int mc) {
#pragma unroll
    for (int q = 0; q < R4; ++q) {
      const int m = mc + r0 + 4 * q < mlast ? mc + r0 + 4 * q : mlast;
      xa[q] = WG_LOAD((const f32x4*)&pa[(size_t)m * lda]);
      xb[q] = WG_LOAD((const f32x4*)&pb[(size_t)m * ldb]);
    }
  };
  auto split_store = [&](char* hi, char* lo, f32x4 v, float sc, int off) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
    const f2 x0 = f2{v[0], v[1]} * sc, x1 = f2{v[2], v[3]} * sc;
    const hh2 h0 = __builtin_convertvector(x0, hh2), h1 = __builtin_convertvector(x1, hh2);
    *(h4*)(hi + off) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
    const hh2 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f2), hh2);
    const hh2 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f2), hh2);
    *(h4*)(lo + off) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
  };
  auto lstore = [&](const f32x4 (&xa)[R4], const f32x4 (&xb)[R4], int buf, int mc) {
    char* base = lds + buf * BUF;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < R4; ++q) {
      const int row = r0 + 4 * q, off = himg<CH>(row, 4 * c4);
      const bool in = mc + row < mend;
      const f32x4 va = (in && cola) ? xa[q] : z4, vb = (in && colb) ? xb[q] : z4;
      bsum += va;
      split_store(base, base + SZ, va, sa, off);
      split_store(base + 2 * SZ, base + 3 * SZ, vb, sb, off);
    }
  };
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int trow = 8 * (g >> 1) + tq, tcol = 16 * (g & 1) + 4 * tp;
  h8 fah[2][MT], fal[2][MT], fbh[2][NT], fbl[2][NT];
  auto readfrag = [&](h8 (&ah)[MT], h8 (&al)[MT], h8 (&bh)[NT], h8 (&bl)[NT], int buf) {
    const char* base = lds + buf * BUF;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int o0 = himg<CH>(trow, n0 + 32 * t + tcol), o1 = himg<CH>(trow + 4, n0 + 32 * t + tcol);
      ah[t] = __builtin_shufflevector(tr_read(base, o0), tr_read(base, o1), 0, 1, 2, 3, 4, 5, 6, 7);
      al[t] = __builtin_shufflevector(tr_read(base + SZ, o0), tr_read(base + SZ, o1), 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int o0 = himg<CH>(trow, k0 + 32 * t + tcol), o1 = himg<CH>(trow + 4, k0 + 32 * t + tcol);
      bh[t] = __builtin_shufflevector(tr_read(base + 2 * SZ, o0), tr_read(base + 2 * SZ, o1), 0, 1, 2, 3, 4, 5, 6, 7);
      bl[t] = __builtin_shufflevector(tr_read(base + 3 * SZ, o0), tr_read(base + 3 * SZ, o1), 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto mma = [&](const h8 (&ah)[MT], const h8 (&al)[MT], const h8 (&bh)[NT], const h8 (&bl)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
      }
  };
  // chunk c lives in image c % 4, fragment set c % 2; its rows travelled in register set c % 2 (rows past mend are staged as zeros)
  gload(ra[0], rb[0], mbeg);
  gload(ra[1], rb[1], mbeg + CH);
  lstore(ra[0], rb[0], 0, mbeg);
  gload(ra[0], rb[0], mbeg + 2 * CH);
  __syncthreads();
  readfrag(fah[0], fal[0], fbh[0], fbl[0], 0);
  lstore(ra[1], rb[1], 1, mbeg + CH);
  gload(ra[1], rb[1], mbeg + 3 * CH);
  __syncthreads();
#pragma unroll 1
  for (int mc = mbeg; mc < mend; mc += 4 * CH) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // chunk c = (mc - mbeg) / CH + u: its images / sets are compile-time names
      readfrag(fah[(u + 1) & 1], fal[(u + 1) & 1], fbh[(u + 1) & 1], fbl[(u + 1) & 1], (u + 1) & 3);  // chunk c + 1
      mma(fah[u & 1], fal[u & 1], fbh[u & 1], fbl[u & 1]);                                              // chunk c
      lstore(ra[u & 1], rb[u & 1], (u + 2) & 3, mc + (u + 2) * CH);                                      // chunk c + 2
      gload(ra[u & 1], rb[u & 1], mc + (u + 4) * CH);                                                    // chunk c + 4
      __syncthreads();
    }
  }
  const float unscale = ldexpf(1.0f, -(ea + eb));
  float* slab = slabs + (size_t)split * TN * TK;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
        slab[n * TK + k0 + 32 * nt + li] = acc[mt][nt][r] * unscale;
      }
  if (bslabs) {  // column sums of A: thread t owns columns 4 * (t % 64) ..+3 of rows t / 64 + 4 q; the four threads sharing them meet in LDS
    __syncthreads();
    f32x4* red = (f32x4*)lds;
    red[tid] = bsum;
    __syncthreads();
    if (tid < 64) *(f32x4*)&bslabs[(size_t)split * TN + 4 * tid] = ((red[tid] + red[tid + 64]) + red[tid + 128]) + red[tid + 192];
  }
}

// ---- fp16-STORED operands (the f16 field mode, upnerf_wgrad_f16p): A16 (and B16 when PKB) hold fp16 values scaled per
// 64-row tile by 2^exp[m / 64] (the LDS planes of the field kernels, copied out as they stood).  A thread moves 8 columns
// (16 bytes) at a time, brings them to the tensor-wide exponent with an exact fp16 power-of-two multiply and drops them into
// the same LDS image the fp32 path builds; one MFMA per block.  Half the HBM bytes of the fp32-stored operands.
// FRAG: the fp16 operands are the operand fragments of the register-resident field kernels (csrc/field16rr.hip) -- [32-row tile]
// [k-block 16][lane 64][8], feature 16 s + 8 (j / 4) + 4 (lane / 32) + j % 4, one exponent per 32 rows.  A thread moves the 16 bytes
// of one lane: sixteen consecutive threads read sixteen consecutive rows of one (k-block, lane half) = 256 contiguous bytes, and
// the two 8-byte halves of the piece land at their natural columns of the same LDS image.  256-wide operands only.
// NP = 2 ("24-bit" storage, upnerf_wgrad_f24p): beside every fp16 tensor a byte tensor of the same shape holds the rounding
// residual the field kernels' lo plane held, quantised to 1/32 of the tile's scaled unit (byte = round(32 lo) + 128): the
// operand is hi + lo again to 2^-20 of its tile's maximum, three MFMAs per block as in the f16x3 kernel; a fp32 B operand is
// split into hi + lo here.  3 bytes per stored element instead of 4.  Row-major operands only (no FRAG).
// HASV (fragment-ordered 256 x 256 problems): a 1-wide head that reads the same B rows rides along (upnerf_wgrad_f16p_chain_v),
// as in wgrad_f16x3_kernel: vsum[col] += v[m] * B[m][col] in fp32 from the pieces as they pass, per-split partials to vslabs.
template <int MTW, int NTW, int PKB, int FRAG, int NP, bool HASV = false>
__global__ __launch_bounds__(FX_THREADS, 1) void wgrad_f16p_kernel(int M, int N, int K, const uint16_t* __restrict__ A, int lda,
                                                                 const int* __restrict__ aexp, const void* __restrict__ Bv, int ldb,
                                                                 const int* __restrict__ bexp, const int* __restrict__ expo_a,
                                                                 const int* __restrict__ expo_b, float* __restrict__ slabs,
                                                                 float* __restrict__ bslabs, int rows_per_split,
                                                                 upnerf_wgrad_pending prev, const uint8_t* __restrict__ Alo,
                                                                 const uint8_t* __restrict__ Blo, const float* __restrict__ vrow = nullptr,
                                                                 float* __restrict__ vslabs = nullptr) {
  static_assert(!HASV || (FRAG && PKB && NP == 1 && MTW == 4 && NTW == 4), "the vector head rides on fragment-ordered 256 x 256 problems");
  constexpr int TN = 64 * MTW, TK = 64 * NTW;
  constexpr int FX_CHUNK = (MTW * NTW == 16) ? FX_CHUNK_BIG : 32;
  constexpr int PN = (TN + 127) / 128, PK = (TK + 127) / 128;
  constexpr int SZA = PN * FX_CHUNK * 256, SZB = PK * FX_CHUNK * 256;
  constexpr int A8 = FX_CHUNK * TN / 8 / FX_THREADS;                       // 16-byte pieces of A per thread
  constexpr int B8 = PKB ? FX_CHUNK * TK / 8 / FX_THREADS : 0;             // ... of a packed B
  constexpr int B4 = PKB ? 0 : FX_CHUNK * TK / 4 / FX_THREADS;             // 16-byte pieces of an fp32 B
  static_assert(A8 >= 1 && (B8 >= 1 || B4 >= 1), "tile too small for 512 threads");
  constexpr int WK = (TK >= 128) ? 4 : 2, WN = 8 / WK;
  constexpr int MT = (TN / WN >= 32) ? TN / WN / 32 : 1, NT = TK / WK / 32;
  static_assert(TN / WN >= 32, "packed variant is built for 256-row blocks");
  static_assert(!FRAG || ((TN == 256 || TN == 128) && (!PKB || TK == TN) && FX_CHUNK % 16 == 0), "fragment-ordered operands are 256 or 128 wide");
  // fragment-ordered operands: a row has TN / 8 (TK / 8) 16-byte pieces, the 512 threads cover RPA (RPB) rows per pass
  constexpr int RPA = FX_THREADS / (TN / 8), RPB = PKB ? FX_THREADS / (TK / 8) : 1;
  static_assert(!FRAG || (FX_CHUNK % RPA == 0 && (!PKB || FX_CHUNK % RPB == 0)), "whole passes per chunk");
  static_assert(NP == 1 || !FRAG, "hi + lo8 operands are row-major");
  constexpr int BUF = NP * (SZA + SZB);                                 // [buffer][A hi | A lo | B hi | B lo] (NP = 1: no lo planes)
  constexpr int OB = NP * SZA;                                          // B planes start here
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int wn = wave / WK, wk = wave % WK;
  const int n0 = wn * 32 * MT, k0 = wk * 32 * NT;
  const int split = blockIdx.x;
  const int nblk = blockIdx.y * TN, kblk = blockIdx.z * TK;
  const int mbeg = split * rows_per_split;
  const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  // Prologue: the first prev.rblocks workgroups sum the slabs the PREVIOUS weight-gradient launch left (upnerf_wgrad_f16p_chain)
  if (prev.nsplit > 0) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wg < prev.rblocks) {
      wgrad_reduce_body(wg, tid, prev, (f32x4(*)[64])lds);
      __syncthreads();
    }
  }
  const int ea = expo_a[0], eb = expo_b[0];
  const float* __restrict__ Bf = (const float*)Bv;
  const uint16_t* __restrict__ Bh = (const uint16_t*)Bv;

  f32x16 acc[MT][NT];
  acc_zero(acc);
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // this thread's 8 columns of A, summed over its rows (natural units)

  // NS register sets: the loads of chunk c + NS are requested when chunk c has been staged.  Two sets (one loop iteration, ~2 us)
  // left the kernel waiting for HBM latency (with nothing waiting for the loads it runs at 7.3 TB/s); the fp16
  // operands are small enough for four (10 registers per set at the 256 x 256 block).
#ifndef WG_P_SETS
#define WG_P_SETS 4
#endif
  constexpr int NS = (NP == 1 && !HASV) ? WG_P_SETS : 2;  // (HASV: eight running sums more; four sets spilled 23 registers)
  static_assert(NS % 2 == 0, "the LDS image is double buffered: set u goes to buffer u & 1");
  h8 ra[NS][A8], rbp[NS][B8 ? B8 : 1];
  f32x4 rbf[NS][B4 ? B4 : 1];
  int xa[NS][A8], xb[NS][B8 ? B8 : 1];  // tile exponents of the rows just loaded
  float rvv[NS][B8 ? B8 : 1];           // HASV: v of the B rows in flight
  float vsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, vtot = 0.0f;  // HASV: this thread's 8 columns of sum_m v[m] B[m][.]; sum of v
  u32x2_t la[NS][A8], lb[NS][B8 ? B8 : 1];  // NP = 2: the residual bytes of the same pieces
  const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const u32x2_t mid8 = {0x80808080u, 0x80808080u};  // (byte 128 = residual 0)
  // loads are unconditional (clamped rows / columns; see wgrad_f16x3_kernel), out-of-range pieces are zeroed at staging
  const int mlast = mend > mbeg ? mend - 1 : (mbeg < M ? mbeg : M - 1);
  auto gload = [&](h8 (&ra)[A8], int (&xa)[A8], h8 (&rbp)[B8 ? B8 : 1], int (&xb)[B8 ? B8 : 1], f32x4 (&rbf)[B4 ? B4 : 1],
                   u32x2_t (&la)[A8], u32x2_t (&lb)[B8 ? B8 : 1], float (&rv)[B8 ? B8 : 1], int mc) {
#pragma unroll
    for (int q = 0; q < A8; ++q) {
      if constexpr (FRAG) {
        const int mr = mc + (tid % RPA) + RPA * q, m = mr < mlast ? mr : mlast, s2 = tid / RPA;  // row of the chunk; (k-block, lane half) of this thread
        ra[q] = NT_LOAD((const h8*)((const char*)A + (((size_t)(m >> 5) * (TN / 16) + (s2 >> 1)) * 64 + (s2 & 1) * 32 + (m & 31)) * 16));
        xa[q] = aexp[m >> 5];
      } else {
        const int idx = tid + q * FX_THREADS, row = idx / (TN / 8), c8 = idx - row * (TN / 8);
        const int m = mc + row < mlast ? mc + row : mlast;
        const int col = nblk + 8 * c8 < N ? nblk + 8 * c8 : 0;
        ra[q] = NT_LOAD((const h8*)&A[(size_t)m * lda + col]);
        if constexpr (NP == 2) la[q] = NT_LOAD((const u32x2_t*)&Alo[(size_t)m * lda + col]);
        xa[q] = aexp[m >> 6];
      }
    }
    if constexpr (PKB) {
#pragma unroll
      for (int q = 0; q < B8; ++q) {
        if constexpr (FRAG) {
          const int mr = mc + (tid % RPB) + RPB * q, m = mr < mlast ? mr : mlast, s2 = tid / RPB;
          rbp[q] = NT_LOAD((const h8*)((const char*)Bh + (((size_t)(m >> 5) * (TK / 16) + (s2 >> 1)) * 64 + (s2 & 1) * 32 + (m & 31)) * 16));
          xb[q] = bexp[m >> 5];
          if constexpr (HASV) rv[q] = vrow[m];
        } else {
          const int idx = tid + q * FX_THREADS, row = idx / (TK / 8), c8 = idx - row * (TK / 8);
          const int m = mc + row < mlast ? mc + row : mlast;
          const int col = kblk + 8 * c8 < K ? kblk + 8 * c8 : 0;
          rbp[q] = NT_LOAD((const h8*)&Bh[(size_t)m * ldb + col]);
          if constexpr (NP == 2) lb[q] = NT_LOAD((const u32x2_t*)&Blo[(size_t)m * ldb + col]);
          xb[q] = bexp[m >> 6];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < B4; ++q) {
        const int idx = tid + q * FX_THREADS, row = idx / (TK / 4), c4 = idx - row * (TK / 4);
        const int m = mc + row < mlast ? mc + row : mlast;
        const int col = kblk + 4 * c4 < K ? kblk + 4 * c4 : 0;
        rbf[q] = NT_LOAD((const f32x4*)&Bf[(size_t)m * ldb + col]);
      }
    }
  };
  auto pw2h = [](int e) {  // 2^e as fp16 (e clamped to what fp16 holds; far-below-maximum tiles flush towards zero)
    e = e > 15 ? 15 : (e < -24 ? -24 : e);
    return (_Float16)ldexpf(1.0f, e);
  };
  // eight residual bytes -> fp16 residuals in units of 2^-e of the tensor: (byte - 128) / 32 * f, f = 2^(e_tensor - e_tile)
  auto lo8 = [](u32x2_t w, _Float16 f) {
    typedef unsigned short us8 __attribute__((ext_vector_type(8)));
    us8 u;
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] = (unsigned short)((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
    h8 v = __builtin_convertvector(u, h8);
    const _Float16 s = f * (_Float16)0.03125f, o = f * (_Float16)-4.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * s + o;
    return v;
  };
  // mc: first row of the chunk the set holds (rows >= mend and columns past the matrix are staged as zeros)
  auto lstore = [&](const h8 (&ra_)[A8], const int (&xa)[A8], const h8 (&rbp_)[B8 ? B8 : 1], const int (&xb)[B8 ? B8 : 1],
                    const f32x4 (&rbf_)[B4 ? B4 : 1], const u32x2_t (&la_)[A8], const u32x2_t (&lb_)[B8 ? B8 : 1],
                    const float (&rv)[B8 ? B8 : 1], int buf, int mc) {
    char* base = lds + buf * BUF;
    // the masks (what the conditional loads used to deliver as zeros)
    h8 ra[A8], rbp[B8 ? B8 : 1];
    f32x4 rbf[B4 ? B4 : 1];
    u32x2_t la[A8], lb[B8 ? B8 : 1];
#pragma unroll
    for (int q = 0; q < A8; ++q) {
      const int idx = tid + q * FX_THREADS;
      const bool ok = FRAG ? (mc + (tid % RPA) + RPA * q < mend) : (mc + idx / (TN / 8) < mend && nblk + 8 * (idx % (TN / 8)) < N);
      ra[q] = ok ? ra_[q] : zero8;
      la[q] = ok ? la_[q] : mid8;
    }
#pragma unroll
    for (int q = 0; q < B8; ++q) {
      const int idx = tid + q * FX_THREADS;
      const bool ok = FRAG ? (mc + (tid % RPB) + RPB * q < mend) : (mc + idx / (TK / 8) < mend && kblk + 8 * (idx % (TK / 8)) < K);
      rbp[q] = ok ? rbp_[q] : zero8;
      lb[q] = ok ? lb_[q] : mid8;
    }
#pragma unroll
    for (int q = 0; q < B4; ++q) {
      const int idx = tid + q * FX_THREADS;
      rbf[q] = (mc + idx / (TK / 4) < mend && kblk + 4 * (idx % (TK / 4)) < K) ? rbf_[q] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < A8; ++q) {
      const int idx = tid + q * FX_THREADS, row = idx / (TN / 8), c8 = idx - row * (TN / 8);
      const float inv = ldexpf(1.0f, -xa[q]);
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum[j] += (float)ra[q][j] * inv;
      const _Float16 f = pw2h(ea - xa[q]);
      h8 v = ra[q];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] * f;
      if constexpr (NP == 2) {
        const h8 vl = lo8(la[q], (_Float16)1.0f);
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += (float)vl[j] * inv;
        *(h8*)(base + SZA + himg<FX_CHUNK>(row, 8 * c8)) = lo8(la[q], f);
      }
      if constexpr (FRAG) {
        const int r = (tid % RPA) + RPA * q, s2 = tid / RPA, col = 16 * (s2 >> 1) + 4 * (s2 & 1);
        *(h4*)(base + himg<FX_CHUNK>(r, col)) = __builtin_shufflevector(v, v, 0, 1, 2, 3);
        *(h4*)(base + himg<FX_CHUNK>(r, col + 8)) = __builtin_shufflevector(v, v, 4, 5, 6, 7);
      } else {
        *(h8*)(base + himg<FX_CHUNK>(row, 8 * c8)) = v;
      }
    }
    if constexpr (PKB) {
#pragma unroll
      for (int q = 0; q < B8; ++q) {
        const int idx = tid + q * FX_THREADS, row = idx / (TK / 8), c8 = idx - row * (TK / 8);
        const _Float16 f = pw2h(eb - xb[q]);
        h8 v = rbp[q];
        if constexpr (HASV) {  // natural units: fp16 value * 2^-tile exponent, times the row's v (zero past mend: rbp is masked)
          const bool in = mc + (tid % RPB) + RPB * q < mend;
          const float sc = in ? ldexpf(rv[q], -xb[q]) : 0.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) vsum[j] = fmaf((float)v[j], sc, vsum[j]);
          vtot += (in && tid < RPB) ? rv[q] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * f;
        if constexpr (FRAG) {
          const int r = (tid % RPB) + RPB * q, s2 = tid / RPB, col = 16 * (s2 >> 1) + 4 * (s2 & 1);
          *(h4*)(base + OB + himg<FX_CHUNK>(r, col)) = __builtin_shufflevector(v, v, 0, 1, 2, 3);
          *(h4*)(base + OB + himg<FX_CHUNK>(r, col + 8)) = __builtin_shufflevector(v, v, 4, 5, 6, 7);
        } else {
          *(h8*)(base + OB + himg<FX_CHUNK>(row, 8 * c8)) = v;
          if constexpr (NP == 2) *(h8*)(base + OB + SZB + himg<FX_CHUNK>(row, 8 * c8)) = lo8(lb[q], f);
        }
      }
    } else {
      const float sb = ldexpf(1.0f, eb);
#pragma unroll
      for (int q = 0; q < B4; ++q) {
        const int idx = tid + q * FX_THREADS, row = idx / (TK / 4), c4 = idx - row * (TK / 4);
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
        const f2 x0 = f2{rbf[q][0], rbf[q][1]} * sb, x1 = f2{rbf[q][2], rbf[q][3]} * sb;
        const hh2 h0 = __builtin_convertvector(x0, hh2), h1 = __builtin_convertvector(x1, hh2);
        *(h4*)(base + OB + himg<FX_CHUNK>(row, 4 * c4)) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
        if constexpr (NP == 2) {  // a fp32 operand (the encoding): split here, as the f16x3 kernel does
          const hh2 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f2), hh2);
          const hh2 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f2), hh2);
          *(h4*)(base + OB + SZB + himg<FX_CHUNK>(row, 4 * c4)) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
        }
      }
    }
  };
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int trow = 8 * (g >> 1) + tq, tcol = 16 * (g & 1) + 4 * tp;
  auto contract = [&](int buf) {
    const char* base = lds + buf * BUF;
#pragma unroll
    for (int kk = 0; kk < FX_CHUNK / 16; ++kk) {
      h8 ah[MT], al[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int o0 = himg<FX_CHUNK>(16 * kk + trow, n0 + 32 * mt + tcol), o1 = himg<FX_CHUNK>(16 * kk + trow + 4, n0 + 32 * mt + tcol);
        const h4 x0 = tr_read(base, o0), x1 = tr_read(base, o1);
        ah[mt] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (NP == 2) {
          const h4 y0 = tr_read(base + SZA, o0), y1 = tr_read(base + SZA, o1);
          al[mt] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int o0 = himg<FX_CHUNK>(16 * kk + trow, k0 + 32 * nt + tcol), o1 = himg<FX_CHUNK>(16 * kk + trow + 4, k0 + 32 * nt + tcol);
        const h4 x0 = tr_read(base + OB, o0), x1 = tr_read(base + OB, o1);
        const h8 bh = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        h8 bl;
        if constexpr (NP == 2) {
          const h4 y0 = tr_read(base + OB + SZB, o0), y1 = tr_read(base + OB + SZB, o1);
          bl = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
          if constexpr (NP == 2) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
          }
        }
      }
    }
  };
  // two register sets: the loads of chunk c+2 are in flight while chunk c is contracted (rows beyond mend load as zeros)
  // NS register sets: the loads of chunk c + NS are in flight while chunk c is contracted (rows beyond mend are staged as zeros)
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    gload(ra[u], xa[u], rbp[u], xb[u], rbf[u], la[u], lb[u], rvv[u], mbeg + u * FX_CHUNK);
    WG_PIN_LOADS_P;  // (the sets are requested in order, and stay where the source puts them: see WG_PIN_LOADS)
  }
#pragma unroll 1
  for (int mc = mbeg; mc < mend; mc += NS * FX_CHUNK) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      lstore(ra[u], xa[u], rbp[u], xb[u], rbf[u], la[u], lb[u], rvv[u], u & 1, mc + u * FX_CHUNK);
      __syncthreads();
      gload(ra[u], xa[u], rbp[u], xb[u], rbf[u], la[u], lb[u], rvv[u], mc + (NS + u) * FX_CHUNK);
      WG_PIN_LOADS_P;
      contract(u & 1);
    }
  }
  const float unscale = ldexpf(1.0f, -(ea + eb));
  const size_t blk = ((size_t)split * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z;
  float* slab = slabs + blk * TN * TK;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
        slab[n * TK + k0 + 32 * nt + li] = acc[mt][nt][r] * unscale;
      }
  if (bslabs && blockIdx.z == 0) {
    // column sums: thread t owns columns 8*(t % (TN/8)) ..+7; the threads sharing them meet in LDS
    __syncthreads();
    float* red = (float*)lds;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bsum[j];
    __syncthreads();
    constexpr int Q = TN / 8, G = FX_THREADS / Q;
    if (tid < TN) {
      float sacc = 0.0f;
      if constexpr (FRAG) {  // column tid = 16 s + 8 (j / 4) + 4 hh + j % 4 is held, as element j, by the RPA threads with tid / RPA = 2 s + hh
        const int s2 = 2 * (tid >> 4) + ((tid >> 2) & 1), j = 4 * ((tid >> 3) & 1) + (tid & 3);
#pragma unroll
        for (int t = 0; t < RPA; ++t) sacc += red[(s2 * RPA + t) * 8 + j];
      } else {
        const int grp = tid >> 3, j = tid & 7;
#pragma unroll
        for (int t = 0; t < G; ++t) sacc += red[(grp + t * Q) * 8 + j];
      }
      bslabs[((size_t)split * gridDim.y + blockIdx.y) * TN + tid] = sacc;
    }
  }
  if constexpr (HASV) {  // column tid = 16 s + 8 (j / 4) + 4 hh + j % 4 is held, as element j, by the RPB threads with tid / RPB = 2 s + hh
    __syncthreads();
    float* red = (float*)lds;
    float* redt = red + 8 * FX_THREADS;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = vsum[j];
    if (tid < RPB) redt[tid] = vtot;
    __syncthreads();
    if (tid < TK) {
      const int s2 = 2 * (tid >> 4) + ((tid >> 2) & 1), j = 4 * ((tid >> 3) & 1) + (tid & 3);
      float sacc = 0.0f;
#pragma unroll
      for (int t = 0; t < RPB; ++t) sacc += red[(s2 * RPB + t) * 8 + j];
      vslabs[(size_t)split * (TK + 4) + tid] = sacc;
    }
    if (tid == 0) {
      float t = 0.0f;
#pragma unroll
      for (int r = 0; r < RPB; ++r) t += redt[r];
      vslabs[(size_t)split * (TK + 4) + TK] = t;
    }
  }
}


// ---- 256 x 256 block from PRODUCER-SPLIT operands, staged by LDS-DMA (round 6) -----------------------------------------------------
// Both operands arrive as the (hi, lo) fp16 planes the f16x3 field kernels hold in LDS -- value = (hi + lo) * 2^-e, one exponent per
// 64-row tile (upnerf_field_fwd_args.h16 / h_lo16 / hexp, upnerf_field_bwd_args.gz16 / gz_lo16 / gzexp): the SAME 4 bytes per element
// as the fp32 rows (which are exactly hi + lo), but already split.  So this kernel has no conversion pass and no staging registers:
// every wave copies its pieces of a 16-row chunk straight into the swizzled LDS image with global_load_lds_dwordx4 -- the XOR of the
// image (himg) is applied on the SOURCE side, through the per-lane global address -- four chunks in a ring, three in flight (96 KB
// per CU requested ahead of the matrix work, against 64 KB in registers in wgrad_f16x3_kernel), ONE counted s_waitcnt vmcnt + ONE
// s_barrier per chunk.  The tile exponents are folded into the B fragments after the transposed read (v_pk_mul_f16 by a power of
// two: exact unless the product sinks below fp16's range, i.e. is 2^-38 of the tensor's maximum), so the MFMA inputs are the ones
// wgrad_f16x3_kernel builds from the fp32 rows: (hi + lo) 2^k splits into hi 2^k and lo 2^k.  Column sums of A (the bias gradient)
// come from the A fragments: each of the four waves that share an n-range sums one of its four n-tiles.  M-range of a split: whole
// 16-row chunks, whole 64-row tiles of exponents (the caller guarantees M % 64 == 0 and rows_per_split % 64 == 0).
#define WP_CH 16
#define WP_SLOTS 5   // the CU's whole 160 KB: four chunks in flight (the stream is LATENCY-bound: period = (latency + re-issue delay) / chunks in flight)
#define WP_SLOT (4 * 8192)  // [A hi | A lo | B hi | B lo], each [2 panels][16 rows][256 bytes]
__global__ __launch_bounds__(FX_THREADS, 1) void wgrad_planes_kernel(int M, const uint16_t* __restrict__ Ah, const uint16_t* __restrict__ Al,
                                                                   const int* __restrict__ aexp, const uint16_t* __restrict__ Bh,
                                                                   const uint16_t* __restrict__ Bl, const int* __restrict__ bexp,
                                                                   const int* __restrict__ expo_a, const int* __restrict__ expo_b,
                                                                   float* __restrict__ slabs, float* __restrict__ bslabs,
                                                                   int rows_per_split, upnerf_wgrad_pending prev) {
  constexpr int TN = 256, TK = 256, MT = 4, NT = 2, SZ = 8192;
  __shared__ __attribute__((aligned(16))) char lds[WP_SLOTS * WP_SLOT];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int wn = wave >> 2, wk = wave & 3;
  const int n0 = wn * 32 * MT, k0 = wk * 32 * NT;
  const int split = blockIdx.x;
  const int mbeg = split * rows_per_split;
  const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  if (prev.nsplit > 0) {
    if ((int)blockIdx.x < prev.rblocks) {
      wgrad_reduce_body(blockIdx.x, tid, prev, (f32x4(*)[64])lds);
      __syncthreads();
    }
  }
  const int ea = expo_a[0], eb = expo_b[0];
  f32x16 acc[MT][NT];
  acc_zero(acc);
  float bacc = 0.0f;  // column sum of A over this split, column n0 + 32 wk + li (both lane halves hold a part)
  const int nchunk = mend > mbeg ? (mend - mbeg) / WP_CH : 0;
  // this wave's pieces of a chunk: plane p = wave / 2 (A hi, A lo, B hi, B lo), panel = wave % 2, four 1 KiB pieces of four rows.
  // Piece rg, lane l: image row 4 rg + l / 16, position l % 16 of the row's sixteen 16-byte chunks, which holds logical chunk
  // position ^ (((row & 3) << 2) | (row >> 2))  (himg).
  const int p = wave >> 1, panel = wave & 1;
  const char* __restrict__ src = (const char*)(p == 0 ? Ah : (p == 1 ? Al : (p == 2 ? Bh : Bl)));
  int soff[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int row = 4 * rg + (lane >> 4), pos = lane & 15;
    soff[rg] = row * 512 + panel * 256 + ((pos ^ (((row & 3) << 2) | (row >> 2))) << 4);
  }
  auto dma_piece = [&](int c, int rg) {  // piece rg of chunk c (see dma)
    const int cc = c < nchunk ? c : nchunk - 1;
    const char* a = src + (size_t)(mbeg + cc * WP_CH) * 512 + soff[rg];
    const unsigned d = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + (c % WP_SLOTS) * WP_SLOT + p * SZ + panel * 4096;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(d + rg * 1024), "v"(a) : "memory", "m0");
  };
  auto dma = [&](int c) {  // chunk c of this split into slot c % 4 (past the end: the last chunk again -- the waits count instructions)
    const int cc = c < nchunk ? c : nchunk - 1;
    const char* g = src + (size_t)(mbeg + cc * WP_CH) * 512;
    // Issued as inline asm ON PURPOSE: behind the builtin hipcc's wait-count pass knows that LDS is being written by a vector-memory
    // operation and -- having no alias information for the transposing reads below -- puts `s_waitcnt vmcnt(0)` in front of them:
    // every chunk then waits for all three in flight (seen in the ISA of the first build).  The counted wait at the top of the
    // loop + the barrier are the synchronisation; M0 = LDS byte address of the piece (lane l lands at M0 + 16 l).
    const unsigned d = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + (c % WP_SLOTS) * WP_SLOT + p * SZ + panel * 4096;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const char* a = g + soff[rg];
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(d + rg * 1024), "v"(a) : "memory", "m0");
    }
  };
  const int g4 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int trow = 8 * (g4 >> 1) + tq, tcol = 16 * (g4 & 1) + 4 * tp;
  if (nchunk > 0) {
    __syncthreads();  // (the reduction prologue is done with the LDS)
    dma(0);
    dma(1);
    dma(2);
    dma(3);
    int nxa = aexp[mbeg >> 6], nxb = bexp[mbeg >> 6];  // tile exponents of the next chunk: requested one chunk ahead (scalar loads)
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
      const int xa = nxa, xb = nxb;
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // this wave's four pieces of chunk c have landed (chunks c + 1 .. c + 3 may fly)
      __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): its reads of chunk c - 1 are done
      __builtin_amdgcn_s_barrier();                      // everybody's pieces have landed; slot (c - 1) % 5 is free
      asm volatile("" ::: "memory");
      dma(c + 4);                                        // at once: what the stream waits for is the request, not the matrix work
      {
        const int mn = mbeg + (c + 1 < nchunk ? c + 1 : c) * WP_CH;
        nxa = aexp[mn >> 6];
        nxb = bexp[mn >> 6];
      }
#ifdef WP_EXP_NOREAD  // timing experiment (wrong results): the DMA stream and the per-chunk synchronisation alone
      continue;
#endif
      const char* base = lds + (c % WP_SLOTS) * WP_SLOT;
      // 2^((ea - xa) + (eb - xb)) as fp16: both tile exponents folded into the B fragments
      int fe = (ea - xa) + (eb - xb);
      fe = fe > 15 ? 15 : (fe < -24 ? -24 : fe);
      const _Float16 f = (_Float16)ldexpf(1.0f, fe);
      h8 ah[MT], al[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int o0 = himg<WP_CH>(trow, n0 + 32 * mt + tcol), o1 = himg<WP_CH>(trow + 4, n0 + 32 * mt + tcol);
        const h4 x0 = tr_read(base, o0), x1 = tr_read(base, o1);
        ah[mt] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        const h4 y0 = tr_read(base + SZ, o0), y1 = tr_read(base + SZ, o1);
        al[mt] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      h8 bhv[NT], blv[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int o0 = himg<WP_CH>(trow, k0 + 32 * nt + tcol), o1 = himg<WP_CH>(trow + 4, k0 + 32 * nt + tcol);
        const h4 x0 = tr_read(base + 2 * SZ, o0), x1 = tr_read(base + 2 * SZ, o1);
        bhv[nt] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        const h4 y0 = tr_read(base + 3 * SZ, o0), y1 = tr_read(base + 3 * SZ, o1);
        blv[nt] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      __builtin_amdgcn_sched_barrier(0);
      // bias: the wave sums n-tile wk of its range (the other three waves of the range hold the same fragments)
      {
        float s = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          if (mt == wk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)ah[mt][j] + (float)al[mt][j];
          }
        bacc = fmaf(s, ldexpf(1.0f, -xa), bacc);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        h8 bh = bhv[nt], bl = blv[nt];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bh[j] = bh[j] * f;
          bl[j] = bl[j] * f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#ifndef WP_EXP_NOMMA  // timing experiment (wrong results): everything but the matrix work
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);
#else
          acc[mt][nt][0] += (float)ah[mt][0] * (float)bh[0] + (float)al[mt][1] * (float)bl[1];
#ifdef WP_EXP_SLEEP  // ... and the wave idles for the time its three MFMAs would hold the pipe when it has it to itself (96 cycles)
          __builtin_amdgcn_s_sleep(1);
          asm volatile("s_nop 15\n\ts_nop 15");
#endif
#endif
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus pieces of the tail
  }
  const float unscale = ldexpf(1.0f, -(ea + eb));
  float* slab = slabs + (size_t)split * TN * TK;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
        slab[n * TK + k0 + 32 * nt + li] = acc[mt][nt][r] * unscale;
      }
  if (bslabs) {
    bacc += __shfl_xor(bacc, 32);  // the two k halves of the fragment
    if (lane < 32) bslabs[(size_t)split * TN + n0 + 32 * wk + li] = bacc;
  }
}

template <int MTW, int NTW, int PKB, int FRAG, int NP = 1>
int launch_p(int M, int N, int K, const uint16_t* A, int lda, const int* aexp, const void* B, int ldb, const int* bexp,
             const int* expo_a, const int* expo_b, float* slabs, float* bslabs, int nsplit, int rows, hipStream_t st,
             const upnerf_wgrad_pending* prevp, const uint8_t* Alo = nullptr, const uint8_t* Blo = nullptr) {
  constexpr int TN = 64 * MTW, TK = 64 * NTW;
  dim3 grid(nsplit, (N + TN - 1) / TN, (K + TK - 1) / TK);
  upnerf_wgrad_pending prev = {};
  if (prevp) prev = *prevp;
  hipLaunchKernelGGL((wgrad_f16p_kernel<MTW, NTW, PKB, FRAG, NP>), grid, dim3(FX_THREADS), 0, st, M, N, K, A, lda, aexp, B, ldb, bexp,
                     expo_a, expo_b, slabs, bslabs, rows, prev, Alo, Blo);
  return (int)hipGetLastError();
}

#ifndef WG_NO_W4
// 0 (make variant EXP=-DWG_NO_W4=0): the 256 x 256 block on wgrad_f16x3_w4_kernel (four waves, one per SIMD).  Built in round 5,
// bitwise the eight-wave kernel's slabs (tests/test_hip_kernels.py runs green on either), measured SLOWER: 0.406 / 0.146 ms against
// 0.366 / 0.135 ms stand-alone (786 432 / 262 144 rows, launch + reduction, tools/bench_wgrad.py) and 239.8 k against 244.1 k rays/s
// on the step: a single in-order wave per SIMD hides neither the operand reads' LDS latency nor the conversion under its own MFMAs
// as well as two waves hide them for each other.  1 = the eight-wave kernel (shipped).
#define WG_NO_W4 1
#endif
template <int MTW, int NTW>
int launch(int planes, int M, int N, int K, const float* A, int lda, const float* B, int ldb, const int* expo_a, const int* expo_b,
           float* slabs, float* bslabs, int nsplit, int rows, hipStream_t st, const upnerf_wgrad_pending* prevp = nullptr) {
  constexpr int TN = 64 * MTW, TK = 64 * NTW;
  dim3 grid(nsplit, (N + TN - 1) / TN, (K + TK - 1) / TK);
  upnerf_wgrad_pending prev = {};
  if (prevp) prev = *prevp;
  if (planes == 1)
    hipLaunchKernelGGL((wgrad_f16x3_kernel<1, MTW, NTW>), grid, dim3(FX_THREADS), 0, st, M, N, K, A, lda, B, ldb, expo_a, expo_b,
                       slabs, bslabs, rows, prev);
  else
    hipLaunchKernelGGL((wgrad_f16x3_kernel<2, MTW, NTW>), grid, dim3(FX_THREADS), 0, st, M, N, K, A, lda, B, ldb, expo_a, expo_b,
                       slabs, bslabs, rows, prev);
  return (int)hipGetLastError();
}

}  // namespace

// Same contract as upnerf_wgrad_partial in gemm.hip: writes nsplit slabs (+ bias slabs) that upnerf_wgrad's reduce
// kernel sums.  expo_a, expo_b: DEVICE pointers to the two exponents.  Returns the block shape through TN/TK.
extern "C" int upnerf_wgrad_f16x3_partial(int M, const float* A, int lda, int N, const float* B, int ldb, int K,
                                          const int* expo_a, const int* expo_b, float* slabs, float* bslabs, int nsplit, int rows, int TN,
                                          int TK, int planes, const upnerf_wgrad_pending* prev, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define WG_ARGS planes, M, N, K, A, lda, B, ldb, expo_a, expo_b, slabs, bslabs, nsplit, rows, st, prev
  if (TN == 256 && TK == 256 && planes == 2 && N <= 256 && K <= 256 && !WG_NO_W4) {  // one wave per SIMD (wgrad_f16x3_w4_kernel)
    upnerf_wgrad_pending pv = {};
    if (prev) pv = *prev;
    hipLaunchKernelGGL((wgrad_f16x3_w4_kernel<2>), dim3(nsplit), dim3(W4_THREADS), 0, st, M, N, K, A, lda, B, ldb, expo_a, expo_b, slabs,
                       bslabs, rows, pv);
    return (int)hipGetLastError();
  }
  if (TN == 256 && TK == 256) return launch<4, 4>(WG_ARGS);
  if (TN == 256 && TK == 128) return launch<4, 2>(WG_ARGS);
  if (TN == 256 && TK == 64) return launch<4, 1>(WG_ARGS);
  if (TN == 128 && TK == 256) return launch<2, 4>(WG_ARGS);
  if (TN == 128 && TK == 128) return launch<2, 2>(WG_ARGS);
  if (TN == 128 && TK == 64) return launch<2, 1>(WG_ARGS);
  if (TN == 64 && TK == 256) return launch<1, 4>(WG_ARGS);
  if (TN == 64 && TK == 128) return launch<1, 2>(WG_ARGS);
  return launch<1, 1>(WG_ARGS);
#undef WG_ARGS
}

// 256 x 256 block from (hi, lo) fp16 planes, staged by LDS-DMA (wgrad_planes_kernel; upnerf_wgrad_planes_chain, gemm.hip)
extern "C" int upnerf_wgrad_planes_partial(int M, const uint16_t* Ah, const uint16_t* Al, const int* aexp, const uint16_t* Bh,
                                           const uint16_t* Bl, const int* bexp, const int* expo_a, const int* expo_b, float* slabs,
                                           float* bslabs, int nsplit, int rows, const upnerf_wgrad_pending* prevp, void* stream) {
  upnerf_wgrad_pending prev = {};
  if (prevp) prev = *prevp;
  hipLaunchKernelGGL(wgrad_planes_kernel, dim3(nsplit, 1, 1), dim3(FX_THREADS), 0, (hipStream_t)stream, M, Ah, Al, aexp, Bh, Bl, bexp, expo_a,
                     expo_b, slabs, bslabs, rows, prev);
  return (int)hipGetLastError();
}

// the same for the fragment-ordered fp16 operands of the register-resident field kernels (upnerf_wgrad_f16p_chain_v, gemm.hip)
extern "C" int upnerf_wgrad_f16p_partial_v(int M, const uint16_t* A16, const int* aexp, const uint16_t* B16, const int* bexp, const float* v,
                                           const int* expo_a, const int* expo_b, float* slabs, float* bslabs, float* vslabs, int nsplit,
                                           int rows, const upnerf_wgrad_pending* prevp, void* stream) {
  upnerf_wgrad_pending prev = {};
  if (prevp) prev = *prevp;
  hipLaunchKernelGGL((wgrad_f16p_kernel<4, 4, 1, 1, 1, true>), dim3(nsplit, 1, 1), dim3(FX_THREADS), 0, (hipStream_t)stream, M, 256, 256, A16,
                     256, aexp, (const void*)B16, 256, bexp, expo_a, expo_b, slabs, bslabs, rows, prev, nullptr, nullptr, v, vslabs);
  return (int)hipGetLastError();
}

// 256 x 256 block with a 1-wide head riding along (upnerf_wgrad_f16x3_chain_v, gemm.hip)
extern "C" int upnerf_wgrad_f16x3_partial_v(int M, const float* A, int lda, const float* B, int ldb, const float* v, const int* expo_a,
                                            const int* expo_b, float* slabs, float* bslabs, float* vslabs, int nsplit, int rows,
                                            const upnerf_wgrad_pending* prevp, void* stream) {
  upnerf_wgrad_pending prev = {};
  if (prevp) prev = *prevp;
  hipLaunchKernelGGL((wgrad_f16x3_kernel<2, 4, 4, true>), dim3(nsplit, 1, 1), dim3(FX_THREADS), 0, (hipStream_t)stream, M, 256, 256, A, lda,
                     B, ldb, expo_a, expo_b, slabs, bslabs, rows, prev, v, vslabs);
  return (int)hipGetLastError();
}

// Packed-operand variant behind upnerf_wgrad_f16p (gemm.hip): blocks of 256 x 256 (both operands fp16-stored) and
// 256 x 64 (fp32 B: the encoding).  Returns UPNERF_EUNSUP for any other block shape.
extern "C" int upnerf_wgrad_f16p_partial(int M, const uint16_t* A16, int lda, const int* aexp, int N, const void* B, int ldb,
                                         const int* bexp, int b_is_f16, int K, const int* expo_a, const int* expo_b, float* slabs,
                                         float* bslabs, int nsplit, int rows, int TN, int TK, const upnerf_wgrad_pending* prev,
                                         const uint8_t* Alo, const uint8_t* Blo, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (Alo) {  // hi + lo8 operands ("24-bit" storage): row-major only; a fp16 B needs its residual bytes too
    if ((b_is_f16 & 2) || ((b_is_f16 & 1) && !Blo)) return UPNERF_EINVAL;
    if (TN == 256 && TK == 256 && (b_is_f16 & 1))
      return launch_p<4, 4, 1, 0, 2>(M, N, K, A16, lda, aexp, B, ldb, bexp, expo_a, expo_b, slabs, bslabs, nsplit, rows, st, prev, Alo, Blo);
    if (TN == 256 && TK == 64 && !(b_is_f16 & 1))
      return launch_p<4, 1, 0, 0, 2>(M, N, K, A16, lda, aexp, B, ldb, bexp, expo_a, expo_b, slabs, bslabs, nsplit, rows, st, prev, Alo, nullptr);
    return UPNERF_EUNSUP;
  }
  const bool frag = (b_is_f16 & 2) != 0;  // bit 1: fp16 operands in the fragment order of the register-resident field kernels
  b_is_f16 &= 1;
  if (frag && ((N != 256 && N != 128) || (b_is_f16 && K != N) || (!b_is_f16 && N != 256))) return UPNERF_EUNSUP;
  if (TN == 128 && TK == 128 && b_is_f16 && frag)  // 128-wide fragments on both sides (candidate_encoding.2: gz_g2 x g1)
    return launch_p<2, 2, 1, 1>(M, N, K, A16, lda, aexp, B, ldb, bexp, expo_a, expo_b, slabs, bslabs, nsplit, rows, st, prev);
  if (TN == 256 && TK == 256 && b_is_f16)
    return frag ? launch_p<4, 4, 1, 1>(M, N, K, A16, lda, aexp, B, ldb, bexp, expo_a, expo_b, slabs, bslabs, nsplit, rows, st, prev)
                : launch_p<4, 4, 1, 0>(M, N, K, A16, lda, aexp, B, ldb, bexp, expo_a, expo_b, slabs, bslabs, nsplit, rows, st, prev);
  if (TN == 256 && TK == 64 && !b_is_f16)
    return frag ? launch_p<4, 1, 0, 1>(M, N, K, A16, lda, aexp, B, ldb, bexp, expo_a, expo_b, slabs, bslabs, nsplit, rows, st, prev)
                : launch_p<4, 1, 0, 0>(M, N, K, A16, lda, aexp, B, ldb, bexp, expo_a, expo_b, slabs, bslabs, nsplit, rows, st, prev);
  return UPNERF_EUNSUP;
}
