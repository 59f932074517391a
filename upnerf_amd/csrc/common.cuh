// Device-side building blocks shared by the gfx950 kernels of libupnerf_hip.so.
//
// Execution model used throughout (MI355X / CDNA4):
//   * 64-lane wavefronts, workgroups of 256 threads = 4 waves = one wave per SIMD of a CU;
//   * dense contractions run on the matrix cores with v_mfma_f32_32x32x2_f32 (exact fp32 in / fp32 accumulate,
//     64 FLOP/clk/SIMD): lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31], and holds
//     D[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31] in its 16 accumulator registers;
//   * a workgroup owns a tile of TILE = 64 rows (samples); the tile's activations live in LDS as
//     [TILE][W] fp32 with a 16-byte-granule XOR swizzle (granule ^= row & 15), which makes both the
//     ds_read_b128 operand reads (16 distinct rows per lane group) and the ds_write_b32 accumulator
//     write-back (32 consecutive columns of one row) bank-conflict free;
//   * weights are streamed from L2 straight into registers, pre-packed per step in MFMA fragment order so that
//     each operand load is one contiguous 1 KiB wave-wide access; they are private to a wave, so an LDS round
//     trip would only add traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "upnerf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NTHREADS 256

#define HIP_TRY(expr)                        \
  do {                                       \
    hipError_t _e = (expr);                  \
    if (_e != hipSuccess) return (int)_e;    \
  } while (0)

__device__ __forceinline__ int swz(int row, int k, int ldw) {
  return row * ldw + ((((k >> 2) ^ (row & 15)) << 2) | (k & 3));
}
// offset of the 16-byte granule holding columns [k, k+4) (k % 4 == 0)
__device__ __forceinline__ int swz4(int row, int k, int ldw) { return row * ldw + (((k >> 2) ^ (row & 15)) << 2); }

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// How the four waves of a workgroup share a [TILE x N] output tile (32x32 MFMA tiles).  When there are fewer
// than four wave-sized pieces (narrow layers of the 64-wide model) the surplus waves recompute a piece another
// wave owns and store identical values -- harmless, and it keeps every wave on the same barrier sequence.
template <int N, int TILE>
struct WaveTile {
  static constexpr int NT = (N >= 256) ? 2 : 1;                        // 32-column tiles per wave
  static constexpr int NG = N / 32 / NT;                               // column groups
  static constexpr int MG = TILE / 32;                                 // 32-row groups in the tile
  static constexpr int WN = NG >= 4 ? 4 : NG;                          // waves along N
  static constexpr int WM = (4 / WN) < MG ? (4 / WN) : MG;             // waves along M
  static constexpr int MT = MG / WM;                                   // 32-row tiles per wave
  __device__ static __forceinline__ int n0(int wave) { return (wave % WN) * 32 * NT; }
  __device__ static __forceinline__ int row0(int wave) { return ((wave / WN) % WM) * 32 * MT; }
};

template <int MT, int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MT][NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
}

// Weight matrices are read in FRAGMENT ORDER (packing.py:NerfPacker.frag): for a row-major [N][Kp] matrix,
//   frag[((n/32) * (Kp/8) + k/8) * 256 + ((k/4)%2 * 32 + n%32) * 4 + k%4] = W[n][k]
// i.e. the 64 x 16 bytes one wave needs for one 32-column tile and 8 consecutive k form one contiguous 1 KiB block:
// every B-operand load is a fully coalesced global_load_dwordx4 and each byte is fetched exactly once per workgroup
// (row-major weights made every load touch 32 different 128-byte lines and thrashed the 32 KiB L1).
//
// acc[TILE rows][n0 .. n0+32*NT) += Hs[:, kA0 .. kA0+K) . W[n][kB0 .. kB0+K)^T
//   Hs: swizzled LDS activations, row stride ldw;  Wf: fragment-ordered matrix with Kp = ldb columns;  K % 8 == 0.
template <int MT, int NT>
__device__ __forceinline__ void mma_step(f32x16 (&acc)[MT][NT], const f32x4 (&a)[MT], const f32x4 (&b)[NT]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
}

// Two-stage software pipeline, written out with ping-pong register sets (no copies): the operands of k-group
// t+1 are requested before the 4*MT*NT MFMAs of k-group t issue, so L2 / LDS latency hides under ~1000 cycles of
// matrix work.  (A rotating `cur = next` copy made hipcc wait for the prefetch inside the same iteration.)
// K/8 must be even (every K in this library is a multiple of 16).
template <int MT, int NT>
__device__ __forceinline__ void mma_lds(f32x16 (&acc)[MT][NT], const float* Hs, int ldw, int row0, int kA0,
                                        const float* __restrict__ Wf, int ldb, int n0, int kB0, int K, int lane) {
  const int i = lane & 31, hh = lane >> 5;
  const int KT = ldb >> 3;
  const float* bp[NT];
  int abase[MT], axor[MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bp[nt] = Wf + ((size_t)((n0 >> 5) + nt) * KT + (kB0 >> 3)) * 256 + lane * 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = row0 + 32 * mt + i;
    abase[mt] = r * ldw;
    axor[mt] = r & 15;
  }
  const int g0 = (kA0 >> 2) + hh;
  const int T = K >> 3;
  f32x4 a0[MT], a1[MT], b0[NT], b1[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b0[nt] = *(const f32x4*)(bp[nt]);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) a0[mt] = *(const f32x4*)&Hs[abase[mt] + ((g0 ^ axor[mt]) << 2)];
#pragma unroll 1  // (a compile-time K would otherwise be unrolled 16x with all LDS addresses hoisted: spills)
  for (int t = 0; t < T; t += 2) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b1[nt] = *(const f32x4*)(bp[nt] + 256 * (t + 1));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a1[mt] = *(const f32x4*)&Hs[abase[mt] + (((g0 + 2 * (t + 1)) ^ axor[mt]) << 2)];
    // keep the requests ABOVE the matrix work: without the fence hipcc sinks them to just before their first
    // use (shorter live ranges) and every half-iteration then waits out the full L2 / LDS latency
    __builtin_amdgcn_sched_barrier(0);
    mma_step(acc, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    const int t2 = (t + 2 < T) ? t + 2 : t;  // last trip: harmless re-read
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b0[nt] = *(const f32x4*)(bp[nt] + 256 * t2);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a0[mt] = *(const f32x4*)&Hs[abase[mt] + (((g0 + 2 * t2) ^ axor[mt]) << 2)];
    __builtin_amdgcn_sched_barrier(0);
    mma_step(acc, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Same contraction with the A operand read from global memory: arow_ptr[mt] points at this lane's row
// (already offset by the column start and by 4*(lane>>5)).  Used for the short side inputs
// (skip-connection encoding, per-ray embedding rows) that are not staged in LDS.
template <int MT, int NT>
__device__ __forceinline__ void mma_glb(f32x16 (&acc)[MT][NT], const float* const (&arow_ptr)[MT],
                                        const float* __restrict__ Wf, int ldb, int n0, int kB0, int K, int lane) {
  const int KT = ldb >> 3;
  const int T = K >> 3;  // even
  const float* bp[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bp[nt] = Wf + ((size_t)((n0 >> 5) + nt) * KT + (kB0 >> 3)) * 256 + lane * 4;
  f32x4 a0[MT], a1[MT], b0[NT], b1[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b0[nt] = *(const f32x4*)(bp[nt]);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) a0[mt] = *(const f32x4*)(arow_ptr[mt]);
#pragma unroll 1
  for (int t = 0; t < T; t += 2) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b1[nt] = *(const f32x4*)(bp[nt] + 256 * (t + 1));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a1[mt] = *(const f32x4*)(arow_ptr[mt] + 8 * (t + 1));
    __builtin_amdgcn_sched_barrier(0);
    mma_step(acc, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    const int t2 = (t + 2 < T) ? t + 2 : t;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b0[nt] = *(const f32x4*)(bp[nt] + 256 * t2);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a0[mt] = *(const f32x4*)(arow_ptr[mt] + 8 * t2);
    __builtin_amdgcn_sched_barrier(0);
    mma_step(acc, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Visit every accumulator element of this lane: v = f(v, row_in_tile, col).
template <int MT, int NT, class F>
__device__ __forceinline__ void acc_map(f32x16 (&acc)[MT][NT], int row0, int n0, int lane, F f) {
  const int i = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int col = n0 + 32 * nt + i;
        acc[mt][nt][r] = f(acc[mt][nt][r], row, col);
      }
}

// ReLU sign bits in the accumulator layout: bit e = (mt*NT + nt)*16 + r of a lane's 64-bit word says whether that
// lane's accumulator element e was positive after bias + ReLU.  The forward kernel stores one word per lane per layer
// (2 KiB per 64-row tile instead of the 64 KiB activation tile); the backward kernel, which uses the same wave tiling
// for the gradient of that activation, reloads its own word and needs no activation read, no mask pass over LDS.
template <int MT, int NT>
__device__ __forceinline__ unsigned long long acc_bias_relu_pack(f32x16 (&acc)[MT][NT], const float* __restrict__ bias,
                                                                 int n0, int lane) {
  static_assert(MT * NT * 16 <= 64, "mask word is 64 bits");
  const int i = lane & 31;
  unsigned int lo = 0u, hi = 0u;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float b = bias[n0 + 32 * nt + i];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int e = (mt * NT + nt) * 16 + r;
        const float v = fmaxf(acc[mt][nt][r] + b, 0.0f);
        acc[mt][nt][r] = v;
        if (e < 32) lo |= (v > 0.0f) ? (1u << e) : 0u;
        else hi |= (v > 0.0f) ? (1u << (e - 32)) : 0u;
      }
    }
  return ((unsigned long long)hi << 32) | lo;
}

template <int MT, int NT>
__device__ __forceinline__ void acc_apply_mask(f32x16 (&acc)[MT][NT], unsigned long long bits) {
  const unsigned int lo = (unsigned int)bits, hi = (unsigned int)(bits >> 32);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int e = (mt * NT + nt) * 16 + r;
        const bool on = e < 32 ? (lo >> e) & 1u : (hi >> (e - 32)) & 1u;
        acc[mt][nt][r] = on ? acc[mt][nt][r] : 0.0f;
      }
}

// max|acc| over the wave, folded into a global running maximum (non-negative floats order like their bit patterns).
template <int MT, int NT>
__device__ __forceinline__ void acc_track_max(const f32x16 (&acc)[MT][NT], float* __restrict__ slot, int lane) {
  if (!slot) return;
  float m = 0.0f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(acc[mt][nt][r]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if (lane == 0) atomicMax((unsigned int*)slot, __float_as_uint(m));
}
__device__ __forceinline__ void wave_track_max(float m, float* __restrict__ slot, int lane) {
  if (!slot) return;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if (lane == 0) atomicMax((unsigned int*)slot, __float_as_uint(m));
}

// Write accumulators into the swizzled LDS tile at column offset c0.
template <int MT, int NT>
__device__ __forceinline__ void acc_to_lds(const f32x16 (&acc)[MT][NT], float* Hs, int ldw, int row0, int n0, int c0,
                                           int lane) {
  const int i = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
        Hs[swz(row, c0 + n0 + 32 * nt + i, ldw)] = acc[mt][nt][r];
      }
}

// Copy LDS columns [c0, c0+ncols) of all TILE rows to global dst[(m0+row)*ldg + col-c0], rows >= M skipped.
template <int TILE>
__device__ __forceinline__ void tile_store(const float* Hs, int ldw, int c0, int ncols, float* __restrict__ dst, int ldg,
                                           int m0, int M, int tid) {
  const int gpr = ncols >> 2;  // granules per row
  for (int idx = tid; idx < TILE * gpr; idx += NTHREADS) {
    const int row = idx / gpr, g = idx - row * gpr;
    if (m0 + row < M) {
      const f32x4 v = *(const f32x4*)&Hs[swz4(row, c0 + 4 * g, ldw)];
      *(f32x4*)&dst[(size_t)(m0 + row) * ldg + 4 * g] = v;
    }
  }
}

// Backward epilogue pass over LDS columns [c0, c0+ncols): zero the entries whose saved forward activation
// (global act[(m0+row)*ldg + col-c0], post-ReLU) is not positive, write the result back to LDS (operand of the
// next contraction) and to global gz (operand of the weight-gradient kernel).  Coalesced 16-byte accesses.
template <int TILE>
__device__ __forceinline__ void tile_mask_store(float* Hs, int ldw, int c0, int ncols, const float* __restrict__ act,
                                                float* __restrict__ gz, int ldg, int m0, int M, int tid) {
  const int gpr = ncols >> 2;
  for (int idx = tid; idx < TILE * gpr; idx += NTHREADS) {
    const int row = idx / gpr, g = idx - row * gpr;
    float* p = &Hs[swz4(row, c0 + 4 * g, ldw)];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m0 + row < M) {
      const size_t off = (size_t)(m0 + row) * ldg + 4 * g;
      const f32x4 av = *(const f32x4*)&act[off];
      v = *(const f32x4*)p;
      v.x = av.x > 0.f ? v.x : 0.f;
      v.y = av.y > 0.f ? v.y : 0.f;
      v.z = av.z > 0.f ? v.z : 0.f;
      v.w = av.w > 0.f ? v.w : 0.f;
      *(f32x4*)&gz[off] = v;
    }
    *(f32x4*)p = v;
  }
}

// Pass over LDS columns [0, ncols): add the rank-1 term wrow[m] * grow[(m / S)][col] (feature-map gradient: per-sample
// compositing weight times the per-ray upstream vector), write back to LDS and store to global gz.
template <int TILE>
__device__ __forceinline__ void tile_rank1_store(float* Hs, int ldw, int ncols, const float* __restrict__ wrow,
                                                 const float* __restrict__ grow, int S, float* __restrict__ gz,
                                                 int m0, int M, int tid, float* __restrict__ maxslot = nullptr) {
  const int gpr = ncols >> 2;
  float lmax = 0.0f;
  for (int idx = tid; idx < TILE * gpr; idx += NTHREADS) {
    const int row = idx / gpr, g = idx - row * gpr, m = m0 + row;
    float* p = &Hs[swz4(row, 4 * g, ldw)];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      v = *(const f32x4*)p;
      if (grow) {
        const float w = wrow[m];
        const f32x4 gv = *(const f32x4*)&grow[(size_t)(m / S) * ncols + 4 * g];
        v.x += w * gv.x; v.y += w * gv.y; v.z += w * gv.z; v.w += w * gv.w;
      }
      *(f32x4*)&gz[(size_t)m * ncols + 4 * g] = v;
    }
    *(f32x4*)p = v;
    lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  wave_track_max(lmax, maxslot, tid & 63);
}

#define PI_F 3.14159274101257324f  // float32(torch.pi)

// o + d*z with two roundings (the reference's `rays_o + rays_d * z` is two ATen ops, rendering.py:251); the
// __f*_rn intrinsics are plain operators in HIP and would be contracted into one fma.
__device__ __forceinline__ float mul_then_add(float o, float d, float z) {
#pragma clang fp contract(off)
  const float p = d * z;
  return o + p;
}

// sin and cos of an fp32 angle (up to ~1e5 rad: 2^9 pi x) to within 1 ulp of fp32: Cody-Waite reduction by pi/2 and
// the fdlibm kernel polynomials, all in fp64 -- a fraction of the instructions of the generic sincosf (whose large-
// argument path is a Payne-Hanek reduction); the result is the correctly rounded fp32 value in 99.998 % of cases.
__device__ __forceinline__ void sincos_f32_via_f64(float arg, float& sn, float& cs) {
  const double x = (double)arg;
  const double q = rint(x * 0.63661977236758134308);
  double r = fma(-q, 1.57079632679489655800e+00, x);
  r = fma(-q, 6.12323399573676603587e-17, r);
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double s = r + r * z * ps;
  const double c = 1.0 - 0.5 * z + z * z * pc;
  const int n = (int)(long long)q & 3;
  const double so = (n & 1) ? c : s, co = (n & 1) ? s : c;
  sn = (float)((n & 2) ? -so : so);
  cs = (float)(((n + 1) & 2) ? -co : co);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// Streaming accesses.  Tensors that one kernel writes and a later kernel reads once, gigabytes of other traffic apart (the stored
// activations and pre-activation gradients: ~30 GB per step), are stored and loaded NON-TEMPORALLY: they then do not push the
// data that IS re-read out of the 4 MB L2s and the Infinity Cache -- the weight fragments every field workgroup streams, the
// weight-gradient slabs the next launch sums.  Measured (round 3, graph-replayed step, three alternating runs): 18.27 -> 17.79 ms.
// -DUPNERF_NO_NT restores plain accesses.
//
// Experiment switches that change RESULTS (timing-only builds: wrong or missing outputs) compile only in a build that says it is
// an experiment (`make variant` passes -DUPNERF_EXPERIMENT); a stray -D in the product build is a compile error, not a silently
// wrong library.
#if (defined(UPNERF_EXP_HALFROW) || defined(UPNERF_EXP_NOSTORE) || defined(UPNERF_EXP_SAMEB) || defined(RR_EXP_NOSTORE) || \
     defined(RR_EXP_NODMA) || defined(RR_EXP_NOEPI) || defined(RR_EXP_NOMMA) || defined(RR_EXP_NOBARRIER) || defined(RR_EXP_NOLDS) || defined(RR_EXP_PLAINLOAD) || defined(WG_EXP_NOLOAD) || defined(WG_EXP_NOSTAGE) || defined(WG_EXP_NOMMA) || defined(WP_EXP_NOREAD) || defined(WP_EXP_NOMMA) || defined(TR_EXP_NOLOAD) || defined(TR_EXP_NOMMA) || defined(TR_EXP_NOSTORE)) && !defined(UPNERF_EXPERIMENT)
#error "UPNERF_EXP_* / RR_EXP_* switches produce wrong results: build them with -DUPNERF_EXPERIMENT (make variant), never into libupnerf_hip.so"
#endif
#ifdef UPNERF_NO_NT
#define NT_LOAD(p) (*(p))
#define NT_STORE(p, v) (*(p) = (v))
#elif defined(UPNERF_ST_POLICY)
// A/B of the cache policy of the streaming stores (make variant EXP=-DUPNERF_ST_POLICY=n): 1 = sc1 (write-through, the line is
// DROPPED from the XCD's L2 -- plain and nt stores keep it: MI355X_MICROARCH.md, fence table, 'stores of each flavour'),
// 2 = sc0 sc1, 3 = sc1 nt.  Inline asm: hipcc does not count these stores in vmcnt (its own waits only get stricter) and the
// data registers are protected by the trailing s_nop (cdna_hip_programming.md 5.7).
template <class T>
__device__ __forceinline__ void store_policy(T* p, const T& v) {
#if UPNERF_ST_POLICY == 1
#define UPNERF_ST_BITS "sc1"
#elif UPNERF_ST_POLICY == 2
#define UPNERF_ST_BITS "sc0 sc1"
#else
#define UPNERF_ST_BITS "sc1 nt"
#endif
  if constexpr (sizeof(T) == 16) asm volatile("global_store_dwordx4 %0, %1, off " UPNERF_ST_BITS "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (sizeof(T) == 8) asm volatile("global_store_dwordx2 %0, %1, off " UPNERF_ST_BITS "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else __builtin_nontemporal_store(v, p);
}
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#define NT_STORE(p, v) store_policy((p), (v))
#else
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#define NT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#endif

// ---- fixed-order reduction of weight-gradient slabs (upnerf_wgrad*): the work of ONE 512-thread block `bid` of the reduce
// grid, callable from the reduce kernel and from the prologue of the NEXT weight-gradient kernel (upnerf_wgrad_f16x3_chain).
// dW[n][k] = sum_split slab[split][by][bz][n%TN][k%TK]; one thread per 4 consecutive k (16-byte loads); the splits are dealt
// round-robin to 8 thread groups whose partial sums meet in LDS (`part`, 8 KiB); each group keeps 8 loads in flight.
#define RED_RG 8
#define RED_THREADS (64 * RED_RG)
// PW: physical waves of the calling workgroup (8, or 4: every wave then plays thread groups w and w + 4 one after the other -- the
// same partial sums, added in the same order: bitwise the 8-wave result).
template <int PW = RED_RG>
__device__ __forceinline__ void wgrad_reduce_body(int bid, int tid, const upnerf_wgrad_pending& P, f32x4 (*part)[64]) {
  static_assert(PW == 8 || PW == 4, "8 thread groups on 8 or 4 waves");
  const int N = P.N, K = P.K, TN = P.TN, TK = P.TK, nsplit = P.nsplit;
  const float* __restrict__ slabs = P.slabs;
  const float* __restrict__ bslabs = P.bslabs;
  const int lane = tid & 63;
  const int q = bid * 64 + lane;  // index of a group of 4 consecutive k
  const int K4 = K >> 2;
  const int gy = (N + TN - 1) / TN, gz = (K + TK - 1) / TK;
  const bool ok = q < N * K4;
  const int n = ok ? q / K4 : 0, k = ok ? (q - n * K4) * 4 : 0;
  const int by = n / TN, bz = k / TK;
  const size_t off = ((size_t)by * gz + bz) * TN * TK + (size_t)(n - by * TN) * TK + (k - bz * TK);
  const size_t stride = (size_t)gy * gz * TN * TK;
  // 8 waves x 8 loads of 16 bytes per lane = 64 KiB in flight per workgroup (one workgroup per CU at 256 x 256): at 16 KiB
  // the 67 MB of slabs of a 256 x 256 layer came in at 2.7 TB/s
#pragma unroll 1
  for (int rg = tid >> 6; rg < RED_RG; rg += PW) {
    f32x4 s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) {
      int sp = rg;
      for (; sp + 7 * RED_RG < nsplit; sp += 8 * RED_RG) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)&slabs[off + (size_t)(sp + u * RED_RG) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += v[u];
      }
      for (; sp < nsplit; sp += RED_RG) s[0] += *(const f32x4*)&slabs[off + (size_t)sp * stride];
    }
    part[rg][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  }
  __syncthreads();
  if ((tid >> 6) == 0 && ok) {
    const f32x4 t = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) +
                    ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
    if (P.n2 > 0 && n >= P.n2) *(f32x4*)&P.dW2[(size_t)(n - P.n2) * P.ldo2 + k] = t;
    else *(f32x4*)&P.dW[(size_t)n * P.ldo + k] = t;
  }
  if (P.db || (P.n2 > 0 && P.db2)) {
#pragma unroll 1
    for (int idx = bid * RED_THREADS + tid; idx < (bid + 1) * RED_THREADS; idx += 64 * PW)
    if (idx < N) {
      const int bby = idx / TN;
      float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* src = bslabs + (size_t)bby * TN + (idx - bby * TN);
      const size_t bst = (size_t)gy * TN;
      int sp = 0;
      for (; sp + 8 <= nsplit; sp += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] += src[(size_t)(sp + u) * bst];
      }
      for (; sp < nsplit; ++sp) p[0] += src[(size_t)sp * bst];
      const float sb = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      if (P.n2 > 0 && idx >= P.n2) { if (P.db2) P.db2[idx - P.n2] = sb; }
      else if (P.db) P.db[idx] = sb;
    }
  }
  if (P.vslabs && bid == 0) {  // the vector head that rode on the problem: K sums + the sum of v, per split [K + 4]
    for (int idx = tid; idx <= K; idx += 64 * PW) {
      float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* src = P.vslabs + idx;
      const size_t st = (size_t)K + 4;
      int sp = 0;
      for (; sp + 8 <= nsplit; sp += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] += src[(size_t)(sp + u) * st];
      }
      for (; sp < nsplit; ++sp) p[0] += src[(size_t)sp * st];
      const float sv = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      if (idx < K) P.dv[idx] = sv;
      else if (P.dbv) P.dbv[0] = sv;
    }
  }
}
