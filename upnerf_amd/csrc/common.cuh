// Device-side building blocks shared by the gfx950 kernels of libupnerf_hip.so.
//
// Execution model used throughout (MI355X / CDNA4):
//   * 64-lane wavefronts, workgroups of 256 threads = 4 waves = one wave per SIMD of a CU;
//   * dense contractions run on the matrix cores with v_mfma_f32_32x32x2_f32 (exact fp32 in / fp32 accumulate,
//     64 FLOP/clk/SIMD): lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31], and holds
//     D[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31] in its 16 accumulator registers;
//   * a workgroup owns a tile of TILE = 128 rows (samples); the tile's activations live in LDS as
//     [128][W] fp32 with a 16-byte-granule XOR swizzle (granule ^= row & 15), which makes both the
//     ds_read_b128 operand reads (16 distinct rows per lane group) and the ds_write_b32 accumulator
//     write-back (32 consecutive columns of one row) bank-conflict free;
//   * weights are streamed from L2 straight into registers in fragment order ([N][K] row-major, each lane
//     reading 4 consecutive k of its own output column): they are private to a wave, so an LDS round trip
//     would only add traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "upnerf_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TILE UPNERF_TILE_ROWS
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

// How the four waves of a workgroup share a [128 x N] output tile (32x32 MFMA tiles).
template <int N>
struct WaveTile {
  static constexpr int NT = (N >= 256) ? 2 : 1;                 // 32-column tiles per wave
  static constexpr int WN = (N / 32 / NT) >= 4 ? 4 : (N / 32 / NT);  // waves along N
  static constexpr int WM = 4 / WN;                             // waves along M
  static constexpr int MT = 4 / WM;                             // 32-row tiles per wave
  __device__ static __forceinline__ int n0(int wave) { return (wave % WN) * 32 * NT; }
  __device__ static __forceinline__ int row0(int wave) { return (wave / WN) * 32 * MT; }
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

// acc[128-row tile][n0 .. n0+32*NT) += Hs[:, kA0 .. kA0+K) . Wp[n][kB0 .. kB0+K)^T
//   Hs: swizzled LDS activations, row stride ldw;  Wp: global [N][ldb] row-major;  K % 8 == 0.
template <int MT, int NT>
__device__ __forceinline__ void mma_lds(f32x16 (&acc)[MT][NT], const float* Hs, int ldw, int row0, int kA0,
                                        const float* __restrict__ Wp, int ldb, int n0, int kB0, int K, int lane) {
  const int i = lane & 31, hh = lane >> 5;
  const float* bp[NT];
  int arow[MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bp[nt] = Wp + (size_t)(n0 + 32 * nt + i) * ldb + kB0 + 4 * hh;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) arow[mt] = row0 + 32 * mt + i;
  f32x4 bcur[NT], bnxt[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bcur[nt] = *(const f32x4*)(bp[nt]);
  const int T = K >> 3;
  for (int t = 0; t < T; ++t) {
    const int tn = (t + 1 < T) ? t + 1 : t;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bnxt[nt] = *(const f32x4*)(bp[nt] + 8 * tn);
    f32x4 a[MT];
    const int g = ((kA0 + 8 * t) >> 2) + hh;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = *(const f32x4*)&Hs[arow[mt] * ldw + ((g ^ (arow[mt] & 15)) << 2)];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], bcur[nt][s], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnxt[nt];
  }
}

// Same contraction with the A operand read from global memory: arow_ptr[mt] points at this lane's row
// (already offset by the column start and by 4*(lane>>5)).  Used for the short side inputs
// (skip-connection encoding, per-ray embedding rows) that are not staged in LDS.
template <int MT, int NT>
__device__ __forceinline__ void mma_glb(f32x16 (&acc)[MT][NT], const float* const (&arow_ptr)[MT],
                                        const float* __restrict__ Wp, int ldb, int n0, int kB0, int K, int lane) {
  const int i = lane & 31, hh = lane >> 5;
  const int T = K >> 3;
  for (int t = 0; t < T; ++t) {
    f32x4 a[MT], b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = *(const f32x4*)(Wp + (size_t)(n0 + 32 * nt + i) * ldb + kB0 + 4 * hh + 8 * t);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = *(const f32x4*)(arow_ptr[mt] + 8 * t);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
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

// Copy LDS columns [c0, c0+ncols) of all 128 rows to global dst[(m0+row)*ldg + col-c0], rows >= M skipped.
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
