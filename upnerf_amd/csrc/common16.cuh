// f16x3 building blocks: fp32-accurate contractions on the f16 matrix cores.
//
// A tile of 64 samples lives in LDS as TWO fp16 planes (hi, lo) with  value = (hi + lo) * 2^-e,  e a per-tile,
// per-tensor power-of-two exponent chosen so that the tile's largest magnitude lands in [2^13, 2^14).  Weights are
// re-packed per step into the same hi/lo form in MFMA fragment order with one exponent per matrix.  Each 32x32x16
// block is three v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi; fp16 products are exact in fp32, accumulation is
// fp32), i.e. 96 matrix cycles instead of the 512 of eight v_mfma_f32_32x32x2_f32.  The split residue (lo*lo and the
// bits below hi+lo) is 2^-22 relative -- the same order as the fp32 rounding of the accumulation itself.
#pragma once
#include "common.cuh"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

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

// byte offset of element (row, k) in a [64][W] fp16 plane; 16-byte chunks XOR-swizzled by the row so that the 16 rows
// one ds_read_b128 lane group touches fall into 16 different bank quads (512-byte rows: bank = 4 * (chunk % 16)).
template <int W>
__device__ __forceinline__ int poff(int row, int k) {
  static_assert(W == 256, "f16x3 field kernels are built for W = 256");
  return row * (W * 2) + ((((k >> 3) ^ (row & 15))) << 4) + ((k & 7) << 1);
}

// max over the 4 waves of a workgroup of per-wave maxima parked in LDS
__device__ __forceinline__ float wg_max4(const float* smax) {
  return fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
}

template <int MT, int NT>
__device__ __forceinline__ float acc_absmax(const f32x16 (&acc)[MT][NT]) {
  float m = 0.0f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(acc[mt][nt][r]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  return m;
}

template <int MT, int NT>
__device__ __forceinline__ void mma16_step(f32x16 (&acc)[MT][NT], const h8 (&ah)[MT], const h8 (&al)[MT],
                                           const h8 (&bh)[NT], const h8 (&bl)[NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
    }
}

// acc[64 rows][n0 .. n0+32*NT) += planes[:, kA0 .. kA0+K) . Wf[n][kB0 .. kB0+K)^T      (scaled integers-in-fp16)
//   Ph, Pl: LDS planes;  Wf: fragment-ordered hi/lo matrix with Kp16 = Kp/16 k-blocks per 32-column tile:
//   byte ((ntile * Kp16 + t) * 2 + plane) * 1024 + lane * 16.   K % 32 == 0 (two k-blocks per pipeline trip).
template <int W, int MT, int NT>
__device__ __forceinline__ void mma16_lds(f32x16 (&acc)[MT][NT], const char* Ph, const char* Pl, int row0, int kA0,
                                          const char* __restrict__ Wf, int Kp16, int n0, int kB0, int K, int lane) {
  const int i = lane & 31, hh = lane >> 5;
  const char* bp[NT];
  int arow[MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bp[nt] = Wf + ((size_t)((n0 >> 5) + nt) * Kp16 + (kB0 >> 4)) * 2048 + lane * 16;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) arow[mt] = row0 + 32 * mt + i;
  const int T = K >> 4;
  h8 ah0[MT], al0[MT], bh0[NT], bl0[NT], ah1[MT], al1[MT], bh1[NT], bl1[NT];
  auto fetch = [&](h8 (&ah)[MT], h8 (&al)[MT], h8 (&bh)[NT], h8 (&bl)[NT], int t) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#ifdef UPNERF_EXP_SAMEB  // timing experiment only (wrong results): every k-block re-reads block 0 -> L1 hits
      bh[nt] = *(const h8*)(bp[nt] + (size_t)(t & 0) * 2048);
      bl[nt] = *(const h8*)(bp[nt] + (size_t)(t & 0) * 2048 + 1024);
#else
      bh[nt] = *(const h8*)(bp[nt] + (size_t)t * 2048);
      bl[nt] = *(const h8*)(bp[nt] + (size_t)t * 2048 + 1024);
#endif
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int o = poff<W>(arow[mt], kA0 + 16 * t + 8 * hh);
      ah[mt] = *(const h8*)(Ph + o);
      al[mt] = *(const h8*)(Pl + o);
    }
  };
  fetch(ah0, al0, bh0, bl0, 0);
#pragma unroll 1
  for (int t = 0; t < T; t += 2) {
    fetch(ah1, al1, bh1, bl1, t + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma16_step(acc, ah0, al0, bh0, bl0);
    __builtin_amdgcn_sched_barrier(0);
    fetch(ah0, al0, bh0, bl0, (t + 2 < T) ? t + 2 : t);
    __builtin_amdgcn_sched_barrier(0);
    mma16_step(acc, ah1, al1, bh1, bl1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Same with the A operand converted on the fly from fp32 rows in global memory (short side inputs: encoding for the
// skip connection, per-ray embedding rows): arow_ptr[mt] points at this lane's row at column 8*(lane>>5); `sc` = 2^e
// is the scale of the LDS planes the same accumulators are fed from.  K % 16 == 0.
template <int MT, int NT>
__device__ __forceinline__ void mma16_glb(f32x16 (&acc)[MT][NT], const float* const (&arow_ptr)[MT], float sc,
                                          const char* __restrict__ Wf, int Kp16, int n0, int kB0, int K, int lane) {
  const int T = K >> 4;
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    h8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const char* p = Wf + ((size_t)((n0 >> 5) + nt) * Kp16 + (kB0 >> 4) + t) * 2048 + lane * 16;
      bh[nt] = *(const h8*)p;
      bl[nt] = *(const h8*)(p + 1024);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 x0 = *(const f32x4*)(arow_ptr[mt] + 16 * t), x1 = *(const f32x4*)(arow_ptr[mt] + 16 * t + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        _Float16 h, l;
        split16(x0[j] * sc, h, l); ah[mt][j] = h; al[mt][j] = l;
        split16(x1[j] * sc, h, l); ah[mt][4 + j] = h; al[mt][4 + j] = l;
      }
    }
    mma16_step(acc, ah, al, bh, bl);
  }
}

// Write accumulator values (already in natural units) into the hi/lo planes at column offset c0 with scale 2^e.
template <int W, int MT, int NT>
__device__ __forceinline__ void acc_to_planes(const f32x16 (&acc)[MT][NT], char* Ph, char* Pl, int row0, int n0, int c0,
                                              float sc, int lane) {
  const int i = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int o = poff<W>(row, c0 + n0 + 32 * nt + i);
        _Float16 h, l;
        split16(acc[mt][nt][r] * sc, h, l);
        *(_Float16*)(Ph + o) = h;
        *(_Float16*)(Pl + o) = l;
      }
}

// dot of plane row segment [c0, c0+K) with w[0..K), split over the TPR adjacent threads that share a row; K is a
// compile-time constant and the weight loads are issued before anything consumes them.
template <int W, int TPR, int K>
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
    const int o = poff<W>(row, c0 + kb + 8 * q);
    const h8 vh = *(const h8*)(Ph + o), vl = *(const h8*)(Pl + o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s += ((float)vh[j] + (float)vl[j]) * w0[q][j];
      s += ((float)vh[4 + j] + (float)vl[4 + j]) * w1[q][j];
    }
  }
#pragma unroll
  for (int d = 1; d < TPR; d <<= 1) s += __shfl_xor(s, d);
  return s * unscale;
}
