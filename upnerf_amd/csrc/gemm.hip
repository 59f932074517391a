// Weight-gradient contraction, the generic fp32 MFMA linear layer, the vector-head weight gradients and the
// fused Adam step for gfx950.
//
// upnerf_wgrad: dW[n][k] = sum_m A[m][n] B[m][k]  (autograd of nn.Linear weights, models/nerf.py:39-78).
//   The reduction runs over ALL samples (M = rays x samples, ~786k), the output is at most 256x256, so the
//   grid splits M: each workgroup keeps a full [TN x TK] output block in accumulator registers (up to
//   16 tiles of 32x32 per wave = 256 AGPRs; one wave per SIMD gets the whole 512-entry register file on CDNA4),
//   streams its slice of A and B through double-buffered LDS in 32-row chunks, and writes one partial slab.
//   A second tiny kernel sums the slabs in a fixed order -> bitwise reproducible, no float atomics.
//   MFMA orientation: lane (i, h) feeds A-operand A[m+h][n0+i] and B-operand B[m+h][k0+i]; both are plain
//   ds_read_b32 of 32 consecutive floats per half-wave (bank-conflict free without padding).
#include "common.cuh"

namespace {

#define WG_CHUNK 32

// One [64 MTW x 64 NTW] block of  sum_{m in [mbeg, mend)} A[m][nblk + n] B[m][kblk + k]  -> slab (row-major [TN][TK]);
// column sums of A -> bslab [TN] when given.  The whole workgroup works on the block.
template <int MTW, int NTW>
__device__ __forceinline__ void wgrad_tile(int N, int K, const float* __restrict__ A, int lda,
                                           const float* __restrict__ B, int ldb, float* __restrict__ slab,
                                           float* __restrict__ bslab, int mbeg, int mend, int nblk, int kblk) {
  constexpr int TN = 64 * MTW, TK = 64 * NTW;
  constexpr int A4 = WG_CHUNK * TN / 4 / NTHREADS, B4 = WG_CHUNK * TK / 4 / NTHREADS;  // float4 per thread per chunk
  __shared__ __attribute__((aligned(16))) float As[2][WG_CHUNK * TN];
  __shared__ __attribute__((aligned(16))) float Bs[2][WG_CHUNK * TK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int n0 = (wave >> 1) * 32 * MTW, k0 = (wave & 1) * 32 * NTW;
  f32x16 acc[MTW][NTW];
  acc_zero(acc);
  float bsum[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) bsum[mt] = 0.f;

  f32x4 ra[A4 > 0 ? A4 : 1], rb[B4 > 0 ? B4 : 1];
  auto gload = [&](int mc) {
#pragma unroll
    for (int q = 0; q < A4; ++q) {
      const int idx = tid + q * NTHREADS, row = idx / (TN / 4), c4 = idx - row * (TN / 4);
      const int m = mc + row;
      ra[q] = (m < mend && nblk + 4 * c4 < N) ? *(const f32x4*)&A[(size_t)m * lda + nblk + 4 * c4]
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < B4; ++q) {
      const int idx = tid + q * NTHREADS, row = idx / (TK / 4), c4 = idx - row * (TK / 4);
      const int m = mc + row;
      rb[q] = (m < mend && kblk + 4 * c4 < K) ? *(const f32x4*)&B[(size_t)m * ldb + kblk + 4 * c4]
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < A4; ++q) *(f32x4*)&As[buf][(tid + q * NTHREADS) * 4] = ra[q];
#pragma unroll
    for (int q = 0; q < B4; ++q) *(f32x4*)&Bs[buf][(tid + q * NTHREADS) * 4] = rb[q];
  };

  int buf = 0;
  if (mbeg < mend) gload(mbeg);
  for (int mc = mbeg; mc < mend; mc += WG_CHUNK) {
    lstore(buf);
    __syncthreads();
    if (mc + WG_CHUNK < mend) gload(mc + WG_CHUNK);
    const float* as = As[buf] + hh * TN + n0 + li;
    const float* bs = Bs[buf] + hh * TK + k0 + li;
    // ping-pong operand sets, requests fenced above the matrix work (see common.cuh:mma_lds)
    float a0[MTW], b0[NTW], a1[MTW], b1[NTW];
    auto fetch = [&](float (&av)[MTW], float (&bv)[NTW], int mm) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) av[mt] = as[mm * TN + 32 * mt];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) bv[nt] = bs[mm * TK + 32 * nt];
    };
    auto step = [&](const float (&av)[MTW], const float (&bv)[NTW]) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        bsum[mt] += av[mt];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
      }
    };
    fetch(a0, b0, 0);
#pragma unroll 2
    for (int mm = 0; mm < WG_CHUNK; mm += 4) {
      fetch(a1, b1, mm + 2);
      __builtin_amdgcn_sched_barrier(0);
      step(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(a0, b0, (mm + 4 < WG_CHUNK) ? mm + 4 : mm);
      __builtin_amdgcn_sched_barrier(0);
      step(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    buf ^= 1;
  }
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
        slab[n * TK + k0 + 32 * nt + li] = acc[mt][nt][r];
      }
  if (bslab && (wave & 1) == 0) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const float s = bsum[mt] + __shfl_xor(bsum[mt], 32);
      if (hh == 0) bslab[n0 + 32 * mt + li] = s;
    }
  }
}

template <int MTW, int NTW>
__global__ __launch_bounds__(NTHREADS, 1) void wgrad_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                            const float* __restrict__ B, int ldb,
                                                            float* __restrict__ slabs, float* __restrict__ bslabs,
                                                            int rows_per_split) {
  constexpr int TN = 64 * MTW, TK = 64 * NTW;
  const int split = blockIdx.x;
  const int mbeg = split * rows_per_split;
  const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  // partial slab [split][by][bz][TN][TK]
  const size_t blk = ((size_t)split * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z;
  wgrad_tile<MTW, NTW>(N, K, A, lda, B, ldb, slabs + blk * TN * TK,
                       (bslabs && blockIdx.z == 0) ? bslabs + ((size_t)split * gridDim.y + blockIdx.y) * TN : nullptr, mbeg,
                       mend, blockIdx.y * TN, blockIdx.z * TK);
}

// ---- grouped variant: many small weight gradients (the per-ray layers: M = rays) in ONE launch + ONE reduction.
// Every group is cut into 128 x 128 output blocks and `nsplit` row ranges; blockIdx.y walks the blocks of all groups.
struct WgradGroups {
  upnerf_wgrad_group g[UPNERF_MAX_WGRAD_GROUPS];
  int tile_start[UPNERF_MAX_WGRAD_GROUPS + 1];  // first output block of each group
  int red_start[UPNERF_MAX_WGRAD_GROUPS + 1];   // first reduction workgroup of each group
  int n;
};
#define GT 128  // block edge of the grouped kernels

__device__ __forceinline__ int find_group(const int* start, int n, int idx) {
  int g = 0;
  while (g + 1 < n && idx >= start[g + 1]) ++g;
  return g;
}

__global__ __launch_bounds__(NTHREADS, 1) void wgrad_grouped_kernel(WgradGroups T, float* __restrict__ slabs,
                                                                    float* __restrict__ bslabs, int nsplit) {
  const int tile = blockIdx.y, split = blockIdx.x;
  const int gi = find_group(T.tile_start, T.n, tile);
  const upnerf_wgrad_group& G = T.g[gi];
  const int gz = (G.K + GT - 1) / GT, local = tile - T.tile_start[gi], by = local / gz, bz = local - by * gz;
  const int rows = (((G.M + nsplit - 1) / nsplit) + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
  const int mbeg = split * rows;
  const int mend = (mbeg + rows < G.M) ? mbeg + rows : G.M;
  const size_t blk = (size_t)tile * nsplit + split;
  wgrad_tile<2, 2>(G.N, G.K, G.A, G.lda, G.B, G.ldb, slabs + blk * GT * GT,
                   (G.db && bz == 0) ? bslabs + blk * GT : nullptr, mbeg < mend ? mbeg : mend, mend, by * GT, bz * GT);
}

// fixed-order sum over the splits, one thread per 4 consecutive k of one group's dW (and per row for db)
__global__ __launch_bounds__(NTHREADS) void wgrad_grouped_reduce_kernel(WgradGroups T, const float* __restrict__ slabs,
                                                                       const float* __restrict__ bslabs, int nsplit) {
  const int gi = find_group(T.red_start, T.n, blockIdx.x);
  const upnerf_wgrad_group& G = T.g[gi];
  const int b = blockIdx.x - T.red_start[gi];
  const int K4 = G.K >> 2, gz = (G.K + GT - 1) / GT;
  const int q = b * NTHREADS + threadIdx.x;
  if (q < G.N * K4) {
    const int n = q / K4, k = (q - n * K4) * 4;
    const int by = n / GT, bz = k / GT;
    const float* src = slabs + ((size_t)(T.tile_start[gi] + by * gz + bz) * nsplit) * GT * GT + (size_t)(n - by * GT) * GT +
                       (k - bz * GT);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int sp = 0;
    for (; sp + 4 <= nsplit; sp += 4) {
      const f32x4 a = *(const f32x4*)&src[(size_t)sp * GT * GT], c = *(const f32x4*)&src[(size_t)(sp + 1) * GT * GT];
      const f32x4 d = *(const f32x4*)&src[(size_t)(sp + 2) * GT * GT], e = *(const f32x4*)&src[(size_t)(sp + 3) * GT * GT];
      s0 += a; s1 += c; s2 += d; s3 += e;
    }
    for (; sp < nsplit; ++sp) s0 += *(const f32x4*)&src[(size_t)sp * GT * GT];
    *(f32x4*)&G.dW[(size_t)n * G.ldo + k] = (s0 + s1) + (s2 + s3);
  }
  if (G.db && q < G.N) {
    const int by = q / GT;
    const float* src = bslabs + ((size_t)(T.tile_start[gi] + by * gz) * nsplit) * GT + (q - by * GT);
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    int sp = 0;
    for (; sp + 4 <= nsplit; sp += 4) {
      p0 += src[(size_t)sp * GT]; p1 += src[(size_t)(sp + 1) * GT];
      p2 += src[(size_t)(sp + 2) * GT]; p3 += src[(size_t)(sp + 3) * GT];
    }
    for (; sp < nsplit; ++sp) p0 += src[(size_t)sp * GT];
    G.db[q] = (p0 + p1) + (p2 + p3);
  }
}

// dW[n][k] = sum_split slab[split][by][bz][n%TN][k%TK]   (fixed order -> bitwise reproducible; common.cuh:wgrad_reduce_body)
__global__ __launch_bounds__(RED_THREADS) void wgrad_reduce_kernel(upnerf_wgrad_pending P) {
  __shared__ f32x4 part[RED_RG][64];
  wgrad_reduce_body(blockIdx.x, threadIdx.x, P, part);
}
static upnerf_wgrad_pending reduce_desc(int N, int K, int TN, int TK, int nsplit, const float* slabs, const float* bslabs, float* dW,
                                        int ldo, float* db) {
  const int quads = N * (K / 4);
  int rblocks = (quads + 63) / 64;
  if (rblocks * NTHREADS < N) rblocks = (N + NTHREADS - 1) / NTHREADS;
  return upnerf_wgrad_pending{slabs, bslabs, dW, db, N, K, TN, TK, nsplit, ldo, rblocks, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr};
}
static void launch_reduce(hipStream_t st, const upnerf_wgrad_pending& P) {
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(P.rblocks), dim3(RED_THREADS), 0, st, P);
}

// ---- N = 1 / 3 heads: dw[c][k] = sum_m v[m][c] X[m][k].  HBM-bound stream of X: every lane owns 4 columns
//      (16-byte loads, one full row per K/4 lanes), row groups run 4 rows ahead, partials meet in LDS.
template <int K>
__global__ __launch_bounds__(NTHREADS) void vec_wgrad_kernel(int M, const float* __restrict__ v, int ldv, int nvec,
                                                            const float* __restrict__ X, int ldx,
                                                            float* __restrict__ part, int rows_per_split) {
  constexpr int CPR = K / 4, RG = NTHREADS / CPR;  // lanes per row, row groups per block
  __shared__ float red[RG][3][K + 4];
  const int tid = threadIdx.x, c4 = tid % CPR, rg = tid / CPR, split = blockIdx.x;
  const int mbeg = split * rows_per_split;
  const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
  f32x4 acc[3];
  float bacc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;  // rows in flight per lane: the stream is latency-bound at one workgroup per CU
  for (int m = mbeg + rg; m < mend; m += U * RG) {
    f32x4 x[U];
    float vv[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int mm = m + u * RG;
      const bool ok = mm < mend;
      const int mc = ok ? mm : mend - 1;  // loads first (clamped), masks after: no branch around a load
      x[u] = NT_LOAD((const f32x4*)&X[(size_t)mc * ldx + 4 * c4]);
#pragma unroll
      for (int c = 0; c < 3; ++c) vv[u][c] = (c < nvec) ? v[(size_t)mc * ldv + c] : 0.f;
      if (!ok) {
        x[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) vv[u][c] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc[c].x += vv[u][c] * x[u].x; acc[c].y += vv[u][c] * x[u].y;
        acc[c].z += vv[u][c] * x[u].z; acc[c].w += vv[u][c] * x[u].w;
        bacc[c] += vv[u][c];
      }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    *(f32x4*)&red[rg][c][4 * c4] = acc[c];
    if (c4 == 0) red[rg][c][K] = bacc[c];
  }
  __syncthreads();
  // part layout [split][4][K+1]: last column = sum of v
  for (int idx = tid; idx < nvec * (K + 1); idx += NTHREADS) {
    const int c = idx / (K + 1), k = idx - c * (K + 1);
    float s = 0.f;
#pragma unroll 4
    for (int g = 0; g < RG; ++g) s += red[g][c][k];
    part[((size_t)split * 4 + c) * (K + 1) + k] = s;
  }
}

__global__ void vec_wgrad_reduce_kernel(int nvec, int K, int nsplit, const float* __restrict__ part,
                                        float* __restrict__ dw, float* __restrict__ dbv) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nvec * (K + 1)) return;
  const int c = idx / (K + 1), k = idx - c * (K + 1);
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int sp = 0;
  for (; sp + 8 <= nsplit; sp += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += part[((size_t)(sp + u) * 4 + c) * (K + 1) + k];
  }
  for (; sp < nsplit; ++sp) p[0] += part[((size_t)sp * 4 + c) * (K + 1) + k];
  const float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  if (k < K) dw[(size_t)c * K + k] = s;
  else if (dbv) dbv[c] = s;
}

// upnerf_vec_wgrad against a fp16 tensor in the operand-fragment order of the register-resident field kernels
// (include/upnerf_hip.h, tile_rows = 256): [32-row tile][k-block s KB][lane 64][8], feature 16 s + 8 (j / 4) + 4 (lane / 32)
// + j % 4, row 32 tile + lane % 32, values scaled by 2^xexp[tile]; KB = 16 (256 wide) or 8 (128 wide).  A thread keeps one
// (s, lane) piece position and walks the tiles of its workgroup's slice (a wave reads 1 KiB contiguous per tile); the 32 rows
// meet in a shuffle tree at the end.  NV vectors share every piece read.
template <int KB, int NV>
__global__ __launch_bounds__(64 * KB) void vec_wgrad_frag16_kernel(int M, const float* __restrict__ v, int ldv,
                                                                 const uint16_t* __restrict__ X16, const int* __restrict__ xexp,
                                                                 float* __restrict__ part, int tiles_per_split) {
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  constexpr int K = 16 * KB;
  const int tid = threadIdx.x, lane = tid & 63, s = tid >> 6, li = lane & 31, hh = lane >> 5;
  const int ntile = (M + 31) >> 5;
  const int t0 = blockIdx.x * tiles_per_split;
  const int t1 = t0 + tiles_per_split < ntile ? t0 + tiles_per_split : ntile;
  float acc[NV][8], vsum[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    vsum[c] = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[c][j] = 0.0f;
  }
  const h8v* __restrict__ src = (const h8v*)X16 + (size_t)s * 64 + lane;
  constexpr int U = NV == 1 ? 4 : 2;  // tiles in flight
  int t = t0;
  for (; t + U <= t1; t += U) {
    h8v x[U];
    float w[U][NV];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x[u] = __builtin_nontemporal_load(src + (size_t)(t + u) * (64 * KB));
      const int m = (t + u) * 32 + li;
      const int e = -xexp[t + u];
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const float wv = m < M ? v[(size_t)m * ldv + c] : 0.0f;
        vsum[c] += wv;
        w[u][c] = ldexpf(wv, e);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < NV; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] = fmaf(w[u][c], (float)x[u][j], acc[c][j]);
  }
  for (; t < t1; ++t) {
    const h8v x = __builtin_nontemporal_load(src + (size_t)t * (64 * KB));
    const int m = t * 32 + li;
    const int e = -xexp[t];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const float wv = m < M ? v[(size_t)m * ldv + c] : 0.0f;
      vsum[c] += wv;
      const float w = ldexpf(wv, e);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[c][j] = fmaf(w, (float)x[j], acc[c][j]);
    }
  }
#pragma unroll
  for (int sh = 1; sh < 32; sh <<= 1) {
#pragma unroll
    for (int c = 0; c < NV; ++c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[c][j] += __shfl_xor(acc[c][j], sh);
      vsum[c] += __shfl_xor(vsum[c], sh);
    }
  }
  if (li == 0) {
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      float* __restrict__ dst = part + ((size_t)blockIdx.x * 4 + c) * (K + 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)] = acc[c][j];
      if (s == 0 && hh == 0) dst[K] = vsum[c];
    }
  }
}

// ---- generic linear: C = act(A B^T + bias) for the per-ray layers (M = rays: a few thousand rows).
// 64 x 64 output tile per workgroup (one 32 x 32 MFMA tile per wave), A and B staged in LDS in K-chunks of 64 with
// coalesced 16-byte loads: 16 x (N/64) x ... = several hundred workgroups even at M = 4096, four or five per CU, so
// the chip is full and the workgroups hide each other's load phases.  (128 x 256 tiles left 3/4 of the CUs idle.)
// act bit 0: ReLU;  act bit 1: B is given transposed, [K][N] with row stride ldb (no host-side transpose copy).
#define LIN_T 64
#define LIN_KC 64
template <bool SLOWB>
__global__ __launch_bounds__(NTHREADS, 4) void linear_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                             const float* __restrict__ B, int ldb,
                                                             const float* __restrict__ bias, float* __restrict__ C,
                                                             int ldc, int act) {
  __shared__ __attribute__((aligned(16))) float As[LIN_T * LIN_KC];
  __shared__ __attribute__((aligned(16))) float Bs[LIN_T * LIN_KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * LIN_T, nb = blockIdx.y * LIN_T;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;  // this wave's 32 x 32 piece of the tile
  const int li = lane & 31, hh = lane >> 5;
  const bool transb = (act & 2) != 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  // this lane's bias value, requested now (unconditionally: a pointer select and a clamped column), consumed behind the K loop
  float bv;
  {
    const int c = nb + wc + li;
    bv = (bias ? bias : A)[bias ? (c < N ? c : N - 1) : 0];
    if (!bias) bv = 0.0f;
  }
  // Register-staged software pipeline: the global loads of chunk c+1 are in flight while chunk c is contracted out of
  // LDS (each chunk used to wait out its own HBM / L2 round trip between two barriers).
  constexpr int LPT = LIN_T * (LIN_KC / 4) / NTHREADS;  // 16-byte loads per thread and operand: 4
  f32x4 ra[LPT], rb[LPT];
  // Round 6: every load of a chunk is UNCONDITIONAL (clamped row / column / k indices, a pointer select between the two layouts of B,
  // masks applied to the loaded values).  The first form branched around each load (`if (!transb)`, `if (k < kk)`, `if (col + 3 < N)`):
  // hipcc put `s_waitcnt vmcnt(0)` behind every such branch -- the ISA had the eight loads of a chunk as eight L2 / HBM round trips in
  // a row (cdna_hip_programming.md, 'Three .s-level traps' (c)), which is what made a 0.8 GFLOP product a 23-38 us launch.  A B^T
  // operand whose rows cannot be read in 16-byte pieces (ldb or N not a multiple of 4) takes the old element-wise path (`slowb`).
  constexpr bool slowb = SLOWB;  // (chosen by the launcher: transb && (ldb % 4 || N % 4))
  auto gload = [&](int kc) {
    const int kk = (K - kc) < LIN_KC ? (K - kc) : LIN_KC;  // multiple of 8
#pragma unroll
    for (int q = 0; q < LPT; ++q) {
      const int idx = tid + q * NTHREADS;
      const int row = idx >> 4, g = idx & 15;
      const bool kin = 4 * g < kk;
      int ma = m0 + row, nbr = nb + row;
      const bool ina = kin && ma < M;
      ma = ma < M ? ma : M - 1;
      nbr = nbr < N ? nbr : N - 1;
      const int kg = kin ? kc + 4 * g : 0;
      ra[q] = *(const f32x4*)&A[(size_t)ma * lda + kg];
      if (!ina) ra[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (!slowb) {
        // B as [N][K] rows (row = n, 16 bytes of k), or B^T as [K][N] rows (row = k of the chunk, 16 bytes of n)
        const int kt = row < kk ? kc + row : kc;          // transb: k of this piece, clamped into the chunk
        int nt4 = nb + 4 * g;
        const bool inbt = row < kk && nt4 + 3 < N;
        nt4 = nt4 + 3 < N ? nt4 : 0;
        const float* __restrict__ bp = transb ? &B[(size_t)kt * ldb + nt4] : &B[(size_t)nbr * ldb + kg];
        const bool inb = transb ? inbt : (kin && nb + row < N);
        rb[q] = *(const f32x4*)bp;
        if (!inb) rb[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if constexpr (slowb) {
#pragma unroll
      for (int q = 0; q < LPT; ++q) {
        const int idx = tid + q * NTHREADS;
        const int k = idx >> 4, g = idx & 15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kk) {
          const float* src = &B[(size_t)(kc + k) * ldb + nb + 4 * g];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (nb + 4 * g + j < N) v[j] = src[j];
        }
        rb[q] = v;
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int q = 0; q < LPT; ++q) {
      const int idx = tid + q * NTHREADS;
      const int row = idx >> 4, g = idx & 15;
      *(f32x4*)&As[swz4(row, 4 * g, LIN_KC)] = ra[q];
      if (!transb) {
        *(f32x4*)&Bs[swz4(row, 4 * g, LIN_KC)] = rb[q];
      } else {  // scattered into the [n][k] image (row = k here)
#pragma unroll
        for (int j = 0; j < 4; ++j) Bs[swz(4 * g + j, row, LIN_KC)] = rb[q][j];
      }
    }
  };
  gload(0);
  for (int kc = 0; kc < K; kc += LIN_KC) {
    const int kk = (K - kc) < LIN_KC ? (K - kc) : LIN_KC;
    __syncthreads();  // everyone is done reading the previous chunk
    lstore();
    __syncthreads();
    gload(kc + LIN_KC < K ? kc + LIN_KC : kc);  // (unconditional: past the end the last chunk is requested again and never staged)
    for (int t = 0; t < (kk >> 3); ++t) {
      const f32x4 a = *(const f32x4*)&As[swz4(wr + li, 8 * t + 4 * hh, LIN_KC)];
      const f32x4 b = *(const f32x4*)&Bs[swz4(wc + li, 8 * t + 4 * hh, LIN_KC)];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
    }
  }
  // bias and activation on all sixteen values BEFORE the guarded stores (the bias was requested at kernel start): with the load
  // behind `bias ? ... : 0` inside the guard, hipcc waited `vmcnt(0)` in front of every one of the sixteen store blocks -- for the
  // bias the first time and for the PREVIOUS STORE every time after: sixteen store round trips in a row (round 6, seen in the ISA)
  const int col = nb + wc + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = acc[r] + bv;
    acc[r] = (act & 1) ? fmaxf(v, 0.f) : v;
  }
  if (col < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (row < M) C[(size_t)row * ldc + col] = acc[r];
    }
  }
}

// ---- per-step re-layout of the weight matrices into MFMA fragment order (see common.cuh:mma_lds)
struct FragDescs {
  upnerf_frag_desc d[UPNERF_MAX_FRAG_DESC];
  int start[UPNERF_MAX_FRAG_DESC + 1];  // prefix sums of rows*cols
  int n;
};
__global__ void frag_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, FragDescs D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.start[D.n]) return;
  int j = 0;
  while (idx >= D.start[j + 1]) ++j;
  const upnerf_frag_desc q = D.d[j];
  const int e = idx - D.start[j];
  const int r = e / q.cols, c = e - r * q.cols;
  const float v = q.transpose ? src[q.src_off + (size_t)c * q.src_ld + r] : src[q.src_off + (size_t)r * q.src_ld + c];
  const int k = q.dst_k0 + c;
  dst[q.dst_off + ((size_t)(r >> 5) * (q.dst_kp >> 3) + (k >> 3)) * 256 + ((((k >> 2) & 1) << 5) + (r & 31)) * 4 + (k & 3)] = v;
}

// ---- packed parameter buffer <-> named parameters (upnerf_amd/packing.py:NerfPacker.pack and its backward)
struct PackDescs {
  upnerf_pack_desc d[UPNERF_MAX_PACK_DESC];
  int start[UPNERF_MAX_PACK_DESC + 1];  // prefix sums of rows * cols
  int n;
};
// P[dst_off + r*dst_ld + c] (=|+=) src[r*src_ld + c]   (padding columns are not touched: P starts zeroed)
__global__ void pack_kernel(float* __restrict__ P, PackDescs D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.start[D.n]) return;
  int j = 0;
  while (idx >= D.start[j + 1]) ++j;
  const upnerf_pack_desc q = D.d[j];
  const int e = idx - D.start[j];
  const int r = e / q.cols, c = e - r * q.cols;
  const float v = q.ptr[(size_t)r * q.src_ld + c];
  float* dst = &P[q.dst_off + (size_t)r * q.dst_ld + c];
  if (q.accumulate) *dst += v;
  else *dst = v;
}
// param_grad[r*src_ld + c] = dP[dst_off + r*dst_ld + c]
__global__ void unpack_kernel(const float* __restrict__ dP, PackDescs D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.start[D.n]) return;
  int j = 0;
  while (idx >= D.start[j + 1]) ++j;
  const upnerf_pack_desc q = D.d[j];
  const int e = idx - D.start[j];
  const int r = e / q.cols, c = e - r * q.cols;
  ((float*)q.ptr)[(size_t)r * q.src_ld + c] = dP[q.dst_off + (size_t)r * q.dst_ld + c];
}

// ---- out_j = a_j + b_j for up to UPNERF_MAX_ADD_PAIRS tensors in one launch: the sum autograd forms when a tensor feeds two
// consumers (ray origins / directions into the coarse and the fine pass; feat_share_layer into the folded colour matrix and the
// per-ray projection), taken over by ops._Fanout so that a step carries ONE such launch per fan-out instead of an ATen add per tensor
struct AddPairs {
  upnerf_add_pair d[UPNERF_MAX_ADD_PAIRS];
  int start[UPNERF_MAX_ADD_PAIRS + 1];
  int n;
};
__global__ void add_pairs_kernel(AddPairs D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D.start[D.n]) return;
  int j = 0;
  while (idx >= D.start[j + 1]) ++j;
  const int e = idx - D.start[j];
  D.d[j].out[e] = D.d[j].a[e] + D.d[j].b[e];
}

// ---- f16x3 weight re-layout (hi/lo fp16, fragment order, one power-of-two exponent per matrix id)
struct Frag16Descs {
  upnerf_frag16_desc d[UPNERF_MAX_FRAG_DESC];
  int start[UPNERF_MAX_FRAG_DESC + 1];
  int n;
};
__device__ __forceinline__ int frag16_exp(float mx) {
  if (!(mx > 0.0f)) return 0;
  int ex;
  (void)frexpf(mx, &ex);
  const int e = 14 - ex;
  return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__device__ __forceinline__ float frag16_src(const float* __restrict__ src, const upnerf_frag16_desc& q, int r, int c) {
  return q.transpose ? src[q.src_off + (size_t)c * q.src_ld + r] : src[q.src_off + (size_t)r * q.src_ld + c];
}
// Zeroing by kernel, not hipMemsetAsync: inside a captured HIP graph a memset node was observed to lose its order against
// the kernel nodes around it when the null stream had run a kernel since the previous replay (the maxima then still held
// what the previous owner of the memory left there; tools/graph_vs_eager_trainer.py found it).  A kernel node is ordered.
__global__ void zero_floats_kernel(float* __restrict__ p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0.f;
}

// Maxima per matrix id.  A thread walks AMAX_PER elements (stride 256: coalesced; a transposed descriptor is walked along
// its source rows -- the maximum does not care about the order) and only touches the table when its id changes; a workgroup
// whose elements all carry one id -- descriptors are long runs -- issues ONE atomic.  (One element per thread was 2 400
// workgroups = 2 400 atomics on a handful of addresses: 32 us for 2.4 MB.)
#define AMAX_PER 8
// (both descriptor sets of upnerf_frag16 -- forward and transposed -- in one launch: blockIdx.y picks the set)
__global__ __launch_bounds__(256) void frag16_amax_kernel(const float* __restrict__ src, Frag16Descs D0, Frag16Descs D1,
                                                          float* __restrict__ amax) {
  const Frag16Descs& D = blockIdx.y ? D1 : D0;
  const int total = D.start[D.n];
  if ((int)(blockIdx.x * AMAX_PER * 256) >= total) return;
  float v = 0.0f;
  int id = -2;  // -2: nothing seen yet
  int j = 0;
#pragma unroll
  for (int u = 0; u < AMAX_PER; ++u) {
    const int idx = (blockIdx.x * AMAX_PER + u) * 256 + threadIdx.x;
    if (idx < total) {
      while (idx >= D.start[j + 1]) ++j;
      const upnerf_frag16_desc q = D.d[j];
      const int e = idx - D.start[j];
      int r, c;
      if (q.transpose) { c = e / q.rows; r = e - c * q.rows; }
      else { r = e / q.cols; c = e - r * q.cols; }
      const float x = fabsf(frag16_src(src, q, r, c));
      if (q.exp_id != id) {
        if (id >= 0) atomicMax((unsigned int*)&amax[id], __float_as_uint(v));
        id = q.exp_id;
        v = x;
      } else {
        v = fmaxf(v, x);
      }
    }
  }
  __shared__ float wmax[4];
  __shared__ int wid[4];
  const int myid = id;
  const float myv = v;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float ov = __shfl_xor(v, d);
    const int oid = __shfl_xor(id, d);
    if (oid != id) id = -1;  // mixed ids inside the wave (spreads to every lane by the end of the butterfly)
    v = fmaxf(v, ov);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    wmax[wave] = v;
    wid[wave] = id;
  }
  __syncthreads();
  const bool uniform = wid[0] >= 0 && wid[0] == wid[1] && wid[1] == wid[2] && wid[2] == wid[3];
  if (uniform) {
    if (threadIdx.x == 0)
      atomicMax((unsigned int*)&amax[wid[0]], __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
  } else if (myid >= 0) {
    atomicMax((unsigned int*)&amax[myid], __float_as_uint(myv));
  }
}
// wnorm[j] = max over the rows r of descriptor j of sum_c |X[r][c]|  (the operator norm that bounds |X h|_inf by
// wnorm * |h|_inf: the register-resident field kernels pick their activation exponents from it before a layer's
// outputs exist).  One wave per row; non-negative floats order like their bit patterns.
// Both fragment sets in one launch (blockIdx.y); a wave takes RN_ROWS consecutive rows and issues one atomic per descriptor it
// touched (one wave per row and one atomic per row made 3 000 atomics on twenty addresses: 35 us per set).
#define RN_ROWS 8
__global__ __launch_bounds__(256) void frag16_rownorm_kernel(const float* __restrict__ src, Frag16Descs D0, Frag16Descs D1,
                                                             float* __restrict__ wnorm) {
  const Frag16Descs& D = blockIdx.y ? D1 : D0;
  float* __restrict__ out = wnorm + 32 * blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int row = (blockIdx.x * 4 + wave) * RN_ROWS, j = 0;
  while (j < D.n && row >= D.d[j].rows) row -= D.d[j++].rows;
  float best = 0.0f;
  for (int i = 0; i < RN_ROWS && j < D.n; ++i) {
    const upnerf_frag16_desc q = D.d[j];
    float s = 0.0f;
    for (int c = lane; c < q.cols; c += 64) s += fabsf(frag16_src(src, q, row, c));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    best = fmaxf(best, s);
    if (++row >= q.rows) {  // last row of this descriptor: flush, move on
      if (lane == 0) atomicMax((unsigned int*)&out[j], __float_as_uint(best));
      best = 0.0f;
      row = 0;
      ++j;
    }
  }
  if (j < D.n && lane == 0 && best > 0.0f) atomicMax((unsigned int*)&out[j], __float_as_uint(best));
}

// perm: k order inside a 16-deep block.  0: element j of lane half h holds k = 8h + j (operands read from memory);
// 1: k = 8(j>>2) + 4h + (j&3) -- the order in which a 32x32 MFMA result, converted in place, presents its rows as the next
// product's operand (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand").
__global__ void frag16_write_kernel(const float* __restrict__ src, char* __restrict__ dst0, char* __restrict__ dst1, Frag16Descs D0,
                                    Frag16Descs D1, const float* __restrict__ amax, int* __restrict__ wexp, int perm0, int perm1) {
  const Frag16Descs& D = blockIdx.y ? D1 : D0;
  char* __restrict__ dst = blockIdx.y ? dst1 : dst0;
  const int perm = blockIdx.y ? perm1 : perm0;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (wexp && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 16) wexp[threadIdx.x] = frag16_exp(amax[threadIdx.x]);
  if (idx >= D.start[D.n]) return;
  int j = 0;
  while (idx >= D.start[j + 1]) ++j;
  const upnerf_frag16_desc q = D.d[j];
  const int e = idx - D.start[j];
  const int r = e / q.cols, c = e - r * q.cols;
  const float x = ldexpf(frag16_src(src, q, r, c), frag16_exp(amax[q.exp_id]));
  const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
  const int k = q.dst_k0 + c;
  const int kk = k & 15;
  const int h = perm ? (kk >> 2) & 1 : kk >> 3, jj = perm ? ((kk >> 3) << 2) | (kk & 3) : kk & 7;
  const size_t base = (size_t)q.dst_off * 4 + ((size_t)(r >> 5) * (q.dst_kp >> 4) + (k >> 4)) * 2048 +
                      ((h << 5) + (r & 31)) * 16 + jj * 2;
  *(_Float16*)(dst + base) = hi;
  *(_Float16*)(dst + base + 1024) = lo;
}

// ---- Adam (torch.optim.Adam, no weight decay / amsgrad): same op order as torch's single-tensor path.  step_size =
// lr / bias_corr1 and bc2_sqrt = sqrt(bias_corr2) are formed by the host in double precision and rounded to fp32, as
// torch forms its Python scalars; dyn2 (device, [step_size, bc2_sqrt]) overrides the by-value pair under graph replay.
// One element's update, shared by the flat and the gathering kernel (one expression tree, one set of roundings: no contraction)
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float gi, float b1, float b2, float eps, float step_size,
                                            float bc2_sqrt) {
#pragma clang fp contract(off)
  const float mi = m + (gi - m) * (1.f - b1);        // exp_avg.lerp_(grad, 1-beta1)
  const float vi = v * b2 + (1.f - b2) * gi * gi;    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
  m = mi;
  v = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p = p - step_size * (mi / denom);
}
__global__ void adam_kernel(long long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float b1, float b2, float eps, float step_size, float bc2_sqrt,
                            const float* __restrict__ dyn2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dyn2) {
    step_size = dyn2[0];
    bc2_sqrt = dyn2[1];
  }
  adam_update(p[i], m[i], v[i], g[i], b1, b2, eps, step_size, bc2_sqrt);
}

struct ScalarPack {
  float v[UPNERF_MAX_SCALARS];
};
// out[i] = 14 - ceil(log2(max(v[i], 1e-30))): the power-of-two exponent that brings a tracked maximum to ~2^14 (the scales
// of the f16x3 weight gradients), the arithmetic of the five ATen launches it replaces, in their order
__global__ void scale_exponents_kernel(const float* __restrict__ v, int n, int* __restrict__ out) {
  const int i = threadIdx.x;
  if (i < n) out[i] = (int)(14.0f - ceilf(log2f(fmaxf(v[i], 1e-30f))));
}

__global__ void set_scalars_kernel(float* __restrict__ dst, int n, ScalarPack s) {
  const int i = threadIdx.x;
  if (i < n) dst[i] = s.v[i];
}

template <int MTW, int NTW>
int launch_wgrad(int M, const float* A, int lda, int N, const float* B, int ldb, int K, float* slabs, float* bslabs,
                 int nsplit, int rows_per_split, hipStream_t st) {
  constexpr int TN = 64 * MTW, TK = 64 * NTW;
  dim3 grid(nsplit, (N + TN - 1) / TN, (K + TK - 1) / TK);
  hipLaunchKernelGGL((wgrad_kernel<MTW, NTW>), grid, dim3(NTHREADS), 0, st, M, N, K, A, lda, B, ldb, slabs, bslabs,
                     rows_per_split);
  return (int)hipGetLastError();
}

}  // namespace

// Scratch needed by upnerf_wgrad for (N, K, nsplit): nsplit * (roundup(N) * roundup(K) + roundup(N)) floats,
// where roundup() is to the block shape chosen below (<= 256).
static void wgrad_shape(int N, int K, int* TN, int* TK) {
  *TN = N >= 256 ? 256 : (N > 64 ? 128 : 64);
  *TK = K >= 256 ? 256 : (K > 64 ? 128 : 64);
}

extern "C" int upnerf_wgrad(int M, const float* A, int lda, int N, const float* B, int ldb, int K, float* dW, int ldo,
                            float* db, float* slabs, int nsplit, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !dW || !slabs || nsplit <= 0) return UPNERF_EINVAL;
  if ((N & 3) || (K & 3) || (lda & 3) || (ldb & 3)) return UPNERF_EINVAL;
  int TN, TK;
  wgrad_shape(N, K, &TN, &TK);
  hipStream_t st = (hipStream_t)stream;
  const int rows = (((M + nsplit - 1) / nsplit) + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
  const int gy = (N + TN - 1) / TN, gz = (K + TK - 1) / TK;
  float* bslabs = slabs + (size_t)nsplit * gy * gz * TN * TK;
  int rc;
  if (TN == 256 && TK == 256) rc = launch_wgrad<4, 4>(M, A, lda, N, B, ldb, K, slabs, bslabs, nsplit, rows, st);
  else if (TN == 256 && TK == 128) rc = launch_wgrad<4, 2>(M, A, lda, N, B, ldb, K, slabs, bslabs, nsplit, rows, st);
  else if (TN == 256 && TK == 64) rc = launch_wgrad<4, 1>(M, A, lda, N, B, ldb, K, slabs, bslabs, nsplit, rows, st);
  else if (TN == 128 && TK == 256) rc = launch_wgrad<2, 4>(M, A, lda, N, B, ldb, K, slabs, bslabs, nsplit, rows, st);
  else if (TN == 128 && TK == 128) rc = launch_wgrad<2, 2>(M, A, lda, N, B, ldb, K, slabs, bslabs, nsplit, rows, st);
  else if (TN == 128 && TK == 64) rc = launch_wgrad<2, 1>(M, A, lda, N, B, ldb, K, slabs, bslabs, nsplit, rows, st);
  else if (TN == 64 && TK == 256) rc = launch_wgrad<1, 4>(M, A, lda, N, B, ldb, K, slabs, bslabs, nsplit, rows, st);
  else if (TN == 64 && TK == 128) rc = launch_wgrad<1, 2>(M, A, lda, N, B, ldb, K, slabs, bslabs, nsplit, rows, st);
  else rc = launch_wgrad<1, 1>(M, A, lda, N, B, ldb, K, slabs, bslabs, nsplit, rows, st);
  if (rc) return rc;
  if (ldo & 3) return UPNERF_EINVAL;  // 16-byte stores into dW
  launch_reduce(st, reduce_desc(N, K, TN, TK, nsplit, slabs, bslabs, dW, ldo, db));  // (the bias sum needs one thread per row)
  return (int)hipGetLastError();
}

extern "C" int upnerf_wgrad_f16x3_partial(int M, const float* A, int lda, int N, const float* B, int ldb, int K,
                                          const int* expo_a, const int* expo_b, float* slabs, float* bslabs,
                                          int nsplit, int rows, int TN, int TK, int planes, const upnerf_wgrad_pending* prev,
                                          void* stream);

// Same contract as upnerf_wgrad, contraction on the f16 matrix cores (wgrad_f16x3.hip): planes 0 / 2 = 3-term hi/lo
// split (fp32-level accuracy), planes 1 = operands rounded to fp16, one MFMA per block.
// expo_a, expo_b: DEVICE ints: A is scaled by 2^*expo_a and B by 2^*expo_b before the conversion to fp16.
extern "C" int upnerf_wgrad_f16x3(int M, const float* A, int lda, int N, const float* B, int ldb, int K, float* dW,
                                  int ldo, float* db, float* slabs, int nsplit, const int* expo_a, const int* expo_b,
                                  int planes, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !dW || !slabs || nsplit <= 0 || !expo_a || !expo_b)
    return UPNERF_EINVAL;
  if (planes != 0 && planes != 1 && planes != 2) return UPNERF_EINVAL;
  if ((N & 3) || (K & 3) || (lda & 3) || (ldb & 3) || (ldo & 3)) return UPNERF_EINVAL;
  int TN, TK;
  wgrad_shape(N, K, &TN, &TK);
  hipStream_t st = (hipStream_t)stream;
  const int rows = (((M + nsplit - 1) / nsplit) + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
  const int gy = (N + TN - 1) / TN, gz = (K + TK - 1) / TK;
  float* bslabs = slabs + (size_t)nsplit * gy * gz * TN * TK;
  int rc = upnerf_wgrad_f16x3_partial(M, A, lda, N, B, ldb, K, expo_a, expo_b, slabs, bslabs, nsplit, rows, TN, TK,
                                      planes, nullptr, stream);
  if (rc) return rc;
  launch_reduce(st, reduce_desc(N, K, TN, TK, nsplit, slabs, bslabs, dW, ldo, db));
  return (int)hipGetLastError();
}

extern "C" int upnerf_wgrad_finish(upnerf_wgrad_pending* p, void* stream) {
  if (!p) return UPNERF_EINVAL;
  if (p->nsplit <= 0) return 0;
  launch_reduce((hipStream_t)stream, *p);
  p->nsplit = 0;
  return (int)hipGetLastError();
}

// Chained upnerf_wgrad_f16x3 (include/upnerf_hip.h): the previous problem's slabs are summed by this launch's first workgroups.
extern "C" int upnerf_wgrad_f16x3_chain2(int M, const float* A, int lda, int N, const float* B, int ldb, int K, float* dW, int ldo,
                                         float* db, int n2, float* dW2, int ldo2, float* db2, float* slabs, int nsplit,
                                         const int* expo_a, const int* expo_b, int planes, upnerf_wgrad_pending* pending,
                                         void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !dW || !slabs || nsplit <= 0 || !expo_a || !expo_b || !pending)
    return UPNERF_EINVAL;
  if (planes != 0 && planes != 1 && planes != 2) return UPNERF_EINVAL;
  if ((N & 3) || (K & 3) || (lda & 3) || (ldb & 3) || (ldo & 3)) return UPNERF_EINVAL;
  if (n2 < 0 || n2 >= N || (n2 > 0 && (!dW2 || (ldo2 & 3)))) return UPNERF_EINVAL;
  if (pending->nsplit > 0 && pending->slabs == slabs) return UPNERF_EINVAL;  // the pending slabs would be overwritten
  int TN, TK;
  wgrad_shape(N, K, &TN, &TK);
  const int rows = (((M + nsplit - 1) / nsplit) + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
  const int gy = (N + TN - 1) / TN, gz = (K + TK - 1) / TK;
  float* bslabs = slabs + (size_t)nsplit * gy * gz * TN * TK;
  if (pending->nsplit > 0 && pending->rblocks > nsplit * gy * gz) {  // grid too small to carry the previous reduction
    int rc = upnerf_wgrad_finish(pending, stream);
    if (rc) return rc;
  }
  int rc = upnerf_wgrad_f16x3_partial(M, A, lda, N, B, ldb, K, expo_a, expo_b, slabs, bslabs, nsplit, rows, TN, TK, planes,
                                      pending->nsplit > 0 ? pending : nullptr, stream);
  if (rc) return rc;
  upnerf_wgrad_pending P = reduce_desc(N, K, TN, TK, nsplit, slabs, bslabs, dW, ldo, db);
  P.n2 = n2;
  P.dW2 = dW2;
  P.db2 = db2;
  P.ldo2 = ldo2;
  *pending = P;
  return 0;
}

extern "C" int upnerf_wgrad_f16x3_partial_v(int M, const float* A, int lda, const float* B, int ldb, const float* v, const int* expo_a,
                                            const int* expo_b, float* slabs, float* bslabs, float* vslabs, int nsplit, int rows,
                                            const upnerf_wgrad_pending* prev, void* stream);  // csrc/wgrad_f16x3.hip

// upnerf_wgrad_f16x3_chain + the 1-wide head that shares B (include/upnerf_hip.h)
extern "C" int upnerf_wgrad_f16x3_chain_v(int M, const float* A, int lda, int N, const float* B, int ldb, int K, float* dW, int ldo,
                                          float* db, const float* v, float* dv, float* dbv, float* slabs, int nsplit,
                                          const int* expo_a, const int* expo_b, int planes, upnerf_wgrad_pending* pending,
                                          void* stream) {
  if (M <= 0 || !A || !B || !dW || !v || !dv || !slabs || nsplit <= 0 || !expo_a || !expo_b || !pending) return UPNERF_EINVAL;
  if (N != 256 || K != 256 || planes != 2) return UPNERF_EUNSUP;
  if ((lda & 3) || (ldb & 3) || (ldo & 3)) return UPNERF_EINVAL;
  if (pending->nsplit > 0 && pending->slabs == slabs) return UPNERF_EINVAL;
  const int TN = 256, TK = 256;
  const int rows = (((M + nsplit - 1) / nsplit) + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
  float* bslabs = slabs + (size_t)nsplit * TN * TK;
  float* vslabs = bslabs + (size_t)nsplit * TN;
  if (pending->nsplit > 0 && pending->rblocks > nsplit) {  // grid too small to carry the previous reduction
    int rc = upnerf_wgrad_finish(pending, stream);
    if (rc) return rc;
  }
  int rc = upnerf_wgrad_f16x3_partial_v(M, A, lda, B, ldb, v, expo_a, expo_b, slabs, bslabs, vslabs, nsplit, rows,
                                        pending->nsplit > 0 ? pending : nullptr, stream);
  if (rc) return rc;
  upnerf_wgrad_pending P = reduce_desc(N, K, TN, TK, nsplit, slabs, bslabs, dW, ldo, db);
  P.vslabs = vslabs;
  P.dv = dv;
  P.dbv = dbv;
  *pending = P;
  return 0;
}

extern "C" int upnerf_wgrad_f16x3_chain(int M, const float* A, int lda, int N, const float* B, int ldb, int K, float* dW, int ldo,
                                        float* db, float* slabs, int nsplit, const int* expo_a, const int* expo_b, int planes,
                                        upnerf_wgrad_pending* pending, void* stream) {
  return upnerf_wgrad_f16x3_chain2(M, A, lda, N, B, ldb, K, dW, ldo, db, 0, nullptr, 0, nullptr, slabs, nsplit, expo_a, expo_b, planes,
                                   pending, stream);
}

extern "C" int upnerf_wgrad_f16p_partial(int M, const uint16_t* A16, int lda, const int* aexp, int N, const void* B, int ldb,
                                         const int* bexp, int b_is_f16, int K, const int* expo_a, const int* expo_b, float* slabs,
                                         float* bslabs, int nsplit, int rows, int TN, int TK, const upnerf_wgrad_pending* prev,
                                         const uint8_t* Alo, const uint8_t* Blo, void* stream);

// upnerf_wgrad with fp16-stored, tile-scaled operands (the f16 field mode): same slabs + fixed-order reduction.
extern "C" int upnerf_wgrad_f16p(int M, const uint16_t* A16, int lda, const int32_t* aexp, int N, const void* B, int ldb,
                                 const int32_t* bexp, int b_is_f16, int K, float* dW, int ldo, float* db, float* slabs, int nsplit,
                                 const int* expo_a, const int* expo_b, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A16 || !aexp || !B || !dW || !slabs || nsplit <= 0 || !expo_a || !expo_b)
    return UPNERF_EINVAL;
  if ((b_is_f16 & 1) && !bexp) return UPNERF_EINVAL;
  if ((N & 7) || (K & 7) || (lda & 7) || (ldb & 7) || (ldo & 3)) return UPNERF_EINVAL;
  int TN, TK;
  wgrad_shape(N, K, &TN, &TK);
  hipStream_t st = (hipStream_t)stream;
  const int rows = (((M + nsplit - 1) / nsplit) + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
  const int gy = (N + TN - 1) / TN, gz = (K + TK - 1) / TK;
  float* bslabs = slabs + (size_t)nsplit * gy * gz * TN * TK;
  int rc = upnerf_wgrad_f16p_partial(M, A16, lda, aexp, N, B, ldb, bexp, b_is_f16, K, expo_a, expo_b, slabs, bslabs, nsplit, rows,
                                     TN, TK, nullptr, nullptr, nullptr, stream);
  if (rc) return rc;
  launch_reduce(st, reduce_desc(N, K, TN, TK, nsplit, slabs, bslabs, dW, ldo, db));
  return (int)hipGetLastError();
}

// Chained upnerf_wgrad_f16p: same pending record as upnerf_wgrad_f16x3_chain (one run may mix both kinds of launches).
static int wgrad_f16p_chain_impl(int M, const uint16_t* A16, const uint8_t* Alo, int lda, const int32_t* aexp, int N, const void* B,
                                 const uint8_t* Blo, int ldb, const int32_t* bexp, int b_is_f16, int K, float* dW, int ldo, float* db,
                                 int n2, float* dW2, int ldo2, float* db2, float* slabs, int nsplit, const int* expo_a,
                                 const int* expo_b, upnerf_wgrad_pending* pending, void* stream) {
  if (n2 < 0 || n2 >= N || (n2 > 0 && (!dW2 || (ldo2 & 3)))) return UPNERF_EINVAL;
  if (M <= 0 || N <= 0 || K <= 0 || !A16 || !aexp || !B || !dW || !slabs || nsplit <= 0 || !expo_a || !expo_b || !pending)
    return UPNERF_EINVAL;
  if ((b_is_f16 & 1) && !bexp) return UPNERF_EINVAL;
  if ((N & 7) || (K & 7) || (lda & 7) || (ldb & 7) || (ldo & 3)) return UPNERF_EINVAL;
  if (pending->nsplit > 0 && pending->slabs == slabs) return UPNERF_EINVAL;  // the pending slabs would be overwritten
  int TN, TK;
  wgrad_shape(N, K, &TN, &TK);
  const int rows = (((M + nsplit - 1) / nsplit) + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
  const int gy = (N + TN - 1) / TN, gz = (K + TK - 1) / TK;
  float* bslabs = slabs + (size_t)nsplit * gy * gz * TN * TK;
  if (pending->nsplit > 0 && pending->rblocks > nsplit * gy * gz) {  // grid too small to carry the previous reduction
    int rc = upnerf_wgrad_finish(pending, stream);
    if (rc) return rc;
  }
  int rc = upnerf_wgrad_f16p_partial(M, A16, lda, aexp, N, B, ldb, bexp, b_is_f16, K, expo_a, expo_b, slabs, bslabs, nsplit, rows,
                                     TN, TK, pending->nsplit > 0 ? pending : nullptr, Alo, Blo, stream);
  if (rc) return rc;
  upnerf_wgrad_pending P = reduce_desc(N, K, TN, TK, nsplit, slabs, bslabs, dW, ldo, db);
  P.n2 = n2;
  P.dW2 = dW2;
  P.db2 = db2;
  P.ldo2 = ldo2;
  *pending = P;
  return 0;
}

extern "C" int upnerf_wgrad_f16p_partial_v(int M, const uint16_t* A16, const int* aexp, const uint16_t* B16, const int* bexp, const float* v,
                                           const int* expo_a, const int* expo_b, float* slabs, float* bslabs, float* vslabs, int nsplit,
                                           int rows, const upnerf_wgrad_pending* prev, void* stream);  // csrc/wgrad_f16x3.hip

// upnerf_wgrad_f16p_chain on fragment-ordered 256 x 256 operands + the 1-wide head that shares B (include/upnerf_hip.h)
extern "C" int upnerf_wgrad_f16p_chain_v(int M, const uint16_t* A16, const int32_t* aexp, const uint16_t* B16, const int32_t* bexp,
                                         float* dW, int ldo, float* db, const float* v, float* dv, float* dbv, float* slabs, int nsplit,
                                         const int* expo_a, const int* expo_b, upnerf_wgrad_pending* pending, void* stream) {
  if (M <= 0 || !A16 || !aexp || !B16 || !bexp || !dW || !v || !dv || !slabs || nsplit <= 0 || !expo_a || !expo_b || !pending || (ldo & 3))
    return UPNERF_EINVAL;
  if (pending->nsplit > 0 && pending->slabs == slabs) return UPNERF_EINVAL;
  const int TN = 256, TK = 256;
  const int rows = (((M + nsplit - 1) / nsplit) + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
  float* bslabs = slabs + (size_t)nsplit * TN * TK;
  float* vslabs = bslabs + (size_t)nsplit * TN;
  if (pending->nsplit > 0 && pending->rblocks > nsplit) {
    int rc = upnerf_wgrad_finish(pending, stream);
    if (rc) return rc;
  }
  int rc = upnerf_wgrad_f16p_partial_v(M, A16, aexp, B16, bexp, v, expo_a, expo_b, slabs, bslabs, vslabs, nsplit, rows,
                                       pending->nsplit > 0 ? pending : nullptr, stream);
  if (rc) return rc;
  upnerf_wgrad_pending P = reduce_desc(256, 256, TN, TK, nsplit, slabs, bslabs, dW, ldo, db);
  P.vslabs = vslabs;
  P.dv = dv;
  P.dbv = dbv;
  *pending = P;
  return 0;
}

extern "C" int upnerf_wgrad_planes_partial(int M, const uint16_t* Ah, const uint16_t* Al, const int* aexp, const uint16_t* Bh,
                                           const uint16_t* Bl, const int* bexp, const int* expo_a, const int* expo_b, float* slabs,
                                           float* bslabs, int nsplit, int rows, const upnerf_wgrad_pending* prev, void* stream);  // csrc/wgrad_f16x3.hip

// dW[256][ldo] = sum_m A[m][:]^T B[m][:] from producer-split operands (include/upnerf_hip.h): a link of a chained run
extern "C" int upnerf_wgrad_planes_chain(int M, const uint16_t* A16, const uint16_t* Alo16, const int32_t* aexp, const uint16_t* B16,
                                         const uint16_t* Blo16, const int32_t* bexp, float* dW, int ldo, float* db, float* slabs, int nsplit,
                                         const int* expo_a, const int* expo_b, upnerf_wgrad_pending* pending, void* stream) {
  if (M <= 0 || !A16 || !Alo16 || !aexp || !B16 || !Blo16 || !bexp || !dW || !slabs || nsplit <= 0 || !expo_a || !expo_b || !pending || (ldo & 3))
    return UPNERF_EINVAL;
  if (M & 63) return UPNERF_EUNSUP;  // whole 64-row tiles (one exponent each) only
  if (pending->nsplit > 0 && pending->slabs == slabs) return UPNERF_EINVAL;
  const int TN = 256, TK = 256;
  const int rows = (((M + nsplit - 1) / nsplit) + 63) / 64 * 64;
  float* bslabs = slabs + (size_t)nsplit * TN * TK;
  if (pending->nsplit > 0 && pending->rblocks > nsplit) {
    int rc = upnerf_wgrad_finish(pending, stream);
    if (rc) return rc;
  }
  int rc = upnerf_wgrad_planes_partial(M, A16, Alo16, aexp, B16, Blo16, bexp, expo_a, expo_b, slabs, bslabs, nsplit, rows,
                                       pending->nsplit > 0 ? pending : nullptr, stream);
  if (rc) return rc;
  *pending = reduce_desc(256, 256, TN, TK, nsplit, slabs, bslabs, dW, ldo, db);
  return 0;
}

extern "C" int upnerf_vec_wgrad_frag16(int M, const float* v, int ldv, int nvec, const uint16_t* X16, const int32_t* xexp, int K,
                                       float* dw, float* dbv, float* scratch, int nsplit, void* stream) {
  if (M <= 0 || !v || !X16 || !xexp || !dw || !scratch || nsplit <= 0 || ldv < nvec) return UPNERF_EINVAL;
  if ((K != 256 && K != 128) || (nvec != 1 && nvec != 3)) return UPNERF_EUNSUP;
  hipStream_t st = (hipStream_t)stream;
  const int ntile = (M + 31) / 32;
  const int per = (ntile + nsplit - 1) / nsplit;
#define VWF(KB, NV) hipLaunchKernelGGL((vec_wgrad_frag16_kernel<KB, NV>), dim3(nsplit), dim3(64 * KB), 0, st, M, v, ldv, X16, xexp, scratch, per)
  if (K == 256 && nvec == 1) VWF(16, 1);
  else if (K == 256) VWF(16, 3);
  else if (nvec == 1) VWF(8, 1);
  else VWF(8, 3);
#undef VWF
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  const int total = nvec * (K + 1);
  hipLaunchKernelGGL(vec_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, nvec, K, nsplit, scratch, dw, dbv);
  return (int)hipGetLastError();
}

extern "C" int upnerf_wgrad_f16p_chain(int M, const uint16_t* A16, int lda, const int32_t* aexp, int N, const void* B, int ldb,
                                       const int32_t* bexp, int b_is_f16, int K, float* dW, int ldo, float* db, int n2, float* dW2,
                                       int ldo2, float* db2, float* slabs, int nsplit, const int* expo_a, const int* expo_b,
                                       upnerf_wgrad_pending* pending, void* stream) {
  return wgrad_f16p_chain_impl(M, A16, nullptr, lda, aexp, N, B, nullptr, ldb, bexp, b_is_f16, K, dW, ldo, db, n2, dW2, ldo2, db2, slabs,
                               nsplit, expo_a, expo_b, pending, stream);
}
// The same with "24-bit" operands: A16 / Alo8 (and B16 / Blo8 when b_is_f16 = 1; a fp32 B is split in the kernel) hold hi + lo8 as
// written by the f16x3 field kernels (upnerf_field_fwd_args.h_lo8, upnerf_field_bwd_args.gz_lo8); three MFMAs per block.
extern "C" int upnerf_wgrad_f24p_chain(int M, const uint16_t* A16, const uint8_t* Alo8, int lda, const int32_t* aexp, int N,
                                       const void* B, const uint8_t* Blo8, int ldb, const int32_t* bexp, int b_is_f16, int K, float* dW,
                                       int ldo, float* db, float* slabs, int nsplit, const int* expo_a, const int* expo_b,
                                       upnerf_wgrad_pending* pending, void* stream) {
  if (!Alo8 || (b_is_f16 & 2)) return UPNERF_EINVAL;
  return wgrad_f16p_chain_impl(M, A16, Alo8, lda, aexp, N, B, Blo8, ldb, bexp, b_is_f16, K, dW, ldo, db, 0, nullptr, 0, nullptr, slabs, nsplit,
                               expo_a, expo_b, pending, stream);
}

// ---- small matrix-vector products of the folded colour layer (packing): y[m] = add[m] + sum_k A[m][k] x[k] (trans = 0, one wave
// per row, lane-strided partial sums + a shuffle tree: fixed order) or y[k] = sum_m A[m][k] x[m] (trans = 1: 16 columns per
// workgroup, 16 row classes m = p (mod 16) per column with all of a class's loads in flight at once, the classes added in order
// out of LDS -- one thread per column walking the 128 rows one dependent load at a time was a 33 us launch).
__global__ __launch_bounds__(256) void matvec_kernel(int M, int K, const float* __restrict__ A, int lda, const float* __restrict__ x,
                                                     const float* __restrict__ add, float* __restrict__ y, int trans,
                                                     float* __restrict__ R1 = nullptr, int ldr = 0, const float* __restrict__ v1 = nullptr) {
  if (trans) {
    __shared__ float part[16][17];
    const int c = threadIdx.x & 15, p = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + c;
    float s = 0.0f;
    if (k < K) {
      for (int m0 = p; m0 < M; m0 += 16 * 8) {
        float a[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int m = m0 + 16 * u;
          const int mc = m < M ? m : M - 1;
          a[u] = A[(size_t)mc * lda + k];
          xv[u] = m < M ? x[mc] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u] * xv[u];
        if (R1) {  // upnerf_matvec_rank1: R[m][k] += x[m] v[k] on the elements this thread visits anyway (each exactly once)
          const float vk = v1[k];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int m = m0 + 16 * u;
            if (m < M) R1[(size_t)m * ldr + k] += xv[u] * vk;
          }
        }
      }
    }
    part[p][c] = s;
    __syncthreads();
    if (p == 0 && k < K) {
      float t = 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += part[q][c];
      y[k] = t + (add ? add[k] : 0.0f);
    }
    return;
  }
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  float s = 0.0f;
  for (int k = lane; k < K; k += 64) s += A[(size_t)m * lda + k] * x[k];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if (lane == 0) y[m] = s + (add ? add[m] : 0.0f);
}
extern "C" int upnerf_matvec(int M, int K, const float* A, int lda, const float* x, const float* add, float* y, int trans,
                             void* stream) {
  if (M <= 0 || K <= 0 || !A || !x || !y || lda < K) return UPNERF_EINVAL;
  const int blocks = trans ? (K + 15) / 16 : (M + 3) / 4;
  hipLaunchKernelGGL(matvec_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, M, K, A, lda, x, add, y, trans);
  return (int)hipGetLastError();
}

extern "C" int upnerf_matvec_rank1(int M, int K, const float* A, int lda, const float* x, float* y, float* R, int ldr, const float* v,
                                   void* stream) {
  if (M <= 0 || K <= 0 || !A || !x || !y || !R || !v || lda < K || ldr < K) return UPNERF_EINVAL;
  hipLaunchKernelGGL(matvec_kernel, dim3((K + 15) / 16), dim3(256), 0, (hipStream_t)stream, M, K, A, lda, x, (const float*)nullptr, y, 1, R,
                     ldr, v);
  return (int)hipGetLastError();
}

extern "C" int upnerf_wgrad_grouped_scratch(const upnerf_wgrad_group* groups, int ngroups, int nsplit) {
  if (!groups || ngroups <= 0 || ngroups > UPNERF_MAX_WGRAD_GROUPS || nsplit <= 0) return UPNERF_EINVAL;
  long long tiles = 0;
  for (int j = 0; j < ngroups; ++j)
    tiles += (long long)((groups[j].N + GT - 1) / GT) * ((groups[j].K + GT - 1) / GT);
  const long long n = tiles * nsplit * (GT * GT + GT);
  return n > 0x7fffffffLL ? UPNERF_EUNSUP : (int)n;
}

extern "C" int upnerf_wgrad_grouped(const upnerf_wgrad_group* groups, int ngroups, float* scratch, int nsplit,
                                    void* stream) {
  if (!groups || ngroups <= 0 || ngroups > UPNERF_MAX_WGRAD_GROUPS || !scratch || nsplit <= 0) return UPNERF_EINVAL;
  WgradGroups T;
  T.n = ngroups;
  T.tile_start[0] = T.red_start[0] = 0;
  for (int j = 0; j < ngroups; ++j) {
    const upnerf_wgrad_group& q = groups[j];
    if (!q.A || !q.B || !q.dW || q.M <= 0 || q.N <= 0 || q.K <= 0 || (q.N & 3) || (q.K & 3) || (q.lda & 3) || (q.ldb & 3) ||
        (q.ldo & 3))
      return UPNERF_EINVAL;
    T.g[j] = q;
    T.tile_start[j + 1] = T.tile_start[j] + ((q.N + GT - 1) / GT) * ((q.K + GT - 1) / GT);
    int rb = (q.N * (q.K / 4) + NTHREADS - 1) / NTHREADS;
    if (rb * NTHREADS < q.N) rb = (q.N + NTHREADS - 1) / NTHREADS;  // the bias sum needs one thread per row
    T.red_start[j + 1] = T.red_start[j] + rb;
  }
  const int tiles = T.tile_start[ngroups];
  float* bslabs = scratch + (size_t)tiles * nsplit * GT * GT;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(nsplit, tiles), dim3(NTHREADS), 0, st, T, scratch, bslabs, nsplit);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_grouped_reduce_kernel, dim3(T.red_start[ngroups]), dim3(NTHREADS), 0, st, T, scratch, bslabs,
                     nsplit);
  return (int)hipGetLastError();
}

extern "C" int upnerf_vec_wgrad(int M, const float* v, int ldv, int nvec, const float* X, int ldx, int K, float* dw,
                                float* dbv, float* scratch, int nsplit, void* stream) {
  if (M <= 0 || nvec <= 0 || nvec > 4 || K <= 0 || K > 256 || !v || !X || !dw || !scratch || nsplit <= 0)
    return UPNERF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int rows = (M + nsplit - 1) / nsplit;
  if (nvec > 3 || (ldx & 3)) return UPNERF_EINVAL;
  if (K == 256) hipLaunchKernelGGL(vec_wgrad_kernel<256>, dim3(nsplit), dim3(NTHREADS), 0, st, M, v, ldv, nvec, X, ldx, scratch, rows);
  else if (K == 128) hipLaunchKernelGGL(vec_wgrad_kernel<128>, dim3(nsplit), dim3(NTHREADS), 0, st, M, v, ldv, nvec, X, ldx, scratch, rows);
  else if (K == 64) hipLaunchKernelGGL(vec_wgrad_kernel<64>, dim3(nsplit), dim3(NTHREADS), 0, st, M, v, ldv, nvec, X, ldx, scratch, rows);
  else if (K == 32) hipLaunchKernelGGL(vec_wgrad_kernel<32>, dim3(nsplit), dim3(NTHREADS), 0, st, M, v, ldv, nvec, X, ldx, scratch, rows);
  else return UPNERF_EUNSUP;
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  const int total = nvec * (K + 1);
  hipLaunchKernelGGL(vec_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, nvec, K, nsplit, scratch, dw,
                     dbv);
  return (int)hipGetLastError();
}

// ---- finishing the per-tile partial sums of upnerf_field_bwd_f16x3 (upnerf_field_bwd_args.tile_part)
#define TP_STRIDE UPNERF_TILE_PART_STRIDE
#define TP_WCOLS 520   // weight part of a row: w_csig | w_r2 | 4 sums | 4 pad
#define TP_NS 128      // first-stage splits
namespace {
// rs[which][r][:] = sum over the tiles that hold rows of ray r of the tile's sums for that ray's slot (ascending tile order)
__device__ __forceinline__ void tile_part_rays_body(int idx, int which, int R, int S, const float* __restrict__ part,
                                                    float* __restrict__ rs_g1, float* __restrict__ rs_r1) {
  float* __restrict__ out = which == 0 ? rs_g1 : rs_r1;
  if (idx >= R * 32 || !out) return;
  const int r = idx >> 5, g = idx & 31;
  const long long b = (long long)r * S, e = b + S - 1;
  const int t0 = (int)(b / 64), t1 = (int)(e / 64);
  const int base = TP_WCOLS + (which == 0 ? 0 : 384) + 4 * g;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int t = t0; t <= t1; ++t) {
    const int j = r - (int)(((long long)t * 64) / S);
    s += *(const f32x4*)&part[(size_t)t * TP_STRIDE + base + 128 * j];
  }
  *(f32x4*)&out[(size_t)r * 128 + 4 * g] = s;
}
__global__ void tile_part_rays_kernel(int R, int S, const float* __restrict__ part, float* __restrict__ rs_g1,
                                      float* __restrict__ rs_r1) {
  tile_part_rays_body(blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y, R, S, part, rs_g1, rs_r1);
}
// out1[split][0..519] = sum of the weight part over the split's tiles: two row groups of 130 16-byte columns, 8 rows in flight
__device__ __forceinline__ void tile_part_sum1_body(int blk, int ntiles, int per, const float* __restrict__ part, float* __restrict__ out1,
                                                    float* red) {
  const int tid = threadIdx.x, c4 = tid % 130, rg = tid / 130;
  const int tb = blk * per, te = (tb + per < ntiles) ? tb + per : ntiles;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (rg < 2) {
    constexpr int U = 8;
    for (int t = tb + rg; t < te; t += 2 * U) {
      f32x4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int tt = t + 2 * u;
        x[u] = *(const f32x4*)&part[(size_t)(tt < te ? tt : te - 1) * TP_STRIDE + 4 * c4];
        if (tt >= te) x[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += x[u];
    }
  }
  if (rg == 1) *(f32x4*)&red[4 * c4] = acc;
  __syncthreads();
  if (rg == 0) *(f32x4*)&out1[(size_t)blk * TP_WCOLS + 4 * c4] = acc + *(const f32x4*)&red[4 * c4];
}
__global__ __launch_bounds__(320) void tile_part_sum1_kernel(int ntiles, int per, const float* __restrict__ part,
                                                             float* __restrict__ out1) {
  __shared__ __attribute__((aligned(16))) float red[TP_WCOLS];
  tile_part_sum1_body(blockIdx.x, ntiles, per, part, out1, red);
}
// both first stages in ONE launch (round 6): blocks [0, nsu) sum the weight part of their split, the rest the per-ray rows -- the
// two read the same table and do not depend on each other
__global__ __launch_bounds__(320) void tile_part_stage1_kernel(int nsu, int nrb, int ntiles, int per, int R, int S,
                                                               const float* __restrict__ part, float* __restrict__ out1,
                                                               float* __restrict__ rs_g1, float* __restrict__ rs_r1) {
  __shared__ __attribute__((aligned(16))) float red[TP_WCOLS];
  if ((int)blockIdx.x < nsu) {
    tile_part_sum1_body(blockIdx.x, ntiles, per, part, out1, red);
    return;
  }
  const int b = blockIdx.x - nsu, which = b >= nrb ? 1 : 0;
  if (threadIdx.x < 256) tile_part_rays_body((b - which * nrb) * 256 + threadIdx.x, which, R, S, part, rs_g1, rs_r1);
}
// column c of the weight part summed over the splits (four groups of splits per column, folded in a fixed order)
__global__ __launch_bounds__(256) void tile_part_sum2_kernel(int ns, const float* __restrict__ out1, float* __restrict__ d_wcsig,
                                                             float* __restrict__ d_bcsig, float* __restrict__ d_wr2,
                                                             float* __restrict__ d_br2) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x, cl = tid & 63, sg = tid >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int per = (ns + 3) / 4, sb = sg * per, se = (sb + per < ns) ? sb + per : ns;
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < TP_WCOLS) {
    int sp = sb;
    for (; sp + 8 <= se; sp += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] += out1[(size_t)(sp + u) * TP_WCOLS + c];
    }
    for (; sp < se; ++sp) p[0] += out1[(size_t)sp * TP_WCOLS + c];
  }
  red[sg][cl] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  __syncthreads();
  if (sg == 0 && c < TP_WCOLS) {
    const float s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    if (c < 128) { if (d_wcsig) d_wcsig[c] = s; }
    else if (c < 512) { if (d_wr2) d_wr2[c - 128] = s; }
    else if (c == 512) { if (d_bcsig) d_bcsig[0] = s; }
    else if (c < 516) { if (d_br2) d_br2[c - 513] = s; }
  }
}
}  // namespace

extern "C" int upnerf_tile_part_finish(int R, int S, const float* tile_part, float* rs_g1, float* rs_r1, float* d_wcsig,
                                       float* d_bcsig, float* d_wr2, float* d_br2, float* scratch, void* stream) {
  if (R <= 0 || S < 32 || !tile_part) return UPNERF_EINVAL;
  const long long M = (long long)R * S;
  if (M > 0x7fffffffLL) return UPNERF_EINVAL;
  const int ntiles = (int)((M + 63) / 64);
  const hipStream_t st = (hipStream_t)stream;
  const bool rays = rs_g1 || rs_r1, wsum = d_wcsig || d_bcsig || d_wr2 || d_br2;
  if (wsum && !scratch) return UPNERF_EINVAL;
  const int ns = ntiles < TP_NS ? ntiles : TP_NS, per = (ntiles + ns - 1) / ns;
  const int nsu = (ntiles + per - 1) / per;  // splits that hold at least one tile
  const int nrb = (R * 32 + 255) / 256;
  if (rays && wsum)
    hipLaunchKernelGGL(tile_part_stage1_kernel, dim3(nsu + 2 * nrb), dim3(320), 0, st, nsu, nrb, ntiles, per, R, S, tile_part, scratch,
                       rs_g1, rs_r1);
  else if (rays)
    hipLaunchKernelGGL(tile_part_rays_kernel, dim3(nrb, 2), dim3(256), 0, st, R, S, tile_part, rs_g1, rs_r1);
  if (wsum) {
    if (!rays) hipLaunchKernelGGL(tile_part_sum1_kernel, dim3(nsu), dim3(320), 0, st, ntiles, per, tile_part, scratch);
    hipLaunchKernelGGL(tile_part_sum2_kernel, dim3((TP_WCOLS + 63) / 64), dim3(256), 0, st, nsu, scratch, d_wcsig, d_bcsig, d_wr2,
                       d_br2);
  }
  return (int)hipGetLastError();
}

extern "C" int upnerf_linear(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* bias,
                             float* C, int ldc, int act, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (lda & 3) || (!(act & 2) && (ldb & 3)) || !A || !B || !C || (act & ~3))
    return UPNERF_EINVAL;
  dim3 grid((M + LIN_T - 1) / LIN_T, (N + LIN_T - 1) / LIN_T);
  if ((act & 2) && (((ldb & 3) != 0) || ((N & 3) != 0)))
    hipLaunchKernelGGL(linear_kernel<true>, grid, dim3(NTHREADS), 0, (hipStream_t)stream, M, N, K, A, lda, B, ldb, bias, C, ldc, act);
  else
    hipLaunchKernelGGL(linear_kernel<false>, grid, dim3(NTHREADS), 0, (hipStream_t)stream, M, N, K, A, lda, B, ldb, bias, C, ldc, act);
  return (int)hipGetLastError();
}

extern "C" int upnerf_frag_copy(const float* src, float* dst, const upnerf_frag_desc* descs, int ndesc, void* stream) {
  if (!src || !dst || !descs || ndesc <= 0 || ndesc > UPNERF_MAX_FRAG_DESC) return UPNERF_EINVAL;
  FragDescs D;
  D.n = ndesc;
  D.start[0] = 0;
  for (int j = 0; j < ndesc; ++j) {
    const upnerf_frag_desc& q = descs[j];
    if (q.rows <= 0 || q.cols <= 0 || (q.rows & 31) || (q.dst_kp & 7) || q.dst_k0 + q.cols > q.dst_kp) return UPNERF_EINVAL;
    D.d[j] = q;
    D.start[j + 1] = D.start[j] + q.rows * q.cols;
  }
  const int total = D.start[ndesc];
  hipLaunchKernelGGL(frag_copy_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, D);
  return (int)hipGetLastError();
}

extern "C" int upnerf_pack(float* P, const upnerf_pack_desc* descs, int ndesc, int unpack, void* stream) {
  if (!P || !descs || ndesc <= 0 || ndesc > UPNERF_MAX_PACK_DESC) return UPNERF_EINVAL;
  PackDescs D;
  D.n = ndesc;
  D.start[0] = 0;
  for (int j = 0; j < ndesc; ++j) {
    const upnerf_pack_desc& q = descs[j];
    if (!q.ptr || q.rows <= 0 || q.cols <= 0 || q.src_ld < q.cols || q.dst_ld < q.cols || q.dst_off < 0) return UPNERF_EINVAL;
    D.d[j] = q;
    D.start[j + 1] = D.start[j] + q.rows * q.cols;
  }
  const int total = D.start[ndesc];
  if (unpack)
    hipLaunchKernelGGL(unpack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)P, D);
  else
    hipLaunchKernelGGL(pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, D);
  return (int)hipGetLastError();
}

// n floats of zeros, 16 bytes per thread and trip (the step's zero arena: zero_pool.py)
__global__ void zero_fill_kernel(float* __restrict__ p, long long n) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) ((f32x4*)p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) p[(n4 << 2) + threadIdx.x] = 0.f;
}
extern "C" int upnerf_zero(float* p, long long n, void* stream) {
  if (!p || n <= 0 || ((uintptr_t)p & 15)) return UPNERF_EINVAL;
  const long long want = ((n >> 2) + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n);
  return (int)hipGetLastError();
}

extern "C" int upnerf_add_pairs(const upnerf_add_pair* pairs, int npairs, void* stream) {
  if (!pairs || npairs <= 0 || npairs > UPNERF_MAX_ADD_PAIRS) return UPNERF_EINVAL;
  AddPairs D;
  D.n = npairs;
  D.start[0] = 0;
  for (int j = 0; j < npairs; ++j) {
    if (!pairs[j].a || !pairs[j].b || !pairs[j].out || pairs[j].n <= 0) return UPNERF_EINVAL;
    D.d[j] = pairs[j];
    D.start[j + 1] = D.start[j] + pairs[j].n;
  }
  hipLaunchKernelGGL(add_pairs_kernel, dim3((D.start[npairs] + 255) / 256), dim3(256), 0, (hipStream_t)stream, D);
  return (int)hipGetLastError();
}

static int frag16_build(const upnerf_frag16_desc* descs, int n, Frag16Descs* D) {
  if (!descs || n <= 0 || n > UPNERF_MAX_FRAG_DESC) return UPNERF_EINVAL;
  D->n = n;
  D->start[0] = 0;
  for (int j = 0; j < n; ++j) {
    const upnerf_frag16_desc& q = descs[j];
    if (q.rows <= 0 || q.cols <= 0 || (q.rows & 31) || (q.dst_kp & 15) || q.dst_k0 + q.cols > q.dst_kp || q.exp_id < 0 ||
        q.exp_id >= 16)
      return UPNERF_EINVAL;
    D->d[j] = q;
    D->start[j + 1] = D->start[j] + q.rows * q.cols;
  }
  return 0;
}

extern "C" int upnerf_frag16(const float* src, void* dst_fwd, void* dst_bwd, const upnerf_frag16_desc* fwd, int nfwd,
                             const upnerf_frag16_desc* bwd, int nbwd, float* amax_scratch, int32_t* wexp, int perm_fwd,
                             int perm_bwd, float* wnorm, void* stream) {
  if (!src || !dst_fwd || !dst_bwd || !amax_scratch || !wexp) return UPNERF_EINVAL;
  Frag16Descs F, Bd;
  int rc = frag16_build(fwd, nfwd, &F);
  if (rc) return rc;
  rc = frag16_build(bwd, nbwd, &Bd);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(zero_floats_kernel, dim3(1), dim3(64), 0, st, amax_scratch, 16);
  // maxima over the forward matrices and over the transposed ones (ids that exist only there, e.g. the fused head)
  const int tmax = F.start[nfwd] > Bd.start[nbwd] ? F.start[nfwd] : Bd.start[nbwd];
  hipLaunchKernelGGL(frag16_amax_kernel, dim3((tmax + 256 * AMAX_PER - 1) / (256 * AMAX_PER), 2), dim3(256), 0, st, src, F, Bd,
                     amax_scratch);
  hipLaunchKernelGGL(frag16_write_kernel, dim3((tmax + 255) / 256, 2), dim3(256), 0, st, src, (char*)dst_fwd, (char*)dst_bwd, F, Bd,
                     amax_scratch, wexp, perm_fwd, perm_bwd);
  if (wnorm) {  // [64]: forward descriptors at 0.., transposed ones at 32..
    hipLaunchKernelGGL(zero_floats_kernel, dim3(1), dim3(64), 0, st, wnorm, 64);
    int rf = 0, rb = 0;
    for (int j = 0; j < nfwd; ++j) rf += fwd[j].rows;
    for (int j = 0; j < nbwd; ++j) rb += bwd[j].rows;
    const int rmax = rf > rb ? rf : rb;
    hipLaunchKernelGGL(frag16_rownorm_kernel, dim3((rmax + 4 * RN_ROWS - 1) / (4 * RN_ROWS), 2), dim3(256), 0, st, src, F, Bd, wnorm);
  }
  return (int)hipGetLastError();
}

// The same update with the gradients read where autograd left them: descriptor j covers flat elements [off, off + n) of p / m / v
// and the n floats at g (no gather of ~70 gradient tensors into a flat buffer first: two multi-tensor copy launches per step).
struct AdamDescs {
  upnerf_adam_desc d[UPNERF_MAX_ADAM_DESC];
  int bstart[UPNERF_MAX_ADAM_DESC + 1];  // first 1024-element block of every descriptor
  int n;
};
__global__ __launch_bounds__(256) void adam_gather_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                          AdamDescs D, float b1, float b2, float eps, float step_size, float bc2_sqrt,
                                                          const float* __restrict__ dyn2) {
  int j = 0;
  while ((int)blockIdx.x >= D.bstart[j + 1]) ++j;  // uniform per workgroup
  const upnerf_adam_desc q = D.d[j];
  const int e0 = ((int)blockIdx.x - D.bstart[j]) * 1024 + threadIdx.x;
  if (dyn2) {
    step_size = dyn2[0];
    bc2_sqrt = dyn2[1];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = e0 + 256 * u;
    if (e < q.n) {
      const size_t i = (size_t)q.off + e;
      adam_update(p[i], m[i], v[i], q.g[e], b1, b2, eps, step_size, bc2_sqrt);
    }
  }
}

extern "C" int upnerf_adam_gather(float* p, float* m, float* v, const upnerf_adam_desc* descs, int ndesc, float beta1, float beta2,
                                  float eps, float step_size, float bc2_sqrt, const float* dyn2, void* stream) {
  if (!p || !m || !v || !descs || ndesc <= 0 || ndesc > UPNERF_MAX_ADAM_DESC) return UPNERF_EINVAL;
  AdamDescs D;
  D.n = ndesc;
  D.bstart[0] = 0;
  for (int j = 0; j < ndesc; ++j) {
    if (!descs[j].g || descs[j].n <= 0 || descs[j].off < 0) return UPNERF_EINVAL;
    D.d[j] = descs[j];
    D.bstart[j + 1] = D.bstart[j] + (descs[j].n + 1023) / 1024;
  }
  hipLaunchKernelGGL(adam_gather_kernel, dim3(D.bstart[ndesc]), dim3(256), 0, (hipStream_t)stream, p, m, v, D, beta1, beta2, eps,
                     step_size, bc2_sqrt, dyn2);
  return (int)hipGetLastError();
}

extern "C" int upnerf_adam(int64_t n, float* p, const float* g, float* m, float* v, float beta1, float beta2, float eps,
                           float step_size, float bc2_sqrt, const float* dyn2, void* stream) {
  if (n <= 0 || !p || !g || !m || !v) return UPNERF_EINVAL;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n, p,
                     g, m, v, beta1, beta2, eps, step_size, bc2_sqrt, dyn2);
  return (int)hipGetLastError();
}

extern "C" int upnerf_scale_exponents(const float* maxima, int n, int32_t* out, void* stream) {
  if (!maxima || !out || n <= 0 || n > 64) return UPNERF_EINVAL;
  hipLaunchKernelGGL(scale_exponents_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, maxima, n, out);
  return (int)hipGetLastError();
}

extern "C" int upnerf_set_scalars(float* dst, int n, const float* vals, void* stream) {
  if (!dst || !vals || n <= 0 || n > UPNERF_MAX_SCALARS) return UPNERF_EINVAL;
  ScalarPack s;
  for (int i = 0; i < UPNERF_MAX_SCALARS; ++i) s.v[i] = i < n ? vals[i] : 0.0f;
  hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(UPNERF_MAX_SCALARS < 64 ? 64 : 128), 0, (hipStream_t)stream, dst, n, s);
  return (int)hipGetLastError();
}
