// TransientNet (models/transient_net.py:5-38) as ONE forward and ONE backward (data-gradient) kernel.
//
// One row per RAY (R = 4096): as nine upnerf_linear launches forward and nine backward, plus a dozen elementwise launches,
// the network costs ~0.5 ms of a 17.7 ms step although it holds 0.1 % of the step's FLOPs -- every launch is a chain of a few
// L2 round trips.  Here a workgroup of four waves owns 16 rows for the whole network: the activations of those rows stay in
// LDS ([16][392] fp32, two buffers), the weights stream L2 -> registers as 16-byte pieces straight out of the nn.Linear
// layout ([out][in] row-major, no repacking), products on v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulate: the
// arithmetic of the launches it replaces).
//
// Operand mapping (the MFMA wants A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16]; which k of the contraction a
// slot carries is free as long as A and B agree):
//   forward  y[m][n] = sum_k x[m][k] W[n][k]:  lane (c = lane % 16, q = lane / 16) reads x[c][16 g + 4 q ..+3] from LDS and, for
//            each of its NT output columns n = base + NT c + t, W[n][16 g + 4 q ..+3]: element j of both feeds MFMA step j.
//   backward gx[m][k] = sum_n gy[m][n] W[n][k]: the lane reads gy[c][16 g + 4 q ..+3] from LDS and the four rows
//            W[16 g + 4 q + j][base + 4 c ..+3]; element x of row j feeds the accumulator of output column base + 4 c + x.
// Either way a lane ends up with consecutive output columns of rows 4 q .. 4 q + 3: 16-byte stores, whole lines per row.
// The weight gradients are left to upnerf_wgrad_grouped on the stored activations / pre-activation gradients.
#include "common.cuh"

namespace {

#define TR_ROWS 16
#define TR_LD 392  // LDS row stride in floats: 384 + 8 (rows start 32 bytes apart modulo the 256-byte bank window)
#define TR_F 384   // feat_dim
#define TR_H 256   // hidden width
#define TR_T 128   // transient embedding / t_encoder width

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Warm this XCD's L2 with the weights: one dword per 128-byte line, the lines of a tensor dealt over the threads of the XCD's
// workgroups (consecutive workgroup ids go round the eight XCDs, so workgroup b is number b / 8 on its XCD), all requests in
// flight at once.  Why (round 6): the weights are cold in every XCD's L2 when the kernel starts (Adam rewrote them), all 32
// workgroups of an XCD walk the same lines in lockstep, and every K-loop group then waits out a full miss whatever the depth of
// the register ring -- `wait_any` 0.59 of the wave cycles at `mfma_busy` 0.30, unchanged by a three-deep ring.  The values are
// summed into a number nobody reads (kept alive by an empty asm at the end of the kernel).  A different workgroup -> XCD mapping
// only lowers the coverage: nothing depends on it.
#ifndef TR_WARM
#define TR_WARM 1
#endif
__device__ __forceinline__ float warm_lines(const float* __restrict__ w, int nfloats, int lane_id, int nlanes) {
  if (!TR_WARM) return 0.f;
  const int lines = nfloats >> 5;
  float s = w[(size_t)(lane_id < lines ? lane_id : lines - 1) << 5];  // unconditional: no branch, no wait (see the kernels' last line)
  for (int l = lane_id + nlanes; l < lines; l += nlanes) s += w[(size_t)l << 5];  // (launches of fewer than 96 workgroups only)
  return s;
}
struct WarmLanes {
  int lane_id, nlanes;
};
__device__ __forceinline__ WarmLanes warm_lanes(int tid) {
  const int per_xcd = ((int)gridDim.x + 7) >> 3, slices = per_xcd < 32 ? per_xcd : 32;
  return WarmLanes{(((int)blockIdx.x >> 3) % slices) * NTHREADS + tid, slices * NTHREADS};
}
// the warmed values are never used: this only keeps their loads (and holds their registers until the loads have long landed)
#define TR_WARM_KEEP(k) asm volatile("" ::"v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(k[4]), "v"(k[5]))

// y = act(x W^T + b) for the 16 rows of the workgroup.  Xs: LDS [16][TR_LD] (K columns used); W [N][ldw]; the wave owns N / 4
// consecutive output columns starting at wave * N / 4, the lane NT = N / 64 consecutive ones of them.  Results go to Ys (LDS,
// column offset ycol0) and, when gout is given, to gout[row][col] (row stride ldg) for rows < R.
template <int N, int K, bool RELU>
__device__ __forceinline__ void layer_fwd(const float* Xs, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                          float* Ys, float* __restrict__ gout, int ldg, int m0, int R, int wave, int lane) {
  constexpr int NT = N / 64;
  static_assert(NT == 4 || NT == 2, "output widths 256 and 128");
  const int c = lane & 15, q = lane >> 4;
  const int col0 = wave * (N / 4) + NT * c;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = W + (size_t)(col0 + t) * ldw + 4 * q;
  const float* xp = Xs + c * TR_LD + 4 * q;
  float bv[NT];  // (requested in front of the K loop, used behind it)
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = bias[col0 + t];
  // Weight fragments through a four-slot register ring, three groups requested ahead, no branch around a request (round 6: the
  // loop had ONE group ahead behind `if (g + 2 < K / 16)`, and a counted wait cannot span the join -- hipcc waited vmcnt(0) in
  // front of every second group; ~1400 cycles per group of 16 MFMAs that take 512).  Past the end the request is clamped to the
  // last group (valid memory, value unused).  Same accumulation order as before: bitwise-identical results.
  constexpr int G = K / 16;
  static_assert(G % 4 == 0, "K is a multiple of 64");
  f32x4 b[4][NT];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) b[s][t] = *(const f32x4*)(wp[t] + 16 * s);
#pragma unroll 1
  for (int g = 0; g < G; g += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gn = g + u + 3 < G ? g + u + 3 : G - 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#ifdef TR_EXP_NOLOAD
        b[(u + 3) & 3][t] = f32x4{1.f, 2.f, 3.f, (float)gn};
#else
        b[(u + 3) & 3][t] = *(const f32x4*)(wp[t] + 16 * gn);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 a = *(const f32x4*)(xp + 16 * (g + u));
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#ifdef TR_EXP_NOMMA
          acc[t][j] += a[j] * b[u][t][j];
#else
          acc[t] = mfma4(a[j], b[u][t][j], acc[t]);
#endif
        }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r;
    float v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      v[t] = acc[t][r] + bv[t];
      if (RELU) v[t] = v[t] > 0.f ? v[t] : 0.f;
    }
    if constexpr (NT == 4) {
      const f32x4 o = {v[0], v[1], v[2], v[3]};
      *(f32x4*)&Ys[row * TR_LD + col0] = o;
#ifndef TR_EXP_NOSTORE
      if (gout && m0 + row < R) *(f32x4*)&gout[(size_t)(m0 + row) * ldg + col0] = o;
#endif
    } else {
      const f32x2 o = {v[0], v[1]};
      *(f32x2*)&Ys[row * TR_LD + col0] = o;
#ifndef TR_EXP_NOSTORE
      if (gout && m0 + row < R) *(f32x2*)&gout[(size_t)(m0 + row) * ldg + col0] = o;
#endif
    }
  }
}

// gx[m][k] = sum_n gy[m][n] W[n][k] for the 64-column block kb of the K output columns.  Gs: LDS [16][TR_LD] (N columns used).
// Returns this lane's columns 64 kb + 4 c .. + 3 of rows 4 q + r in acc[x][r].
template <int N>
__device__ __forceinline__ void block_bwd(f32x4 (&acc)[4], const float* Gs, const float* __restrict__ W, int ldw, int kb, int lane) {
  const int c = lane & 15, q = lane >> 4;
#pragma unroll
  for (int x = 0; x < 4; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wp = W + (size_t)(4 * q) * ldw + 64 * kb + 4 * c;
  const float* gp = Gs + c * TR_LD + 4 * q;
  // (four-slot register ring, three groups ahead, branch-free: see layer_fwd)
  constexpr int G = N / 16;
  static_assert(G % 4 == 0, "N is a multiple of 64");
  f32x4 b[4][4];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) b[s][j] = *(const f32x4*)(wp + (size_t)(16 * s + j) * ldw);
#pragma unroll 1
  for (int g = 0; g < G; g += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gn = g + u + 3 < G ? g + u + 3 : G - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) b[(u + 3) & 3][j] = *(const f32x4*)(wp + (size_t)(16 * gn + j) * ldw);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 a = *(const f32x4*)(gp + 16 * (g + u));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[x] = mfma4(a[j], b[u][j][x], acc[x]);
    }
  }
}

// rows of a global [R][ncols] tensor -> LDS tile columns [col0, col0 + ncols), zero rows past R
__device__ __forceinline__ void tile_load(float* Xs, int col0, const float* __restrict__ src, int ncols, int m0, int R, int tid) {
  const int g4 = ncols >> 2;
  for (int idx = tid; idx < TR_ROWS * g4; idx += NTHREADS) {
    const int row = idx / g4, g = idx - row * g4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m0 + row < R) v = *(const f32x4*)&src[(size_t)(m0 + row) * ncols + 4 * g];
    *(f32x4*)&Xs[row * TR_LD + col0 + 4 * g] = v;
  }
}

// dot products of the workgroup's 16 rows (LDS tile, K columns) with NV weight vectors: 16 lanes per row, each K / 16 columns
template <int K, int NV>
__device__ __forceinline__ void row_dots(float (&out)[NV], const float* Xs, const float* __restrict__ w, int tid) {
  const int row = tid >> 4, l16 = tid & 15;
#pragma unroll
  for (int v = 0; v < NV; ++v) out[v] = 0.f;
  for (int k = 4 * l16; k < K; k += 64) {
    const f32x4 x = *(const f32x4*)&Xs[row * TR_LD + k];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const f32x4 ww = *(const f32x4*)&w[v * K + k];
      out[v] += x[0] * ww[0] + x[1] * ww[1] + x[2] * ww[2] + x[3] * ww[3];
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) out[v] += __shfl_xor(out[v], d);
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
// torch.nn.functional.softplus (beta 1, threshold 20)
__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(NTHREADS) void transient_fwd_kernel(upnerf_transient_args a) {
  __shared__ __attribute__((aligned(16))) float bufA[TR_ROWS * TR_LD];
  __shared__ __attribute__((aligned(16))) float bufB[TR_ROWS * TR_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * TR_ROWS, R = a.R;
  const size_t RH = (size_t)R * TR_H;
  const WarmLanes wl = warm_lanes(tid);
  const float wk[6] = {warm_lines(a.w0, TR_H * TR_F, wl.lane_id, wl.nlanes), warm_lines(a.w1, TR_H * TR_H, wl.lane_id, wl.nlanes),
                       warm_lines(a.w2, TR_H * TR_H, wl.lane_id, wl.nlanes), warm_lines(a.w3, TR_H * TR_H, wl.lane_id, wl.nlanes),
                       warm_lines(a.wf, TR_H * TR_H, wl.lane_id, wl.nlanes), warm_lines(a.wt, TR_T * TR_F, wl.lane_id, wl.nlanes)};
  tile_load(bufA, 0, a.feat, TR_F, m0, R, tid);
  __syncthreads();
  layer_fwd<TR_H, TR_F, true>(bufA, a.w0, TR_F, a.b0, bufB, a.h, TR_H, m0, R, wave, lane);
  __syncthreads();
  layer_fwd<TR_H, TR_H, true>(bufB, a.w1, TR_H, a.b1, bufA, a.h + RH, TR_H, m0, R, wave, lane);
  __syncthreads();
  layer_fwd<TR_H, TR_H, true>(bufA, a.w2, TR_H, a.b2, bufB, a.h + 2 * RH, TR_H, m0, R, wave, lane);
  __syncthreads();
  layer_fwd<TR_H, TR_H, true>(bufB, a.w3, TR_H, a.b3, bufA, a.h + 3 * RH, TR_H, m0, R, wave, lane);  // h4 in bufA
  __syncthreads();
  layer_fwd<TR_H, TR_H, false>(bufA, a.wf, TR_H, a.bf, bufB, a.e, TR_H, m0, R, wave, lane);  // final_encoding in bufB[:, :256]
  tile_load(bufB, TR_H, a.t_emb, TR_T, m0, R, tid);                                          // | t_emb
  float da[1];
  row_dots<TR_H, 1>(da, bufA, a.wa, tid);  // alpha head on h4
  const int row = tid >> 4;
  const float alpha = sigmoidf(da[0] + a.ba[0]);
  if ((tid & 15) == 0 && m0 + row < R) a.alpha[m0 + row] = alpha;
  __syncthreads();
  layer_fwd<TR_T, TR_F, true>(bufB, a.wt, TR_F, a.bt, bufA, a.t, TR_T, m0, R, wave, lane);  // t in bufA[:, :128]
  __syncthreads();
  float dr[3], db[1];
  row_dots<TR_T, 3>(dr, bufA, a.wr, tid);
  row_dots<TR_T, 1>(db, bufA, a.wb, tid);
  if ((tid & 15) == 0 && m0 + row < R) {
    const int m = m0 + row;
    const float sp = db[0] + a.bb[0];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) a.rgb[(size_t)m * 3 + cc] = sigmoidf(dr[cc] + a.br[cc]);
    a.spre[m] = sp;
    a.beta[m] = softplusf(sp) * alpha + a.beta_min;
  }
  TR_WARM_KEEP(wk);
}

// The stored activations that mask one output block of a backward layer (this lane's four rows), requested BEFORE the block's K
// loop and branch-free (rows past R read the last row; their mask is forced to zero in emit_bwd): inside emit_bwd, as
// `if (in) h = act[...]` per row, each of the four loads was a branch with `s_waitcnt vmcnt(0)` behind it -- four memory round
// trips in a row per layer (round 6, seen in the ISA).
struct ActRows {
  f32x4 h[4];
  f32x4 wv;
};
__device__ __forceinline__ ActRows act_rows(const float* __restrict__ act, int ldg, const float* __restrict__ r1w, int kb, int m0,
                                            int R, int lane) {
  const int c = lane & 15, q = lane >> 4, col = 64 * kb + 4 * c;
  ActRows A;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 4 * q + r < R ? m0 + 4 * q + r : R - 1;
    A.h[r] = *(const f32x4*)&act[(size_t)m * ldg + col];
  }
  A.wv = r1w ? *(const f32x4*)&r1w[col] : f32x4{0.f, 0.f, 0.f, 0.f};
  return A;
}

// One output block of a backward layer: += extra (a rank-1 term), masked by the stored activation, to LDS and to global.
template <bool MASK>
__device__ __forceinline__ void emit_bwd(const f32x4 (&acc)[4], int kb, float* Ys, float* __restrict__ gout, int ldg,
                                         const ActRows& A, const float* r1s, int m0, int R, int lane) {
  const int c = lane & 15, q = lane >> 4, col = 64 * kb + 4 * c;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r;
    f32x4 o = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    if (r1s) o += A.wv * r1s[row];
    const bool in = m0 + row < R;
    if constexpr (MASK) {
#pragma unroll
      for (int x = 0; x < 4; ++x) o[x] = (in && A.h[r][x] > 0.f) ? o[x] : 0.f;
    }
    if (Ys) *(f32x4*)&Ys[row * TR_LD + col] = o;
    if (gout && in) *(f32x4*)&gout[(size_t)(m0 + row) * ldg + col] = o;
  }
}

__global__ __launch_bounds__(NTHREADS) void transient_bwd_kernel(upnerf_transient_args a, upnerf_transient_grads g) {
  __shared__ __attribute__((aligned(16))) float bufA[TR_ROWS * TR_LD];
  __shared__ __attribute__((aligned(16))) float bufB[TR_ROWS * TR_LD];
  __shared__ float dz_s[TR_ROWS][8];  // dz_alpha, dz_beta, dz_rgb[3]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * TR_ROWS, R = a.R;
  const size_t RH = (size_t)R * TR_H;
  const WarmLanes wl = warm_lanes(tid);  // (the order the layers are walked in: t-encoder, final, 3, 2, 1, 0)
  const float wk[6] = {warm_lines(a.wt, TR_T * TR_F, wl.lane_id, wl.nlanes), warm_lines(a.wf, TR_H * TR_H, wl.lane_id, wl.nlanes),
                       warm_lines(a.w3, TR_H * TR_H, wl.lane_id, wl.nlanes), warm_lines(a.w2, TR_H * TR_H, wl.lane_id, wl.nlanes),
                       warm_lines(a.w1, TR_H * TR_H, wl.lane_id, wl.nlanes), warm_lines(a.w0, TR_H * TR_F, wl.lane_id, wl.nlanes)};
  // head pre-activation gradients (transient_net.py:33-37): beta = softplus(s) alpha + beta_min
  if (tid < TR_ROWS) {
    const int m = m0 + tid;
    float dz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < R) {
      const float al = a.alpha[m], s = a.spre[m], dbeta = g.d_beta ? g.d_beta[m] : 0.f;
      const float dal = (g.d_alpha ? g.d_alpha[m] : 0.f) + dbeta * softplusf(s);
      dz[0] = dal * al * (1.0f - al);
      dz[1] = dbeta * al * sigmoidf(s);
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const float y = a.rgb[(size_t)m * 3 + cc];
        dz[2 + cc] = (g.d_rgb ? g.d_rgb[(size_t)m * 3 + cc] : 0.f) * y * (1.0f - y);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) g.dz_heads[(size_t)m * 8 + j] = dz[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) dz_s[tid][j] = dz[j];
  }
  __syncthreads();
  // gz_t = (dz_rgb . W_rgb + dz_beta w_beta) * (t > 0)  -> bufA[:, :128]
  for (int idx = tid; idx < TR_ROWS * (TR_T / 4); idx += NTHREADS) {
    const int row = idx / (TR_T / 4), c4 = idx - row * (TR_T / 4);
    const bool in = m0 + row < R;
    f32x4 o = *(const f32x4*)&a.wb[4 * c4] * dz_s[row][1];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) o += *(const f32x4*)&a.wr[cc * TR_T + 4 * c4] * dz_s[row][2 + cc];
    const f32x4 t = *(const f32x4*)&a.t[(size_t)(in ? m0 + row : R - 1) * TR_T + 4 * c4];
#pragma unroll
    for (int x = 0; x < 4; ++x) o[x] = (in && t[x] > 0.f) ? o[x] : 0.f;
    *(f32x4*)&bufA[row * TR_LD + 4 * c4] = o;
    if (in) *(f32x4*)&g.gz_t[(size_t)(m0 + row) * TR_T + 4 * c4] = o;
  }
  __syncthreads();
  f32x4 acc[4];
  // [gz_e | g_temb] = gz_t . W_t  (384 outputs: six 64-column blocks over four waves)
  for (int kb = wave; kb < TR_F / 64; kb += 4) {
    block_bwd<TR_T>(acc, bufA, a.wt, TR_F, kb, lane);
    if (kb < TR_H / 64) {
      emit_bwd<false>(acc, kb, bufB, g.gz_e, TR_H, ActRows{}, nullptr, m0, R, lane);
    } else {  // the embedding's gradient rows: global only
      const int c = lane & 15, q = lane >> 4, col = 64 * kb + 4 * c - TR_H;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;
        if (g.g_temb && m0 + row < R) *(f32x4*)&g.g_temb[(size_t)(m0 + row) * TR_T + col] = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      }
    }
  }
  __syncthreads();
  // gz_4 = (gz_e . W_final + dz_alpha w_alpha) * (h4 > 0)
  __shared__ float dza_s[TR_ROWS];
  if (tid < TR_ROWS) dza_s[tid] = dz_s[tid][0];
  __syncthreads();
  ActRows A = act_rows(a.h + 3 * RH, TR_H, a.wa, wave, m0, R, lane);
  block_bwd<TR_H>(acc, bufB, a.wf, TR_H, wave, lane);
  emit_bwd<true>(acc, wave, bufA, g.gz_h + 3 * RH, TR_H, A, dza_s, m0, R, lane);
  __syncthreads();
  A = act_rows(a.h + 2 * RH, TR_H, nullptr, wave, m0, R, lane);
  block_bwd<TR_H>(acc, bufA, a.w3, TR_H, wave, lane);
  emit_bwd<true>(acc, wave, bufB, g.gz_h + 2 * RH, TR_H, A, nullptr, m0, R, lane);
  __syncthreads();
  A = act_rows(a.h + RH, TR_H, nullptr, wave, m0, R, lane);
  block_bwd<TR_H>(acc, bufB, a.w2, TR_H, wave, lane);
  emit_bwd<true>(acc, wave, bufA, g.gz_h + RH, TR_H, A, nullptr, m0, R, lane);
  __syncthreads();
  A = act_rows(a.h, TR_H, nullptr, wave, m0, R, lane);
  block_bwd<TR_H>(acc, bufA, a.w1, TR_H, wave, lane);
  emit_bwd<true>(acc, wave, bufB, g.gz_h, TR_H, A, nullptr, m0, R, lane);
  __syncthreads();
  if (g.g_feat) {
    for (int kb = wave; kb < TR_F / 64; kb += 4) {
      block_bwd<TR_H>(acc, bufB, a.w0, TR_F, kb, lane);
      emit_bwd<false>(acc, kb, nullptr, g.g_feat, TR_F, ActRows{}, nullptr, m0, R, lane);
    }
  }
  TR_WARM_KEEP(wk);
}

}  // namespace

static int check_args(const upnerf_transient_args* a) {
  if (!a || a->R <= 0 || !a->feat || !a->t_emb || !a->w0 || !a->b0 || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->w3 || !a->b3 ||
      !a->wf || !a->bf || !a->wt || !a->bt || !a->wa || !a->ba || !a->wb || !a->bb || !a->wr || !a->br || !a->h || !a->e || !a->t ||
      !a->alpha || !a->rgb || !a->beta || !a->spre)
    return UPNERF_EINVAL;
  return 0;
}

extern "C" int upnerf_transient_fwd(const upnerf_transient_args* a, void* stream) {
  int rc = check_args(a);
  if (rc) return rc;
  hipLaunchKernelGGL(transient_fwd_kernel, dim3((a->R + TR_ROWS - 1) / TR_ROWS), dim3(NTHREADS), 0, (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}

extern "C" int upnerf_transient_bwd(const upnerf_transient_args* a, const upnerf_transient_grads* g, void* stream) {
  int rc = check_args(a);
  if (rc) return rc;
  if (!g || !g->dz_heads || !g->gz_t || !g->gz_e || !g->gz_h) return UPNERF_EINVAL;
  hipLaunchKernelGGL(transient_bwd_kernel, dim3((a->R + TR_ROWS - 1) / TR_ROWS), dim3(NTHREADS), 0, (hipStream_t)stream, *a, *g);
  return (int)hipGetLastError();
}
