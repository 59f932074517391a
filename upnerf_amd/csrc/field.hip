// Fused NeRF field for gfx950: positional encoding -> D x W trunk (skip connection) -> density / feature /
// candidate / colour heads, forward and data-gradient backward, one 64-sample tile per workgroup (fp32 MFMA variant;
// csrc/field16.hip is the f16x3 variant used for 256-wide fields).
//
// Reference behaviour: models/nerf.py:80-124 (NeRF.forward) and 126-147 (positional_encoding), evaluated by
// models/rendering.py:102-122 on xyz = o + d*z.  What is different by design (not by arithmetic):
//   * the whole per-sample network runs inside one kernel; activations move layer to layer through LDS and
//     are written to HBM only as the tensors the backward pass needs;
//   * the 384-wide feature heads are NOT evaluated per sample: feat maps are composited in the W-wide space
//     and projected once per ray (upnerf_composite_fwd + host), and the colour head consumes a per-step folded
//     weight  W_rgb[:, :384] . W_feat  (exact algebra);
//   * per-ray inputs of the heads (candidate embedding row; [PE(dir) | appearance row]) are read as MFMA
//     A-fragments straight from the per-ray tables instead of being repeated per sample.
#include "common.cuh"

// Rows of samples per workgroup.  64 rows x 256 floats = 64 KiB of LDS, so TWO workgroups share a CU (two waves per
// SIMD): while one is in a layer epilogue (bias/ReLU, LDS write-back, activation store, barriers) the other keeps
// the matrix pipe busy.  With 128-row tiles (one workgroup per CU) those phases were exposed: 61 % / 40 % of the
// fp32 MFMA peak in the forward / backward kernels (profiles/r01_a_*).
#define FIELD_TILE UPNERF_TILE_ROWS

namespace {

// dot of LDS row segment [c0, c0+K) with w[0..K), split over the TPR adjacent threads that share a row
template <int TPR>
__device__ __forceinline__ float rowdot(const float* Hs, int ldw, int row, int part, int c0, int K, const float* __restrict__ w) {
  float s = 0.0f;
  const int kb = part * (K / TPR);
  for (int k = 0; k < K / TPR; k += 4) {
    const f32x4 a = *(const f32x4*)&Hs[swz4(row, c0 + kb + k, ldw)];
    const f32x4 ww = *(const f32x4*)&w[kb + k];
    s += a.x * ww.x + a.y * ww.y + a.z * ww.z + a.w * ww.w;
  }
#pragma unroll
  for (int d = 1; d < TPR; d <<= 1) s += __shfl_xor(s, d);
  return s;
}

template <int W, int TILE>
__global__ __launch_bounds__(NTHREADS, (TILE * W * 4 <= 32768) ? 4 : ((TILE * W * 4 <= 65536) ? 2 : 1)) void field_fwd_kernel(upnerf_layout L, upnerf_field_fwd_args a) {
  __shared__ __attribute__((aligned(16))) float Hs[TILE * W];
  constexpr int W2 = W / 2;
  constexpr int TPR = NTHREADS / TILE;  // threads per row in the per-row stages
  using TW = WaveTile<W, TILE>;
  using TH = WaveTile<W2, TILE>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int S = a.S, M = a.R * a.S, m0 = blockIdx.x * TILE;
  const float* __restrict__ P = a.P;
  const int n0 = TW::n0(wave), row0 = TW::row0(wave);
  const int hn0 = TH::n0(wave), hrow0 = TH::row0(wave);

  // ---- sample positions xyz = o + d*z (rendering.py:251 / 308), two roundings like the reference
  if (tid < TILE) {
    const int m = m0 + tid;
    float x = 0.f, y = 0.f, zc = 0.f;
    if (m < M) {
      const int r = m / S;
      const float zz = a.z[m];
      x = mul_then_add(a.rays_o[3 * r + 0], a.rays_d[3 * r + 0], zz);
      y = mul_then_add(a.rays_o[3 * r + 1], a.rays_d[3 * r + 1], zz);
      zc = mul_then_add(a.rays_o[3 * r + 2], a.rays_d[3 * r + 2], zz);
    }
    Hs[swz(tid, 0, W)] = x;
    Hs[swz(tid, 1, W)] = y;
    Hs[swz(tid, 2, W)] = zc;
    Hs[swz(tid, 63, W)] = 0.0f;
    // max|x0| = max(|xyz|, |sin/cos| <= 1)
    wave_track_max(fmaxf(fmaxf(fabsf(x), fabsf(y)), fmaxf(fabsf(zc), 1.0f)), a.amax ? a.amax + L.D + 4 : nullptr, lane);
  }
  __syncthreads();
  // ---- BARF-masked encoding (nerf.py:126-147): [x, sin(2^k pi x_n) w_k, cos(2^k pi x_n) w_k]
  // band-major, one item per thread and trip: a wave's band is uniform (TILE * 3 items per band = whole waves), a band whose
  // weight is exactly zero is written as zeros without the sincos (csrc/field16.hip has the reasoning)
  static_assert((TILE * 3) % 64 == 0, "a wave's items share one band");
#pragma unroll 1
  for (int i0 = 64 * __builtin_amdgcn_readfirstlane(tid >> 6); i0 < TILE * 3 * 10; i0 += NTHREADS) {
    const int k = i0 / (TILE * 3);
    const int it = i0 + lane - k * (TILE * 3), row = it / 3, n = it - row * 3;
    const float wk = a.wk_xyz_dev ? a.wk_xyz_dev[k] : a.wk_xyz[k];  // device copy: graph replay
    float sv = 0.0f, cv = 0.0f;
    if (__builtin_amdgcn_readfirstlane(__float_as_uint(wk)) != 0u) {
      sincos_f32_via_f64(Hs[swz(row, n, W)] * ldexpf(PI_F, k), sv, cv);
      sv *= wk;
      cv *= wk;
    }
    Hs[swz(row, 3 + 20 * n + k, W)] = sv;
    Hs[swz(row, 3 + 20 * n + 10 + k, W)] = cv;
  }
  __syncthreads();
  tile_store<TILE>(Hs, W, 0, UPNERF_X0, a.x0, UPNERF_X0, m0, M, tid);

  // ---- trunk (nerf.py:84-87)
  for (int l = 0; l < L.D; ++l) {
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    if (l == 0) {
      mma_lds(acc, Hs, W, row0, 0, P + L.w[0], UPNERF_X0, n0, 0, UPNERF_X0, lane);
    } else if (l == L.skip) {
      const float* ap[TW::MT];
#pragma unroll
      for (int mt = 0; mt < TW::MT; ++mt) {
        int m = m0 + row0 + 32 * mt + li;
        m = m < M ? m : M - 1;
        ap[mt] = a.x0 + (size_t)m * UPNERF_X0 + 4 * hh;
      }
      mma_glb(acc, ap, P + L.w[l], UPNERF_X0 + W, n0, 0, UPNERF_X0, lane);
      mma_lds(acc, Hs, W, row0, 0, P + L.w[l], UPNERF_X0 + W, n0, UPNERF_X0, W, lane);
    } else {
      mma_lds(acc, Hs, W, row0, 0, P + L.w[l], W, n0, 0, W, lane);
    }
    const unsigned long long bits = acc_bias_relu_pack(acc, P + L.b[l], n0, lane);
    acc_track_max(acc, a.amax ? a.amax + l : nullptr, lane);
    if (a.hmask) ((unsigned long long*)a.hmask)[((size_t)l * gridDim.x + blockIdx.x) * NTHREADS + tid] = bits;
    __syncthreads();
    acc_to_lds(acc, Hs, W, row0, n0, 0, lane);
    __syncthreads();
    if (a.h) tile_store<TILE>(Hs, W, 0, W, a.h + (size_t)l * M * W, W, m0, M, tid);
  }

  const int prow = tid / TPR, phalf = tid % TPR, pm = m0 + prow;
  // ---- shared density head (nerf.py:89): softplus(w . h + b)
  {
    const float pre = rowdot<TPR>(Hs, W, prow, phalf, 0, W, P + L.wsig) + P[L.bsig];
    if (phalf == 0 && pm < M) a.sigma_s[pm] = softplus_f(pre);
  }
  // density-only pass (nerf.py:90-91 `sigma_only`): nobody consumes e, so the pass ends here
  if (!a.e && !a.use_rgb && !a.use_cand) return;
  // ---- xyz_encoding_final (nerf.py:93), no activation
  {
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    mma_lds(acc, Hs, W, row0, 0, P + L.we, W, n0, 0, W, lane);
    const float* __restrict__ bias = P + L.be;
    acc_map(acc, row0, n0, lane, [&](float v, int, int col) { return v + bias[col]; });
    acc_track_max(acc, a.amax ? a.amax + L.D : nullptr, lane);
    __syncthreads();
    acc_to_lds(acc, Hs, W, row0, n0, 0, lane);
    __syncthreads();
    if (a.e) tile_store<TILE>(Hs, W, 0, W, a.e, W, m0, M, tid);
  }
  if (!a.use_rgb && !a.use_cand) return;

  // ---- first layer of the colour head (folded, nerf.py:95+102-109) and of the candidate head (nerf.py:97-98)
  f32x16 accr[TH::MT][TH::NT], accc[TH::MT][TH::NT];
  int rayrow[TH::MT];
#pragma unroll
  for (int mt = 0; mt < TH::MT; ++mt) {
    int m = m0 + hrow0 + 32 * mt + li;
    m = m < M ? m : M - 1;
    rayrow[mt] = m / S;
  }
  if (a.use_rgb) {
    acc_zero(accr);
    mma_lds(accr, Hs, W, hrow0, 0, P + L.wr1, W + UPNERF_AUXK, hn0, 0, W, lane);
    const float* ap[TH::MT];
#pragma unroll
    for (int mt = 0; mt < TH::MT; ++mt) ap[mt] = a.aux + (size_t)rayrow[mt] * UPNERF_AUXK + 4 * hh;
    mma_glb(accr, ap, P + L.wr1, W + UPNERF_AUXK, hn0, W, UPNERF_AUXK, lane);
  }
  if (a.use_cand) {
    acc_zero(accc);
    mma_lds(accc, Hs, W, hrow0, 0, P + L.wc1, W + UPNERF_CK, hn0, 0, W, lane);
    const float* ap[TH::MT];
#pragma unroll
    for (int mt = 0; mt < TH::MT; ++mt) ap[mt] = a.c_rows + (size_t)rayrow[mt] * UPNERF_CK + 4 * hh;
    mma_glb(accc, ap, P + L.wc1, W + UPNERF_CK, hn0, W, UPNERF_CK, lane);
  }
  __syncthreads();
  if (a.use_rgb) {
    const float* __restrict__ bias = P + L.br1;
    acc_map(accr, hrow0, hn0, lane, [&](float v, int, int col) { return fmaxf(v + bias[col], 0.0f); });
    acc_track_max(accr, a.amax ? a.amax + L.D + 3 : nullptr, lane);
    acc_to_lds(accr, Hs, W, hrow0, hn0, 0, lane);
  }
  if (a.use_cand) {
    const float* __restrict__ bias = P + L.bc1;
    acc_map(accc, hrow0, hn0, lane, [&](float v, int, int col) { return fmaxf(v + bias[col], 0.0f); });
    acc_track_max(accc, a.amax ? a.amax + L.D + 1 : nullptr, lane);
    acc_to_lds(accc, Hs, W, hrow0, hn0, W2, lane);
  }
  __syncthreads();
  if (a.use_rgb) {
    if (a.r1) tile_store<TILE>(Hs, W, 0, W2, a.r1, W2, m0, M, tid);
    // rgb_share_layer.2 + sigmoid (nerf.py:56-61)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float pre = rowdot<TPR>(Hs, W, prow, phalf, 0, W2, P + L.wr2 + c * W2) + P[L.br2 + c];
      if (phalf == 0 && pm < M) a.rgb[(size_t)pm * 3 + c] = sigmoid_f(pre);
    }
  }
  if (a.use_cand) {
    if (a.g1) tile_store<TILE>(Hs, W, W2, W2, a.g1, W2, m0, M, tid);
    f32x16 acc[TH::MT][TH::NT];
    acc_zero(acc);
    mma_lds(acc, Hs, W, hrow0, W2, P + L.wc2, W2, hn0, 0, W2, lane);
    const float* __restrict__ bias = P + L.bc2;
    acc_map(acc, hrow0, hn0, lane, [&](float v, int, int col) { return fmaxf(v + bias[col], 0.0f); });
    __syncthreads();
    acc_to_lds(acc, Hs, W, hrow0, hn0, W2, lane);
    __syncthreads();
    if (a.g2) tile_store<TILE>(Hs, W, W2, W2, a.g2, W2, m0, M, tid);
    const float pre = rowdot<TPR>(Hs, W, prow, phalf, W2, W2, P + L.wcsig) + P[L.bcsig];
    if (phalf == 0 && pm < M) a.sigma_c[pm] = softplus_f(pre);
  }
}

// ------------------------------------------------------------------------------------------------------
// Backward data-gradient chain.  Every stage leaves the pre-activation gradient of one layer in LDS (the A
// operand of the next contraction) and in HBM (the A operand of upnerf_wgrad).
template <int W, int TILE>
__global__ __launch_bounds__(NTHREADS, (TILE * W * 4 <= 32768) ? 4 : ((TILE * W * 4 <= 65536) ? 2 : 1)) void field_bwd_kernel(upnerf_layout L, upnerf_field_bwd_args a) {
  __shared__ __attribute__((aligned(16))) float Gs[TILE * W];
  __shared__ float pre_s[TILE];
  // per-row scalars of the head stages, computed once per row instead of once per 16-byte column group
  __shared__ float dpc_s[TILE], cwj_s[TILE];
  __shared__ __attribute__((aligned(16))) float dprgb_s[TILE][4];
  constexpr int W2 = W / 2;
  using TW = WaveTile<W, TILE>;
  using TH = WaveTile<W2, TILE>;
  using TX = WaveTile<UPNERF_X0, TILE>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = a.S, M = a.R * a.S, m0 = blockIdx.x * TILE, D = L.D;
  const float* __restrict__ P = a.P;
  const float* __restrict__ PT = a.PT;
  const int n0 = TW::n0(wave), row0 = TW::row0(wave);
  const int hn0 = TH::n0(wave), hrow0 = TH::row0(wave);
  const int xn0 = TX::n0(wave), xrow0 = TX::row0(wave);

  // softplus'(x) = 1 - exp(-softplus(x)); per-row scalars of the head stages
  if (tid < TILE) {
    const int m = m0 + tid;
    float v = 0.0f, dpc = 0.0f, cwj = 0.0f;
    f32x4 dprgb = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      v = a.d_sigma_s[m] * (1.0f - expf(-a.sigma_s[m]));
      a.dpre_sig_s[m] = v;
      if (a.use_cand) {
        dpc = a.d_sigma_c[m] * (1.0f - expf(-a.sigma_c[m]));
        a.dpre_sig_c[m] = dpc;
        if (a.g_G_c) cwj = a.w_cj[m];
      }
      if (a.use_rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float y = a.rgb[(size_t)m * 3 + c];
          dprgb[c] = a.d_rgb[(size_t)m * 3 + c] * (y * (1.0f - y));
        }
        *(f32x4*)&a.dpre_rgb[(size_t)m * 4] = dprgb;
      }
    }
    pre_s[tid] = v;
    dpc_s[tid] = dpc;
    cwj_s[tid] = cwj;
    *(f32x4*)&dprgb_s[tid][0] = dprgb;
  }
  __syncthreads();
  // column group / first row of this thread in the element-wise head stages; all loads of a stage are issued before
  // anything consumes them (a branch around each load made hipcc wait for every one in turn)
  constexpr int GPR = W2 / 4, ERS = NTHREADS / GPR, EPT = (TILE * GPR + NTHREADS - 1) / NTHREADS;
  const int eg = tid % GPR, er0 = tid / GPR;

  if (a.use_cand) {
    // d g2 = w_csig * dpre_c + w_cj * g_G_c[ray]   (candidate_sigma / feat_candidate_layer, nerf.py:99-100)
    float lmax = 0.0f;
    const f32x4 wv = *(const f32x4*)&P[L.wcsig + 4 * eg];
    f32x4 gv[EPT], gg[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      int m = m0 + er0 + ERS * q;
      m = m < M ? m : M - 1;
      gv[q] = *(const f32x4*)&a.g2[(size_t)m * W2 + 4 * eg];
      gg[q] = a.g_G_c ? *(const f32x4*)&a.g_G_c[(size_t)(m / S) * W2 + 4 * eg] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int row = er0 + ERS * q, m = m0 + row;
      if (row < TILE) {
        const float dp = dpc_s[row], cw = cwj_s[row];
        f32x4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = (m < M && gv[q][c] > 0.f) ? wv[c] * dp + cw * gg[q][c] : 0.f;
        if (m < M) *(f32x4*)&a.gz_g2[(size_t)m * W2 + 4 * eg] = out;
        *(f32x4*)&Gs[swz4(row, W2 + 4 * eg, W)] = out;
        lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(out.x), fabsf(out.y)), fmaxf(fabsf(out.z), fabsf(out.w))));
      }
    }
    wave_track_max(lmax, a.gmax ? a.gmax + D + 2 : nullptr, lane);
    __syncthreads();
    f32x16 acc[TH::MT][TH::NT];
    acc_zero(acc);
    mma_lds(acc, Gs, W, hrow0, W2, PT + L.t_wc2, W2, hn0, 0, W2, lane);
    acc_track_max(acc, a.gmax ? a.gmax + D + 1 : nullptr, lane);  // (before the ReLU mask: an upper bound)
    __syncthreads();
    acc_to_lds(acc, Gs, W, hrow0, hn0, W2, lane);
    __syncthreads();
    tile_mask_store<TILE>(Gs, W, W2, W2, a.g1, a.gz_g1, W2, m0, M, tid);
  }
  if (a.use_rgb) {
    // d r1 = W_r2^T (d rgb * rgb (1-rgb))   (rgb_share_layer.2 + sigmoid)
    float lmax = 0.0f;
    f32x4 wr[3], rv[EPT];
#pragma unroll
    for (int c = 0; c < 3; ++c) wr[c] = *(const f32x4*)&P[L.wr2 + c * W2 + 4 * eg];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      int m = m0 + er0 + ERS * q;
      m = m < M ? m : M - 1;
      rv[q] = *(const f32x4*)&a.r1[(size_t)m * W2 + 4 * eg];
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int row = er0 + ERS * q, m = m0 + row;
      if (row < TILE) {
        const f32x4 dp = *(const f32x4*)&dprgb_s[row][0];
        f32x4 out;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float t = 0.0f;
#pragma unroll
          for (int c = 0; c < 3; ++c) t += wr[c][u] * dp[c];
          out[u] = (m < M && rv[q][u] > 0.f) ? t : 0.f;
        }
        if (m < M) *(f32x4*)&a.gz_r1[(size_t)m * W2 + 4 * eg] = out;
        *(f32x4*)&Gs[swz4(row, 4 * eg, W)] = out;
        lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(out.x), fabsf(out.y)), fmaxf(fabsf(out.z), fabsf(out.w))));
      }
    }
    wave_track_max(lmax, a.gmax ? a.gmax + D + 3 : nullptr, lane);
  }
  __syncthreads();

  // ---- d e = [gz_r1 | gz_g1] . [W_fold | W_c1e] + w_feat * g_E_s[ray]   (e has no activation)
  {
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    const int ks = a.use_rgb ? 0 : W2;
    const int kl = (a.use_rgb ? W2 : 0) + (a.use_cand ? W2 : 0);
    if (kl > 0) mma_lds(acc, Gs, W, row0, ks, PT + L.t_head, W, n0, ks, kl, lane);
    __syncthreads();
    acc_to_lds(acc, Gs, W, row0, n0, 0, lane);
    __syncthreads();
    // + w_feat[m] * g_E_s[ray] in a coalesced pass (per-element loads in the accumulator layout were 64 dependent
    // L2 round trips per lane)
    tile_rank1_store<TILE>(Gs, W, W, a.w_feat_s, a.g_E_s, S, a.gz_e, m0, M, tid, a.gmax ? a.gmax + D : nullptr);
    __syncthreads();
  }
  // ---- d h_{D-1} = gz_e . W_e + w_sig * dpre_s, masked by relu (sign bits from the forward, in this lane's layout)
  const unsigned long long* __restrict__ hm = (const unsigned long long*)a.hmask + (size_t)blockIdx.x * NTHREADS + tid;
  const size_t hm_stride = (size_t)gridDim.x * NTHREADS;
  {
    const unsigned long long bits = hm[(size_t)(D - 1) * hm_stride];
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    mma_lds(acc, Gs, W, row0, 0, PT + L.t_we, W, n0, 0, W, lane);
    const float* __restrict__ ws = P + L.wsig;
    acc_map(acc, row0, n0, lane, [&](float v, int row, int col) { return v + ws[col] * pre_s[row]; });
    acc_apply_mask(acc, bits);
    acc_track_max(acc, a.gmax ? a.gmax + (D - 1) : nullptr, lane);
    __syncthreads();
    acc_to_lds(acc, Gs, W, row0, n0, 0, lane);
    __syncthreads();
    tile_store<TILE>(Gs, W, 0, W, a.gz_h + (size_t)(D - 1) * M * W, W, m0, M, tid);
  }
  // ---- trunk, last layer to first
  f32x16 accx[TX::MT][TX::NT];
  acc_zero(accx);
  for (int l = D - 1; l >= 1; --l) {
    const unsigned long long bits = hm[(size_t)(l - 1) * hm_stride];  // arrives under the contraction below
    if (a.need_dxyz && l == L.skip) mma_lds(accx, Gs, W, xrow0, 0, PT + L.t_skipx, W, xn0, 0, W, lane);
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    mma_lds(acc, Gs, W, row0, 0, PT + L.t_w[l], W, n0, 0, W, lane);
    acc_apply_mask(acc, bits);
    acc_track_max(acc, a.gmax ? a.gmax + (l - 1) : nullptr, lane);
    __syncthreads();
    acc_to_lds(acc, Gs, W, row0, n0, 0, lane);
    __syncthreads();
    tile_store<TILE>(Gs, W, 0, W, a.gz_h + (size_t)(l - 1) * M * W, W, m0, M, tid);
  }
  if (!a.need_dxyz) return;
  // ---- d x0 (first layer + skip) -> d xyz through the encoding
  mma_lds(accx, Gs, W, xrow0, 0, PT + L.t_w[0], W, xn0, 0, W, lane);
  __syncthreads();
  acc_to_lds(accx, Gs, W, xrow0, xn0, 0, lane);
  __syncthreads();
  for (int it = tid; it < TILE * 3; it += NTHREADS) {
    const int row = it / 3, n = it - row * 3, m = m0 + row;
    if (m >= M) continue;
    const float* __restrict__ x0 = a.x0 + (size_t)m * UPNERF_X0 + 3 + 20 * n;
    float xs[10], xc[10];  // all 20 loads in flight at once
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      xs[k] = x0[k];
      xc[k] = x0[10 + k];
    }
    float g = Gs[swz(row, n, W)];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const float f = ldexpf(PI_F, k);
      g += f * (xc[k] * Gs[swz(row, 3 + 20 * n + k, W)] - xs[k] * Gs[swz(row, 3 + 20 * n + 10 + k, W)]);
    }
    a.dxyz[(size_t)m * 3 + n] = g;
  }
}

int check_layout(const upnerf_layout* L) {
  if (!L) return UPNERF_EINVAL;
  if (L->W != 64 && L->W != 256) return UPNERF_EUNSUP;
  if (L->D < 1 || L->D > UPNERF_MAX_D) return UPNERF_EUNSUP;
  if (L->skip >= L->D) return UPNERF_EINVAL;
  return 0;
}

}  // namespace

extern "C" int upnerf_field_fwd(const upnerf_layout* L, const upnerf_field_fwd_args* a, void* stream) {
  int rc = check_layout(L);
  if (rc) return rc;
  if (!a || a->R <= 0 || a->S <= 0 || !a->rays_o || !a->rays_d || !a->z || !a->P || !a->x0 || !a->sigma_s)
    return UPNERF_EINVAL;
  if (a->use_cand && (!a->c_rows || !a->sigma_c)) return UPNERF_EINVAL;
  if (a->use_rgb && (!a->aux || !a->rgb)) return UPNERF_EINVAL;
  const long long M = (long long)a->R * a->S;
  if (M > 0x7fffffffLL) return UPNERF_EINVAL;
  const int grid = (int)((M + FIELD_TILE - 1) / FIELD_TILE);
  hipStream_t st = (hipStream_t)stream;
  if (L->W == 256)
    hipLaunchKernelGGL((field_fwd_kernel<256, FIELD_TILE>), dim3(grid), dim3(NTHREADS), 0, st, *L, *a);
  else
    hipLaunchKernelGGL((field_fwd_kernel<64, FIELD_TILE>), dim3(grid), dim3(NTHREADS), 0, st, *L, *a);
  return (int)hipGetLastError();
}

extern "C" int upnerf_field_bwd(const upnerf_layout* L, const upnerf_field_bwd_args* a, void* stream) {
  int rc = check_layout(L);
  if (rc) return rc;
  if (!a || a->R <= 0 || a->S <= 0 || !a->P || !a->PT || !a->d_sigma_s || !a->sigma_s || !a->h || !a->gz_h ||
      !a->gz_e || !a->dpre_sig_s || !a->hmask)
    return UPNERF_EINVAL;
  if (a->use_cand && (!a->d_sigma_c || !a->sigma_c || !a->g1 || !a->g2 || !a->gz_g1 || !a->gz_g2 || !a->dpre_sig_c))
    return UPNERF_EINVAL;
  if (a->use_cand && a->g_G_c && !a->w_cj) return UPNERF_EINVAL;
  if (a->use_rgb && (!a->d_rgb || !a->rgb || !a->r1 || !a->gz_r1 || !a->dpre_rgb)) return UPNERF_EINVAL;
  if (a->g_E_s && !a->w_feat_s) return UPNERF_EINVAL;
  if (a->need_dxyz && (!a->dxyz || !a->x0)) return UPNERF_EINVAL;
  if (a->tile_part || a->gz_rg_ld) return UPNERF_EUNSUP;  // per-tile partial sums, joined gz_r1 / gz_g1: f16x3 variant only
  const long long M = (long long)a->R * a->S;
  if (M > 0x7fffffffLL) return UPNERF_EINVAL;
  const int grid = (int)((M + FIELD_TILE - 1) / FIELD_TILE);
  hipStream_t st = (hipStream_t)stream;
  if (L->W == 256)
    hipLaunchKernelGGL((field_bwd_kernel<256, FIELD_TILE>), dim3(grid), dim3(NTHREADS), 0, st, *L, *a);
  else
    hipLaunchKernelGGL((field_bwd_kernel<64, FIELD_TILE>), dim3(grid), dim3(NTHREADS), 0, st, *L, *a);
  return (int)hipGetLastError();
}
