// Fused NeRF field on the f16 matrix cores (csrc/common16.cuh): PE -> 8x256 trunk (skip) -> density / final / candidate /
// colour heads in one kernel per pass, forward and backward data-gradient chain, in two arithmetic modes selected by
// `planes` in the argument structs:
//   f16x3 (NP = 2, default)  fp32-accurate 3-term hi/lo split, fp32 in / fp32 out: the headline configuration;
//   f16   (NP = 1)           fp16 weights and activations, one MFMA per product, fp32 accumulate, fp32 encoding / heads /
//                            stores (BASELINE.json configs[3]).
//
// Reference behaviour: models/nerf.py:80-124 (NeRF.forward), 126-147 (positional_encoding), evaluated by
// models/rendering.py:102-122.  Layout of the work:
//   * a workgroup owns 64 samples; their activations live in LDS as fp16 planes carrying  value * 2^e  with one exponent
//     per tile and stage, picked from the tile's running maximum (wave max -> 4 floats in LDS -> the barrier every
//     epilogue already has);
//   * weights arrive pre-split (upnerf_frag16) with one exponent per matrix (table wexp) and stream L2 -> registers in MFMA
//     fragment order; the epilogue folds both exponents into the fma that adds the bias: v = fma(acc, 2^-(e_tile+e_w), b);
//   * the contraction is issued TRANSPOSED (weights as the A operand): a lane owns one sample row and four consecutive
//     features per register quad, so the epilogue works on packed pairs / quads (common16.cuh header);
//   * trunk / final activations and all pre-activation gradients are stored straight from the accumulators (16 bytes per
//     lane and quad), the 128-wide head activations from the planes.
#include "common16.cuh"

#define F16_TILE 64
#define F16_WAVES 4
#define F16_THREADS (64 * F16_WAVES)
// two workgroups per CU = 2 waves per SIMD; hipcc takes the second __launch_bounds__ argument as the minimum number of
// waves per SIMD
#define F16_WAVES_PER_EU 2
// Weight fragments are requested this many 16-deep k-blocks ahead of the MFMAs that consume them.  Measured with the stamps
// build (per wave and trunk layer): f16 mode 10.4k cycles per K loop one block ahead = 650 cycles per k-block = the L2
// latency, 8.1k three ahead and 8.3k seven ahead -- from there on the loop is bound by the bytes the CU can pull from L2
// (~32 B/clk: 128 KB of fp16 weights per 64-row tile and layer).  The f16x3 mode moves twice the bytes (16.5k cycles per K
// loop at every depth: bandwidth-bound already one block ahead), deeper rings only add register pressure there.
#ifndef F16_AHEAD_X3
#define F16_AHEAD_X3 1
#endif
#ifndef F16_AHEAD_F16
#define F16_AHEAD_F16 3
#endif

// Diagnostic build only (make -C upnerf_amd/csrc stamps, -DUPNERF_STAMPS): per-phase shader-clock stamps of the forward
// trunk loop, accumulated in registers and flushed once per workgroup (tools/stamps_field16.py).  Never compiled into
// the shipped library.
#ifdef UPNERF_STAMPS
__device__ unsigned long long upnerf_stamp_acc[16];  // [0..7] forward trunk phases, [8..15] backward stages
#define STAMP_DECL                                         \
  unsigned long long _t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  unsigned long long _t_prev = __builtin_amdgcn_s_memtime()
#define STAMP(i)                                                \
  do {                                                          \
    const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    _t_acc[i] += _t - _t_prev;                                  \
    _t_prev = _t;                                               \
  } while (0)
#define STAMP_FLUSH_AT(base)                                                               \
  do {                                                                                     \
    if (lane == 0 && (blockIdx.x & 15) == 0)                                               \
      for (int _i = 0; _i < 8; ++_i) atomicAdd(&upnerf_stamp_acc[(base) + _i], _t_acc[_i]); \
  } while (0)
#define STAMP_FLUSH STAMP_FLUSH_AT(0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#define STAMP_FLUSH_AT(base)
#endif

namespace {

// How the 4 waves of a workgroup share a [TILE x N] output tile (32 x 32 MFMA tiles); same rules as WaveTile in
// common.cuh: surplus waves of a narrow layer recompute a piece another wave owns.
template <int N, int TILE>
struct WaveTile16 {
  static constexpr int NW = F16_WAVES;
  static constexpr int NT = (N >= 256) ? 2 : 1;
  static constexpr int NG = N / 32 / NT;
  static constexpr int MG = TILE / 32;
  static constexpr int WN = NG >= NW ? NW : NG;
  static constexpr int WM = (NW / WN) < MG ? (NW / WN) : MG;
  static constexpr int MT = MG / WM;
  __device__ static __forceinline__ int n0(int wave) { return (wave % WN) * 32 * NT; }
  __device__ static __forceinline__ int row0(int wave) { return ((wave / WN) % WM) * 32 * MT; }
};

__device__ __forceinline__ float wg_max(const float* smax) {
  float m = smax[0];
#pragma unroll
  for (int w = 1; w < F16_WAVES; ++w) m = fmaxf(m, smax[w]);
  return m;
}

__device__ __forceinline__ float pow2f(int n) { return ldexpf(1.0f, n); }

// LDS planes -> row-major fp32 global tensor, coalesced (8 columns = 32 bytes per thread, whole rows per wave).  The
// LDS reads of BATCH row groups are issued before the first conversion, so their latency is paid once per batch.
template <int NP, int W, int TILE, int NCOLS, int BATCH = 1>
__device__ __forceinline__ void tile_store16(const char* Ph, const char* Pl, int c0, float unscale,
                                             float* __restrict__ dst, int ldg, int m0, int M, int tid) {
  constexpr int GPR = NCOLS >> 3, ITER = TILE * GPR / F16_THREADS;
  static_assert(TILE * GPR % F16_THREADS == 0, "whole passes of the workgroup");
  constexpr int B = (BATCH == 0 || BATCH > ITER) ? ITER : BATCH;
  static_assert(ITER % B == 0, "whole batches");
  const bool whole = m0 + TILE <= M;
#pragma unroll 1
  for (int it0 = 0; it0 < ITER; it0 += B) {
    h8 wh[B], wl[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int idx = tid + (it0 + j) * F16_THREADS, row = idx / GPR, g = idx % GPR;
      const int o = poff<W>(row, c0 + 8 * g);
      wh[j] = *(const h8*)(Ph + o);
      if constexpr (NP == 2) wl[j] = *(const h8*)(Pl + o);
    }
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int idx = tid + (it0 + j) * F16_THREADS, row = idx / GPR, g = idx % GPR;
      f32x4 o0, o1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (NP == 2) {
          o0[c] = ((float)wh[j][c] + (float)wl[j][c]) * unscale;
          o1[c] = ((float)wh[j][4 + c] + (float)wl[j][4 + c]) * unscale;
        } else {
          o0[c] = (float)wh[j][c] * unscale;
          o1[c] = (float)wh[j][4 + c] * unscale;
        }
      }
      if (whole || m0 + row < M) {
        float* p = &dst[(size_t)(m0 + row) * ldg + 8 * g];
        *(f32x4*)p = o0;
        *(f32x4*)(p + 4) = o1;
      }
    }
  }
}

// The hi plane of the tile (columns [0, W)) -> row-major fp16 global tensor, as it stands (16 bytes per thread and step);
// the tile's exponent goes to its slot of the exponent table.  f16 mode only.
template <int W, int TILE>
__device__ __forceinline__ void tile_copy16(const char* Ph, int e, uint16_t* __restrict__ dst, int32_t* __restrict__ dexp, int m0,
                                            int M, int tid) {
  constexpr int GPR = W >> 3, ITER = TILE * GPR / F16_THREADS;
  if (tid == 0) dexp[m0 / TILE] = e;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * F16_THREADS, row = idx / GPR, g = idx % GPR;
    const f32x4 v = *(const f32x4*)(Ph + poff<W>(row, 8 * g));
    if (m0 + row < M) *(f32x4*)((char*)dst + ((size_t)(m0 + row) * W + 8 * g) * 2) = v;
  }
}

__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  return m;
}

__device__ __forceinline__ void track(float* __restrict__ slot, float mx, int tid) {
  if (slot && tid == 0) atomicMax((unsigned int*)slot, __float_as_uint(mx));
}

// four fp32 values * 2^e -> one 8-byte write per plane at (row, col % 4 == 0)
template <int NP, int W>
__device__ __forceinline__ void put_quad(char* Ph, char* Pl, int row, int col, const f32x4& v, int e) {
  h4 hi, lo;
  split_quad<NP>(ldexpf(v[0], e), ldexpf(v[1], e), ldexpf(v[2], e), ldexpf(v[3], e), hi, lo);
  const int o = poff<W>(row, col);
  *(h4*)(Ph + o) = hi;
  if constexpr (NP == 2) *(h4*)(Pl + o) = lo;
}

// ------------------------------------------------------------------------------------------------------------------
template <int NP, int TILE>
__global__ __launch_bounds__(F16_THREADS, F16_WAVES_PER_EU) void field16_fwd_kernel(upnerf_layout L, upnerf_field_fwd_args a) {
  constexpr int W = 256, W2 = 128;
  constexpr int AH = NP == 2 ? F16_AHEAD_X3 : F16_AHEAD_F16;  // weight k-blocks in flight (common16.cuh:mma16_lds)
  constexpr int TPR = F16_THREADS / TILE;
  __shared__ __attribute__((aligned(16))) char planes[NP * TILE * W * 2];
  __shared__ float smax[F16_WAVES], smaxb[F16_WAVES];
  __shared__ float xyz_s[TILE * 3];
  // per-layer offsets of the layout: read from LDS inside the layer loop.  Indexing the by-value struct with the runtime
  // layer makes hipcc copy it to scratch and fetch the entry with a VMEM load + s_waitcnt vmcnt(0) -- a full drain of the
  // previous layer's activation stores at the top of every layer.
  __shared__ int loff_s[2 * UPNERF_MAX_D];
  char* Ph = planes;
  char* Pl = planes + (NP - 1) * TILE * W * 2;  // NP == 1: never dereferenced
  using TW = WaveTile16<W, TILE>;
  using TH = WaveTile16<W2, TILE>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int S = a.S, M = a.R * a.S, m0 = blockIdx.x * TILE;
  const float* __restrict__ P = a.P;
  const char* __restrict__ P16 = (const char*)a.P16;
  const int* __restrict__ wexp = a.wexp;
  const int n0 = TW::n0(wave), row0 = TW::row0(wave);
  const int hn0 = TH::n0(wave), hrow0 = TH::row0(wave);
  const int D = L.D;
  STAMP_DECL;
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < UPNERF_MAX_D; ++l) {
      loff_s[l] = L.w[l];
      loff_s[UPNERF_MAX_D + l] = L.b[l];
    }
  }

  // ---- sample positions (rendering.py:251 / 308) and the maxima that bound the side inputs of this tile
  {
    float xm = 0.0f;
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
      xyz_s[tid * 3 + 0] = x;
      xyz_s[tid * 3 + 1] = y;
      xyz_s[tid * 3 + 2] = zc;
      xm = fmaxf(fmaxf(fabsf(x), fabsf(y)), fmaxf(fabsf(zc), 1.0f));  // |sin|, |cos| <= 1
    }
    float sm = 0.0f;
    const int mlast = (m0 + TILE < M ? m0 + TILE : M) - 1;
    const int ray0 = m0 / S, nr = mlast / S - ray0 + 1;
    if (a.use_rgb)
      for (int idx = tid; idx < nr * UPNERF_AUXK; idx += F16_THREADS) sm = fmaxf(sm, fabsf(a.aux[(size_t)ray0 * UPNERF_AUXK + idx]));
    if (a.use_cand)
      for (int idx = tid; idx < nr * UPNERF_CK; idx += F16_THREADS) sm = fmaxf(sm, fabsf(a.c_rows[(size_t)ray0 * UPNERF_CK + idx]));
    xm = wave_max(xm);
    sm = wave_max(sm);
    if (lane == 0) {
      smax[wave] = xm;
      smaxb[wave] = sm;
    }
  }
  __syncthreads();
  const float x0max = wg_max(smax), sidemax = wg_max(smaxb);
  track(a.amax ? a.amax + D + 4 : nullptr, x0max, tid);
  int ecur = scale_exp(x0max);
  // ---- BARF-masked encoding (nerf.py:126-147) straight into the planes
  {
    const float sc = pow2f(ecur);
    auto put = [&](int row, int col, float v) {
      _Float16 h, l;
      split16(v * sc, h, l);
      const int o = poff<W>(row, col);
      *(_Float16*)(Ph + o) = h;
      if constexpr (NP == 2) *(_Float16*)(Pl + o) = l;
    };
    const float* __restrict__ wkd = a.wk_xyz_dev;  // per-step band weights from device memory under graph replay
    for (int it = tid; it < TILE * 3; it += F16_THREADS) {
      const int row = it / 3, n = it - row * 3;
      const float xv = xyz_s[row * 3 + n];
      put(row, n, xv);
      if (n == 0) put(row, 63, 0.0f);
#pragma unroll 1
      for (int k = 0; k < 10; ++k) {
        const float arg = xv * ldexpf(PI_F, k);
        float sv, cv;
        sincos_f32_via_f64(arg, sv, cv);
        const float wk = wkd ? wkd[k] : a.wk_xyz[k];
        put(row, 3 + 20 * n + k, sv * wk);
        put(row, 3 + 20 * n + 10 + k, cv * wk);
      }
    }
  }
  __syncthreads();
  tile_store16<NP, W, TILE, UPNERF_X0>(Ph, Pl, 0, pow2f(-ecur), a.x0, UPNERF_X0, m0, M, tid);

  // ---- trunk (nerf.py:84-87)
  STAMP(7);  // sample positions + encoding + x0 store
  for (int l = 0; l < D; ++l) {
    STAMP(0);
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    const int wel = __builtin_amdgcn_readfirstlane(wexp[l]);
    const int wl = __builtin_amdgcn_readfirstlane(loff_s[l]), bl_off = __builtin_amdgcn_readfirstlane(loff_s[UPNERF_MAX_D + l]);
    if (l == 0) {
      mma16_lds<NP, W, UPNERF_X0 / 16, AH>(acc, Ph, Pl, row0, 0, P16 + 4 * (size_t)wl, UPNERF_X0 / 16, n0, 0, lane);
    } else if (l == L.skip) {
      const float* ap[TW::MT];
#pragma unroll
      for (int mt = 0; mt < TW::MT; ++mt) {
        int m = m0 + row0 + 32 * mt + li;
        m = m < M ? m : M - 1;
        ap[mt] = a.x0 + (size_t)m * UPNERF_X0 + 8 * hh;
      }
      mma16_glb<NP>(acc, ap, ecur, P16 + 4 * (size_t)wl, (UPNERF_X0 + W) / 16, n0, 0, UPNERF_X0, lane);
      mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, P16 + 4 * (size_t)wl, (UPNERF_X0 + W) / 16, n0, UPNERF_X0, lane);
    } else {
      mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, P16 + 4 * (size_t)wl, W / 16, n0, 0, lane);
    }
    STAMP(1);
    f32x4 bl[TW::NT][4];
    load_cols(bl, P + bl_off, n0, hh);
    const unsigned long long bits = acc_fma_relu_pack(acc, pow2f(-(ecur + wel)), bl);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    if (a.hmask) ((unsigned long long*)a.hmask)[((size_t)l * gridDim.x + blockIdx.x) * F16_THREADS + tid] = bits;
    if (a.h && !(a.h16 && a.h_last_only && l != D - 1))
      acc_store_global(acc, a.h + (a.h_last_only ? 0 : (size_t)l * M * W), W, m0, M, row0, n0, lane);
    STAMP(2);
    __syncthreads();
    STAMP(3);
    float mx = wg_max(smax);
    track(a.amax ? a.amax + l : nullptr, mx, tid);
    if (l + 1 == L.skip) mx = fmaxf(mx, x0max);  // the skip layer feeds x0 rows through the same accumulators
    ecur = scale_exp(mx);
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    STAMP(4);
    __syncthreads();
    STAMP(5);
    // fp16 storage: the (hi) plane IS the stored tile -- the next write to it is a barrier away.  (f16x3 mode: an option;
    // the weight-gradient operand then is the activation rounded to fp16, the forward chain keeps hi + lo.)
    if (a.h16) tile_copy16<W, TILE>(Ph, ecur, a.h16 + (size_t)l * M * W, a.hexp + (size_t)l * gridDim.x, m0, M, tid);
    STAMP(6);
  }

  STAMP_FLUSH;
#ifdef UPNERF_STAMPS
  for (int _i = 0; _i < 8; ++_i) _t_acc[_i] = 0;
  _t_prev = __builtin_amdgcn_s_memtime();
#endif
  const int prow = tid / TPR, phalf = tid % TPR, pm = m0 + prow;
  // ---- shared density head (nerf.py:89): softplus(w . h + b)
  {
    const float pre = rowdot16<NP, W, TPR, W>(Ph, Pl, prow, phalf, 0, P + L.wsig, pow2f(-ecur)) + P[L.bsig];
    if (phalf == 0 && pm < M) a.sigma_s[pm] = softplus_f(pre);
  }
  // density-only pass (nerf.py:90-91 `sigma_only`): nobody consumes e, so the pass ends here
  if (!a.e && !a.use_rgb && !a.use_cand) {
    STAMP_FLUSH_AT(8);
    return;
  }
  // ---- xyz_encoding_final (nerf.py:93), no activation
  {
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    f32x4 be[TW::NT][4];
    load_cols(be, P + L.be, n0, hh);
    mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, P16 + 4 * (size_t)L.we, W / 16, n0, 0, lane);
    acc_fma_bias<false>(acc, pow2f(-(ecur + wexp[8])), be);
    if (a.e) acc_store_global(acc, a.e, W, m0, M, row0, n0, lane);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    const float mx = wg_max(smax);
    track(a.amax ? a.amax + D : nullptr, mx, tid);
    ecur = scale_exp(fmaxf(mx, sidemax));  // the heads feed per-ray rows through the same accumulators
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    __syncthreads();
  }
  STAMP(5);  // density head + xyz_encoding_final
  if (!a.use_rgb && !a.use_cand) {
    STAMP_FLUSH_AT(8);
    return;
  }

  // ---- first layer of the colour head (folded, nerf.py:95+102-109) and of the candidate head (nerf.py:97-98)
  f32x16 accr[TH::MT][TH::NT], accc[TH::MT][TH::NT];
  int rayrow[TH::MT];
#pragma unroll
  for (int mt = 0; mt < TH::MT; ++mt) {
    int m = m0 + hrow0 + 32 * mt + li;
    m = m < M ? m : M - 1;
    rayrow[mt] = m / S;
  }
  float mr = 0.0f, mc = 0.0f;
  if (a.use_rgb) {
    acc_zero(accr);
    f32x4 br[TH::NT][4];
    load_cols(br, P + L.br1, hn0, hh);
    mma16_lds<NP, W, W / 16, AH>(accr, Ph, Pl, hrow0, 0, P16 + 4 * (size_t)L.wr1, (W + UPNERF_AUXK) / 16, hn0, 0, lane);
    const float* ap[TH::MT];
#pragma unroll
    for (int mt = 0; mt < TH::MT; ++mt) ap[mt] = a.aux + (size_t)rayrow[mt] * UPNERF_AUXK + 8 * hh;
    mma16_glb<NP>(accr, ap, ecur, P16 + 4 * (size_t)L.wr1, (W + UPNERF_AUXK) / 16, hn0, W, UPNERF_AUXK, lane);
    acc_fma_bias<true>(accr, pow2f(-(ecur + wexp[11])), br);
    mr = acc_absmax(accr);
  }
  if (a.use_cand) {
    acc_zero(accc);
    f32x4 bc[TH::NT][4];
    load_cols(bc, P + L.bc1, hn0, hh);
    mma16_lds<NP, W, W / 16, AH>(accc, Ph, Pl, hrow0, 0, P16 + 4 * (size_t)L.wc1, (W + UPNERF_CK) / 16, hn0, 0, lane);
    const float* ap[TH::MT];
#pragma unroll
    for (int mt = 0; mt < TH::MT; ++mt) ap[mt] = a.c_rows + (size_t)rayrow[mt] * UPNERF_CK + 8 * hh;
    mma16_glb<NP>(accc, ap, ecur, P16 + 4 * (size_t)L.wc1, (W + UPNERF_CK) / 16, hn0, W, UPNERF_CK, lane);
    const unsigned long long bits = acc_fma_relu_pack(accc, pow2f(-(ecur + wexp[9])), bc);
    if (a.hmask) ((unsigned long long*)a.hmask)[((size_t)D * gridDim.x + blockIdx.x) * F16_THREADS + tid] = bits;
    mc = acc_absmax(accc);
  }
  if (lane == 0) {
    smax[wave] = mr;
    smaxb[wave] = mc;
  }
  __syncthreads();
  {
    const float mxr = wg_max(smax), mxc = wg_max(smaxb);
    track(a.amax && a.use_rgb ? a.amax + D + 3 : nullptr, mxr, tid);
    track(a.amax && a.use_cand ? a.amax + D + 1 : nullptr, mxc, tid);
    ecur = scale_exp(fmaxf(mxr, mxc));
  }
  if (a.use_rgb) acc_to_planes<NP, W>(accr, Ph, Pl, hrow0, hn0, 0, ecur, lane);
  if (a.use_cand) acc_to_planes<NP, W>(accc, Ph, Pl, hrow0, hn0, W2, ecur, lane);
  __syncthreads();
  if (a.use_rgb) {
    if (a.r1) tile_store16<NP, W, TILE, W2>(Ph, Pl, 0, pow2f(-ecur), a.r1, W2, m0, M, tid);
    // rgb_share_layer.2 + sigmoid (nerf.py:56-61)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float pre = rowdot16<NP, W, TPR, W2>(Ph, Pl, prow, phalf, 0, P + L.wr2 + c * W2, pow2f(-ecur)) + P[L.br2 + c];
      if (phalf == 0 && pm < M) a.rgb[(size_t)pm * 3 + c] = sigmoid_f(pre);
    }
  }
  if (a.use_cand) {
    if (a.g1) tile_store16<NP, W, TILE, W2>(Ph, Pl, W2, pow2f(-ecur), a.g1, W2, m0, M, tid);
    f32x16 acc[TH::MT][TH::NT];
    acc_zero(acc);
    f32x4 b2[TH::NT][4];
    load_cols(b2, P + L.bc2, hn0, hh);
    mma16_lds<NP, W, W2 / 16, AH>(acc, Ph, Pl, hrow0, W2, P16 + 4 * (size_t)L.wc2, W2 / 16, hn0, 0, lane);
    acc_fma_bias<true>(acc, pow2f(-(ecur + wexp[10])), b2);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    ecur = scale_exp(wg_max(smax));
    acc_to_planes<NP, W>(acc, Ph, Pl, hrow0, hn0, W2, ecur, lane);
    __syncthreads();
    if (a.g2) tile_store16<NP, W, TILE, W2>(Ph, Pl, W2, pow2f(-ecur), a.g2, W2, m0, M, tid);
    const float pre = rowdot16<NP, W, TPR, W2>(Ph, Pl, prow, phalf, W2, P + L.wcsig, pow2f(-ecur)) + P[L.bcsig];
    if (phalf == 0 && pm < M) a.sigma_c[pm] = softplus_f(pre);
  }
  STAMP(6);  // colour / candidate heads
  STAMP_FLUSH_AT(8);
}

// ------------------------------------------------------------------------------------------------------------------
// Backward data-gradient chain (autograd of nerf.py:80-124), stage for stage as field.hip:field_bwd_kernel.
template <int NP, int TILE>
__global__ __launch_bounds__(F16_THREADS, F16_WAVES_PER_EU) void field16_bwd_kernel(upnerf_layout L, upnerf_field_bwd_args a) {
  constexpr int W = 256, W2 = 128;
  constexpr int AH = NP == 2 ? F16_AHEAD_X3 : F16_AHEAD_F16;
  constexpr int MAXRAYS = 3;            // rays a 64-sample tile can touch when S >= 32
  constexpr int GPR = W2 / 4;           // 16-byte groups per half-width row
  constexpr int EPT = TILE * GPR / F16_THREADS;  // groups per thread in the elementwise stages
  constexpr int PLANE_BYTES = NP * TILE * W * 2;
  constexpr int LDS_BYTES = PLANE_BYTES < TILE * UPNERF_X0 * 4 ? TILE * UPNERF_X0 * 4 : PLANE_BYTES;
  __shared__ __attribute__((aligned(16))) char planes[LDS_BYTES];
  __shared__ float smax[F16_WAVES], smaxb[F16_WAVES];
  __shared__ float pre_s[TILE];
  __shared__ __attribute__((aligned(16))) float wfj[MAXRAYS][TILE];  // w_feat[row] on the row's ray slot, else 0
  // per-row scalars of the head stages, computed once per row (not once per 16-byte column group): d pre-activation of
  // the candidate density, its compositing weight, the three d pre-activations of the colour output
  __shared__ float dpc_s[TILE], cwj_s[TILE];
  __shared__ __attribute__((aligned(16))) float dprgb_s[TILE][4];
  __shared__ int loff_s[UPNERF_MAX_D];  // t_w[l] (see the forward kernel: no runtime index into the by-value struct)
  char* Ph = planes;
  char* Pl = planes + (NP - 1) * TILE * W * 2;
  using TW = WaveTile16<W, TILE>;
  using TH = WaveTile16<W2, TILE>;
  using TX = WaveTile16<UPNERF_X0, TILE>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int S = a.S, M = a.R * a.S, m0 = blockIdx.x * TILE, D = L.D;
  const float* __restrict__ P = a.P;
  const char* __restrict__ PT16 = (const char*)a.PT16;
  const int* __restrict__ wexp = a.wexp;
  const int n0 = TW::n0(wave), row0 = TW::row0(wave);
  const int hn0 = TH::n0(wave), hrow0 = TH::row0(wave);
  const int xn0 = TX::n0(wave), xrow0 = TX::row0(wave);
  const int ray0 = m0 / S;
  const unsigned long long* __restrict__ hm = (const unsigned long long*)a.hmask + (size_t)blockIdx.x * F16_THREADS + tid;
  const size_t hm_stride = (size_t)gridDim.x * F16_THREADS;

  if (tid == 64) {
#pragma unroll
    for (int l = 0; l < UPNERF_MAX_D; ++l) loff_s[l] = L.t_w[l];
  }
  // softplus'(x) = 1 - exp(-softplus(x)); per-row feature weight on its ray slot; per-row scalars of the head stages
  if (tid < TILE) {
    const int m = m0 + tid;
    float v = 0.0f, wf = 0.0f, dpc = 0.0f, cwj = 0.0f;
    f32x4 dprgb = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    if (m < M) {
      v = a.d_sigma_s[m] * (1.0f - expf(-a.sigma_s[m]));
      a.dpre_sig_s[m] = v;
      if (a.g_E_s) wf = a.w_feat_s[m];
      j = m / S - ray0;
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
#pragma unroll
    for (int q = 0; q < MAXRAYS; ++q) wfj[q][tid] = (q == j) ? wf : 0.0f;
  }
  __syncthreads();
  // column group and first row of this thread in the elementwise head stages (row advances by F16_THREADS / GPR per step)
  const int eg = tid % GPR, er0 = tid / GPR;
  constexpr int ERS = F16_THREADS / GPR;

  STAMP_DECL;
  int erg = 0;  // exponent of the [gz_r1 | gz_g1] planes
  {
    f32x16 accg[TH::MT][TH::NT];
    acc_zero(accg);
    float mg1 = 0.0f;
    if (a.use_cand) {
      // d g2 = w_csig * dpre_c + w_cj * g_G_c[ray]   (candidate_sigma / feat_candidate_layer, nerf.py:99-100)
      f32x4 vals[EPT];
      float lmax = 0.0f;
      {
        // all loads first, unconditionally (rows past M are clamped and masked afterwards): a branch around each load
        // makes hipcc wait for every one of them in turn
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
          const float dp = dpc_s[row], cw = cwj_s[row];
          f32x4 out;
#pragma unroll
          for (int c = 0; c < 4; ++c) out[c] = (m < M && gv[q][c] > 0.f) ? wv[c] * dp + cw * gg[q][c] : 0.f;
          if (m < M) *(f32x4*)&a.gz_g2[(size_t)m * W2 + 4 * eg] = out;
          vals[q] = out;
          lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(out[0]), fabsf(out[1])), fmaxf(fabsf(out[2]), fabsf(out[3]))));
        }
      }
      lmax = wave_max(lmax);
      if (lane == 0) smax[wave] = lmax;
      __syncthreads();
      const float mx = wg_max(smax);
      track(a.gmax ? a.gmax + D + 2 : nullptr, mx, tid);
      const int eg2 = scale_exp(mx);
#pragma unroll
      for (int q = 0; q < EPT; ++q) put_quad<NP, W>(Ph, Pl, er0 + ERS * q, W2 + 4 * eg, vals[q], eg2);
      __syncthreads();
      mma16_lds<NP, W, W2 / 16, AH>(accg, Ph, Pl, hrow0, W2, PT16 + 4 * (size_t)L.t_wc2, W2 / 16, hn0, 0, lane);
      acc_scale(accg, pow2f(-(eg2 + wexp[10])));
      acc_apply_mask(accg, hm[(size_t)D * hm_stride]);
      mg1 = acc_absmax(accg);
    }
    f32x4 valr[EPT];
    float mr1 = 0.0f;
    if (a.use_rgb) {
      // d r1 = W_r2^T (d rgb * rgb (1-rgb))   (rgb_share_layer.2 + sigmoid); loads first, as above
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
        valr[q] = out;
        mr1 = fmaxf(mr1, fmaxf(fmaxf(fabsf(out[0]), fabsf(out[1])), fmaxf(fabsf(out[2]), fabsf(out[3]))));
      }
      mr1 = wave_max(mr1);
    }
    if (lane == 0) {
      smax[wave] = mg1;
      smaxb[wave] = mr1;
    }
    __syncthreads();  // also: every wave is done reading the gz_g2 planes
    {
      const float mxg = wg_max(smax), mxr = wg_max(smaxb);
      track(a.gmax && a.use_cand ? a.gmax + D + 1 : nullptr, mxg, tid);
      track(a.gmax && a.use_rgb ? a.gmax + D + 3 : nullptr, mxr, tid);
      erg = scale_exp(fmaxf(mxg, mxr));
    }
    if (a.use_cand) acc_to_planes<NP, W>(accg, Ph, Pl, hrow0, hn0, W2, erg, lane);
    if (a.use_rgb) {
#pragma unroll
      for (int q = 0; q < EPT; ++q) put_quad<NP, W>(Ph, Pl, er0 + ERS * q, 4 * eg, valr[q], erg);
    }
    __syncthreads();
    if (a.use_cand) tile_store16<NP, W, TILE, W2>(Ph, Pl, W2, pow2f(-erg), a.gz_g1, W2, m0, M, tid);
  }

  STAMP(0);  // head stages (elementwise d g2 / d r1, 128-wide contraction, plane writes)
  int ecur;
  // ---- d e = [gz_r1 | gz_g1] . [W_fold | W_c1e] + w_feat * g_E_s[ray]   (e has no activation)
  {
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    const int ks = a.use_rgb ? 0 : W2;
    const int kl = (a.use_rgb ? W2 : 0) + (a.use_cand ? W2 : 0);
    if (kl == W)
      mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, PT16 + 4 * (size_t)L.t_head, W / 16, n0, 0, lane);
    else if (kl == W2)
      mma16_lds<NP, W, W2 / 16, AH>(acc, Ph, Pl, row0, ks, PT16 + 4 * (size_t)L.t_head, W / 16, n0, ks, lane);;
    acc_scale(acc, pow2f(-(erg + wexp[12])));
    if (a.g_E_s) {
      const int mlast = (m0 + TILE < M ? m0 + TILE : M) - 1;
      const int nr = mlast / S - ray0 + 1;
#pragma unroll
      for (int j = 0; j < MAXRAYS; ++j) {
        if (j < nr) {
          f32x4 gv[TW::NT][4];
          load_cols(gv, a.g_E_s + (size_t)(ray0 + j) * W, n0, hh);
#pragma unroll
          for (int mt = 0; mt < TW::MT; ++mt) {
            const float wv = wfj[j][row0 + 32 * mt + li];  // zero unless this lane's row belongs to ray slot j
#pragma unroll
            for (int nt = 0; nt < TW::NT; ++nt)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(wv, gv[nt][r >> 2][r & 3], acc[mt][nt][r]);
          }
        }
      }
    }
    acc_store_global(acc, a.gz_e, W, m0, M, row0, n0, lane);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    const float mx = wg_max(smax);
    track(a.gmax ? a.gmax + D : nullptr, mx, tid);
    ecur = scale_exp(mx);
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    __syncthreads();
  }
  STAMP(1);  // d e
  // ---- d h_{D-1} = gz_e . W_e + w_sig * dpre_s, masked by relu (sign bits from the forward, in this lane's layout)
  {
    const unsigned long long bits = hm[(size_t)(D - 1) * hm_stride];
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    f32x4 ws[TW::NT][4];
    load_cols(ws, P + L.wsig, n0, hh);
    mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, PT16 + 4 * (size_t)L.t_we, W / 16, n0, 0, lane);
    const float un = pow2f(-(ecur + wexp[8]));
#pragma unroll
    for (int mt = 0; mt < TW::MT; ++mt) {
      const float ps = pre_s[row0 + 32 * mt + li];
#pragma unroll
      for (int nt = 0; nt < TW::NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(acc[mt][nt][r], un, ws[nt][r >> 2][r & 3] * ps);
    }
    acc_apply_mask(acc, bits);
    if (a.gz_h) acc_store_global(acc, a.gz_h + (size_t)(D - 1) * M * W, W, m0, M, row0, n0, lane);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    const float mx = wg_max(smax);
    track(a.gmax ? a.gmax + (D - 1) : nullptr, mx, tid);
    ecur = scale_exp(mx);
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    __syncthreads();
    if (a.gz16) tile_copy16<W, TILE>(Ph, ecur, a.gz16 + (size_t)(D - 1) * M * W, a.gzexp + (size_t)(D - 1) * gridDim.x, m0, M, tid);
  }
  STAMP(2);  // d h_{D-1}
  // ---- trunk, last layer to first
  f32x16 accx[TX::MT][TX::NT];
  acc_zero(accx);
  for (int l = D - 1; l >= 1; --l) {
    const unsigned long long bits = hm[(size_t)(l - 1) * hm_stride];  // arrives under the contraction below
    if (a.need_dxyz && l == L.skip) {
      mma16_lds<NP, W, W / 16, AH>(accx, Ph, Pl, xrow0, 0, PT16 + 4 * (size_t)L.t_skipx, W / 16, xn0, 0, lane);
      acc_scale(accx, pow2f(-(ecur + wexp[l])));  // natural units: the layer-0 term arrives at another exponent
    }
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    const int wel = __builtin_amdgcn_readfirstlane(wexp[l]);  // wave-uniform; asked for before the contraction
    mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, PT16 + 4 * (size_t)__builtin_amdgcn_readfirstlane(loff_s[l]), W / 16, n0, 0, lane);
    acc_scale(acc, pow2f(-(ecur + wel)));
    acc_apply_mask(acc, bits);
    if (a.gz_h) acc_store_global(acc, a.gz_h + (size_t)(l - 1) * M * W, W, m0, M, row0, n0, lane);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    const float mx = wg_max(smax);
    track(a.gmax ? a.gmax + (l - 1) : nullptr, mx, tid);
    ecur = scale_exp(mx);
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    __syncthreads();
    if (a.gz16) tile_copy16<W, TILE>(Ph, ecur, a.gz16 + (size_t)(l - 1) * M * W, a.gzexp + (size_t)(l - 1) * gridDim.x, m0, M, tid);
  }
  STAMP(3);  // D-1 trunk layers
  if (!a.need_dxyz) {
    STAMP_FLUSH_AT(8);
    return;
  }
  // ---- d x0 (first layer + skip) -> d xyz through the encoding
  {
    f32x16 acc0[TX::MT][TX::NT];
    acc_zero(acc0);
    mma16_lds<NP, W, W / 16, AH>(acc0, Ph, Pl, xrow0, 0, PT16 + 4 * (size_t)L.t_w[0], W / 16, xn0, 0, lane);
    const float un = pow2f(-(ecur + wexp[0]));
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[0][0][r] = fmaf(acc0[0][0][r], un, accx[0][0][r]);
  }
  static_assert(TX::MT == 1 && TX::NT == 1, "d x0 tiling");
  __syncthreads();
  float* Gs = (float*)planes;  // fp32 [TILE][64] scratch over the (now dead) planes
  acc_to_lds_t(accx, Gs, UPNERF_X0, xrow0, xn0, lane);
  __syncthreads();
  for (int it = tid; it < TILE * 3; it += F16_THREADS) {
    const int row = it / 3, n = it - row * 3, m = m0 + row;
    if (m >= M) continue;
    const float* __restrict__ x0 = a.x0 + (size_t)m * UPNERF_X0 + 3 + 20 * n;
    float xs[10], xc[10];  // all 20 loads in flight at once (two per trip left ten dependent L2 round trips)
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      xs[k] = x0[k];
      xc[k] = x0[10 + k];
    }
    float g = Gs[swz(row, n, UPNERF_X0)];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const float f = ldexpf(PI_F, k);
      g += f * (xc[k] * Gs[swz(row, 3 + 20 * n + k, UPNERF_X0)] - xs[k] * Gs[swz(row, 3 + 20 * n + 10 + k, UPNERF_X0)]);
    }
    a.dxyz[(size_t)m * 3 + n] = g;
  }
  STAMP(4);  // d x0 -> d xyz
  STAMP_FLUSH_AT(8);
}

int check_layout16(const upnerf_layout* L) {
  if (!L) return UPNERF_EINVAL;
  if (L->W != 256) return UPNERF_EUNSUP;
  if (L->D < 1 || L->D > UPNERF_MAX_D) return UPNERF_EUNSUP;
  if (L->skip >= L->D) return UPNERF_EINVAL;
  return 0;
}

}  // namespace

#ifdef UPNERF_STAMPS
extern "C" int upnerf_stamps_read(unsigned long long* out16, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(upnerf_stamp_acc), 16 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(upnerf_stamp_acc), z, sizeof(z)));
  }
  return 0;
}
#endif

int upnerf_field16r_fwd_launch(const upnerf_layout* L, const upnerf_field_fwd_args* a, void* stream);  // field16r.hip

extern "C" int upnerf_field_fwd_f16x3(const upnerf_layout* L, const upnerf_field_fwd_args* a, void* stream) {
  int rc = check_layout16(L);
  if (rc) return rc;
  if (!a || a->R <= 0 || a->S <= 0 || !a->rays_o || !a->rays_d || !a->z || !a->P || !a->P16 || !a->wexp || !a->x0 ||
      !a->sigma_s)
    return UPNERF_EINVAL;
  if (a->use_cand && (!a->c_rows || !a->sigma_c)) return UPNERF_EINVAL;
  if (a->use_rgb && (!a->aux || !a->rgb)) return UPNERF_EINVAL;
  if (a->planes != 0 && a->planes != 1 && a->planes != 2) return UPNERF_EINVAL;
  const long long M = (long long)a->R * a->S;
  if (M > 0x7fffffffLL) return UPNERF_EINVAL;
  if (a->h16 && (!a->hexp || a->wnorm)) return UPNERF_EINVAL;  // fp16 storage: LDS-tile kernel only
  if (a->wnorm) return upnerf_field16r_fwd_launch(L, a, stream);  // register-resident kernel (field16r.hip)
  const int grid = (int)((M + F16_TILE - 1) / F16_TILE);
  if (a->planes == 1)
    hipLaunchKernelGGL((field16_fwd_kernel<1, F16_TILE>), dim3(grid), dim3(F16_THREADS), 0, (hipStream_t)stream, *L, *a);
  else
    hipLaunchKernelGGL((field16_fwd_kernel<2, F16_TILE>), dim3(grid), dim3(F16_THREADS), 0, (hipStream_t)stream, *L, *a);
  return (int)hipGetLastError();
}

extern "C" int upnerf_field_bwd_f16x3(const upnerf_layout* L, const upnerf_field_bwd_args* a, void* stream) {
  int rc = check_layout16(L);
  if (rc) return rc;
  if (!a || a->R <= 0 || a->S <= 0 || !a->P || !a->PT16 || !a->wexp || !a->d_sigma_s || !a->sigma_s || (!a->gz_h && !a->gz16) ||
      !a->gz_e || !a->dpre_sig_s || !a->hmask)
    return UPNERF_EINVAL;
  if (a->S < 32) return UPNERF_EUNSUP;  // at most 3 rays per 64-sample tile
  if (a->use_cand && (!a->d_sigma_c || !a->sigma_c || !a->g2 || !a->gz_g1 || !a->gz_g2 || !a->dpre_sig_c))
    return UPNERF_EINVAL;
  if (a->use_cand && a->g_G_c && !a->w_cj) return UPNERF_EINVAL;
  if (a->use_rgb && (!a->d_rgb || !a->rgb || !a->r1 || !a->gz_r1 || !a->dpre_rgb)) return UPNERF_EINVAL;
  if (a->g_E_s && !a->w_feat_s) return UPNERF_EINVAL;
  if (a->need_dxyz && (!a->dxyz || !a->x0)) return UPNERF_EINVAL;
  if (a->planes != 0 && a->planes != 1 && a->planes != 2) return UPNERF_EINVAL;
  if (a->gz16 && !a->gzexp) return UPNERF_EINVAL;
  const long long M = (long long)a->R * a->S;
  if (M > 0x7fffffffLL) return UPNERF_EINVAL;
  const int grid = (int)((M + F16_TILE - 1) / F16_TILE);
  if (a->planes == 1)
    hipLaunchKernelGGL((field16_bwd_kernel<1, F16_TILE>), dim3(grid), dim3(F16_THREADS), 0, (hipStream_t)stream, *L, *a);
  else
    hipLaunchKernelGGL((field16_bwd_kernel<2, F16_TILE>), dim3(grid), dim3(F16_THREADS), 0, (hipStream_t)stream, *L, *a);
  return (int)hipGetLastError();
}
