// Alpha compositing along rays for gfx950 (models/rendering.py:125-218), forward and backward.
//
// One 64-lane wavefront owns one ray.  Samples are processed in chunks of 64 (one sample per lane): the
// exclusive transmittance product T_i = prod_{j<i} (1 - alpha_j) is a wave-level scan built from __shfl_up
// with a running carry between chunks, so any S works (64 coarse, 192 or 256 fine).  Feature maps are
// accumulated in the W-wide space of xyz_encoding_final (e) and W/2-wide candidate encoding (g2): for each
// sample the wave reads one contiguous row (16 bytes per lane) and scales it by the broadcast weight; the
// 384-wide projection happens once per ray on the host side of the boundary.
//
// Backward: division-free reverse scan of SURVEY Appendix A.3 -- with X_j = sum of (weight_j * upstream_j)
//   d sigma_s,i = delta_i [ e_s,i T_i Gs_i + e_a,i T_i Gw_i - sum_{j>i} X_j ]  (+ the shared-only field's term)
// where e_* = exp(-delta sigma) so that no (1 - alpha) ever appears in a denominator.
#include "common.cuh"
#pragma clang fp contract(off)

// fp32 rows of e / g2 a wave (= a ray) requests before it uses the first, in the kernels that read them as rows (f16x3 / fp32
// field modes).  Backward: each row ends in a wave-wide sum, and with one row in flight the launch sat at 4.9 TB/s; four rows:
// 177 -> 157 us per launch (round 6, same box; eight: 204 us).  Forward: 145 us at 5.8 TB/s with one row, unchanged with four
// or eight -- left at one.  Summation order per sample is the same for every value: bitwise-identical results.
#ifndef COMPOSITE_ROWS_FWD
#define COMPOSITE_ROWS_FWD 1
#endif
#ifndef COMPOSITE_ROWS_BWD
#define COMPOSITE_ROWS_BWD 4
#endif

namespace {

__device__ __forceinline__ float excl_prod_scan(float x, int lane, float& total) {
  // inclusive Hillis-Steele product over the wave, then shift by one lane
  float v = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(v, d);
    if (lane >= d) v *= t;
  }
  total = __shfl(v, 63);
  const float e = __shfl_up(v, 1);
  return lane == 0 ? 1.0f : e;
}

__device__ __forceinline__ float excl_suffix_sum(float x, int lane, float& total) {
  float v = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_down(v, d);
    if (lane + d < 64) v += t;
  }
  total = __shfl(v, 0);
  const float e = __shfl_down(v, 1);
  return lane == 63 ? 0.0f : e;
}

__device__ __forceinline__ double excl_prod_scan_d(double x, int lane, double& total) {
  double v = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double t = __shfl_up(v, d);
    if (lane >= d) v *= t;
  }
  total = __shfl(v, 63);
  const double e = __shfl_up(v, 1);
  return lane == 0 ? 1.0 : e;
}

__device__ __forceinline__ double excl_suffix_sum_d(double x, int lane, double& total) {
  double v = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double t = __shfl_down(v, d);
    if (lane + d < 64) v += t;
  }
  total = __shfl(v, 0);
  const double e = __shfl_down(v, 1);
  return lane == 63 ? 0.0 : e;
}

struct Alphas {
  float delta, es, ec, ea;  // exp(-delta*sigma_s), exp(-delta*sigma_c), exp(-delta*(sigma_s+sigma_c))
  float a_s, a_c, a_all;    // 1 - exp(...)
};

__device__ __forceinline__ Alphas alphas_at(const float* __restrict__ z, const float* __restrict__ sig_s,
                                            const float* __restrict__ sig_c, size_t base, int i, int S, bool joint) {
  Alphas A;
  A.delta = (i == S - 1) ? 1e2f : z[base + i + 1] - z[base + i];  // rendering.py:125-129
  const float ss = sig_s[base + i];
  A.es = expf(-A.delta * ss);
  A.a_s = 1.0f - A.es;
  A.ec = 1.0f; A.ea = A.es; A.a_c = 0.0f; A.a_all = A.a_s;
  if (joint) {
    const float sc = sig_c[base + i];
    A.ec = expf(-A.delta * sc);
    A.a_c = 1.0f - A.ec;
    A.ea = expf(-A.delta * (ss + sc));
    A.a_all = 1.0f - A.ea;
  }
  return A;
}

// e [M][256] as the fp16 operand fragments of the register-resident field kernels (include/upnerf_hip.h, tile_rows = 256):
// [32-sample tile][k-block s 16][lane 64][8], feature 16 s + 8 (u / 4) + 4 (lane / 32) + u % 4, sample 32 tile + lane % 32, scaled
// by 2^eexp[tile].  A wave walks the tiles its ray touches: sixteen coalesced 1 KiB loads per tile, lane = (sample, feature half)
// as stored, so a sample's share of E_s accumulates in 128 per-lane registers (ONE cross-lane reduction per ray) and a sample's
// dot product with g_E_s closes inside two lanes -- no per-sample shuffles at all, half the bytes of the fp32 rows.
typedef _Float16 h8c __attribute__((ext_vector_type(8)));
#define E16_MAXS 1024  // = 64 * MAX_CHUNKS
// eight k-blocks (s0 .. s0 + 7) of tile t of a fragment-ordered tensor with `bpt` k-blocks per tile (16: 256 wide, 8: 128 wide)
__device__ __forceinline__ void frag_half(const uint16_t* __restrict__ x16, int bpt, int t, int s0, int lane, h8c (&x)[8]) {
  const h8c* __restrict__ p = (const h8c*)x16 + ((size_t)t * bpt + s0) * 64 + lane;
#pragma unroll
  for (int s = 0; s < 8; ++s) x[s] = __builtin_nontemporal_load(p + s * 64);
}
// out[16 s0 + 0 .. 127] = sum over the ray's samples of wl[sample] * x[sample][16 s0 + ..]  (wl: LDS, already scaled by 2^-exp[tile])
__device__ __forceinline__ void frag_weighted_sum(const uint16_t* __restrict__ x16, int bpt, int s0, const float* wl, size_t base, int S,
                                                  int lane, float* __restrict__ out) {
  const int t0 = (int)(base >> 5), t1 = (int)((base + S - 1) >> 5);
  float acc[8][8];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[s][u] = 0.0f;
  for (int t = t0; t <= t1; ++t) {
    const long long idx = (long long)t * 32 + (lane & 31) - (long long)base;
    const float w = (idx >= 0 && idx < S) ? wl[idx] : 0.0f;
    h8c x[8];
    frag_half(x16, bpt, t, s0, lane, x);
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[s][u] = fmaf(w, (float)x[s][u], acc[s][u]);
  }
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float v = acc[s][u];
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) v += __shfl_xor(v, d);
      acc[s][u] = v;
    }
  if ((lane & 31) == 0) {
    float* __restrict__ dst = out + 4 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      *(f32x4*)(dst + 16 * s) = f32x4{acc[s][0], acc[s][1], acc[s][2], acc[s][3]};
      *(f32x4*)(dst + 16 * s + 8) = f32x4{acc[s][4], acc[s][5], acc[s][6], acc[s][7]};
    }
  }
}
// dl[sample] (=, or += when `add`) <g[16 s0 + 0 .. 127], x[sample][16 s0 + ..]> * 2^-xexp[tile], for every sample of the ray
__device__ __forceinline__ void frag_dots(const uint16_t* __restrict__ x16, const int* __restrict__ xexp, int bpt, int s0,
                                          const float* __restrict__ g, size_t base, int S, int lane, float* dl, bool add) {
  const int t0 = (int)(base >> 5), t1 = (int)((base + S - 1) >> 5);
  // this lane's 64 entries of g for these k-blocks, in the order its fragment pieces hold the features
  f32x4 ga[8], gb[8];
  const float* __restrict__ gsrc = g + 4 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    ga[s] = *(const f32x4*)(gsrc + 16 * s);
    gb[s] = *(const f32x4*)(gsrc + 16 * s + 8);
  }
  for (int t = t0; t <= t1; ++t) {
    h8c x[8];
    frag_half(x16, bpt, t, s0, lane, x);
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      p0 += ga[s].x * (float)x[s][0] + ga[s].y * (float)x[s][1] + ga[s].z * (float)x[s][2] + ga[s].w * (float)x[s][3];
      p1 += gb[s].x * (float)x[s][4] + gb[s].y * (float)x[s][5] + gb[s].z * (float)x[s][6] + gb[s].w * (float)x[s][7];
    }
    float pe = p0 + p1;
    pe += __shfl_xor(pe, 32);
    const long long idx = (long long)t * 32 + (lane & 31) - (long long)base;
    if (lane < 32 && idx >= 0 && idx < S) {
      pe *= ldexpf(1.0f, -xexp[t]);
      dl[idx] = add ? dl[idx] + pe : pe;
    }
  }
}

template <int W, bool EFRAG>
__global__ __launch_bounds__(NTHREADS) void composite_fwd_kernel(upnerf_composite_fwd_args a) {
  constexpr int W2 = W / 2;
  __shared__ float wf_lds[EFRAG ? 4 : 1][EFRAG ? E16_MAXS : 1];  // (e16: the feature weights of the ray, scaled per tile)
  __shared__ float wc_lds[EFRAG ? 4 : 1][EFRAG ? E16_MAXS : 1];  // (g2_16: the candidate weights, likewise)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= a.R) return;
  const int S = a.S;
  const size_t base = (size_t)r * S;
  const bool joint = a.mode <= 1;
  const bool feat_from_ws = a.mode == 3;       // feature map weighted by the shared-only weights
  const bool want_feat = a.mode != 2;
  float carryT = 1.0f, carryTs = 1.0f;
  float acc_cd = 0.f, acc_sd = 0.f, acc_tw = 0.f, acc_sf = 0.f, acc_rgb[3] = {0.f, 0.f, 0.f}, acc_rgbj[3] = {0.f, 0.f, 0.f};
  const bool rgb_joint = joint && a.has_rgb && a.rgb_joint_map;  // encode_feat = False: the shared half of c_rgb
  f32x4 accE = {0.f, 0.f, 0.f, 0.f}, accG = {0.f, 0.f, 0.f, 0.f};
  const bool laneE = lane < W / 4, laneG = lane < W2 / 4;
  for (int c0 = 0; c0 < S; c0 += 64) {
    const int i = c0 + lane;
    const bool valid = i < S;
    float zi = 0.f, w_all = 0.f, w_sj = 0.f, w_cj = 0.f, w_s = 0.f;
    Alphas A;
    A.a_s = A.a_c = A.a_all = 0.f;
    if (valid) {
      A = alphas_at(a.z, a.sigma_s, a.sigma_c, base, i, S, joint);
      zi = a.z[base + i];
    }
    float tot;
    if (joint) {
      // 1 - alpha is formed from alpha (not from the exponential) like the reference (rendering.py:156-161)
      const float T = carryT * excl_prod_scan(valid ? 1.0f - A.a_all : 1.0f, lane, tot);
      carryT *= tot;
      w_all = A.a_all * T; w_sj = A.a_s * T; w_cj = A.a_c * T;
      if (valid) {
        a.w_all[base + i] = w_all; a.w_sj[base + i] = w_sj; a.w_cj[base + i] = w_cj;
      }
      acc_cd += w_all * zi;
      acc_tw += w_cj;
      if (rgb_joint && valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc_rgbj[c] += w_sj * a.rgb[(base + i) * 3 + c];
      }
    }
    const float Ts = carryTs * excl_prod_scan(valid ? 1.0f - A.a_s : 1.0f, lane, tot);
    carryTs *= tot;
    w_s = A.a_s * Ts;
    if (valid) a.w_s[base + i] = w_s;
    acc_sd += w_s * zi;
    if (a.has_rgb && valid) {
#pragma unroll
      for (int c = 0; c < 3; ++c) acc_rgb[c] += w_s * a.rgb[(base + i) * 3 + c];
    }
    if (want_feat) {
      const float wf = feat_from_ws ? w_s : w_sj;
      acc_sf += wf;
      const int nv = (S - c0) < 64 ? (S - c0) : 64;
      if constexpr (EFRAG) {
        if (valid) wf_lds[wave][i] = wf * ldexpf(1.0f, -a.eexp[(base + i) >> 5]);
        if (valid && joint && a.g2_16) wc_lds[wave][i] = w_cj * ldexpf(1.0f, -a.g2exp[(base + i) >> 5]);
      }
      const bool g_rows = joint && !(EFRAG && a.g2_16);  // g2 as fp32 rows: the per-sample loop below
      // four samples' rows are requested before the first is used (the compiler keeps ONE load in flight otherwise, and a wave
      // per ray then waits out a full memory latency per sample); same summation order as a plain loop
      constexpr int B = EFRAG ? 4 : COMPOSITE_ROWS_FWD;
      if (!EFRAG || g_rows)
      for (int j0 = 0; j0 < nv; j0 += B) {
        f32x4 ev[B], gv[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
          const size_t m = base + c0 + (j0 + u < nv ? j0 + u : nv - 1);
          if constexpr (!EFRAG) {
            if (laneE) ev[u] = NT_LOAD((const f32x4*)&a.e[m * W + 4 * lane]);
          }
          if (g_rows && laneG) gv[u] = NT_LOAD((const f32x4*)&a.g2[m * W2 + 4 * lane]);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
          if (j0 + u < nv) {
            const float wj = __shfl(wf, j0 + u);
            if constexpr (!EFRAG) {
              if (laneE) { accE.x += wj * ev[u].x; accE.y += wj * ev[u].y; accE.z += wj * ev[u].z; accE.w += wj * ev[u].w; }
            }
            if (g_rows) {
              const float cj = __shfl(w_cj, j0 + u);
              if (laneG) { accG.x += cj * gv[u].x; accG.y += cj * gv[u].y; accG.z += cj * gv[u].z; accG.w += cj * gv[u].w; }
            }
          }
        }
      }
    }
  }
  const float cd = wave_sum(acc_cd), sd = wave_sum(acc_sd), tw = wave_sum(acc_tw), sf = wave_sum(acc_sf);
  float rg[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) rg[c] = wave_sum(acc_rgb[c]);
  if (lane == 0) {
    a.s_depth[r] = sd;
    if (joint) { a.c_depth[r] = cd; a.t_weight[r] = tw; }
    if (want_feat) a.sum_sfeat[r] = sf;
    if (a.has_rgb) { a.rgb_map[r * 3] = rg[0]; a.rgb_map[r * 3 + 1] = rg[1]; a.rgb_map[r * 3 + 2] = rg[2]; }
  }
  if (rgb_joint) {  // (wave-uniform)
    float rj[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rj[c] = wave_sum(acc_rgbj[c]);
    if (lane == 0) { a.rgb_joint_map[r * 3] = rj[0]; a.rgb_joint_map[r * 3 + 1] = rj[1]; a.rgb_joint_map[r * 3 + 2] = rj[2]; }
  }
  if constexpr (EFRAG) {
    if (want_feat) {
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
      // passes over the ray's tiles, eight k-blocks each (every piece is read once): 64 accumulators at a time
#pragma unroll 1
      for (int hf = 0; hf < 2; ++hf) frag_weighted_sum(a.e16, 16, 8 * hf, wf_lds[wave], base, S, lane, a.E_s + (size_t)r * W + 128 * hf);
      if (joint && a.g2_16) frag_weighted_sum(a.g2_16, 8, 0, wc_lds[wave], base, S, lane, a.G_c + (size_t)r * W2);
    }
  }
  if (want_feat) {
    if (!EFRAG && laneE) *(f32x4*)&a.E_s[(size_t)r * W + 4 * lane] = accE;
    if (joint && laneG && !(EFRAG && a.g2_16)) *(f32x4*)&a.G_c[(size_t)r * W2 + 4 * lane] = accG;
  }
}

#define MAX_CHUNKS 16
template <int W, bool EFRAG>
__global__ __launch_bounds__(NTHREADS) void composite_bwd_kernel(upnerf_composite_bwd_args a) {
  constexpr int W2 = W / 2;
  __shared__ double carry_s[4][2][MAX_CHUNKS];
  __shared__ float dot_lds[EFRAG ? 4 : 1][EFRAG ? E16_MAXS : 1];  // (e16: <g_E_s, e_i> of every sample of the ray)
  __shared__ float dotg_lds[EFRAG ? 4 : 1][EFRAG ? E16_MAXS : 1];  // (g2_16: <g_G_c, g2_i>)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= a.R) return;
  const int S = a.S;
  const size_t base = (size_t)r * S;
  const bool joint = a.mode <= 1;
  const bool feat_from_ws = a.mode == 3;
  const bool want_feat = a.mode != 2;
  const bool laneE = lane < W / 4, laneG = lane < W2 / 4;
  const int nchunk = (S + 63) >> 6;
  // The bracket  (1-alpha_i) T_i G_i - sum_{j>i} w_j G_j  is a small residual of two nearly equal terms (it
  // telescopes to T_end G when G is constant along the ray), so transmittances and suffix sums of the backward are
  // carried in fp64: the result is then the exact gradient of the fp32 forward values, like the reference's CPU
  // autograd whose cumsum accumulates in double.  (This kernel is latency bound; the fp64 scans cost nothing.)
  {
    double cT = 1.0, cTs = 1.0;
    for (int c = 0; c < nchunk; ++c) {
      if (lane == 0) { carry_s[wave][0][c] = cT; carry_s[wave][1][c] = cTs; }
      const int i = c * 64 + lane;
      double om = 1.0, oms = 1.0;
      if (i < S) {
        const Alphas A = alphas_at(a.z, a.sigma_s, a.sigma_c, base, i, S, joint);
        om = (double)(1.0f - A.a_all); oms = (double)(1.0f - A.a_s);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { om *= __shfl_xor(om, d); oms *= __shfl_xor(oms, d); }
      cT *= om; cTs *= oms;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  f32x4 gE = {0.f, 0.f, 0.f, 0.f}, gG = {0.f, 0.f, 0.f, 0.f};
  if (!EFRAG && want_feat && a.g_E_s && laneE) gE = *(const f32x4*)&a.g_E_s[(size_t)r * W + 4 * lane];
  if (joint && a.g_G_c && laneG) gG = *(const f32x4*)&a.g_G_c[(size_t)r * W2 + 4 * lane];
  const float g_sf = (want_feat && a.g_sum_sfeat) ? a.g_sum_sfeat[r] : 0.f;
  const float g_tw = (joint && a.g_t_weight) ? a.g_t_weight[r] : 0.f;
  const float g_cd = (joint && a.g_c_depth) ? a.g_c_depth[r] : 0.f;
  const float g_sd = a.g_s_depth ? a.g_s_depth[r] : 0.f;
  float g_rm[3] = {0.f, 0.f, 0.f};
  if (a.has_rgb && a.g_rgb_map) { g_rm[0] = a.g_rgb_map[r * 3]; g_rm[1] = a.g_rgb_map[r * 3 + 1]; g_rm[2] = a.g_rgb_map[r * 3 + 2]; }
  float g_rj[3] = {0.f, 0.f, 0.f};  // encode_feat = False: gradient of the joint-weight colour map (the shared half of c_rgb)
  const bool rgb_joint = joint && a.has_rgb && a.g_rgb_joint_map;
  if (rgb_joint) { g_rj[0] = a.g_rgb_joint_map[r * 3]; g_rj[1] = a.g_rgb_joint_map[r * 3 + 1]; g_rj[2] = a.g_rgb_joint_map[r * 3 + 2]; }
  const bool need_dots = want_feat && (a.g_E_s || (joint && a.g_G_c));
  const bool g_frag = EFRAG && joint && a.g_G_c && a.g2_16;
  if constexpr (EFRAG) {
    if (want_feat) {
      if (a.g_E_s) {
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf)
          frag_dots(a.e16, a.eexp, 16, 8 * hf, a.g_E_s + (size_t)r * W + 128 * hf, base, S, lane, dot_lds[wave], hf != 0);
      }
      if (g_frag) frag_dots(a.g2_16, a.g2exp, 8, 0, a.g_G_c + (size_t)r * W2, base, S, lane, dotg_lds[wave], false);
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
    }
  }

  double sufX = 0.0, sufY = 0.0;  // suffix sums over later chunks
  for (int c = nchunk - 1; c >= 0; --c) {
    const int c0 = c * 64, i = c0 + lane;
    const bool valid = i < S;
    // <g_E_s, e_i> and <g_G_c, g2_i> for the 64 samples of the chunk: lane j keeps sample j's result
    float dotE = 0.f, dotG = 0.f;
    if (need_dots) {
      const int nv = (S - c0) < 64 ? (S - c0) : 64;
      constexpr int B = EFRAG ? 4 : COMPOSITE_ROWS_BWD;  // (rows in flight: see COMPOSITE_ROWS_* above)
      const bool g_rows = joint && a.g_G_c && !g_frag;
      if (!EFRAG || g_rows)
      for (int j0 = 0; j0 < nv; j0 += B) {
        f32x4 ev[B], gv[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
          const size_t m = base + c0 + (j0 + u < nv ? j0 + u : nv - 1);
          if constexpr (!EFRAG) {
            if (laneE) ev[u] = NT_LOAD((const f32x4*)&a.e[m * W + 4 * lane]);
          }
          if (g_rows && laneG) gv[u] = NT_LOAD((const f32x4*)&a.g2[m * W2 + 4 * lane]);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
          if (j0 + u < nv) {
            float pe = 0.f, pg = 0.f;
            if constexpr (!EFRAG) {
              if (laneE) pe = gE.x * ev[u].x + gE.y * ev[u].y + gE.z * ev[u].z + gE.w * ev[u].w;
              pe = wave_sum(pe);
            }
            if (g_rows) {
              if (laneG) pg = gG.x * gv[u].x + gG.y * gv[u].y + gG.z * gv[u].z + gG.w * gv[u].w;
              pg = wave_sum(pg);
            }
            if (lane == j0 + u) { dotE = pe; dotG = pg; }
          }
        }
      }
      if constexpr (EFRAG) {
        if (valid && g_frag) dotG = dotg_lds[wave][i];
        if (valid && a.g_E_s) dotE = dot_lds[wave][i];
      }
    }
    Alphas A;
    A.delta = 0.f; A.es = A.ec = A.ea = 1.f; A.a_s = A.a_c = A.a_all = 0.f;
    float zi = 0.f;
    if (valid) {
      A = alphas_at(a.z, a.sigma_s, a.sigma_c, base, i, S, joint);
      zi = a.z[base + i];
    }
    double tot, ds = 0.0, dc = 0.0, w_sj_d = 0.0;
    if (joint) {
      const double om = valid ? (double)(1.0f - A.a_all) : 1.0;
      const double T = carry_s[wave][0][c] * excl_prod_scan_d(om, lane, tot);
      double Gs = (double)dotE + g_sf;
      const double Gc = (double)dotG + g_tw;
      if (rgb_joint && valid) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) Gs += (double)g_rj[ch] * (double)a.rgb[(base + i) * 3 + ch];
        w_sj_d = (double)A.a_s * T;
      }
      const double Gw = (double)g_cd * zi + ((a.g_w_all && valid) ? (double)a.g_w_all[base + i] : 0.0);
      const double X = valid ? ((double)A.a_s * Gs + (double)A.a_c * Gc + (double)A.a_all * Gw) * T : 0.0;
      const double suf = excl_suffix_sum_d(X, lane, tot) + sufX;
      sufX += tot;
      // d alpha_s / d sigma_s = delta e_s etc.; the Gw term uses (1 - alpha) = the factor T_{i+1} is built from,
      // so it telescopes against the suffix sum
      ds = (double)A.delta * ((double)A.es * T * Gs + om * T * Gw - suf);
      dc = (double)A.delta * ((double)A.ec * T * Gc + om * T * Gw - suf);
    }
    {
      const double oms = valid ? (double)(1.0f - A.a_s) : 1.0;
      const double Ts = carry_s[wave][1][c] * excl_prod_scan_d(oms, lane, tot);
      const double w_s = (double)A.a_s * Ts;
      double Gws = (double)g_sd * zi + ((a.g_w_s && valid) ? (double)a.g_w_s[base + i] : 0.0);
      if (a.has_rgb && valid) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) Gws += (double)g_rm[ch] * (double)a.rgb[(base + i) * 3 + ch];
      }
      if (feat_from_ws) Gws += (double)dotE + g_sf;
      const double Y = valid ? w_s * Gws : 0.0;
      const double suf = excl_suffix_sum_d(Y, lane, tot) + sufY;
      sufY += tot;
      ds += (double)A.delta * (oms * Ts * Gws - suf);
      if (a.has_rgb && valid) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) a.d_rgb[(base + i) * 3 + ch] = rgb_joint ? (float)(w_s * g_rm[ch] + w_sj_d * g_rj[ch]) : (float)w_s * g_rm[ch];
      }
    }
    if (valid) {
      a.d_sigma_s[base + i] = (float)ds;
      if (joint) a.d_sigma_c[base + i] = (float)dc;
    }
  }
}

int check_common(int R, int S, int W, int mode) {
  if (R <= 0 || S <= 0 || S > 64 * MAX_CHUNKS) return UPNERF_EINVAL;
  if (W != 64 && W != 256) return UPNERF_EUNSUP;
  if (mode < 0 || mode > 3) return UPNERF_EINVAL;
  return 0;
}

}  // namespace

extern "C" int upnerf_composite_fwd(const upnerf_composite_fwd_args* a, void* stream) {
  if (!a) return UPNERF_EINVAL;
  int rc = check_common(a->R, a->S, a->W, a->mode);
  if (rc) return rc;
  const bool joint = a->mode <= 1, want_feat = a->mode != 2;
  if (!a->z || !a->sigma_s || !a->w_s || !a->s_depth) return UPNERF_EINVAL;
  if (joint && (!a->sigma_c || !a->w_all || !a->w_sj || !a->w_cj || !a->c_depth || !a->t_weight || (!a->g2 && !a->g2_16) || !a->G_c))
    return UPNERF_EINVAL;
  if (want_feat && ((!a->e && !a->e16) || (a->e16 && (!a->eexp || a->W != 256)) || !a->E_s || !a->sum_sfeat)) return UPNERF_EINVAL;
  if (a->g2_16 && (!a->e16 || !a->g2exp)) return UPNERF_EINVAL;  // (the fragment walk lives in the e16 kernels)
  if (a->has_rgb && (!a->rgb || !a->rgb_map)) return UPNERF_EINVAL;
  const dim3 grid((a->R + 3) / 4), block(NTHREADS);
  if (a->W == 256 && a->e16 && want_feat)
    hipLaunchKernelGGL((composite_fwd_kernel<256, true>), grid, block, 0, (hipStream_t)stream, *a);
  else if (a->W == 256)
    hipLaunchKernelGGL((composite_fwd_kernel<256, false>), grid, block, 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL((composite_fwd_kernel<64, false>), grid, block, 0, (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}

extern "C" int upnerf_composite_bwd(const upnerf_composite_bwd_args* a, void* stream) {
  if (!a) return UPNERF_EINVAL;
  int rc = check_common(a->R, a->S, a->W, a->mode);
  if (rc) return rc;
  const bool joint = a->mode <= 1, want_feat = a->mode != 2;
  if (!a->z || !a->sigma_s || !a->d_sigma_s) return UPNERF_EINVAL;
  if (joint && (!a->sigma_c || !a->d_sigma_c)) return UPNERF_EINVAL;
  if (want_feat && a->g_E_s && ((!a->e && !a->e16) || (a->e16 && (!a->eexp || a->W != 256)))) return UPNERF_EINVAL;
  if (a->g2_16 && (!a->e16 || !a->g2exp)) return UPNERF_EINVAL;
  if (joint && a->g_G_c && !a->g2 && !a->g2_16) return UPNERF_EINVAL;
  if (a->has_rgb && (!a->rgb || !a->d_rgb)) return UPNERF_EINVAL;
  const dim3 grid((a->R + 3) / 4), block(NTHREADS);
  if (a->W == 256 && a->e16 && want_feat)
    hipLaunchKernelGGL((composite_bwd_kernel<256, true>), grid, block, 0, (hipStream_t)stream, *a);
  else if (a->W == 256)
    hipLaunchKernelGGL((composite_bwd_kernel<256, false>), grid, block, 0, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL((composite_bwd_kernel<64, false>), grid, block, 0, (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}
