// Depth-prior affine + UPNeRFLoss on per-ray maps, forward (8 reductions, fixed-order two-stage sum) and backward
// (elementwise).  Reference: models/nerf_system.py:169-177 and losses.py:21-64.  Everything here is O(R*F) with
// R = 4096 rays: launch-latency bound; the point of the kernel is one launch instead of ~40 ATen launches.
#include "common.cuh"

namespace {

#define LOSS_BLOCKS 64
enum { T_DEPTH_C = 0, T_FEAT_C, T_RGB_C, T_DEPTH_F, T_FEAT_F, T_RGB_F, T_BETA, T_ALPHA, T_N };

__device__ __forceinline__ float prior_depth(const upnerf_loss_args& a, int r, float* dscale, float* dshift) {
  // p = inv * exp(scale) + shift, clamped from below at 1/far; depth = 1/p clamped from below at near; the clamps are
  // masked assignments in the reference, i.e. zero gradient where they fire.
  if (a.depth_direct) {
    if (dscale) { *dscale = 0.f; *dshift = 0.f; }
    return a.depth_direct[r];
  }
  const float sc = a.depth_scale_rows[2 * r], sh = a.depth_scale_rows[2 * r + 1];
  const float es = expf(sc);
  float p = a.inv_depth[r] * es + sh;
  bool live = true;
  if (p < 1.0f / a.far) { p = 1.0f / a.far; live = false; }
  float d = 1.0f / p;
  if (d < a.near) { d = a.near; live = false; }
  if (dscale) {
    const float dd = live ? -1.0f / (p * p) : 0.0f;
    *dscale = dd * a.inv_depth[r] * es;
    *dshift = dd;
  }
  return d;
}

__global__ __launch_bounds__(NTHREADS) void loss_fwd_kernel(upnerf_loss_args a, float* __restrict__ depth_out,
                                                           float* __restrict__ part) {
  __shared__ float red[T_N][NTHREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (a.R + LOSS_BLOCKS - 1) / LOSS_BLOCKS;
  const int r0 = blockIdx.x * per, r1 = (r0 + per < a.R) ? r0 + per : a.R;
  float acc[T_N];
#pragma unroll
  for (int k = 0; k < T_N; ++k) acc[k] = 0.f;
  const bool p0 = a.sched < 1.0f, p1 = a.sched > 0.0f;
  for (int r = r0 + tid; r < r1; r += NTHREADS) {
    const float d = prior_depth(a, r, nullptr, nullptr);
    depth_out[r] = d;
    if (p0) {
      float l = fabsf(a.s_depth_c[r] - d);
      if (a.has_tw) l *= 1.0f - a.t_weight_c[r];
      acc[T_DEPTH_C] += l;
      if (a.fine) {
        float lf = fabsf(a.s_depth_f[r] - d);
        if (a.has_tw) lf *= 1.0f - a.t_weight_f[r];
        acc[T_DEPTH_F] += lf;
      }
    }
    if (p1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float e = a.rgb_c[r * 3 + c] - a.rgb_gt[r * 3 + c];
        acc[T_RGB_C] += e * e;
      }
      if (a.fine) {
        const float b = a.beta[r];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float e = a.rgb_f[r * 3 + c] - a.rgb_gt[r * 3 + c];
          acc[T_RGB_F] += e * e / (2.0f * b * b);
        }
        acc[T_BETA] += logf(b);
        acc[T_ALPHA] += a.alpha[r];
      }
    }
  }
  if (p0) {
    const int n = (r1 - r0) * a.F;
    const size_t base = (size_t)r0 * a.F;
    int i0 = 0;
    if ((a.F & 3) == 0) {
      // 16-byte loads, four pieces per thread in flight: with 64 workgroups this loop is a chain of L2 round trips (96 of them
      // per thread at 4 bytes and one in flight: 75 us for 19 MB)
      typedef float f4 __attribute__((ext_vector_type(4)));
      const int n4 = n >> 2;
      constexpr int U = 4;
      for (int i = tid; i < n4; i += U * NTHREADS) {
        f4 g[U], c[U], f[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = i + u * NTHREADS;
          const size_t off = base + 4 * (size_t)(j < n4 ? j : n4 - 1);
          g[u] = *(const f4*)&a.feat_gt[off];
          c[u] = *(const f4*)&a.feat_c[off];
          f[u] = a.fine ? *(const f4*)&a.feat_f[off] : g[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (i + u * NTHREADS < n4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float e = c[u][q] - g[u][q], ef = f[u][q] - g[u][q];
              acc[T_FEAT_C] += e * e;
              acc[T_FEAT_F] += ef * ef;  // zero without a fine pass
            }
          }
        }
      }
      i0 = n;
    }
    for (int i = i0 + tid; i < n; i += NTHREADS) {
      const size_t off = base + i;
      const float g = a.feat_gt[off];
      const float e = a.feat_c[off] - g;
      acc[T_FEAT_C] += e * e;
      if (a.fine) {
        const float ef = a.feat_f[off] - g;
        acc[T_FEAT_F] += ef * ef;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < T_N; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[k][wave] = s;
  }
  __syncthreads();
  if (tid < T_N) part[blockIdx.x * T_N + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

__global__ void loss_finish_kernel(upnerf_loss_args a, const float* __restrict__ part, float* __restrict__ terms) {
  const int k = threadIdx.x;
  if (k >= T_N) return;
  float s = 0.f;
  for (int b = 0; b < LOSS_BLOCKS; ++b) s += part[b * T_N + k];
  const float m = a.sched_dev ? *a.sched_dev : a.sched, R = (float)a.R;
  float scale = 0.f;
  switch (k) {
    case T_DEPTH_C: case T_DEPTH_F: scale = a.depth_mult * (1.f - m) / R; break;
    case T_FEAT_C: case T_FEAT_F: scale = (1.f - m) / (R * a.F); break;
    case T_RGB_C: scale = m / 2.f / (R * 3.f); break;
    case T_RGB_F: scale = m / (R * 3.f); break;
    case T_BETA: scale = m / R; break;
    case T_ALPHA: scale = a.alpha_reg * m / R; break;
  }
  const float t = s * scale;
  terms[k] = t;
  if (a.total) {  // (one wave: the masked terms are added in term order by lane 0)
    const float v = ((a.term_mask >> k) & 1) ? t : 0.f;
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < T_N; ++j) tot += __shfl(v, j);
    if (k == 0) a.total[0] = tot;
  }
}

__global__ __launch_bounds__(NTHREADS) void loss_bwd_kernel(upnerf_loss_args a, const float* __restrict__ gt_in,
                                                           upnerf_loss_grads g) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  float gt[T_N];  // upstream gradient of every term: the caller's vector + the total's gradient on the masked terms
  {
    const float gtot = a.g_total ? a.g_total[0] : 0.f;
#pragma unroll
    for (int k = 0; k < T_N; ++k) gt[k] = (gt_in ? gt_in[k] : 0.f) + (((a.term_mask >> k) & 1) ? gtot : 0.f);
  }
  const float m = a.sched_dev ? *a.sched_dev : a.sched, R = (float)a.R;
  const bool p0 = a.sched < 1.0f, p1 = a.sched > 0.0f;  // the phase is static; only the multiplier follows the step
  if (idx < a.R) {
    const int r = idx;
    float dsc, dsh;
    const float d = prior_depth(a, r, &dsc, &dsh);
    float gd = 0.f;  // gradient w.r.t. the depth target
    if (p0) {
      const float k = a.depth_mult * (1.f - m) / R;
      const float ec = a.s_depth_c[r] - d;
      const float wc = (a.has_tw ? 1.0f - a.t_weight_c[r] : 1.0f) * k * gt[T_DEPTH_C];
      const float sc = ec > 0.f ? 1.f : (ec < 0.f ? -1.f : 0.f);
      if (g.d_s_depth_c) g.d_s_depth_c[r] = sc * wc;
      gd -= sc * wc;
      if (a.fine) {
        const float ef = a.s_depth_f[r] - d;
        const float wf = (a.has_tw ? 1.0f - a.t_weight_f[r] : 1.0f) * k * gt[T_DEPTH_F];
        const float sf = ef > 0.f ? 1.f : (ef < 0.f ? -1.f : 0.f);
        if (g.d_s_depth_f) g.d_s_depth_f[r] = sf * wf;
        gd -= sf * wf;
      }
    } else {
      if (g.d_s_depth_c) g.d_s_depth_c[r] = 0.f;
      if (g.d_s_depth_f) g.d_s_depth_f[r] = 0.f;
    }
    if (g.d_depth) g.d_depth[r] = gd;
    if (g.d_depth_scale_rows) {
      g.d_depth_scale_rows[2 * r] = gd * dsc;
      g.d_depth_scale_rows[2 * r + 1] = gd * dsh;
    }
    if (p1) {
      const float kc = m / 2.f / (R * 3.f) * gt[T_RGB_C];
      float dbeta = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float t = a.rgb_gt[r * 3 + c];
        if (g.d_rgb_c) g.d_rgb_c[r * 3 + c] = 2.f * (a.rgb_c[r * 3 + c] - t) * kc;
        if (a.fine) {
          const float b = a.beta[r], e = a.rgb_f[r * 3 + c] - t;
          const float kf = m / (R * 3.f) * gt[T_RGB_F];
          if (g.d_rgb_f) g.d_rgb_f[r * 3 + c] = e / (b * b) * kf;
          dbeta += -e * e / (b * b * b) * kf;
        }
      }
      if (a.fine) {
        if (g.d_beta) g.d_beta[r] = dbeta + gt[T_BETA] * m / R / a.beta[r];
        if (g.d_alpha) g.d_alpha[r] = gt[T_ALPHA] * a.alpha_reg * m / R;
      }
    }
  }
  if (p0) {
    const float kf = 2.f * (1.f - m) / (R * a.F);
    const long long n = (long long)a.R * a.F;
    for (long long i = idx; i < n; i += (long long)gridDim.x * blockDim.x) {
      const float t = a.feat_gt[i];
      if (g.d_feat_c) g.d_feat_c[i] = (a.feat_c[i] - t) * kf * gt[T_FEAT_C];
      if (a.fine && g.d_feat_f) g.d_feat_f[i] = (a.feat_f[i] - t) * kf * gt[T_FEAT_F];
    }
  }
}

int check(const upnerf_loss_args* a) {
  if (!a || a->R <= 0) return UPNERF_EINVAL;
  if (!a->depth_direct && (!a->inv_depth || !a->depth_scale_rows)) return UPNERF_EINVAL;
  const bool p0 = a->sched < 1.f, p1 = a->sched > 0.f;
  if (p0 && (!a->s_depth_c || !a->feat_c || !a->feat_gt || a->F <= 0)) return UPNERF_EINVAL;
  if (p0 && a->fine && (!a->s_depth_f || !a->feat_f)) return UPNERF_EINVAL;
  if (p0 && a->has_tw && (!a->t_weight_c || (a->fine && !a->t_weight_f))) return UPNERF_EINVAL;
  if (p1 && (!a->rgb_c || !a->rgb_gt)) return UPNERF_EINVAL;
  if (p1 && a->fine && (!a->rgb_f || !a->beta || !a->alpha)) return UPNERF_EINVAL;
  return 0;
}

}  // namespace

extern "C" int upnerf_loss_fwd(const upnerf_loss_args* a, float* depth_out, float* terms, float* scratch, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!depth_out || !terms || !scratch) return UPNERF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(LOSS_BLOCKS), dim3(NTHREADS), 0, st, *a, depth_out, scratch);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, *a, scratch, terms);
  return (int)hipGetLastError();
}

extern "C" int upnerf_loss_bwd(const upnerf_loss_args* a, const float* g_terms, const upnerf_loss_grads* g, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if ((!g_terms && !a->g_total) || !g) return UPNERF_EINVAL;
  const long long n = (long long)a->R * (a->sched < 1.f ? a->F : 1);
  int blocks = (int)((n + NTHREADS - 1) / NTHREADS);
  if (blocks > 2048) blocks = 2048;
  const int minb = (a->R + NTHREADS - 1) / NTHREADS;
  if (blocks < minb) blocks = minb;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(blocks), dim3(NTHREADS), 0, (hipStream_t)stream, *a, g_terms, *g);
  return (int)hipGetLastError();
}
