// Per-ray kernels of the UP-NeRF hot path on gfx950: SE(3) pose refinement + ray generation (forward and
// analytic backward), stratified coarse depths, inverse-CDF resampling, per-row sort, per-ray side inputs and
// the per-ray reductions of the backward pass.  All of them are HBM/latency bound and tiny next to the MLP;
// they exist so that the path has no eager ATen launches and no host round trips.
//
// Floating-point contraction is disabled in this file: the reference evaluates these expressions as separate
// fp32 ATen ops (one rounding per op), and the sampled depths feed sin(2^9 pi x), so a fused multiply-add here
// would show up as ~1e-4 differences in the highest encoding band.
#include "common.cuh"
#pragma clang fp contract(off)

namespace {

// ---- a2: Taylor coefficients of sin(x)/x, (1-cos x)/x^2, (x-sin x)/x^3 as polynomials in q = |w|^2
//      (utils/camera.py:126-152, nth=10).  c_i = (-1)^i / (2i+1)!, / (2i+2)!, / (2i+3)!
struct Taylor {
  float A, B, C, dA, dB, dC;  // values and derivatives with respect to q
};

__device__ __forceinline__ Taylor taylor_abc(float q) {
  double fa = 1.0, fb = 2.0, fc = 6.0;  // (2i+1)!, (2i+2)!, (2i+3)! at i = 0
  float ca[11], cb[11], cc[11];
#pragma unroll
  for (int i = 0; i <= 10; ++i) {
    if (i > 0) {
      fa *= (double)(2 * i) * (2 * i + 1);
      fb *= (double)(2 * i + 1) * (2 * i + 2);
      fc *= (double)(2 * i + 2) * (2 * i + 3);
    }
    const double sgn = (i & 1) ? -1.0 : 1.0;
    ca[i] = (float)(sgn / fa);
    cb[i] = (float)(sgn / fb);
    cc[i] = (float)(sgn / fc);
  }
  Taylor t;
  t.A = ca[10]; t.B = cb[10]; t.C = cc[10];
  t.dA = 10.f * ca[10]; t.dB = 10.f * cb[10]; t.dC = 10.f * cc[10];
#pragma unroll
  for (int i = 9; i >= 0; --i) {
    t.A = t.A * q + ca[i];
    t.B = t.B * q + cb[i];
    t.C = t.C * q + cc[i];
    if (i >= 1) {
      t.dA = t.dA * q + (float)i * ca[i];
      t.dB = t.dB * q + (float)i * cb[i];
      t.dC = t.dC * q + (float)i * cc[i];
    }
  }
  return t;
}

__device__ __forceinline__ void hat3(const float w[3], float K[3][3]) {
  K[0][0] = 0.f;   K[0][1] = -w[2]; K[0][2] = w[1];
  K[1][0] = w[2];  K[1][1] = 0.f;   K[1][2] = -w[0];
  K[2][0] = -w[1]; K[2][1] = w[0];  K[2][2] = 0.f;
}

__device__ __forceinline__ void mat3mul(const float a[3][3], const float b[3][3], float c[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

// refinement pose [R | V u] from one se(3) row (camera.py:87-98)
__device__ __forceinline__ void se3_exp_dev(const float* wu, float Rm[3][3], float Vm[3][3], float t[3], float K[3][3],
                                            float K2[3][3], Taylor& ty) {
  const float w[3] = {wu[0], wu[1], wu[2]};
  const float q = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  ty = taylor_abc(q);
  hat3(w, K);
  mat3mul(K, K, K2);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float id = (i == j) ? 1.f : 0.f;
      Rm[i][j] = id + ty.A * K[i][j] + ty.B * K2[i][j];
      Vm[i][j] = id + ty.B * K[i][j] + ty.C * K2[i][j];
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = Vm[i][0] * wu[3] + Vm[i][1] * wu[4] + Vm[i][2] * wu[5];
}

__global__ void pose_rays_fwd_kernel(int R, const float* __restrict__ se3, const float* __restrict__ c2w,
                                     const float* __restrict__ dirs, float* __restrict__ rays_o,
                                     float* __restrict__ rays_d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float Rc[3][3], tc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Rc[i][j] = c2w[r * 12 + i * 4 + j];
    tc[i] = c2w[r * 12 + i * 4 + 3];
  }
  float Rn[3][3], tn[3];
  if (se3) {
    float Rm[3][3], Vm[3][3], t[3], K[3][3], K2[3][3];
    Taylor ty;
    se3_exp_dev(se3 + r * 6, Rm, Vm, t, K, K2, ty);
    mat3mul(Rc, Rm, Rn);  // compose([refine, c2w]) = c2w o refine (camera.py:51-58, nerf_system.py:160)
#pragma unroll
    for (int i = 0; i < 3; ++i) tn[i] = (Rc[i][0] * t[0] + Rc[i][1] * t[1] + Rc[i][2] * t[2]) + tc[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Rn[i][j] = Rc[i][j];
      tn[i] = tc[i];
    }
  }
  // get_rays (ray.py:44-56): d = R dir / |R dir|, o = t
  const float dx = dirs[r * 3], dy = dirs[r * 3 + 1], dz = dirs[r * 3 + 2];
  float v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = dx * Rn[i][0] + dy * Rn[i][1] + dz * Rn[i][2];
  const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rays_d[r * 3 + i] = v[i] / nrm;
    rays_o[r * 3 + i] = tn[i];
  }
}

__global__ void pose_rays_bwd_kernel(int R, const float* __restrict__ se3, const float* __restrict__ c2w,
                                     const float* __restrict__ dirs, const float* __restrict__ g_o,
                                     const float* __restrict__ g_d, float* __restrict__ g_se3) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float Rc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Rc[i][j] = c2w[r * 12 + i * 4 + j];
  float Rm[3][3], Vm[3][3], t[3], K[3][3], K2[3][3];
  Taylor ty;
  const float* wu = se3 + r * 6;
  se3_exp_dev(wu, Rm, Vm, t, K, K2, ty);
  float Rn[3][3];
  mat3mul(Rc, Rm, Rn);
  const float dir[3] = {dirs[r * 3], dirs[r * 3 + 1], dirs[r * 3 + 2]};
  float v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = dir[0] * Rn[i][0] + dir[1] * Rn[i][1] + dir[2] * Rn[i][2];
  const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float d[3] = {v[0] / nrm, v[1] / nrm, v[2] / nrm};
  const float gd[3] = {g_d[r * 3], g_d[r * 3 + 1], g_d[r * 3 + 2]};
  const float go[3] = {g_o[r * 3], g_o[r * 3 + 1], g_o[r * 3 + 2]};
  const float dg = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];
  float gv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) gv[i] = (gd[i] - d[i] * dg) / nrm;
  // Rn = Rc Rm, v = Rn dir:  g_Rm = Rc^T (gv dir^T);  tn = Rc t + tc: g_t = Rc^T go
  float gR[3][3], gt[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float rg = Rc[0][i] * gv[0] + Rc[1][i] * gv[1] + Rc[2][i] * gv[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) gR[i][j] = rg * dir[j];
    gt[i] = Rc[0][i] * go[0] + Rc[1][i] * go[1] + Rc[2][i] * go[2];
  }
  // t = V u
  const float u[3] = {wu[3], wu[4], wu[5]};
  float gu[3], gV[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gu[i] = Vm[0][i] * gt[0] + Vm[1][i] * gt[1] + Vm[2][i] * gt[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) gV[i][j] = gt[i] * u[j];
  }
  // R = I + A K + B K2, V = I + B K + C K2
  float gA = 0.f, gB = 0.f, gC = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      gA += gR[i][j] * K[i][j];
      gB += gR[i][j] * K2[i][j] + gV[i][j] * K[i][j];
      gC += gV[i][j] * K2[i][j];
    }
  // gradient reaching K directly and through K2 = K K:  g_K += G K^T + K^T G with G = B gR + C gV
  float G[3][3], gK[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) G[i][j] = ty.B * gR[i][j] + ty.C * gV[i][j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float s = ty.A * gR[i][j] + ty.B * gV[i][j];
#pragma unroll
      for (int k = 0; k < 3; ++k) s += G[i][k] * K[j][k] + K[k][i] * G[k][j];
      gK[i][j] = s;
    }
  const float w[3] = {wu[0], wu[1], wu[2]};
  const float gq = 2.f * (gA * ty.dA + gB * ty.dB + gC * ty.dC);
  g_se3[r * 6 + 0] = (gK[2][1] - gK[1][2]) + gq * w[0];
  g_se3[r * 6 + 1] = (gK[0][2] - gK[2][0]) + gq * w[1];
  g_se3[r * 6 + 2] = (gK[1][0] - gK[0][1]) + gq * w[2];
  g_se3[r * 6 + 3] = gu[0];
  g_se3[r * 6 + 4] = gu[1];
  g_se3[r * 6 + 5] = gu[2];
}

// ---- a5: stratified depths (rendering.py:232-249)
__device__ __forceinline__ float base_z(float near, float far, float s, int use_disp) {
  if (!use_disp) return near * (1.f - s) + far * s;
  return 1.f / (1.f / near * (1.f - s) + 1.f / far * s);
}

__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1);
__device__ __forceinline__ float keyed_uniform(unsigned int seed_lo, unsigned int seed_hi, int step, unsigned int grow, int draw, int col);
// rng (by value; seed_lo == seed_hi == 0 and row_stride == 0: none): the jitter draws are GENERATED here (draw 0 of
// upnerf_uniform_keyed, the same numbers) instead of read from `u` -- one launch less per step
struct RngKey {
  unsigned int seed_lo, seed_hi;
  int step, row0, row_stride;
  const float* step_dev;
};
__global__ void sample_coarse_kernel(int R, int S, const float* __restrict__ near_far, const float* __restrict__ steps,
                                     const float* __restrict__ u, float perturb, int use_disp, float* __restrict__ z, RngKey rng) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * S) return;
  const int r = idx / S, i = idx - r * S;
  const float near = near_far[2 * r], far = near_far[2 * r + 1];
  const float zi = base_z(near, far, steps[i], use_disp);
  if (perturb > 0.f && rng.row_stride > 0) {
    const int step = rng.step_dev ? (int)rng.step_dev[0] : rng.step;
    const float uk = keyed_uniform(rng.seed_lo, rng.seed_hi, step, (unsigned int)(rng.row0 + r * rng.row_stride), 0, i);
    const float zl = i > 0 ? base_z(near, far, steps[i - 1], use_disp) : zi;
    const float zr = i < S - 1 ? base_z(near, far, steps[i + 1], use_disp) : zi;
    const float upper = i < S - 1 ? 0.5f * (zi + zr) : zi;
    const float lower = i > 0 ? 0.5f * (zl + zi) : zi;
    z[idx] = lower + (upper - lower) * (perturb * uk);
  } else if (perturb > 0.f && u) {
    const float zl = i > 0 ? base_z(near, far, steps[i - 1], use_disp) : zi;
    const float zr = i < S - 1 ? base_z(near, far, steps[i + 1], use_disp) : zi;
    const float upper = i < S - 1 ? 0.5f * (zi + zr) : zi;
    const float lower = i > 0 ? 0.5f * (zl + zi) : zi;
    z[idx] = lower + (upper - lower) * (perturb * u[idx]);
  } else {
    z[idx] = zi;
  }
}

// ---- uniform draws keyed by (seed, step, global ray, draw, column): Philox4x32-10 (Salmon et al., SC'11; the generator
// family torch.rand uses on the GPU), counter = (global ray, column / 4, step, draw), key = seed.  The value depends on the
// ray's GLOBAL row in the data-parallel batch, not on the rank that renders it: 1 rank x 8192 rays and 2 ranks x 4096 draw
// the same numbers.  u = (x >> 8) * 2^-24, 24 random bits in [0, 1) like torch.rand.
__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (unsigned int)p1;
  c[3] = (unsigned int)p0;
  c[0] = n0;
  c[2] = n2;
}
// u(seed, step, global row, draw, column c): the value uniform_keyed_kernel writes to out[r][c] (element c & 3 of block c >> 2)
__device__ __forceinline__ float keyed_uniform(unsigned int seed_lo, unsigned int seed_hi, int step, unsigned int grow, int draw, int col) {
  unsigned int c[4] = {grow, (unsigned int)(col >> 2), (unsigned int)step, (unsigned int)draw};
  unsigned int k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const int j = col & 3;
  const unsigned int x = j == 0 ? c[0] : (j == 1 ? c[1] : (j == 2 ? c[2] : c[3]));
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__global__ void uniform_keyed_kernel(int R, int n, unsigned int seed_lo, unsigned int seed_hi, int step,
                                     const float* __restrict__ step_dev, int row0, int row_stride, int draw,
                                     float* __restrict__ out) {
  const int q4 = (n + 3) >> 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * q4) return;
  const int r = idx / q4, q = idx - r * q4;
  if (step_dev) step = (int)step_dev[0];  // graph replay: the step counter lives in device memory (exact in fp32 below 2^24)
  unsigned int c[4] = {(unsigned int)(row0 + r * row_stride), (unsigned int)q, (unsigned int)step, (unsigned int)draw};
  unsigned int k0 = seed_lo, k1 = seed_hi;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 * q + j < n) out[(size_t)r * n + 4 * q + j] = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
}

// ---- a11: sample_pdf (rendering.py:7-50); one wave per ray, cdf kept in LDS.
#define PDF_MAXS 1024
// One ray by one wave: the cdf of weights[1 .. S-1) over the mid-points of z (both rows of S entries), n inverse-CDF samples to
// out[0 .. n).  U(k) yields the k-th uniform (a buffer read or a keyed Philox value); out may be global or LDS.
template <class U>
__device__ __forceinline__ void pdf_row(int S, const float* __restrict__ zrow, const float* __restrict__ wrow, int n, U&& uget,
                                        float* out, float* cdf, float* bins, int lane) {
  const int B = S - 2;  // number of weights; cdf and bins have B+1 entries
  const float eps = 1e-5f;
  double part = 0.0;
  for (int j = lane; j < B; j += 64) {
    const float w = wrow[1 + j] + eps;
    cdf[j + 1] = w;  // staged; turned into the running sum below
    part += (double)w;
  }
  for (int j = lane; j <= B; j += 64) bins[j] = 0.5f * (zrow[j] + zrow[j + 1]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
  const float total = (float)part;  // correctly rounded sum of the row
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    // torch.cumsum on CPU accumulates float rows in double and rounds each prefix to float
    double run = 0.0;
    cdf[0] = 0.f;
    for (int j = 0; j < B; ++j) {
      run += (double)(cdf[j + 1] / total);
      cdf[j + 1] = (float)run;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int k = lane; k < n; k += 64) {
    const float uk = uget(k);
    // searchsorted(cdf, u, right=True): number of entries <= u
    int lo = 0, hi = B + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= uk) lo = mid + 1; else hi = mid;
    }
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < B ? lo : B;
    const float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
    float den = c1 - c0;
    if (den < eps) den = 1.f;
    out[k] = b0 + (uk - c0) / den * (b1 - b0);
  }
  __builtin_amdgcn_wave_barrier();  // (the next call of this wave rewrites cdf / bins)
}

__global__ __launch_bounds__(NTHREADS) void sample_pdf_kernel(int R, int S, const float* __restrict__ z,
                                                             const float* __restrict__ weights,
                                                             const float* __restrict__ u, int u_rows, int n,
                                                             float* __restrict__ out, int out_stride) {
  __shared__ float cdf_s[4][PDF_MAXS];
  __shared__ float bins_s[4][PDF_MAXS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const float* __restrict__ urow = u + (size_t)(u_rows == 1 ? 0 : r) * n;
  pdf_row(S, z + (size_t)r * S, weights + (size_t)r * S, n, [&](int k) { return urow[k]; }, out + (size_t)r * out_stride,
          cdf_s[wave], bins_s[wave], lane);
}

// ---- a11 + a12 in one launch (rendering.py:262-308): the fine depths of a ray = sort(z_coarse | samples of set A | samples of
// set B).  One wave per ray: both inverse-CDF sets (sample_pdf_kernel's arithmetic, call for call) land in an LDS row next to
// the coarse depths, the rank-counting sort of sort_rows_kernel writes the row out.  Uniforms: buffers (u_a / u_b, the
// parity tests inject them) or, with a key, generated here (draws 1, 2 of upnerf_uniform_keyed in call order: set A is drawn
// first).  Replaces a strided copy, two sample_pdf launches, a sort and two uniform launches of a training step.
struct ResampleSet {
  const float* w;  // [R][Nc] weights (detached)
  const float* u;  // [R or 1][n] uniforms, or NULL with a key
  int n, col, u_rows, draw;
};
__global__ __launch_bounds__(NTHREADS) void resample_sort_kernel(int R, int Nc, int S, const float* __restrict__ z, ResampleSet A,
                                                                ResampleSet Bs, RngKey rng, float* __restrict__ zf) {
  __shared__ float cdf_s[4][PDF_MAXS];
  __shared__ float bins_s[4][PDF_MAXS];
  __shared__ float v_s[4][PDF_MAXS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  float* v = v_s[wave];
  const float* __restrict__ zrow = z + (size_t)r * Nc;
  for (int j = lane; j < Nc; j += 64) v[j] = zrow[j];
  const int step = rng.step_dev ? (int)rng.step_dev[0] : rng.step;
  const unsigned int grow = (unsigned int)(rng.row0 + r * rng.row_stride);
  auto one = [&](const ResampleSet& q) {
    if (q.n <= 0) return;
    const float* __restrict__ urow = q.u ? q.u + (size_t)(q.u_rows == 1 ? 0 : r) * q.n : nullptr;
    pdf_row(Nc, zrow, q.w + (size_t)r * Nc, q.n,
            [&](int k) { return urow ? urow[k] : keyed_uniform(rng.seed_lo, rng.seed_hi, step, grow, q.draw, k); }, v + q.col,
            cdf_s[wave], bins_s[wave], lane);
  };
  one(A);
  one(Bs);
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int e = lane; e < S; e += 64) {
    const float x = v[e];
    int rank = 0;
    for (int j = 0; j < S; ++j) {
      const float y = v[j];
      rank += (y < x || (y == x && j < e)) ? 1 : 0;
    }
    zf[(size_t)r * S + rank] = x;
  }
}

// ---- a12: per-row ascending sort (values only) by rank counting in LDS; stable for ties.
__global__ __launch_bounds__(NTHREADS) void sort_rows_kernel(int R, int S, float* __restrict__ z) {
  __shared__ float v_s[4][PDF_MAXS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  float* v = v_s[wave];
  for (int j = lane; j < S; j += 64) v[j] = z[(size_t)r * S + j];
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int e = lane; e < S; e += 64) {
    const float x = v[e];
    int rank = 0;
    for (int j = 0; j < S; ++j) {
      const float y = v[j];
      rank += (y < x || (y == x && j < e)) ? 1 : 0;
    }
    z[(size_t)r * S + rank] = x;
  }
}

// ---- per-ray side input of the colour head: [PE(dir, L=4) | a | 0]
__global__ void ray_aux_kernel(int R, const float* __restrict__ rays_d, const float* __restrict__ a_rows,
                               float w0, float w1, float w2, float w3, const float* __restrict__ wk_dev,
                               float* __restrict__ aux) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float* o = aux + (size_t)r * UPNERF_AUXK;
  float wk[4] = {w0, w1, w2, w3};
  if (wk_dev) {  // per-step scalars from device memory (graph replay)
#pragma unroll
    for (int k = 0; k < 4; ++k) wk[k] = wk_dev[k];
  }
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const float x = rays_d[r * 3 + n];
    o[n] = x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float arg = x * ldexpf(3.14159274101257324f, k);
      float sv, cv;
      sincosf(arg, &sv, &cv);
      o[3 + 8 * n + k] = sv * wk[k];
      o[3 + 8 * n + 4 + k] = cv * wk[k];
    }
  }
  // the 48 appearance-row floats: twelve 16-byte loads requested together, then the stores (the element-wise copy was compiled to
  // load / wait / store forty-eight times over: one memory round trip per element, round 6); rows that do not start on 16 bytes
  // (a table slice at an odd offset) keep the element-wise copy
  if (a_rows && ((size_t)a_rows & 15) == 0) {
    f32x4 av[12];
    const f32x4* ap = (const f32x4*)(a_rows + (size_t)r * 48);
#pragma unroll
    for (int j = 0; j < 12; ++j) av[j] = ap[j];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      o[27 + 4 * j] = av[j].x; o[28 + 4 * j] = av[j].y; o[29 + 4 * j] = av[j].z; o[30 + 4 * j] = av[j].w;
    }
  } else {
    for (int j = 0; j < 48; ++j) o[27 + j] = a_rows ? a_rows[(size_t)r * 48 + j] : 0.f;
  }
  for (int j = 75; j < UPNERF_AUXK; ++j) o[j] = 0.f;
}

// ---- out[r][c] = sum_i X[r*S+i][c]; one block per ray, one thread per column (C <= 256)
__global__ void ray_sum_kernel(int S, const float* __restrict__ X, int C, float* __restrict__ out) {
  const int r = blockIdx.x, c = threadIdx.x;
  if (c >= C) return;
  const float* p = X + (size_t)r * S * C + c;
  float s0 = 0.f, s1 = 0.f;
  int i = 0;
  for (; i + 1 < S; i += 2) {
    s0 += p[(size_t)i * C];
    s1 += p[(size_t)(i + 1) * C];
  }
  if (i < S) s0 += p[(size_t)i * C];
  out[(size_t)r * C + c] = s0 + s1;
}

// ---- (d_o, d_d) = (sum_i dxyz_i, sum_i z_i dxyz_i); one wave per ray
__global__ __launch_bounds__(NTHREADS) void ray_geom_bwd_kernel(int R, int S, const float* __restrict__ dxyz,
                                                               const float* __restrict__ z, float* __restrict__ d_o,
                                                               float* __restrict__ d_d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
  for (int i = lane; i < S; i += 64) {
    const size_t m = (size_t)r * S + i;
    const float zz = z[m];
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const float g = dxyz[m * 3 + n];
      so[n] += g;
      sd[n] += zz * g;
    }
  }
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const float a = wave_sum(so[n]), b = wave_sum(sd[n]);
    if (lane == 0) {
      d_o[r * 3 + n] = a;
      d_d[r * 3 + n] = b;
    }
  }
}


// Dense gradient of an embedding table from the gradients of the gathered rows (autograd of nn.Embedding(idx),
// models/nerf_system.py:79-91 tables and models/transient_net.py): out[n] = sum over r with idx[r] == n of g[r], summed
// in increasing r (deterministic, no atomics, no sort).  One wave per table row scans idx 64 entries at a time.
__global__ void embed_bwd_kernel(int R, int N, int dim, const long long* __restrict__ idx, const float* __restrict__ g,
                                 float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // dim <= 256
  for (int base = 0; base < R; base += 64) {
    const int r = base + lane;
    const bool hit = r < R && idx[r] == (long long)n;
    unsigned long long m = __ballot(hit);
    while (m) {
      const int j = __ffsll((long long)m) - 1;
      m &= m - 1;
      const float* __restrict__ row = g + (size_t)(base + j) * dim;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lane + 64 * q < dim) acc[q] += row[lane + 64 * q];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (lane + 64 * q < dim) out[(size_t)n * dim + lane + 64 * q] = acc[q];
}

// Several tables indexed by the SAME idx (the per-image tables of a training step all are): one scan of idx per table
// row serves every table, one launch instead of one per table.
struct EmbedGroups {
  const float* g[UPNERF_MAX_EMBED_GROUPS];
  float* out[UPNERF_MAX_EMBED_GROUPS];
  int dim[UPNERF_MAX_EMBED_GROUPS];
  int n;
};

// The workgroup's four table rows scan the SAME indices: they are staged in LDS once (coalesced, 4096 rays per trip) and each
// wave ballots its row out of LDS -- every wave walking idx through its own chain of global loads was most of a 52 us launch
// for a batch in which a table row is hit five times.  The rows a wave adds up are still taken in ray order.
#define EMB_CHUNK 4096
__global__ __launch_bounds__(NTHREADS) void embed_bwd_grouped_kernel(int R, int N, const long long* __restrict__ idx, EmbedGroups T) {
  __shared__ int idx_s[EMB_CHUNK];
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (NTHREADS >> 6) + (threadIdx.x >> 6);
  float acc[UPNERF_MAX_EMBED_GROUPS][4];
#pragma unroll
  for (int t = 0; t < UPNERF_MAX_EMBED_GROUPS; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[t][q] = 0.f;
  for (int c0 = 0; c0 < R; c0 += EMB_CHUNK) {
    if (c0) __syncthreads();
#pragma unroll
    for (int u = 0; u < EMB_CHUNK / NTHREADS; ++u) {
      const int r = c0 + u * NTHREADS + (int)threadIdx.x;
      const long long v = idx[r < R ? r : R - 1];  // (branch-free: sixteen guarded loads were sixteen memory round trips in a row)
      idx_s[u * NTHREADS + threadIdx.x] = (r < R && v >= 0 && v < N) ? (int)v : -1;
    }
    __syncthreads();
    const int cnt = (R - c0) < EMB_CHUNK ? (R - c0) : EMB_CHUNK;
    if (n < N) {
      for (int base = 0; base < cnt; base += 64) {
        unsigned long long m = __ballot(idx_s[base + lane] == n);
        while (m) {
          const int j = __ffsll((long long)m) - 1;
          m &= m - 1;
          const size_t ray = (size_t)(c0 + base + j);
          float v[UPNERF_MAX_EMBED_GROUPS][4];
#pragma unroll
          for (int t = 0; t < UPNERF_MAX_EMBED_GROUPS; ++t) {
            const int dim = T.dim[t];  // 0 for an unused slot
            const float* __restrict__ row = T.g[t] + ray * dim;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[t][q] = (lane + 64 * q < dim) ? row[lane + 64 * q] : 0.f;
          }
#pragma unroll
          for (int t = 0; t < UPNERF_MAX_EMBED_GROUPS; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][q] += v[t][q];
        }
      }
    }
  }
  if (n >= N) return;
#pragma unroll
  for (int t = 0; t < UPNERF_MAX_EMBED_GROUPS; ++t) {
    if (t < T.n) {
      const int dim = T.dim[t];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lane + 64 * q < dim) T.out[t][(size_t)n * dim + lane + 64 * q] = acc[t][q];
    }
  }
}

// Train-split ray sampler (datasets/phototourism.py:420-454 + default collate): one wave per ray gathers the per-ray
// scalars and interpolates the image's feature map bilinearly -- same operand order and roundings as the reference's
// scalar code (w11 p11 + w12 p12 + w21 p21 + w22 p22, no fma), including its zero weights on the last row / column.
__global__ void gather_rays_kernel(upnerf_gather_rays_args a) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= a.R) return;
  const long long i = a.idx[r];
  const int img = (int)a.all_ray_infos[i * 3 + 2];
  if (lane < 2) a.ray_infos[r * 2 + lane] = a.all_ray_infos[i * 3 + lane];
  if (lane < 3) {
    a.directions[r * 3 + lane] = a.all_directions[i * 3 + lane];
    a.rgbs[r * 3 + lane] = a.all_rgbs[i * 3 + lane];
  }
  if (lane < 12) a.c2w[r * 12 + lane] = a.poses[img * 12 + lane];
  if (lane == 0) {
    a.img_idx[r] = img;
    if (a.inv_depths) a.inv_depths[r] = a.all_inv_depths[i];
  }
  if (!a.feats) return;
  const int h = a.h, C = a.C;
  const float y = a.all_pxl_coords[i * 2 + 0] * (float)(h - 1), x = a.all_pxl_coords[i * 2 + 1] * (float)(h - 1);
  const int y1 = (int)floorf(y), x1 = (int)floorf(x);
  const int y2 = y1 + 1 < h - 1 ? y1 + 1 : h - 1, x2 = x1 + 1 < h - 1 ? x1 + 1 : h - 1;
  const float wy2 = (float)y2 - y, wy1 = y - (float)y1, wx2 = (float)x2 - x, wx1 = x - (float)x1;
  const float w11 = wy2 * wx2, w12 = wy2 * wx1, w21 = wy1 * wx2, w22 = wy1 * wx1;
  const float* __restrict__ fm = a.feat_maps + (size_t)img * h * h * C;
  const float* __restrict__ p11 = fm + ((size_t)y1 * h + x1) * C;
  const float* __restrict__ p12 = fm + ((size_t)y1 * h + x2) * C;
  const float* __restrict__ p21 = fm + ((size_t)y2 * h + x1) * C;
  const float* __restrict__ p22 = fm + ((size_t)y2 * h + x2) * C;
  for (int c = lane; c < C; c += 64) {
    float v = w11 * p11[c];
    v = v + w12 * p12[c];
    v = v + w21 * p21[c];
    v = v + w22 * p22[c];
    a.feats[(size_t)r * C + c] = v;
  }
}
}  // namespace

extern "C" int upnerf_abi_version(void) { return UPNERF_ABI_VERSION; }

extern "C" int upnerf_pose_rays_fwd(int R, const float* se3, const float* c2w, const float* dirs, float* rays_o,
                                    float* rays_d, void* stream) {
  if (R <= 0 || !c2w || !dirs || !rays_o || !rays_d) return UPNERF_EINVAL;
  hipLaunchKernelGGL(pose_rays_fwd_kernel, dim3((R + 127) / 128), dim3(128), 0, (hipStream_t)stream, R, se3, c2w, dirs,
                     rays_o, rays_d);
  return (int)hipGetLastError();
}

extern "C" int upnerf_pose_rays_bwd(int R, const float* se3, const float* c2w, const float* dirs, const float* g_o,
                                    const float* g_d, float* g_se3, void* stream) {
  if (R <= 0 || !se3 || !c2w || !dirs || !g_o || !g_d || !g_se3) return UPNERF_EINVAL;
  hipLaunchKernelGGL(pose_rays_bwd_kernel, dim3((R + 127) / 128), dim3(128), 0, (hipStream_t)stream, R, se3, c2w, dirs,
                     g_o, g_d, g_se3);
  return (int)hipGetLastError();
}

extern "C" int upnerf_sample_coarse(int R, int S, const float* near_far, const float* steps, const float* u,
                                    float perturb, int use_disp, float* z_out, void* stream) {
  if (R <= 0 || S <= 0 || !near_far || !steps || !z_out) return UPNERF_EINVAL;
  if (perturb > 0.f && !u) return UPNERF_EINVAL;
  const long long n = (long long)R * S;
  hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, R, S,
                     near_far, steps, u, perturb, use_disp, z_out, RngKey{0u, 0u, 0, 0, 0, nullptr});
  return (int)hipGetLastError();
}

static int rng_key(const upnerf_rng* g, RngKey* k) {
  if (!g || g->step < 0 || g->row0 < 0 || g->row_stride < 1) return UPNERF_EINVAL;
  *k = RngKey{(unsigned int)g->seed, (unsigned int)(g->seed >> 32), g->step, g->row0, g->row_stride, g->step_dev};
  return 0;
}

extern "C" int upnerf_sample_coarse_keyed(int R, int S, const float* near_far, const float* steps, const upnerf_rng* rng,
                                          float perturb, int use_disp, float* z_out, void* stream) {
  if (R <= 0 || S <= 0 || !near_far || !steps || !z_out || !(perturb > 0.f)) return UPNERF_EINVAL;
  RngKey k;
  if (rng_key(rng, &k)) return UPNERF_EINVAL;
  const long long n = (long long)R * S;
  hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, R, S,
                     near_far, steps, (const float*)nullptr, perturb, use_disp, z_out, k);
  return (int)hipGetLastError();
}

extern "C" int upnerf_uniform_keyed(int R, int n, uint64_t seed, int step, const float* step_dev, int row0, int row_stride,
                                    int draw, float* out, void* stream) {
  if (R <= 0 || n <= 0 || !out || step < 0 || row0 < 0 || row_stride < 1 || draw < 0) return UPNERF_EINVAL;
  const long long total = (long long)R * ((n + 3) / 4);
  hipLaunchKernelGGL(uniform_keyed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, R, n,
                     (unsigned int)seed, (unsigned int)(seed >> 32), step, step_dev, row0, row_stride, draw, out);
  return (int)hipGetLastError();
}

extern "C" int upnerf_sample_pdf(int R, int S, const float* z, const float* weights, const float* u, int u_rows, int n,
                                 float* out, int out_stride, void* stream) {
  if (R <= 0 || S < 3 || S > PDF_MAXS || n < 0 || !z || !weights || !out) return UPNERF_EINVAL;
  if (n == 0) return 0;
  if (!u || (u_rows != 1 && u_rows != R) || out_stride < n) return UPNERF_EINVAL;
  hipLaunchKernelGGL(sample_pdf_kernel, dim3((R + 3) / 4), dim3(NTHREADS), 0, (hipStream_t)stream, R, S, z, weights, u,
                     u_rows, n, out, out_stride);
  return (int)hipGetLastError();
}

extern "C" int upnerf_resample_sort(int R, int Nc, const float* z, const float* w_a, int n_a, int col_a, const float* u_a, int draw_a,
                                    const float* w_b, int n_b, int col_b, const float* u_b, int draw_b, int u_rows,
                                    const upnerf_rng* rng, float* zf, void* stream) {
  const long long S = (long long)Nc + n_a + n_b;
  if (R <= 0 || Nc < 3 || n_a < 0 || n_b < 0 || S > PDF_MAXS || !z || !zf) return UPNERF_EINVAL;
  if ((n_a > 0 && !w_a) || (n_b > 0 && !w_b)) return UPNERF_EINVAL;
  // the sets tile the columns behind the coarse depths
  if (n_a > 0 && (col_a < Nc || col_a + n_a > S)) return UPNERF_EINVAL;
  if (n_b > 0 && (col_b < Nc || col_b + n_b > S)) return UPNERF_EINVAL;
  if (n_a > 0 && n_b > 0 && !(col_a + n_a <= col_b || col_b + n_b <= col_a)) return UPNERF_EINVAL;
  RngKey k{0u, 0u, 0, 0, 0, nullptr};
  const bool need_key = (n_a > 0 && !u_a) || (n_b > 0 && !u_b);
  if (need_key && rng_key(rng, &k)) return UPNERF_EINVAL;
  if (!need_key && (u_rows != 1 && u_rows != R)) return UPNERF_EINVAL;
  if (draw_a < 0 || draw_b < 0) return UPNERF_EINVAL;
  ResampleSet A{w_a, u_a, n_a, col_a, u_rows, draw_a}, B{w_b, u_b, n_b, col_b, u_rows, draw_b};
  hipLaunchKernelGGL(resample_sort_kernel, dim3((R + 3) / 4), dim3(NTHREADS), 0, (hipStream_t)stream, R, Nc, (int)S, z, A, B, k, zf);
  return (int)hipGetLastError();
}

extern "C" int upnerf_sort_rows(int R, int S, float* z, void* stream) {
  if (R <= 0 || S <= 0 || S > PDF_MAXS || !z) return UPNERF_EINVAL;
  hipLaunchKernelGGL(sort_rows_kernel, dim3((R + 3) / 4), dim3(NTHREADS), 0, (hipStream_t)stream, R, S, z);
  return (int)hipGetLastError();
}

extern "C" int upnerf_ray_aux(int R, const float* rays_d, const float* a_rows, const float* wk_dir,
                              const float* wk_dir_dev, float* aux, void* stream) {
  if (R <= 0 || !rays_d || !wk_dir || !aux) return UPNERF_EINVAL;
  // wk_dir is a HOST pointer to 4 floats (band weights are host scalars derived from the step counter); wk_dir_dev, when
  // given, is read by the kernel instead (same values, but live across replays of a captured graph)
  hipLaunchKernelGGL(ray_aux_kernel, dim3((R + 127) / 128), dim3(128), 0, (hipStream_t)stream, R, rays_d, a_rows,
                     wk_dir[0], wk_dir[1], wk_dir[2], wk_dir[3], wk_dir_dev, aux);
  return (int)hipGetLastError();
}

// the register-resident backward kernel's per-32-sample partial sums (include/upnerf_hip.h, UPNERF_RR_PART_STRIDE) -> per-ray sums
__global__ void ray_part_finish_kernel(int S, const float* __restrict__ part, float* __restrict__ rs_g1, float* __restrict__ rs_r1) {
  const int r = blockIdx.x, t = threadIdx.x >> 7, col = threadIdx.x & 127;
  float* __restrict__ out = t ? rs_g1 : rs_r1;
  if (!out) return;
  const int w0 = (int)(((long long)r * S) >> 5), w1 = (int)((((long long)(r + 1) * S) - 1) >> 5);
  float acc = 0.0f;
  for (int w = w0; w <= w1; ++w) {
    const int slot = r - (int)(((long long)w << 5) / S);  // 0: the 32 samples start in this ray, 1: the ray starts inside them
    acc += part[(size_t)w * UPNERF_RR_PART_STRIDE + (2 * t + slot) * 128 + col];
  }
  out[(size_t)r * 128 + col] = acc;
}
extern "C" int upnerf_ray_part_finish(int R, int S, const float* tile_part, float* rs_g1, float* rs_r1, void* stream) {
  if (R <= 0 || S < 32 || !tile_part) return UPNERF_EINVAL;
  hipLaunchKernelGGL(ray_part_finish_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, S, tile_part, rs_g1, rs_r1);
  return (int)hipGetLastError();
}

extern "C" int upnerf_ray_sum(int R, int S, const float* X, int C, float* out, void* stream) {
  if (R <= 0 || S <= 0 || C <= 0 || C > 256 || !X || !out) return UPNERF_EINVAL;
  const int threads = ((C + 63) / 64) * 64;
  hipLaunchKernelGGL(ray_sum_kernel, dim3(R), dim3(threads), 0, (hipStream_t)stream, S, X, C, out);
  return (int)hipGetLastError();
}

extern "C" int upnerf_ray_geom_bwd(int R, int S, const float* dxyz, const float* z, float* d_o, float* d_d,
                                   void* stream) {
  if (R <= 0 || S <= 0 || !dxyz || !z || !d_o || !d_d) return UPNERF_EINVAL;
  hipLaunchKernelGGL(ray_geom_bwd_kernel, dim3((R + 3) / 4), dim3(NTHREADS), 0, (hipStream_t)stream, R, S, dxyz, z, d_o,
                     d_d);
  return (int)hipGetLastError();
}

extern "C" int upnerf_embed_bwd(int R, int N, int dim, const int64_t* idx, const float* g, float* out, void* stream) {
  if (R <= 0 || N <= 0 || dim <= 0 || dim > 256 || !idx || !g || !out) return UPNERF_EINVAL;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((N + 3) / 4), dim3(NTHREADS), 0, (hipStream_t)stream, R, N, dim,
                     (const long long*)idx, g, out);
  return (int)hipGetLastError();
}

extern "C" int upnerf_embed_bwd_grouped(int R, int N, const int64_t* idx, const upnerf_embed_group* groups, int ngroups,
                                        void* stream) {
  if (R <= 0 || N <= 0 || !idx || !groups || ngroups <= 0 || ngroups > UPNERF_MAX_EMBED_GROUPS) return UPNERF_EINVAL;
  EmbedGroups T;
  T.n = ngroups;
  for (int j = 0; j < UPNERF_MAX_EMBED_GROUPS; ++j) {
    const bool live = j < ngroups;
    if (live && (!groups[j].g || !groups[j].out || groups[j].dim <= 0 || groups[j].dim > 256)) return UPNERF_EINVAL;
    T.g[j] = live ? groups[j].g : nullptr;
    T.out[j] = live ? groups[j].out : nullptr;
    T.dim[j] = live ? groups[j].dim : 0;
  }
  hipLaunchKernelGGL(embed_bwd_grouped_kernel, dim3((N + 3) / 4), dim3(NTHREADS), 0, (hipStream_t)stream, R, N,
                     (const long long*)idx, T);
  return (int)hipGetLastError();
}

// Forward gathers of the per-image tables: one wave per ray copies its row of every table.
struct EmbedRowGroups {
  const float* table[UPNERF_MAX_EMBED_GROUPS];
  float* rows[UPNERF_MAX_EMBED_GROUPS];
  int dim[UPNERF_MAX_EMBED_GROUPS];
  int n;
};
__global__ __launch_bounds__(NTHREADS) void embed_fwd_grouped_kernel(int R, int N, const long long* __restrict__ idx, EmbedRowGroups T) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (NTHREADS >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const long long i = idx[r];
  const bool ok = i >= 0 && i < N;
#pragma unroll
  for (int t = 0; t < UPNERF_MAX_EMBED_GROUPS; ++t) {
    if (t < T.n) {
      const int dim = T.dim[t];
      const float* __restrict__ src = T.table[t] + (size_t)(ok ? i : 0) * dim;
      float* __restrict__ dst = T.rows[t] + (size_t)r * dim;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lane + 64 * q < dim) dst[lane + 64 * q] = ok ? src[lane + 64 * q] : __builtin_nanf("");
    }
  }
}
extern "C" int upnerf_embed_fwd_grouped(int R, int N, const int64_t* idx, const upnerf_embed_rows_group* groups, int ngroups,
                                        void* stream) {
  if (R <= 0 || N <= 0 || !idx || !groups || ngroups <= 0 || ngroups > UPNERF_MAX_EMBED_GROUPS) return UPNERF_EINVAL;
  EmbedRowGroups T;
  T.n = ngroups;
  for (int j = 0; j < UPNERF_MAX_EMBED_GROUPS; ++j) {
    const bool live = j < ngroups;
    if (live && (!groups[j].table || !groups[j].rows || groups[j].dim <= 0 || groups[j].dim > 256)) return UPNERF_EINVAL;
    T.table[j] = live ? groups[j].table : nullptr;
    T.rows[j] = live ? groups[j].rows : nullptr;
    T.dim[j] = live ? groups[j].dim : 0;
  }
  hipLaunchKernelGGL(embed_fwd_grouped_kernel, dim3((R + 3) / 4), dim3(NTHREADS), 0, (hipStream_t)stream, R, N,
                     (const long long*)idx, T);
  return (int)hipGetLastError();
}

extern "C" int upnerf_gather_rays(const upnerf_gather_rays_args* a, void* stream) {
  if (!a || a->R <= 0 || !a->idx || !a->all_ray_infos || !a->all_directions || !a->all_rgbs || !a->poses || !a->ray_infos ||
      !a->directions || !a->img_idx || !a->c2w || !a->rgbs)
    return UPNERF_EINVAL;
  if (a->feats && (!a->feat_maps || !a->all_pxl_coords || a->h < 2 || a->C <= 0)) return UPNERF_EINVAL;
  if (a->inv_depths && !a->all_inv_depths) return UPNERF_EINVAL;
  hipLaunchKernelGGL(gather_rays_kernel, dim3((a->R + 3) / 4), dim3(NTHREADS), 0, (hipStream_t)stream, *a);
  return (int)hipGetLastError();
}
