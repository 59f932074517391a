/*
 * upnerf_hip.h -- C ABI of libupnerf_hip.so, the MI355X (gfx950) implementation of the UP-NeRF
 * render_rays training hot path.
 *
 * The reference (mlvlab/UP-NeRF) is pure Python on PyTorch: it has no native boundary of its own
 * (SURVEY.md 2.2).  The entry points below are therefore the leaf operations its Python hot path
 * performs, one per group of ATen launches, and each cites the reference lines it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add to call them from models/rendering.py.
 *
 * Conventions
 *  - every function returns 0 on success, a hipError_t (>0) on a HIP failure, or a negative
 *    UPNERF_E* code on an argument error; nothing is printed, nothing throws.
 *  - all pointers are DEVICE pointers to fp32 (or int64 where said), row-major, contiguous; they
 *    are borrowed for the duration of the call (stream-ordered) -- the library allocates nothing
 *    and keeps no global state.  `stream` is a hipStream_t passed as void*.
 *  - "rows" M = R * S (rays x samples per ray), sample i of ray r is row r*S + i.
 */
#ifndef UPNERF_HIP_H
#define UPNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UPNERF_ABI_VERSION 10
#define UPNERF_EINVAL (-1)   /* bad size / null pointer */
#define UPNERF_EUNSUP (-2)   /* unsupported width/depth combination */

#ifndef UPNERF_TILE_ROWS
#define UPNERF_TILE_ROWS 64  /* rows of samples per workgroup in the fused field kernels */
#endif
#define UPNERF_X0 64         /* positional encoding 3+6*10 = 63 padded to 64 floats per row */
#define UPNERF_AUXK 80       /* per-ray rgb-head side input [dirPE(27) | appearance(48) | 0 x5] */
#define UPNERF_CK 16         /* candidate embedding width */
#define UPNERF_MAX_D 8

int upnerf_abi_version(void);

/* ---- a2-a4: pose refinement + ray generation (utils/camera.py:87-98,113-152 se3_to_SE3;
 *      camera.py:51-58 compose_pair; utils/ray.py:44-56 get_rays batched branch) -------------------
 * se3 [R][6] (already gathered rows of se3_refine, or NULL = identity refinement), c2w [R][3][4],
 * dirs [R][3] camera-space directions -> rays_o [R][3], rays_d [R][3] (unit norm).
 * bwd: g_o, g_d [R][3] -> g_se3 [R][6] (per ray; the caller scatter-adds rows by img_idx). */
int upnerf_pose_rays_fwd(int R, const float* se3, const float* c2w, const float* dirs,
                         float* rays_o, float* rays_d, void* stream);
int upnerf_pose_rays_bwd(int R, const float* se3, const float* c2w, const float* dirs,
                         const float* g_o, const float* g_d, float* g_se3, void* stream);

/* ---- a5: stratified coarse depths (models/rendering.py:232-249) ---------------------------------
 * near_far [R][2]; steps [S] = linspace(0,1,S); u [R][S] uniform draws or NULL (perturb == 0);
 * z_out [R][S]. */
int upnerf_sample_coarse(int R, int S, const float* near_far, const float* steps, const float* u,
                         float perturb, int use_disp, float* z_out, void* stream);

/* ---- uniform draws for a5 / a11 (the reference calls torch.rand / rand_like, models/rendering.py:248, 29) --------------
 * out[r][c] = u(seed, step, row0 + r * row_stride, draw, c) in [0, 1), c < n: Philox4x32-10 with counter (global row, c / 4,
 * step, draw) and key seed, 24 bits per value.  row0 + r * row_stride = GLOBAL row of local ray r in the data-parallel batch, so
 * the numbers do not depend on how the batch is split over ranks: contiguous shards pass (rank * R, 1), shards
 * dealt like DistributedSampler (local ray r = global ray r * world + rank) pass (rank, world).  draw = 0 coarse jitter, 1, 2 =
 * the sample_pdf sets in call order.  step_dev (DEVICE [1] float or NULL) overrides `step` at execution time (graph replay). */
int upnerf_uniform_keyed(int R, int n, uint64_t seed, int step, const float* step_dev, int row0, int row_stride, int draw,
                         float* out, void* stream);

/* Key of the uniform draws, for kernels that GENERATE their draws instead of reading a buffer upnerf_uniform_keyed wrote (round
 * 6: one launch less per consumer).  Same fields, same numbers: u(seed, step, row0 + r * row_stride, draw, column). */
typedef struct upnerf_rng {
  uint64_t seed;
  int32_t step;           /* optimisation step; overridden by step_dev[0] when step_dev != NULL (graph replay) */
  int32_t row0, row_stride;
  const float* step_dev;  /* DEVICE [1] float or NULL */
} upnerf_rng;

/* a5 with the jitter draws generated in the kernel (draw 0 of upnerf_uniform_keyed): z_out as upnerf_sample_coarse would compute
 * it from u = upnerf_uniform_keyed(R, S, ..., draw 0), bit for bit.  perturb > 0. */
int upnerf_sample_coarse_keyed(int R, int S, const float* near_far, const float* steps, const upnerf_rng* rng, float perturb,
                               int use_disp, float* z_out, void* stream);

/* ---- a11: inverse-CDF resampling (models/rendering.py:7-50 sample_pdf) ---------------------------
 * z [R][S] coarse depths (bins = midpoints, computed inside), weights [R][S] (only [1:S-1] used),
 * u [u_rows][n] with u_rows == R, or u_rows == 1 for the deterministic linspace; writes n values per
 * ray to out + r*out_stride. */
int upnerf_sample_pdf(int R, int S, const float* z, const float* weights, const float* u, int u_rows,
                      int n, float* out, int out_stride, void* stream);

/* ---- a12: ascending sort of each row (models/rendering.py:275,290,298,307 torch.sort values) ----- */
int upnerf_sort_rows(int R, int S, float* z, void* stream);

/* ---- a11 + a12 in one launch (models/rendering.py:262-308): zf[r] = sort(z[r] | set A | set B), S = Nc + n_a + n_b columns.
 * z [R][Nc] coarse depths; set X = n_x inverse-CDF samples of the weights w_x [R][Nc] (entries 1..Nc-2 used, as upnerf_sample_pdf)
 * that upnerf_sample_pdf would write to columns [col_x, col_x + n_x) of zf before upnerf_sort_rows sorts the row -- the same
 * arithmetic call for call, so zf equals the three-launch sequence bit for bit.  n_b may be 0 (one set).  Uniforms: u_x
 * [u_rows][n_x] (u_rows = R, or 1 for the deterministic linspace) or, where u_x is NULL, generated from `rng` as draw number
 * draw_x of upnerf_uniform_keyed (the caller numbers its draws in call order; draw 0 is the coarse jitter). */
int upnerf_resample_sort(int R, int Nc, const float* z, const float* w_a, int n_a, int col_a, const float* u_a, int draw_a,
                         const float* w_b, int n_b, int col_b, const float* u_b, int draw_b, int u_rows, const upnerf_rng* rng,
                         float* zf, void* stream);

/* ---- per-ray rgb-head side input: [PE(rays_d, L=4, masked) | appearance row | 0]  (nerf.py:102-107;
 *      the reference repeats it per sample, rendering.py:104-109) ---------------------------------- */
int upnerf_ray_aux(int R, const float* rays_d, const float* a_rows /*[R][48] or NULL*/,
                   const float* wk_dir /*[4] HOST*/, const float* wk_dir_dev /*[4] DEVICE or NULL: overrides wk_dir (graph
                   replay: per-step scalars live in device memory, see upnerf_set_scalars)*/,
                   float* aux /*[R][UPNERF_AUXK]*/, void* stream);

/* ---- a6-a9: fused NeRF field, forward (models/nerf.py:80-124 + 126-147) --------------------------
 * Layout of the packed parameter buffers (floats): offsets below; a matrix is [N][Kp] with Kp a multiple
 * of 8 (zero padded), N a multiple of 32.  Two orderings of the SAME offsets are used:
 *   row-major      W[n][k] at off + n*Kp + k           -- gradients (upnerf_wgrad output) and host code
 *   fragment order W[n][k] at off + ((n/32)*(Kp/8) + k/8)*256 + (((k/4)%2)*32 + n%32)*4 + k%4
 *                                                        -- what upnerf_field_fwd/bwd READ (one 1 KiB block =
 *                                                           one wave-wide MFMA B-operand load)
 * Vectors (biases, the 1- and 3-wide heads wsig/wcsig/wr2) are stored plainly in both.
 * See upnerf_amd/packing.py (pack / frag / pack_t / frag_t) for the packing from the reference's
 * state_dict names. */
typedef struct {
  int32_t W, D, skip;            /* width (64 or 256), depth (<= 8), index of the skip layer or -1 */
  int32_t w[UPNERF_MAX_D];       /* trunk layer l: [W][Kp_l], Kp_0 = 64, Kp_skip = 64 + W, else W */
  int32_t b[UPNERF_MAX_D];       /* trunk biases [W] */
  int32_t we, be;                /* xyz_encoding_final [W][W], [W] */
  int32_t wsig, bsig;            /* share_sigma.0 [W], [1] */
  int32_t wc1, bc1;              /* candidate_encoding.0 [W/2][W + 16], [W/2] */
  int32_t wc2, bc2;              /* candidate_encoding.2 [W/2][W/2], [W/2] */
  int32_t wcsig, bcsig;          /* candidate_sigma.0 [W/2], [1] */
  int32_t wr1, br1;              /* folded rgb_share_layer.0 [W/2][W + 80], [W/2] */
  int32_t wr2, br2;              /* rgb_share_layer.2 [3][W/2], [3] (padded to 4) */
  int32_t total;                 /* floats in P */
  /* transposed copies for the backward data-gradient chain, in buffer PT */
  int32_t t_w[UPNERF_MAX_D];     /* layer l: [Kin_l][W] with Kin_0 = 64; skip: h part [W][W] */
  int32_t t_skipx;               /* skip layer, encoding part [64][W] */
  int32_t t_we;                  /* [W][W] */
  int32_t t_head;                /* [W][W]: cols [0,W/2) = wr1[:, :W]^T, cols [W/2,W) = wc1[:, :W]^T */
  int32_t t_wc2;                 /* [W/2][W/2] */
  int32_t t_total;
} upnerf_layout;

typedef struct {
  int32_t R, S;                  /* rays, samples per ray */
  int32_t use_cand, use_rgb;     /* sched_mult < 1 && encode_candidate ; sched_mult > 0 */
  const float* rays_o;           /* [R][3] */
  const float* rays_d;           /* [R][3] */
  const float* z;                /* [R][S] */
  const float* c_rows;           /* [R][16] candidate embedding rows (use_cand) */
  const float* aux;              /* [R][80] from upnerf_ray_aux (use_rgb) */
  float wk_xyz[10];              /* BARF band weights for the xyz encoding */
  const float* P;                /* packed parameters, matrices in fragment order */
  /* per-sample outputs */
  float* sigma_s;                /* [M] softplus output */
  float* sigma_c;                /* [M] (use_cand) */
  float* rgb;                    /* [M][3] sigmoid output (use_rgb) */
  /* activations kept for compositing and for the backward pass.  Inference (no gradient wanted): h, hmask, g1, r1 may be
   * NULL = not stored; e and g2 may be NULL when the compositing mode does not build a feature map (mode 2).  x0 is
   * always required (the skip layer re-reads it). */
  float* x0;                     /* [M][64] */
  float* h;                      /* [D][M][W] post-ReLU trunk activations */
  uint64_t* hmask;               /* ReLU sign bits of h in the kernels' accumulator layout, private to a forward / backward kernel
                                    pair: [D + 1][tiles][threads per workgroup] words.  (D + 1) * ceil(M/128) * 512 words cover
                                    either tiling */
  float* amax;                   /* [16] or NULL: running max|.| (atomicMax; zero it first) of h_0..h_{D-1} (slots 0..D-1),
                                    e (D), g1 (D+1), r1 (D+3), x0 (D+4) -- scale exponents of upnerf_wgrad_f16x3 */
  float* e;                      /* [M][W]   xyz_encoding_final output */
  float* g1;                     /* [M][W/2] (use_cand) */
  float* g2;                     /* [M][W/2] (use_cand) */
  float* r1;                     /* [M][W/2] (use_rgb) */
  /* f16x3 variant only (upnerf_field_fwd_f16x3): */
  const void* P16;               /* matrices of P as scaled fp16 (hi, lo) fragments, from upnerf_frag16 (forward set) */
  const int32_t* wexp;           /* [16] per-matrix exponents from upnerf_frag16 */
  const float* wk_xyz_dev;       /* [10] DEVICE or NULL: overrides wk_xyz (read at execution time, so a captured HIP graph
                                    follows the schedule from one replay to the next) */
  int32_t planes;                /* upnerf_field_fwd_f16x3 only: 0 or 2 = f16x3 (fp32-accurate hi/lo split, three MFMAs per
                                    product); 1 = f16 (fp16 weights and activations, one MFMA per product, fp32 accumulate:
                                    BASELINE.json configs[3]); the lo halves of P16 are then never read */
  int32_t tile_rows;             /* upnerf_field_fwd_f16x3 only: samples per workgroup.  0 or 64: four waves, two workgroups per CU.
                                    (128, round 3's software-pipelined trunk, left the library in round 4: UPNERF_EINVAL.)  The
                                    backward pass of the same evaluation must be given the same value (the hmask layout follows
                                    the tile).
                                    256 (planes = 1 only, S >= 32): the REGISTER-RESIDENT kernels of csrc/field16rr.hip -- eight waves
                                    of 32 samples whose activations stay in registers, every weight slab staged once per workgroup
                                    in an LDS ring by LDS-DMA.  Contract of that variant: P16 / PT16 from upnerf_frag16 with
                                    perm_fwd = perm_bwd = 1 and `wnorm` from the same call; EVERY per-sample tensor with more than
                                    4 floats per sample (x0, e, g1, g2, r1, h, h16, hmask and the backward pass's gz_*, gz16) has room
                                    for Mp = ceil(M / 256) * 256 rows (rows >= M are written with padding values); h16 / gz16 are in
                                    FRAGMENT order -- [layer][Mp / 32][k-block 0..15][lane 0..63][8] fp16, feature of element j of
                                    lane l in k-block s = 16 s + 8 (j / 4) + 4 (l / 32) + j % 4, row = 32 tile + l % 32 -- with one
                                    exponent per 32 rows (hexp / gzexp [D][Mp / 32]); hmask holds (D + 3) * Mp * 4 words in the
                                    kernel pair's own layout; `h` is [Mp][W] (last layer, h_last_only = 1) */
  const float* wnorm;            /* [64] row 1-norms from upnerf_frag16 (forward set at 0.., transposed set at 32..): required by the
                                    register-resident kernels (tile_rows = 256), must be NULL otherwise */
  /* fp16 STORAGE of the trunk activations (always in the f16 mode; an option in the f16x3 mode, where it rounds only the
   * operands of the weight gradients): halves what the pass writes and what the weight-gradient kernels read back
   * (upnerf_wgrad_f16p).  h16[l][m][k] = fp16(h_l[m][k] * 2^hexp[l][m / 64]): the content of the
   * LDS plane of the 64-sample tile, copied out as it stands, with the tile's power-of-two exponent beside it. */
  uint16_t* h16;                 /* [D][M][W] fp16 bits, or NULL (fp32 `h` as above) */
  int32_t* hexp;                 /* [D][ceil(M/64)] */
  int32_t h_last_only;           /* with h16: `h` (if non-NULL) receives layer D-1 only, as [M][W] fp32 (density-head and
                                    final-layer weight gradients read it) */
  void* x0f;                     /* reserved (was the 128-sample tiling's encoding scratch): ignored */
  uint16_t* e16;                 /* tile_rows = 256 only, or NULL: e as fp16 operand fragments [ceil(M/256) * 8][16][64][8] like one
                                    layer of h16 (then `e` may be NULL); upnerf_composite_fwd / _bwd and upnerf_wgrad_f16p read it */
  int32_t* eexp;                 /* [ceil(M/256) * 8] */
  uint16_t* g2_16;               /* the same for g2 (128 wide: [..][8][64][8]; then `g2` may be NULL) and for r1: compositing / */
  int32_t* g2exp;                /* upnerf_vec_wgrad_frag16 read them; the backward kernel works from the sign bits in hmask */
  uint16_t* r1_16;
  int32_t* r1exp;
  uint16_t* g1_16;               /* and for g1 (candidate_encoding.2's weight gradient reads it: upnerf_wgrad_f16p, 128-wide fragments) */
  int32_t* g1exp;
  uint8_t* h_lo8;                /* f16x3 mode with h16, or NULL: [D][M][W] bytes, the rounding residual of every h16 element in 1/32 of
                                    its tile's scaled unit (byte = round(32 lo) + 128): h16 + h_lo8 = the trunk activation to 2^-20
                                    of its tile's maximum in 3 bytes ("24-bit" weight-gradient operands, upnerf_wgrad_f24p_chain) */
  int64_t rows_capacity;         /* (ABI 9) tile_rows = 256: the number of rows every per-sample tensor passed here was allocated with.
                                    The register-resident kernels write whole 256-sample tiles: tensors need ceil(M / 256) * 256 rows.
                                    A caller that says so here gets UPNERF_EINVAL instead of a write past the end when it is less;
                                    0 = not stated (unchecked, as before) */
} upnerf_field_fwd_args;

int upnerf_field_fwd(const upnerf_layout* L, const upnerf_field_fwd_args* a, void* stream);
/* Same contract, contractions on the f16 matrix cores with the 3-term hi/lo split (fp32-level accuracy, see
 * csrc/common16.cuh).  W = 256 only; `P` is read for the vectors only (biases, wsig, wcsig, wr2: any ordering of P has
 * them in place); hmask needs D+1 slots (slot D = sign bits of g1). */
int upnerf_field_fwd_f16x3(const upnerf_layout* L, const upnerf_field_fwd_args* a, void* stream);

/* ---- a10: alpha compositing, forward (models/rendering.py:125-218) -------------------------------
 * Features are composited in the W-wide space of xyz_encoding_final / candidate_encoding and projected
 * once per ray by the caller (exact algebra: feat = W_f (sum w e) + b_f sum w). */
typedef struct {
  int32_t R, S, W;
  int32_t mode;                  /* 0: candidate+shared (sched==0), 1: both + rgb (0<sched<1),
                                    2: shared only (sched==1), 3: shared only, feature map (no candidate, sched<1) */
  const float* z;                /* [R][S] */
  const float* sigma_s;          /* [M] */
  const float* sigma_c;          /* [M] modes 0,1 */
  const float* rgb;              /* [M][3] modes 1,2,3 when sched>0 */
  int32_t has_rgb;
  const float* e;                /* [M][W] */
  const float* g2;               /* [M][W/2] modes 0,1 */
  /* per-sample outputs (also the saved state of the backward) */
  float* w_all;                  /* [M] alpha*T       -> c_weights      (modes 0,1) */
  float* w_sj;                   /* [M] alpha_s*T     (joint transmittance, modes 0,1) */
  float* w_cj;                   /* [M] alpha_c*T     (modes 0,1) */
  float* w_s;                    /* [M] alpha_s*T_s   -> s_weights */
  /* per-ray outputs */
  float* E_s;                    /* [R][W]   sum_i ws_feat_i e_i   (ws_feat = w_sj in modes 0,1; w_s in mode 3) */
  float* G_c;                    /* [R][W/2] sum_i w_cj g2_i       (modes 0,1) */
  float* sum_sfeat;              /* [R] sum_i ws_feat_i */
  float* t_weight;               /* [R] sum_i w_cj */
  float* c_depth;                /* [R] sum_i w_all z */
  float* s_depth;                /* [R] sum_i w_s z */
  float* rgb_map;                /* [R][3] sum_i w_s rgb_i */
  /* W = 256: e as the fp16 operand fragments of the register-resident field kernels instead of fp32 rows (then `e` is
   * ignored): upnerf_field_fwd_args.e16 / eexp */
  const uint16_t* e16;
  const int32_t* eexp;
  const uint16_t* g2_16;         /* with e16 (modes 0, 1): g2 the same way, 128 wide = 8 k-blocks per 32 samples (then `g2` is ignored) */
  const int32_t* g2exp;
  /* encode_feat = False (models/rendering.py:177-190): the shared colour composited with the JOINT-transmittance weights,
   * sum_i w_sj rgb_i -- the shared half of `c_rgb`.  NULL = not wanted; modes 0, 1 with has_rgb only.  (ABI 9) */
  float* rgb_joint_map;          /* [R][3] */
} upnerf_composite_fwd_args;

int upnerf_composite_fwd(const upnerf_composite_fwd_args* a, void* stream);

/* ---- a10 backward (division-free reverse scan) ------------------------------ */
typedef struct {
  int32_t R, S, W, mode, has_rgb;
  const float* z; const float* sigma_s; const float* sigma_c; const float* rgb;
  const float* e; const float* g2;
  const float* w_all; const float* w_sj; const float* w_cj; const float* w_s;
  /* upstream gradients (any may be NULL = zero) */
  const float* g_E_s;            /* [R][W] */
  const float* g_G_c;            /* [R][W/2] */
  const float* g_sum_sfeat;      /* [R] */
  const float* g_t_weight;       /* [R] */
  const float* g_c_depth;        /* [R] */
  const float* g_s_depth;        /* [R] */
  const float* g_rgb_map;        /* [R][3] */
  const float* g_w_all;          /* [M] */
  const float* g_w_s;            /* [M] */
  /* outputs */
  float* d_sigma_s;              /* [M] */
  float* d_sigma_c;              /* [M] modes 0,1 */
  float* d_rgb;                  /* [M][3] has_rgb */
  const uint16_t* e16;           /* as in upnerf_composite_fwd_args */
  const int32_t* eexp;
  const uint16_t* g2_16;
  const int32_t* g2exp;
  const float* g_rgb_joint_map;  /* [R][3] upstream gradient of upnerf_composite_fwd_args.rgb_joint_map, or NULL (ABI 9) */
} upnerf_composite_bwd_args;

int upnerf_composite_bwd(const upnerf_composite_bwd_args* a, void* stream);

/* ---- a7-a9 backward: data-gradient chain through the fused field (autograd of nerf.py:80-124) ----
 * Consumes the per-sample gradients from upnerf_composite_bwd plus the rank-1 feature terms
 * (w_feat_s[m] * g_E_s[ray], w_cj[m] * g_G_c[ray]) and writes the pre-activation gradient of every layer
 * (inputs of upnerf_wgrad) and d(xyz). */
typedef struct {
  int32_t R, S, use_cand, use_rgb, need_dxyz;
  const float* PT;               /* transposed parameter copies (layout t_*), fragment order */
  const float* P;                /* forward parameters (vectors wsig, wcsig, wr2) */
  const float* d_sigma_s; const float* d_sigma_c; const float* d_rgb;
  const float* sigma_s; const float* sigma_c; const float* rgb;
  const float* w_feat_s;         /* [M] weight multiplying e_i in the feature map (w_sj or w_s), or NULL */
  const float* w_cj;             /* [M] */
  const float* g_E_s;            /* [R][W] or NULL */
  const float* g_G_c;            /* [R][W/2] or NULL */
  const float* x0; const float* h; const float* g1; const float* g2; const float* r1;
  const uint64_t* hmask;         /* from upnerf_field_fwd */
  float* gmax;                   /* [16] or NULL: running max|.| of gz_h[0..D-1] (slots 0..D-1), gz_e (D), gz_g1 (D+1),
                                    gz_g2 (D+2), gz_r1 (D+3) */
  /* outputs */
  float* gz_h;                   /* [D][M][W] */
  float* gz_e;                   /* [M][W] */
  float* gz_g1; float* gz_g2;    /* [M][W/2] */
  float* gz_r1;                  /* [M][W/2] */
  float* dpre_sig_s;             /* [M] */
  float* dpre_sig_c;             /* [M] */
  float* dpre_rgb;               /* [M][4] (3 used) */
  float* dxyz;                   /* [M][3] (need_dxyz) */
  /* f16x3 variant only (upnerf_field_bwd_f16x3): */
  const void* PT16;              /* transposed set of upnerf_frag16 */
  const int32_t* wexp;           /* [16] */
  int32_t planes;                /* as in upnerf_field_fwd_args: 0 / 2 = f16x3, 1 = f16 */
  int32_t tile_rows;             /* as in upnerf_field_fwd_args; must equal the forward pass's */
  uint16_t* gz16;                /* [D][M][W] fp16 bits of gz_h, tile-scaled like h16 (then gz_h may be NULL).  tile_rows = 256
                                    with gz_e == NULL: [D + 1] layers, the last one d e in the same form (and gzexp [D + 1] rows) */
  int32_t* gzexp;                /* [D][ceil(M/64)] */
  void* xs;                      /* reserved (was the 128-sample tiling's scratch): ignored */
  float* tile_part;              /* NULL, or [ceil(M/64)][UPNERF_TILE_PART_STRIDE] (f16x3 variant, tile_rows 0 / 64 only): per-tile
                                    partial sums of what upnerf_vec_wgrad (dpre_sig_c x g2, dpre_rgb x r1) and upnerf_ray_sum
                                    (gz_g1, gz_r1) would re-read M x W/2 tensors for; finished by upnerf_tile_part_finish.
                                    tile_rows = 256: NULL, or [ceil(M/256) * 8][UPNERF_RR_PART_STRIDE]: the per-ray sums only, per
                                    32 samples, finished by upnerf_ray_part_finish */
  int32_t gz_rg_ld;              /* f16x3 variant: row stride (floats) of gz_r1 and gz_g1; 0 = W/2 (two dense tensors).  With
                                    gz_g1 = gz_r1 + W/2 and a stride of W the two form ONE [M][W] tensor [gz_r1 | gz_g1], whose
                                    weight gradient against e is one launch (upnerf_wgrad_f16x3_chain2); its running maximum is
                                    tracked in gmax slot D+4 */
  int32_t reserved_;
  const float* wnorm;            /* as in upnerf_field_fwd_args (tile_rows = 256: required; else NULL) */
  uint16_t* gz_rg16;             /* tile_rows = 256 with both heads, or NULL: [gz_r1 | gz_g1] as ONE 256-wide tensor of fp16 operand
                                    fragments (then gz_r1 / gz_g1 may be NULL: with tile_part their per-ray sums still leave) */
  int32_t* gzrgexp;              /* [ceil(M/256) * 8] */
  uint16_t* gz_g2_16;            /* tile_rows = 256, or NULL: gz_g2 as 128-wide fp16 operand fragments (then gz_g2 may be NULL) */
  int32_t* gzg2exp;
  uint8_t* gz_lo8;               /* f16x3 mode with gz16, or NULL: [D][M][W] residual bytes of gz16 (as upnerf_field_fwd_args.h_lo8) */
  int64_t rows_capacity;         /* (ABI 9) as in upnerf_field_fwd_args */
} upnerf_field_bwd_args;

/* Layout of one row of tile_part (floats): d w_csig [W/2] | d w_r2 [3][W/2] | sum dpre_sig_c, sum dpre_rgb[0..2] | 4 pad |
 * sums of gz_g1 over the tile's rows of ray slot 0, 1, 2 [3][W/2] | the same for gz_r1 [3][W/2]; ray slot of row i of tile t =
 * (64 t + i) / S - (64 t) / S.  W/2 = 128. */
#define UPNERF_TILE_PART_STRIDE 1288
/* Finishes the per-tile partial sums: rs_g1 / rs_r1 [R][128] = per-ray sums of gz_g1 / gz_r1 (upnerf_ray_sum's result),
 * d_wcsig [128], d_bcsig [1], d_wr2 [3][128], d_br2 [3] = upnerf_vec_wgrad's results; any output may be NULL.  Fixed summation
 * order (bitwise reproducible).  scratch: 128 * 520 floats. */
int upnerf_tile_part_finish(int R, int S, const float* tile_part, float* rs_g1, float* rs_r1, float* d_wcsig, float* d_bcsig,
                            float* d_wr2, float* d_br2, float* scratch, void* stream);

/* tile_rows = 256: one row of tile_part per 32 samples: sums of gz_r1 over the rows of the first / second ray of those 32 samples
 * [2][128], then the same for gz_g1 (S >= 32: at most two rays; the second block is only written when there is a second ray). */
#define UPNERF_RR_PART_STRIDE 512
/* rs_g1 / rs_r1 [R][128] from those rows, in a fixed order (either may be NULL). */
int upnerf_ray_part_finish(int R, int S, const float* tile_part, float* rs_g1, float* rs_r1, void* stream);

int upnerf_field_bwd(const upnerf_layout* L, const upnerf_field_bwd_args* a, void* stream);
/* f16x3 variant: W = 256 and S >= 32 (at most 3 rays per 64-sample tile); hmask from upnerf_field_fwd_f16x3. */
int upnerf_field_bwd_f16x3(const upnerf_layout* L, const upnerf_field_bwd_args* a, void* stream);

/* ---- weight gradients: dW[N][ldo] (+)= sum_m A[m][n] * B[m][k], db[n] = sum_m A[m][n] ------------
 * A [M][lda] (N columns used), B [M][ldb] (K columns used); K, N multiples of 32 (N <= 256, K <= 256).
 * `slabs` is scratch for nsplit partial results (nsplit * N * K floats); reduced in fixed order, so the
 * result is bitwise reproducible.  db may be NULL. */
int upnerf_wgrad(int M, const float* A, int lda, int N, const float* B, int ldb, int K,
                 float* dW, int ldo, float* db, float* slabs, int nsplit, void* stream);

/* Many small weight gradients in one launch + one fixed-order reduction (the per-ray layers of TransientNet and the
 * feature projections, M = rays: separately each is a 25 us launch that fills a quarter of the GPU).  Same arithmetic
 * as upnerf_wgrad (fp32 MFMA); N, K, lda, ldb, ldo multiples of 4, any N and K (cut into 128 x 128 blocks).
 * scratch: upnerf_wgrad_grouped_scratch(...) floats (a negative return is an error code). */
#define UPNERF_MAX_WGRAD_GROUPS 32
typedef struct {
  const float* A;                /* [M][lda], N columns used */
  const float* B;                /* [M][ldb], K columns used */
  float* dW;                     /* [N][ldo] */
  float* db;                     /* [N] or NULL */
  int32_t M, N, K, lda, ldb, ldo;
} upnerf_wgrad_group;
int upnerf_wgrad_grouped_scratch(const upnerf_wgrad_group* groups, int ngroups, int nsplit);
int upnerf_wgrad_grouped(const upnerf_wgrad_group* groups, int ngroups, float* scratch, int nsplit, void* stream);

/* Same contract, contraction on the f16 matrix cores: A and B are scaled by 2^*expo_a, 2^*expo_b (DEVICE ints, chosen so
 * that the scaled maxima are ~2^14).  planes 0 / 2 (f16x3): split into fp16 hi + lo parts, Ah Bh + Ah Bl + Al Bh
 * accumulated in fp32 -- fp32-level accuracy at 5.3x fewer matrix cycles than the fp32 MFMA (HBM-bound).  planes 1 (f16):
 * operands rounded to fp16, one MFMA per block, fp32 accumulate (the "f16" field mode). */
int upnerf_wgrad_f16x3(int M, const float* A, int lda, int N, const float* B, int ldb, int K,
                       float* dW, int ldo, float* db, float* slabs, int nsplit, const int* expo_a,
                       const int* expo_b, int planes, void* stream);

/* ---- TransientNet (models/transient_net.py:5-38) as one forward and one backward launch: feat_dim 384, hidden width 256,
 * transient embedding width 128 (the reference's defaults), one row per ray.  Weights in the nn.Linear layout ([out][in],
 * row-major); fp32 MFMA, fp32 accumulate.  The forward pass stores what the backward pass and the weight gradients read. */
typedef struct {
  int32_t R; float beta_min;
  const float* feat;             /* [R][384] */
  const float* t_emb;            /* [R][128] rows of embedding_t */
  const float* w0; const float* b0;   /* feat_encoder.0  [256][384], [256] */
  const float* w1; const float* b1;   /* feat_encoder.2  [256][256] */
  const float* w2; const float* b2;   /* feat_encoder.4 */
  const float* w3; const float* b3;   /* feat_encoder.6 */
  const float* wf; const float* bf;   /* final_encoder   [256][256] */
  const float* wt; const float* bt;   /* t_encoder.0     [128][384] over [final_encoding | t_emb] */
  const float* wa; const float* ba;   /* alpha_layer.0   [1][256] */
  const float* wb; const float* bb;   /* beta_layer.0    [1][128] */
  const float* wr; const float* br;   /* rgb_layer.0     [3][128] */
  float* h;                      /* [4][R][256] post-ReLU outputs of the four feat_encoder layers */
  float* e;                      /* [R][256] final_encoding */
  float* t;                      /* [R][128] post-ReLU output of t_encoder */
  float* alpha; float* rgb; float* beta;   /* [R], [R][3], [R]: the module's outputs */
  float* spre;                   /* [R] pre-activation of the beta head */
} upnerf_transient_args;
typedef struct {
  const float* d_alpha; const float* d_rgb; const float* d_beta;   /* [R], [R][3], [R]; NULL = zero */
  float* dz_heads;               /* [R][8]: pre-activation gradients of alpha, beta, rgb[3] (3 unused) */
  float* gz_t;                   /* [R][128] */
  float* gz_e;                   /* [R][256] */
  float* gz_h;                   /* [4][R][256] */
  float* g_temb;                 /* [R][128] or NULL */
  float* g_feat;                 /* [R][384] or NULL */
} upnerf_transient_grads;
int upnerf_transient_fwd(const upnerf_transient_args* a, void* stream);
/* data gradients only; the weight gradients are upnerf_wgrad_grouped over (gz_*, stored inputs) */
int upnerf_transient_bwd(const upnerf_transient_args* a, const upnerf_transient_grads* g, void* stream);

/* Chained form of upnerf_wgrad_f16x3: the slabs of one weight gradient are summed by the first workgroups of the NEXT
 * weight-gradient launch (a prologue that costs it a few microseconds) instead of a reduction launch of their own (20+ us
 * each, 40 per step).  `pending` describes the problem whose slabs are written but not summed (nsplit == 0: none): the call
 * sums it -- in its kernel's prologue when the grid is large enough, by a reduction launch otherwise -- and replaces it by
 * the description of ITS problem.  upnerf_wgrad_finish sums what is pending and clears it.  The caller alternates between two
 * slab buffers: `slabs` must differ from pending->slabs.  Same arithmetic and summation order as upnerf_wgrad_f16x3. */
typedef struct {
  const float* slabs; const float* bslabs;
  float* dW; float* db;
  int32_t N, K, TN, TK, nsplit, ldo, rblocks;
  int32_t n2;                    /* > 0: rows n >= n2 of the result go to dW2[(n - n2) * ldo2 + k], db2[n - n2] */
  float* dW2; float* db2;
  int32_t ldo2, pad;
  /* a vector head that shares B with the problem (upnerf_wgrad_f16x3_chain_v): per-split partial sums [nsplit][K + 4]
   * (vslabs: K sums of v[m] B[m][k], then the sum of v) -> dv [K], dbv [1]; NULL: none */
  const float* vslabs; float* dv; float* dbv;
} upnerf_wgrad_pending;
int upnerf_wgrad_f16x3_chain(int M, const float* A, int lda, int N, const float* B, int ldb, int K, float* dW, int ldo,
                             float* db, float* slabs, int nsplit, const int* expo_a, const int* expo_b, int planes,
                             upnerf_wgrad_pending* pending, void* stream);
/* the same with a result split by rows between two destinations (two layers that share B and whose A operands sit side by
 * side in one tensor: the colour and candidate heads' first layers, both fed by e) */
int upnerf_wgrad_f16x3_chain2(int M, const float* A, int lda, int N, const float* B, int ldb, int K, float* dW, int ldo,
                              float* db, int n2, float* dW2, int ldo2, float* db2, float* slabs, int nsplit, const int* expo_a,
                              const int* expo_b, int planes, upnerf_wgrad_pending* pending, void* stream);
/* upnerf_wgrad_f16x3_chain for a 256 x 256 problem, plus the gradient of a 1-wide head fed by the same B rows, in the same pass
 * over B:  dv[k] = sum_m v[m] B[m][k],  dbv[0] = sum_m v[m]  (the shared density head: share_sigma reads the last trunk
 * activation, which is also the B operand of xyz_encoding_final's weight gradient -- models/nerf.py:89, 93 -- so the separate
 * upnerf_vec_wgrad launch and its second read of that tensor, 1 KB per sample, go away).  fp32 arithmetic for the vector (as
 * upnerf_vec_wgrad), fixed summation order.  slabs: nsplit * (256 * 256 + 256 + 260) floats.  N = K = 256, planes = 2 only. */
int upnerf_wgrad_f16x3_chain_v(int M, const float* A, int lda, int N, const float* B, int ldb, int K, float* dW, int ldo,
                               float* db, const float* v, float* dv, float* dbv, float* slabs, int nsplit, const int* expo_a,
                               const int* expo_b, int planes, upnerf_wgrad_pending* pending, void* stream);
/* the same on the fragment-ordered fp16 operands of the register-resident field kernels (upnerf_wgrad_f16p_chain with
 * b_is_f16 = 3, N = K = 256): replaces upnerf_vec_wgrad_frag16 for the shared density head of the f16 mode */
int upnerf_wgrad_f16p_chain_v(int M, const uint16_t* A16, const int32_t* aexp, const uint16_t* B16, const int32_t* bexp, float* dW,
                              int ldo, float* db, const float* v, float* dv, float* dbv, float* slabs, int nsplit, const int* expo_a,
                              const int* expo_b, upnerf_wgrad_pending* pending, void* stream);
int upnerf_wgrad_finish(upnerf_wgrad_pending* pending, void* stream);

/* Same contraction for the f16 field mode with fp16-STORED operands: A16 [M][lda] fp16 bits scaled per 64-row tile by
 * 2^aexp[m / 64] (upnerf_field_bwd_f16x3's gz16 / gzexp); B either fp16 the same way (b_is_f16 = 1: B16 / bexp, from h16 /
 * hexp) or fp32 row-major (b_is_f16 = 0: x0).  b_is_f16 | 2: the fp16 operands (A16, and B16 when bit 0 is set) are the operand
 * FRAGMENTS of the register-resident field kernels (upnerf_field_fwd_args.tile_rows = 256: [32-row tile][k-block][lane][8], one
 * exponent per 32 rows, N = 256 and, for a fp16 B, K = 256; lda / ldb are ignored for them).  Operands are brought to the tensor-wide exponents *expo_a / *expo_b on load
 * (exact power-of-two scaling in fp16), one MFMA per block, fp32 accumulate.  Reads 1 KB per sample and layer instead of 2. */
int upnerf_wgrad_f16p(int M, const uint16_t* A16, int lda, const int32_t* aexp, int N, const void* B, int ldb,
                      const int32_t* bexp, int b_is_f16, int K, float* dW, int ldo, float* db, float* slabs, int nsplit,
                      const int* expo_a, const int* expo_b, void* stream);
/* chained like upnerf_wgrad_f16x3_chain, on the same pending record (a run may mix the two kinds of launches) */
int upnerf_wgrad_f16p_chain(int M, const uint16_t* A16, int lda, const int32_t* aexp, int N, const void* B, int ldb,
                            const int32_t* bexp, int b_is_f16, int K, float* dW, int ldo, float* db, int n2, float* dW2, int ldo2,
                            float* db2, float* slabs, int nsplit, const int* expo_a, const int* expo_b,
                            upnerf_wgrad_pending* pending, void* stream);  /* n2 > 0: rows [n2, N) -> dW2 / db2 (as chain2) */

/* A 256 x 256 weight gradient from PRODUCER-SPLIT operands (round 6): A16 / Alo16 and B16 / Blo16 are the (hi, lo) fp16 planes of the
 * f16x3 field kernels' tiles, row-major [M][256], value = (hi + lo) * 2^-exp[m / 64] (upnerf_field_fwd_args.h16 / h_lo16 / hexp,
 * upnerf_field_bwd_args.gz16 / gz_lo16 / gzexp) -- the 4 bytes per element of the fp32 rows, which ARE hi + lo, already split.  The
 * kernel stages them by LDS-DMA (no conversion pass, no staging registers, three 16-row chunks in flight) and contracts as
 * upnerf_wgrad_f16x3 does (three MFMAs per block); dW / db / slabs / pending as upnerf_wgrad_f16x3_chain.  M % 64 == 0
 * (UPNERF_EUNSUP otherwise: the caller keeps the fp32 rows for such shapes). */
int upnerf_wgrad_planes_chain(int M, const uint16_t* A16, const uint16_t* Alo16, const int32_t* aexp, const uint16_t* B16,
                              const uint16_t* Blo16, const int32_t* bexp, float* dW, int ldo, float* db, float* slabs, int nsplit,
                              const int* expo_a, const int* expo_b, upnerf_wgrad_pending* pending, void* stream);
/* "24-bit" operands (f16x3 mode): A16 / Alo8 [M][lda] and, with b_is_f16 = 1, B16 / Blo8 [M][ldb] hold hi + lo8 as the f16x3 field
 * kernels write them (h16 + h_lo8, gz16 + gz_lo8; exponents per 64 rows); with b_is_f16 = 0 B is fp32 rows, split into hi + lo
 * on load.  Three MFMAs per block as upnerf_wgrad_f16x3: the operands are exact to 2^-20 of their tile's maximum.  256 x 256
 * and 256 x 64 blocks; chained on the same pending record as the other two kinds. */
int upnerf_wgrad_f24p_chain(int M, const uint16_t* A16, const uint8_t* Alo8, int lda, const int32_t* aexp, int N, const void* B,
                            const uint8_t* Blo8, int ldb, const int32_t* bexp, int b_is_f16, int K, float* dW, int ldo, float* db,
                            float* slabs, int nsplit, const int* expo_a, const int* expo_b, upnerf_wgrad_pending* pending, void* stream);

/* dw[c][k] = sum_m v[m*ldv + c] * X[m][k], c < nvec <= 3; dbv[c] = sum_m v[m*ldv + c]   (N=1/3 heads);
 * K in {32, 64, 128, 256}; scratch: nsplit * 4 * (K+1) floats */
int upnerf_vec_wgrad(int M, const float* v, int ldv, int nvec, const float* X, int ldx, int K,
                     float* dw /*[nvec][K]*/, float* dbv /*[nvec]*/, float* scratch, int nsplit, void* stream);

/* upnerf_vec_wgrad (same v / ldv / nvec / dw / dbv) against a 256- or 128-wide fp16 tensor in the operand-fragment order of the
 * register-resident field kernels (upnerf_field_fwd_args.tile_rows = 256: h16 / hexp of one layer, g2_16, r1_16; padded to whole
 * 32-sample tiles); scratch: nsplit * 4 * (K + 1) floats.  Fixed summation order. */
int upnerf_vec_wgrad_frag16(int M, const float* v, int ldv, int nvec, const uint16_t* X16, const int32_t* xexp, int K, float* dw,
                            float* dbv, float* scratch, int nsplit, void* stream);  /* nvec <= 3, K = 256 or 128 */

/* out[r][c] = sum_{i<S} X[(r*S+i)][c]  (per-ray sums of a per-sample tensor; embedding-row gradients) */
int upnerf_ray_sum(int R, int S, const float* X, int C, float* out, void* stream);
/* (d_o, d_d)[r] = (sum_i dxyz_i, sum_i z_i dxyz_i)    */
int upnerf_ray_geom_bwd(int R, int S, const float* dxyz, const float* z, float* d_o, float* d_d, void* stream);

/* ---- f1: train-split ray sampler (datasets/phototourism.py:420-454, PhototourismDataset.__getitem__ + default
 * collate): gathers a batch of rays from the flat per-ray buffers, resident in HBM, and interpolates each ray's
 * feature bilinearly from its image's [h][h][C] map with the reference's weights (incl. its vanishing weights on the
 * last row / column).  Bit-exact with the reference.  inv_depths / feats may be NULL (depth / feature supervision off). */
typedef struct {
  int32_t R, h, C;                 /* rays in the batch, feature-map side, channels */
  const int64_t* idx;              /* [R] indices into the flat ray buffers */
  const float* all_ray_infos;      /* [N][3] near, far, image index */
  const float* all_directions;     /* [N][3] */
  const float* all_rgbs;           /* [N][3] */
  const float* all_pxl_coords;     /* [N][2] (row, column) in [0, 1] */
  const float* all_inv_depths;     /* [N] or NULL */
  const float* feat_maps;          /* [I][h][h][C] or NULL */
  const float* poses;              /* [I][3][4] camera-to-world per image index */
  float* ray_infos;                /* [R][2] */
  float* directions;               /* [R][3] */
  int64_t* img_idx;                /* [R] */
  float* c2w;                      /* [R][3][4] */
  float* rgbs;                     /* [R][3] */
  float* feats;                    /* [R][C] or NULL */
  float* inv_depths;               /* [R] or NULL */
} upnerf_gather_rays_args;
int upnerf_gather_rays(const upnerf_gather_rays_args* a, void* stream);

/* ---- dense gradient of an embedding table (autograd of nn.Embedding(img_idx): the per-image appearance / candidate /
 * transient rows, se3_refine and depth_scale of models/nerf_system.py:79-91): out[n][:] = sum_{r: idx[r]==n} g[r][:],
 * rows without a hit are written as zeros; summation in increasing r (bitwise reproducible).  dim <= 256. */
int upnerf_embed_bwd(int R, int N, int dim, const int64_t* idx, const float* g, float* out, void* stream);
/* The same for up to UPNERF_MAX_EMBED_GROUPS tables of N rows gathered with the SAME idx (every per-image table of a
 * training step): one scan of idx serves all of them. */
#define UPNERF_MAX_EMBED_GROUPS 8
typedef struct {
  const float* g;                /* [R][dim] gradient of the gathered rows */
  float* out;                    /* [N][dim] dense table gradient */
  int32_t dim;                   /* <= 256 */
} upnerf_embed_group;
int upnerf_embed_bwd_grouped(int R, int N, const int64_t* idx, const upnerf_embed_group* groups, int ngroups, void* stream);
/* The forward side of the same tables (nn.Embedding.forward, models/nerf_system.py:79-91 called at :160, :170 and in
 * rendering.py:177-181 / transient_net.py:31): rows[r][:] = table[idx[r]][:] for up to UPNERF_MAX_EMBED_GROUPS tables of N
 * rows in one launch instead of an index_select launch per table.  An index outside [0, N) is UPNERF_EINVAL-free on the host
 * (it lives in device memory): its row is written as NaN, which no test or loss survives unnoticed. */
typedef struct {
  const float* table;            /* [N][dim] */
  float* rows;                   /* [R][dim] gathered rows */
  int32_t dim;                   /* <= 256 */
} upnerf_embed_rows_group;
int upnerf_embed_fwd_grouped(int R, int N, const int64_t* idx, const upnerf_embed_rows_group* groups, int ngroups, void* stream);


/* ---- generic fp32 MFMA linear layer: C[M][N] = act(A[M][K] . B[N][K]^T + bias) --------------------
 * (TransientNet, models/transient_net.py:27-38, and the per-ray feature projection.)  K multiple of 8,
 * act bit 0: ReLU; act bit 1: B is given as [K][N] (row stride ldb), i.e. C = act(A . B + bias) -- the data-gradient
 * form, no transposed copy needed.  N arbitrary (ldb/ldc are row strides). */
int upnerf_linear(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                  const float* bias, float* C, int ldc, int act, void* stream);
/* Matrix-vector products in a fixed summation order (the bias fold of the colour layer, both directions):
 * trans = 0: y[m] = add[m] + sum_k A[m][k] x[k], m < M;  trans = 1: y[k] = add[k] + sum_m A[m][k] x[m], k < K.  add may be NULL. */
int upnerf_matvec(int M, int K, const float* A, int lda, const float* x, const float* add, float* y, int trans, void* stream);
/* p[0 .. n) = 0 (p 16-byte aligned): the one fill of a training step's zero arena (upnerf_amd/zero_pool.py). */
int upnerf_zero(float* p, long long n, void* stream);
/* out_j[i] = a_j[i] + b_j[i], i < n_j, for up to UPNERF_MAX_ADD_PAIRS tensors in one launch (the gradient sum of a tensor with two
 * consumers: ops._Fanout; fp32 addition is commutative, so the result is autograd's own a + b bit for bit). */
#define UPNERF_MAX_ADD_PAIRS 8
typedef struct upnerf_add_pair {
  const float* a;
  const float* b;
  float* out;   /* may alias a or b */
  int32_t n;
} upnerf_add_pair;
int upnerf_add_pairs(const upnerf_add_pair* pairs, int npairs, void* stream);
/* The backward of the bias fold in one launch: y[k] = sum_m A[m][k] x[m] (upnerf_matvec, trans = 1, add = NULL, bit for bit) AND the
 * rank-1 update R[m][k] += x[m] v[k] (m < M, k < K; row stride ldr) that the caller used to issue as a separate addr_ launch
 * (d W_r1[:, :F] += g_br1 (x) b_feat, the second term of the folded colour layer's gradient). */
int upnerf_matvec_rank1(int M, int K, const float* A, int lda, const float* x, float* y, float* R, int ldr, const float* v,
                        void* stream);

/* ---- a15 + a17: depth-prior affine (models/nerf_system.py:169-177) fused with UPNeRFLoss (losses.py:21-64) --------
 * Per-ray inputs only ([R], [R,3], [R,F]); any absent tensor is NULL.  `terms` receives the 8 loss terms in the order
 * l_depth_c, l_feat_c, l_rgb_c, l_depth_f, l_feat_f, l_rgb_f, l_beta, l_alpha (zero where a term does not exist in
 * the phase).  bwd: g_terms[8] are the upstream gradients of the terms (device memory); every non-NULL d_* receives
 * the gradient of sum_k g_terms[k] * terms[k]. */
typedef struct {
  int32_t R, F, fine, has_tw;            /* rays, feature width, fine network present, t_weight_* present */
  float sched, depth_mult, alpha_reg, near, far;
  const float* depth_direct;             /* [R] depth targets given directly (losses.py calling convention), or NULL: */
  const float* inv_depth;                /* [R] mono-depth prior, and */
  const float* depth_scale_rows;         /* [R][2] per-image (scale, shift) rows -> target computed here (a15) */
  const float* s_depth_c; const float* s_depth_f;     /* [R] */
  const float* t_weight_c; const float* t_weight_f;   /* [R] (treated as constants, losses.py:27,48) */
  const float* feat_c; const float* feat_f; const float* feat_gt;   /* [R][F] */
  const float* rgb_c; const float* rgb_f; const float* rgb_gt;      /* [R][3] */
  const float* beta; const float* alpha;                             /* [R] */
  const float* sched_dev;                /* [1] DEVICE or NULL: the multiplier m of the terms is read from here at execution
                                            time (graph replay); `sched` then only selects the phase (== 0, in (0,1), == 1) */
  /* (ABI 9) the sum of the terms a phase uses, as part of the same two launches instead of a select + reduce + product on the
   * caller's side: bit k of term_mask = term k counts (`sum(loss_d.values())`, models/nerf_system.py:183).  upnerf_loss_fwd
   * writes total[0] = sum of the masked terms in term order when `total` is given; upnerf_loss_bwd adds g_total[0] to the
   * upstream gradient of every masked term when `g_total` is given (g_terms may then be NULL). */
  int32_t term_mask, reserved_;
  float* total;                          /* [1] or NULL */
  const float* g_total;                  /* [1] DEVICE or NULL (backward only) */
} upnerf_loss_args;
int upnerf_loss_fwd(const upnerf_loss_args* a, float* depth_out /*[R]*/, float* terms /*[8]*/,
                    float* scratch /*[64*8]*/, void* stream);
typedef struct {
  float* d_depth_scale_rows;             /* [R][2] */
  float* d_depth;                        /* [R] gradient w.r.t. the depth target (depth_direct mode) */
  float* d_s_depth_c; float* d_s_depth_f;
  float* d_feat_c; float* d_feat_f;
  float* d_rgb_c; float* d_rgb_f;
  float* d_beta; float* d_alpha;
} upnerf_loss_grads;
int upnerf_loss_bwd(const upnerf_loss_args* a, const float* g_terms /*[8] device*/, const upnerf_loss_grads* g,
                    void* stream);

/* ---- parameter re-layout: row-major matrices of `src` -> MFMA fragment order in `dst` (optionally transposed) ------
 * One launch re-packs every matrix of a field (forward copies and the transposed copies the backward chain reads).
 * Logical matrix X[r][c], r < rows (multiple of 32), c < cols:
 *     X[r][c] = transpose ? src[src_off + c*src_ld + r] : src[src_off + r*src_ld + c]
 * is written as columns [dst_k0, dst_k0+cols) of a fragment-ordered [rows][dst_kp] matrix at dst + dst_off. */
typedef struct {
  int32_t src_off, src_ld, transpose, rows, cols, dst_off, dst_kp, dst_k0;
} upnerf_frag_desc;
#define UPNERF_MAX_FRAG_DESC 32
int upnerf_frag_copy(const float* src, float* dst, const upnerf_frag_desc* descs /*host*/, int ndesc, void* stream);

/* ---- packed parameter buffer P <-> the network's named parameters (one launch instead of ~40 cat / pad / copy launches) ----
 * pack   (unpack = 0): P[dst_off + r*dst_ld + c] = ptr[r*src_ld + c] for c < cols (padding columns are left alone: start
 *                      from a zeroed P); with `accumulate` the value is added to what P holds (bias of the folded layer).
 * unpack (unpack = 1): ((float*)ptr)[r*src_ld + c] = P[dst_off + r*dst_ld + c]   -- the backward of pack on dP.
 * Descriptors are host memory, `ptr` are device pointers; at most UPNERF_MAX_PACK_DESC per call. */
#define UPNERF_MAX_PACK_DESC 48
typedef struct {
  const float* ptr;              /* device pointer of the parameter (pack: source, unpack: destination) */
  int32_t rows, cols, src_ld;    /* logical shape and row stride of the parameter */
  int32_t dst_off, dst_ld;       /* float offset and row stride inside P (dst_ld >= cols) */
  int32_t accumulate;            /* pack only */
} upnerf_pack_desc;
int upnerf_pack(float* P, const upnerf_pack_desc* descs, int ndesc, int unpack, void* stream);

/* ---- f16x3 weight re-layout: every matrix of `src` -> scaled fp16 (hi, lo) pairs in MFMA fragment order ------------
 * Same descriptors as upnerf_frag_copy plus `exp_id`: matrices sharing an id share one power-of-two exponent
 * (2^14 / max|.| over all their elements), written to wexp[exp_id].  Destination element (r, k) of a [rows][dst_kp]
 * matrix at float offset dst_off:  byte dst_off*4 + (((r/32)*(dst_kp/16) + k/16)*2 + plane)*1024
 *                                       + (((k/8)%2)*32 + r%32)*16 + (k%8)*2,   plane 0 = hi, 1 = lo.
 * `amax_scratch` [16] floats is zeroed and used inside.
 * perm_fwd / perm_bwd = 1 write the k index inside every 16-deep block in the order in which a converted 32x32 MFMA result
 * presents its rows as the next product's operand -- element j of lane half h holds k = 8(j/4) + 4h + j%4 instead of
 * 8h + j: (k%8) above becomes ((k%16)/8)*4 + k%4 and ((k/8)%2) becomes ((k%16)/4)%2 -- what the register-resident field
 * kernels (upnerf_field_fwd_f16x3 with activations chained through registers) read.
 * wnorm (DEVICE [64] or NULL): receives max_r sum_c |X[r][c]| per descriptor, forward set at [0..), transposed set at [32..). */
typedef struct {
  int32_t src_off, src_ld, transpose, rows, cols, dst_off, dst_kp, dst_k0, exp_id;
} upnerf_frag16_desc;
int upnerf_frag16(const float* src, void* dst_fwd, void* dst_bwd, const upnerf_frag16_desc* fwd, int nfwd,
                  const upnerf_frag16_desc* bwd, int nbwd, float* amax_scratch /*[16]*/, int32_t* wexp /*[16]*/,
                  int perm_fwd, int perm_bwd, float* wnorm, void* stream);

/* ---- a18: fused Adam on a flat fp32 buffer (torch.optim.Adam semantics, utils/optim.py:20-33) ----
 * step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t), both formed by the host in double precision and rounded
 * to fp32 (as torch forms them).  dyn2 (DEVICE [step_size, bc2_sqrt], or NULL) overrides the by-value pair at execution
 * time: a captured graph follows the step count and the learning-rate schedule from one replay to the next. */
int upnerf_adam(int64_t n, float* p, const float* g, float* m, float* v, float beta1, float beta2, float eps,
                float step_size, float bc2_sqrt, const float* dyn2, void* stream);
/* The same update over ndesc pieces of the flat buffers, each with its gradient wherever autograd left it: piece j covers
 * elements [off, off + n) of p / m / v and reads the n floats at g.  Bitwise the result of gathering the gradients first. */
#define UPNERF_MAX_ADAM_DESC 96
typedef struct { const float* g; int32_t off, n; } upnerf_adam_desc;
int upnerf_adam_gather(float* p, float* m, float* v, const upnerf_adam_desc* descs /*host*/, int ndesc, float beta1, float beta2,
                       float eps, float step_size, float bc2_sqrt, const float* dyn2, void* stream);

/* ---- per-step scalars for captured HIP graphs: dst[i] = vals[i], i < n <= UPNERF_MAX_SCALARS --------------------------
 * `vals` is HOST memory, copied into the kernel arguments at call time (no staging buffer whose lifetime the caller would
 * have to manage, no host-device synchronisation): one small launch in front of every graph replay carries the learning
 * rates, Adam bias corrections, BARF band weights and the schedule multiplier of that step. */
#define UPNERF_MAX_SCALARS 96
int upnerf_set_scalars(float* dst, int n, const float* vals, void* stream);

/* out[i] = 14 - ceil(log2(max(maxima[i], 1e-30))), i < n <= 64: the power-of-two exponents that upnerf_wgrad_f16x3 /
 * upnerf_wgrad_f16p take (expo_a / expo_b) from the maxima the field kernels track in `amax` / `gmax`; device to device, no
 * host synchronisation (replaces rendering.py's five ATen launches per table). */
int upnerf_scale_exponents(const float* maxima, int n, int32_t* out, void* stream);

#ifdef UPNERF_STAMPS
/* Diagnostic build only (make -C upnerf_amd/csrc stamps -> libupnerf_hip_stamps.so, never the shipped library): per-phase
 * shader-clock sums accumulated by the f16x3 field kernels; out16[0..7] forward trunk phases, [8..15] backward stages. */
int upnerf_stamps_read(unsigned long long* out16, int reset);
/* same for the register-resident forward kernel (field16r.hip): 8 phase sums of its slab loop */
int upnerf_stamps_read_r(unsigned long long* out8, int reset);
#endif

#ifdef __cplusplus
}
#endif
#endif /* UPNERF_HIP_H */
