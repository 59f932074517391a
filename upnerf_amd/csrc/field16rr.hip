// Register-resident field kernels of the fp16 mode (BASELINE.json configs[3]: fp16 MLP weights on MFMA): the activations of a
// sample never leave the registers of the wave that owns it; what streams through LDS is the WEIGHTS -- once per workgroup of
// 256 samples, by LDS-DMA, shared by its eight waves (north_star: "LDS staging of MLP weights").
//
// Why (profiles/r04_pmc_path_summary.md): the tile-in-LDS kernels (csrc/field16.hip) pull every weight fragment L2 -> registers
// once per 64 samples; a CU's texture-data return path then runs 0.81-0.84 busy at 19-22 B/clk of fragments over the WHOLE
// kernel, in the fp16 mode at a third of the matrix work (MFMA busy 0.23).  Only fewer fragment bytes per sample move that.
// Here a weight byte enters the CU once per 256 samples (a quarter of the fragment traffic of a 64-sample tile) and is read
// from LDS by the eight waves (128 B/clk of ds_read_b128: half the LDS rate).
//
// Reference behaviour: models/nerf.py:80-124 (NeRF.forward) + 126-147 (positional_encoding), evaluated by
// models/rendering.py:102-122; the backward kernel is the autograd of the same lines, stage for stage as
// field16.hip:field16_bwd_kernel.
//
// How the layers chain without LDS (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand"): the contraction
// is issued transposed, D^T[n][m] = sum_k W[n][k] X[m][k] (weights = A operand), so a lane holds ITS sample (column m =
// lane & 31) and, in the 16 registers of a 32 x 32 result, rows n = 8 (r / 4) + 4 (lane / 32) + r % 4 of the feature tile.
// Converted to fp16 in place, registers 0..7 / 8..15 ARE the B-operand fragments of k-blocks 2j / 2j + 1 of the next layer,
// with the k order inside a 16-block permuted to 8 (j / 4) + 4 h + j % 4 -- upnerf_frag16(perm = 1) writes the weights in
// that order (both the forward and the transposed set).  A wave owns 32 samples; two waves share a SIMD (<= 256 registers).
//
// Exponents.  Activations travel as fp16 value * 2^e with one exponent per WAVE (32 samples) and stage, known BEFORE the
// stage runs so that tiles are converted as they complete: |W x + b|_inf <= wnorm |x|_inf + |b|_inf with wnorm = max_n sum_k
// |W[n][k]| from upnerf_frag16 and |x|_inf the wave's exact input maximum (tracked by the epilogue).  The bound is loose by the
// usual gap between the 1-norm bound and the attained maximum (2^3 .. 2^5): fp16 keeps 11 bits down to 2^-28 of the bound.
//
// Weight stream.  One slab = one 32-feature output tile of one matrix = K/16 k-blocks x 1 KiB (the hi planes of the
// fragment buffer).  The eight waves DMA the slab's chunks (global_load_lds_dwordx4, 1 KiB per wave instruction) into a ring
// of four LDS slots, three slabs ahead of the MFMAs; per slab ONE counted s_waitcnt vmcnt + ONE raw s_barrier.  vmcnt retires
// in order, so the count is exact: every vector-memory instruction of the slab loop is issued unconditionally by every wave
// (all per-sample tensors are padded to whole workgroup tiles) and counted where it is issued.  Under-counting is safe (it
// only waits longer), over-counting is not: conditional stores are never counted.
//
// What leaves the kernel (all written in whole 1 KiB pieces):
//   h16 / gz16   trunk activations / pre-activation gradients as the operand fragments themselves -- [layer][32-row tile]
//                [k-block 0..15][lane][8 fp16], the tile's exponent beside them (hexp / gzexp [layer][tile]) -- what
//                upnerf_wgrad_f16p(frag = 1) contracts;
//   fp32 rows    (x0, e, g1, g2, r1, h_{D-1}; gz_e, gz_g1, gz_g2, gz_r1) through a 4 KiB per-wave LDS transposer, 8 rows x
//                128 B per store instruction;
//   hmask        ReLU sign bits, 128 per lane and layer (two 64-bit words), in this kernel pair's own layout.
#include "common16.cuh"
#include <type_traits>

#define RR_NSLOT 4                 // ring slots
#define RR_AHEAD 3                 // slabs in flight ahead of the one being contracted
#define RR_MAXKB 21                // k-blocks of the widest matrix row (colour head: 256 + 80)
#define RR_SLOT (RR_MAXKB * 1024)  // bytes per ring slot
#define RR_STG 4096                // per-wave transposer: 32 rows x 32 fp32
#ifndef RR_PF
#define RR_PF 2                    // weight fragments requested from LDS this many k-blocks ahead
#endif
#ifndef RR_FILL_VALU
#define RR_FILL_VALU 7
#endif
#define RR_PE_LD 68                // floats per row of the encoding exchange scratch

// staged vectors (floats): trunk biases [8][256], final bias, rgb1 / cand1 / cand2 biases, w_sigma, w_rgb2 [3][128],
// w_csigma, then per-vector maxima |b|_inf [16]
#define RR_V_BE 2048
#define RR_V_BR1 2304
#define RR_V_BC1 2432
#define RR_V_BC2 2560
#define RR_V_WSIG 2688
#define RR_V_WR2 2944
#define RR_V_WCSIG 3328
#define RR_V_BMAX 3456
#define RR_V_TOTAL 3472

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int NW>
struct RRCfg {
  static constexpr int THREADS = 64 * NW, TILE = 32 * NW;
  static constexpr int MAXR = TILE / 32 + 1;  // rays a tile can touch when S >= 32
  static constexpr int RING = RR_NSLOT * RR_SLOT;
  static constexpr int STG0 = RING;
  static constexpr int VEC0 = STG0 + NW * RR_STG;
  static constexpr int ROW0 = VEC0 + RR_V_TOTAL * 4;          // per-ray side rows
  static constexpr int ROWF = 256 + 128;                      // floats per ray slot (fwd: aux 80 + cand 16; bwd: g_E_s 256 + g_G_c 128)
  static constexpr int INT0 = ROW0 + MAXR * ROWF * 4;         // small integer tables
  static constexpr int LDS = INT0 + 64 * 4;
  static_assert(NW * 32 * RR_PE_LD * 4 <= RING, "encoding exchange scratch lives in the ring");
};

__device__ __forceinline__ float pow2r(int n) { return ldexpf(1.0f, n); }

// maximum of non-negative floats over the wave as a wave-uniform value (DPP butterflies + four readlanes)
__device__ __forceinline__ float wave_max_rr(float m) {
  int v = __builtin_bit_cast(int, m);
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false));
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return __builtin_bit_cast(float, max(max(a, b), max(c, d)));
}

// s_waitcnt vmcnt(n), n wave-uniform (values above 62 wait for 62); expcnt / lgkmcnt untouched.  gfx9 encoding:
// vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
__device__ __forceinline__ void wait_vmcnt(int n) {
#ifdef RR_SAFE_WAIT  // diagnostic: drain everything (A/B against the counted waits)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  return;
#endif
#define RR_W(N) case N: __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14)); break;
  switch (n) {
    RR_W(0) RR_W(1) RR_W(2) RR_W(3) RR_W(4) RR_W(5) RR_W(6) RR_W(7) RR_W(8) RR_W(9) RR_W(10) RR_W(11) RR_W(12) RR_W(13) RR_W(14)
    RR_W(15) RR_W(16) RR_W(17) RR_W(18) RR_W(19) RR_W(20) RR_W(21) RR_W(22) RR_W(23) RR_W(24) RR_W(25) RR_W(26) RR_W(27) RR_W(28)
    RR_W(29) RR_W(30) RR_W(31) RR_W(32) RR_W(33) RR_W(34) RR_W(35) RR_W(36) RR_W(37) RR_W(38) RR_W(39) RR_W(40) RR_W(41) RR_W(42)
    RR_W(43) RR_W(44) RR_W(45) RR_W(46) RR_W(47)
    default: __builtin_amdgcn_s_waitcnt(0x0F70 | (48 & 15) | ((48 >> 4) << 14)); break;
  }
#undef RR_W
}

// ---- slab sequence -------------------------------------------------------------------------------------------------------
// A pass is a list of STAGES (one weight matrix each) of `tiles` 32-feature slabs; every stage has a multiple of RR_NSLOT
// tiles, so slab i of a stage always sits in ring slot i % RR_NSLOT (compile-time in the unrolled tile loops).  The tables
// live in LDS (sq_*): byte offset of the matrix in the fragment buffer, k-blocks per row, tiles.
struct SlabIt {
  int stage, tile;
};
#define RR_MAXSTAGE 16

// ---- per-wave ring state -------------------------------------------------------------------------------------------------
// vmc: vector-memory instructions this wave has issued in the slab loop so far (DMA + counted stores); mark[s]: its value
// right after the DMA into slot s was issued.  vmc - mark[s] instructions are younger than that DMA.
template <int NW>
struct Ring {
  const char* src;       // fragment buffer (P16 or PT16)
  const int* sq_off;     // [stages] byte offsets (LDS table)
  const int* sq_kb;      // [stages] k-blocks per tile row
  const int* sq_tiles;   // [stages]
  const int* sq_wrap;    // [stages] or nullptr: source tile = tile % wrap (a 2-tile stage issued twice keeps the slots aligned)
  int nstage;
  int wave, lane;
  int vmc;
  int mark[RR_NSLOT];    // (every index below is a compile-time constant: the struct lives in scalar registers)
  SlabIt pre;            // next slab to request

  // request the next slab of the sequence into ring slot SLOT (`ring`: the kernel's LDS array, passed in so that the compiler
  // keeps the address space)
  template <int SLOT>
  __device__ __forceinline__ void issue_next(char* ring) {
    if (pre.stage < nstage) {
      const int kb = __builtin_amdgcn_readfirstlane(sq_kb[pre.stage]);
      const int off = __builtin_amdgcn_readfirstlane(sq_off[pre.stage]);
      const int per = (kb + NW - 1) / NW;
      const int gt = sq_wrap ? pre.tile % __builtin_amdgcn_readfirstlane(sq_wrap[pre.stage]) : pre.tile;
      const char* g = src + (size_t)off + (size_t)gt * kb * 2048 + lane * 16;
      char* d = ring + SLOT * RR_SLOT;
      asm volatile("" ::: "memory");
      for (int q = 0; q < per; ++q) {
        int ch = wave + NW * q;
        ch = ch < kb ? ch : kb - 1;  // surplus waves repeat the last chunk (same bytes to the same place)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)ch * 2048),
                                         (__attribute__((address_space(3))) void*)(d + ch * 1024), 16, 0, 0);
      }
      asm volatile("" ::: "memory");
      vmc += per;
      if (++pre.tile >= __builtin_amdgcn_readfirstlane(sq_tiles[pre.stage])) {
        pre.tile = 0;
        do ++pre.stage;
        while (pre.stage < nstage && __builtin_amdgcn_readfirstlane(sq_tiles[pre.stage]) == 0);
      }
    }
    mark[SLOT] = vmc;
  }

  __device__ __forceinline__ void start(char* ring) {
    vmc = 0;
    pre.stage = 0;
    pre.tile = 0;
    static_assert(RR_AHEAD == 3, "start() requests slabs 0, 1, 2");
    issue_next<0>(ring);
    issue_next<1>(ring);
    issue_next<2>(ring);
  }

  // top of a slab whose data sits in slot SLOT: wait for its DMA (this wave's part), meet the other waves (their parts have
  // landed too, and everybody is done with the slot refilled next), request the slab RR_AHEAD further on
  template <int SLOT>
  __device__ __forceinline__ void begin(char* ring) {
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt(vmc - mark[SLOT]);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS traffic (transposer, previous fragments) is done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_next<(SLOT + RR_AHEAD) % RR_NSLOT>(ring);
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void count(int n) { vmc += n; }
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// acc (32 features x 32 samples, transposed) += slab k-blocks [p, p + T KiB) . o[0 .. T)
// Weight fragments come from the LDS slot through a ring RR_PF k-blocks ahead of their MFMAs.  With FILL, `fill(g)` -- a quarter
// of the PREVIOUS tile's epilogue -- is called once in each of the first four regions of four k-blocks, and sched_group_barrier
// lays a region out as LDS read, MFMA, a few vector instructions, ... so that the vector work issues in the shadow of the
// matrix pipe.
template <int T, bool FILL, int N, class F>
__device__ __forceinline__ void kpart(f32x16& acc, const char* p, const h8 (&o)[N], F fill) {
  static_assert(T <= N, "operand array");
  constexpr int SETS = RR_PF + 1, G = (T + 3) / 4;
  h8 af[SETS];
#pragma unroll
  for (int t = 0; t < RR_PF && t < T; ++t) af[t] = *(const h8*)(p + t * 1024);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int t = 4 * g; t < 4 * g + 4 && t < T; ++t) {
      if (t + RR_PF < T) af[(t + RR_PF) % SETS] = *(const h8*)(p + (t + RR_PF) * 1024);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t % SETS], o[t], acc, 0, 0, 0);
    }
    if constexpr (FILL) {
      if (g < 4) fill(g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // one LDS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, RR_FILL_VALU, 0);  // vector instructions in its shadow
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void acc_clear(f32x16& acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
}

// 8 fp32 values (natural units, operand order) -> one B-operand fragment at exponent e
__device__ __forceinline__ h8 make_op(const float (&v)[8], int e) {
  h4 a, b, c, d;
  split_quad<1>(ldexpf(v[0], e), ldexpf(v[1], e), ldexpf(v[2], e), ldexpf(v[3], e), a, c);
  split_quad<1>(ldexpf(v[4], e), ldexpf(v[5], e), ldexpf(v[6], e), ldexpf(v[7], e), b, d);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// operand fragment of k-block s of an fp32 row (LDS or global) in the k order of the register chain
__device__ __forceinline__ h8 row_op(const float* row, int s, int hh, int e) {
  const f32x4 a = *(const f32x4*)(row + 16 * s + 4 * hh), b = *(const f32x4*)(row + 16 * s + 8 + 4 * hh);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return make_op(v, e);
}

// ---- per-wave transposer: a 32 x 32 fp32 tile (accumulator layout: a lane = a row, four consecutive columns per quad) ->
// whole 128-byte rows of a row-major tensor.  16-byte chunks XOR-swizzled by the row (writes: eight rows per lane group hit
// eight bank quads; reads: the eight chunks of a row).
__device__ __forceinline__ void stg_put(char* stg, int li, int hh, int q, const f32x4& v) {
  *(f32x4*)(stg + li * 128 + (((2 * q + hh) ^ (li & 7)) << 4)) = v;
}
// rows [0, 32) of the tile -> dst[(row0 + r) * ld + col0 + 0..31]; four store instructions of 8 rows x 128 B
__device__ __forceinline__ void stg_flush(const char* stg, int lane, float* __restrict__ dst, size_t ld, int col0) {
  const int c = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3);
    const f32x4 v = *(const f32x4*)(stg + row * 128 + ((c ^ (row & 7)) << 4));
    NT_STORE((f32x4*)(dst + (size_t)row * ld + col0 + 4 * c), v);
  }
}

enum { EP_RELU = 1, EP_MASK = 2, EP_CONV = 4 };

// Epilogue of register quad q of the 32-feature tile j of a stage: v = act(fma(acc, un, bias)); optional sign bits (word
// j / 2 of bits[4], bit 16 (j % 2) + 4 q + u), running maximum, fp32 copy into the transposer, conversion into the next
// operand fragments (k-blocks 2j, 2j + 1) at exponent eo, up to three dot products with LDS-staged vectors.
template <int FLAGS, int NDOT, int NB>
__device__ __forceinline__ void quad_epilogue(const f32x16& acc, int j, int q, float un, const float* bias_s, unsigned int (&bits)[4],
                                              float& vmax, char* stg /* transposer, or nullptr: no fp32 copy */, int eo, h8 (&nx)[NB],
                                              const float* dotw_s, int dot_ld, float (&dot)[3], int li, int hh) {
  const int col = 32 * j + 8 * q + 4 * hh;
  const f32x4 b = *(const f32x4*)&bias_s[col];
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    v[u] = fmaf(acc[4 * q + u], un, b[u]);
    if (FLAGS & EP_RELU) v[u] = fmaxf(v[u], 0.0f);
    if (FLAGS & EP_MASK) {
      const unsigned int one = min(__float_as_uint(v[u]), 1u);  // v >= 0 after the ReLU: positive <=> non-zero bit pattern
      bits[(j >> 1) & 3] |= one << (16 * (j & 1) + 4 * q + u);
    }
  }
  vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  if (stg) stg_put(stg, li, hh, q, f32x4{v[0], v[1], v[2], v[3]});
  if constexpr (NDOT > 0) {
#pragma unroll
    for (int c = 0; c < NDOT; ++c) {
      const f32x4 w = *(const f32x4*)&dotw_s[c * dot_ld + col];
      dot[c] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
    }
  }
  if constexpr ((FLAGS & EP_CONV) != 0) {
    h4 hi, lo;
    split_quad<1>(ldexpf(v[0], eo), ldexpf(v[1], eo), ldexpf(v[2], eo), ldexpf(v[3], eo), hi, lo);
    const int blk = 2 * j + (q >> 1);
    if (q & 1) nx[blk] = __builtin_shufflevector(nx[blk], __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 12, 13, 14, 15);
    else nx[blk] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3);
  }
}

// one operand fragment (k-block blk of this wave's 32-row tile) -> the fragment-ordered fp16 tensor: 1 KiB per instruction
__device__ __forceinline__ void frag_store(uint16_t* __restrict__ base, size_t tile32, int blk, int lane, const h8& v) {
  NT_STORE((f32x4*)((char*)base + (tile32 * 16 + blk) * 1024 + lane * 16), __builtin_bit_cast(f32x4, v));
}

__device__ __forceinline__ void track_lds(unsigned int* mx_s, int slot, float wave_mx, int lane) {
  if (lane == 0) atomicMax(&mx_s[slot], __float_as_uint(wave_mx));
}

// ================================================================================================================================
// forward
// ================================================================================================================================
template <int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void rr16_fwd_kernel(upnerf_layout L, upnerf_field_fwd_args a) {
  constexpr int W = 256, W2 = 128;
  using C = RRCfg<NW>;
  __shared__ __attribute__((aligned(16))) char lds[C::LDS];  // ONE object: [ring | transposers | vectors | ray rows | tables]
  char* ring = lds;
  float* vec_s = (float*)(lds + C::VEC0);
  float* rows_s = (float*)(lds + C::ROW0);
  int* int_s = (int*)(lds + C::INT0);  // [0,16) stage offsets (bytes), [16,32) k-blocks, [32,48) tiles, [48,64) running maxima
  unsigned int* mx_s = (unsigned int*)(int_s + 48);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  char* stg = lds + C::STG0 + wave * RR_STG;
  const int S = a.S, M = a.R * a.S, D = L.D;
  const int m0 = blockIdx.x * C::TILE + 32 * wave;   // first row of this wave
  const int m = m0 + li;
  const int mc = m < M ? m : M - 1;                  // rows past the end repeat the last sample (their stores land in padding)
  const bool valid = m < M;
  const int ray = mc / S, ray0 = (blockIdx.x * C::TILE) / S;
  const int rs = ray - ray0;                         // ray slot of this lane's sample
  const size_t t32 = (size_t)blockIdx.x * NW + wave; // 32-row tile index
  const size_t nt32 = (size_t)gridDim.x * NW;
  const float* __restrict__ P = a.P;
  const int* __restrict__ wexp = a.wexp;
  const float* __restrict__ wnorm = a.wnorm;
  const bool use_rgb = a.use_rgb != 0, use_cand = a.use_cand != 0;
  const bool train = a.h16 != nullptr;
  const int last_stage = (!a.e && !use_rgb && !use_cand) ? D - 1 : ((!use_rgb && !use_cand) ? D : D + 3);

  // ---- stage tables, vectors, per-ray rows (ordinary loads: all of them BEFORE the first DMA is in flight)
  if (tid < RR_MAXSTAGE) {
    int off = 0, kb = 16, tiles = 0;
    const int st = tid;
    if (st < D) {
      // (no runtime index into the by-value struct: hipcc would copy it to scratch)
      int w = L.w[0];
#pragma unroll
      for (int l = 1; l < UPNERF_MAX_D; ++l) w = st == l ? L.w[l] : w;
      off = w;
      kb = st == 0 ? UPNERF_X0 / 16 : (st == L.skip ? (UPNERF_X0 + W) / 16 : W / 16);
      tiles = 8;
    } else if (st == D) {
      off = L.we, kb = W / 16, tiles = last_stage >= D ? 8 : 0;
    } else if (st == D + 1) {
      off = L.wr1, kb = (W + UPNERF_AUXK) / 16, tiles = (last_stage > D && use_rgb) ? 4 : 0;
    } else if (st == D + 2) {
      off = L.wc1, kb = (W + UPNERF_CK) / 16, tiles = (last_stage > D && use_cand) ? 4 : 0;
    } else if (st == D + 3) {
      off = L.wc2, kb = W2 / 16, tiles = (last_stage > D && use_cand) ? 4 : 0;
    }
    int_s[st] = 4 * off;
    int_s[16 + st] = kb;
    int_s[32 + st] = tiles;
    mx_s[st] = 0u;
  }
#pragma unroll
  for (int l = 0; l < UPNERF_MAX_D; ++l)
    for (int c = tid; c < W; c += C::THREADS) vec_s[256 * l + c] = l < D ? P[L.b[l] + c] : 0.0f;
  for (int c = tid; c < W; c += C::THREADS) {
    vec_s[RR_V_BE + c] = P[L.be + c];
    vec_s[RR_V_WSIG + c] = P[L.wsig + c];
  }
  for (int c = tid; c < W2; c += C::THREADS) {
    vec_s[RR_V_BR1 + c] = use_rgb ? P[L.br1 + c] : 0.0f;
    vec_s[RR_V_BC1 + c] = use_cand ? P[L.bc1 + c] : 0.0f;
    vec_s[RR_V_BC2 + c] = use_cand ? P[L.bc2 + c] : 0.0f;
    vec_s[RR_V_WCSIG + c] = use_cand ? P[L.wcsig + c] : 0.0f;
  }
  for (int c = tid; c < 3 * W2; c += C::THREADS) vec_s[RR_V_WR2 + c] = use_rgb ? P[L.wr2 + c] : 0.0f;
  // per-ray side inputs of the heads: [aux 80 | candidate row 16] per ray slot; their largest magnitude bounds the exponent of e
  float sidemax = 0.0f;
  {
    const int mlast = (blockIdx.x * C::TILE + C::TILE < M ? blockIdx.x * C::TILE + C::TILE : M) - 1;
    const int nr = mlast / S - ray0 + 1;
    if (use_rgb)
      for (int i = tid; i < nr * UPNERF_AUXK; i += C::THREADS) {
        const float v = a.aux[(size_t)ray0 * UPNERF_AUXK + i];
        rows_s[(i / UPNERF_AUXK) * C::ROWF + i % UPNERF_AUXK] = v;
        sidemax = fmaxf(sidemax, fabsf(v));
      }
    if (use_cand)
      for (int i = tid; i < nr * UPNERF_CK; i += C::THREADS) {
        const float v = a.c_rows[(size_t)ray0 * UPNERF_CK + i];
        rows_s[(i / UPNERF_CK) * C::ROWF + 96 + i % UPNERF_CK] = v;
        sidemax = fmaxf(sidemax, fabsf(v));
      }
  }
  // ---- sample position (rendering.py:251 / 308) and its encoding (nerf.py:126-147): each lane half evaluates 15 of the 30
  // (coordinate, band) pairs of ITS sample once; the halves meet in an LDS scratch (the ring, not yet in use)
  float xm;
  {
    const float zz = a.z[mc];
    float xyz[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) xyz[n] = mul_then_add(a.rays_o[3 * ray + n], a.rays_d[3 * ray + n], zz);
    xm = fmaxf(fmaxf(fabsf(xyz[0]), fabsf(xyz[1])), fmaxf(fabsf(xyz[2]), 1.0f));  // |sin|, |cos| <= 1
    float* pe = (float*)ring + (32 * wave + li) * RR_PE_LD;
    const float* __restrict__ wkd = a.wk_xyz_dev;
    if (hh == 0) {
      pe[0] = xyz[0];
      pe[1] = xyz[1];
      pe[2] = xyz[2];
      pe[63] = 0.0f;
    }
#pragma unroll 1
    for (int p = 0; p < 15; ++p) {
      const int pp = 15 * hh + p, n = pp / 10, k = pp - 10 * n;
      const float xv = n == 0 ? xyz[0] : (n == 1 ? xyz[1] : xyz[2]);
      float sv, cv;
      sincos_f32_via_f64(xv * ldexpf(PI_F, k), sv, cv);
      const float wk = wkd ? wkd[k] : a.wk_xyz[k];
      pe[3 + 20 * n + k] = sv * wk;
      pe[3 + 20 * n + 10 + k] = cv * wk;
    }
  }
  __syncthreads();
  const float x0max = wave_max_rr(xm);
  const int e0 = scale_exp(x0max);
  {
    // this lane's 32 encoding features in operand order: operand of layer 0 (re-scaled for the skip layer) and the row-major x0
    // tensor (backward pass, weight gradients)
    const float* pe = (const float*)ring + (32 * wave + li) * RR_PE_LD;
    float* __restrict__ xrow = a.x0 + (size_t)m * UPNERF_X0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float x0v[8];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 v = *(const f32x4*)&pe[16 * s + 8 * g + 4 * hh];
#pragma unroll
        for (int u = 0; u < 4; ++u) x0v[4 * g + u] = v[u];
        *(f32x4*)&xrow[16 * s + 8 * g + 4 * hh] = v;  // (rows past M land in the padding of x0)
      }
      *(h8*)(stg + s * 1024 + lane * 16) = make_op(x0v, e0);  // parked in the transposer (4 KiB: the wave's four fragments)
    }
  }
  // per-vector maxima |b|_inf (bounds of the stage outputs), one wave
  if (wave == 0) {
    for (int l = 0; l < 12; ++l) {  // 0..7 trunk, 8 final, 9 rgb1, 10 cand1, 11 cand2
      const int base = l < 8 ? 256 * l : (l == 8 ? RR_V_BE : (l == 9 ? RR_V_BR1 : (l == 10 ? RR_V_BC1 : RR_V_BC2)));
      const int n = l <= 8 ? 256 : 128;
      float mx = 0.0f;
      for (int i = lane; i < n; i += 64) mx = fmaxf(mx, fabsf(vec_s[base + i]));
      mx = wave_max_rr(mx);
      if (lane == 0) vec_s[RR_V_BMAX + l] = mx;
    }
  }
  sidemax = wave_max_rr(sidemax);
  if (lane == 0) atomicMax(&mx_s[15], __float_as_uint(sidemax));
  track_lds(mx_s, D + 4, x0max, lane);
  __syncthreads();  // scratch reads done (the ring is free for the weight stream), tables / vector maxima visible
  sidemax = __uint_as_float(mx_s[15]);

  const float bsig = P[L.bsig];
  float br2[3] = {0.f, 0.f, 0.f}, bcsig = 0.0f;
  if (use_rgb) {
#pragma unroll
    for (int c = 0; c < 3; ++c) br2[c] = P[L.br2 + c];
  }
  if (use_cand) bcsig = P[L.bcsig];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every ordinary load / store of the prologue has retired: the counter starts at 0

  Ring<NW> rg;
  rg.src = (const char*)a.P16;
  rg.sq_off = int_s;
  rg.sq_kb = int_s + 16;
  rg.sq_tiles = int_s + 32;
  rg.sq_wrap = nullptr;
  rg.nstage = D + 4;
  rg.wave = wave;
  rg.lane = lane;
  rg.start(lds);

  h8 Bh[16];  // operand of the running stage (previous stage's outputs)
  h8 Nh[16];  // operand of the next stage, filled tile by tile
  float dot3[3] = {0.f, 0.f, 0.f};
  int e_in = e0;          // exponent of the operand the running stage reads
  float amax_in = x0max;  // its exact largest magnitude in this wave
  auto nofill = [](int) {};
  // The two waves of a SIMD (w and w + NW/2) run the tile loop half a tile apart: the leading half contracts tile j and then
  // runs its epilogue, the lagging half runs the epilogue of tile j - 1 and then contracts tile j -- one wave's vector work
  // beside the other's matrix work on every SIMD, one accumulator per wave (MI355X_MICROARCH.md, "try a stagger").
  const bool lag = wave >= NW / 2;

  // ---- trunk (nerf.py:84-87)
#pragma unroll 1
  for (int l = 0; l < D; ++l) {
    const bool has_x = l == 0 || l == L.skip, has_h = l > 0;
    h8 Xh[4];
    if (has_x) {
      if (l > 0) amax_in = fmaxf(amax_in, x0max);
      // the encoding enters at THIS stage's operand exponent (e_in covers x0max: chosen one layer earlier): exact fp16
      // power-of-two rescale of the fragments parked in the transposer
      const _Float16 f = (_Float16)ldexpf(1.0f, max(e_in - e0, -24));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        h8 x = *(const h8*)(stg + s * 1024 + lane * 16);
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = x[u] * f;
        Xh[s] = x;
      }
    }
    const float un = pow2r(-(e_in + wexp[l]));
    float bound = wnorm[l] * amax_in + vec_s[RR_V_BMAX + l];
    if (l + 1 == L.skip) bound = fmaxf(bound, x0max);
    const int eo = scale_exp(bound);
    const float* bias_s = vec_s + 256 * l;
    const bool rows32 = train && a.h != nullptr && l == D - 1;  // fp32 copy of the last trunk layer (density-head / final-layer gradients)
    char* stg_l = rows32 ? stg : nullptr;
    const int xkb = (has_x && has_h) ? UPNERF_X0 / 16 : 0;  // k-blocks of the encoding part in front of the h part
    unsigned int bits[4] = {0u, 0u, 0u, 0u};
    float vmax = 0.0f;
    f32x16 acc;
    // epilogue of tile jp, then its two operand fragments and (last layer) its fp32 rows leave
    auto tile_epi = [&](auto JP) {
      constexpr int jp = decltype(JP)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        quad_epilogue<EP_RELU | EP_MASK | EP_CONV, 0>(acc, jp, q, un, bias_s, bits, vmax, stg_l, eo, Nh, nullptr, 0, dot3, li, hh);
      if (train) {
        uint16_t* __restrict__ h16l = a.h16 + (size_t)l * nt32 * 16 * 512;
        frag_store(h16l, t32, 2 * jp, lane, Nh[2 * jp]);
        frag_store(h16l, t32, 2 * jp + 1, lane, Nh[2 * jp + 1]);
        rg.count(2);
      }
      if (rows32) {
        stg_flush(stg, lane, a.h + (size_t)m0 * W, W, 32 * jp);
        rg.count(4);
      }
    };
    static_for<0, 8>([&](auto J) {
      constexpr int j = decltype(J)::value;
      rg.template begin<(j & 3)>(lds);
      const char* p = lds + (j & 3) * RR_SLOT + lane * 16;
      if constexpr (j > 0) {
        if (lag) tile_epi(std::integral_constant<int, (j > 0 ? j - 1 : 0)>{});
      }
      acc_clear(acc);
      if (has_x) kpart<4, false>(acc, p, Xh, nofill);
      if (has_h) kpart<16, false>(acc, p + xkb * 1024, Bh, nofill);
      if (!lag) tile_epi(J);
    });
    if (lag) tile_epi(std::integral_constant<int, 7>{});
    if (train) {
      NT_STORE((u32x4_t*)((char*)a.hmask + (((size_t)l * nt32 + t32) * 64 + lane) * 16), (u32x4_t{bits[0], bits[1], bits[2], bits[3]}));
      rg.count(1);
      if (lane == 0) a.hexp[(size_t)l * nt32 + t32] = eo;  // (one lane: not counted)
    }
    amax_in = wave_max_rr(vmax);
    track_lds(mx_s, l, amax_in, lane);
    e_in = eo;
#pragma unroll
    for (int s = 0; s < 16; ++s) Bh[s] = Nh[s];
  }

  // ---- shared density head (nerf.py:89): softplus(w . h + b), from the operand fragments
  {
    float sdot = 0.0f;
    const float* wsig_s = vec_s + RR_V_WSIG;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const f32x4 w0 = *(const f32x4*)&wsig_s[16 * s + 4 * hh], w1 = *(const f32x4*)&wsig_s[16 * s + 8 + 4 * hh];
#pragma unroll
      for (int u = 0; u < 4; ++u) sdot += (float)Bh[s][u] * w0[u] + (float)Bh[s][4 + u] * w1[u];
    }
    sdot += __shfl_xor(sdot, 32);
    if (hh == 0 && valid) a.sigma_s[m] = softplus_f(sdot * pow2r(-e_in) + bsig);
  }
  if (last_stage >= D) {
    // ---- xyz_encoding_final (nerf.py:93), no activation; its operand form E feeds both heads
    float amax_e;
    int e_E;
    {
      const float un = pow2r(-(e_in + wexp[8]));
      const float bound = fmaxf(wnorm[D] * amax_in + vec_s[RR_V_BMAX + 8], sidemax);  // the heads add per-ray rows at E's exponent
      e_E = scale_exp(bound);
      char* stg_e = a.e ? stg : nullptr;
      unsigned int nobits[4] = {0u, 0u, 0u, 0u};
      float vmax = 0.0f;
      f32x16 acc;
      auto tile_epi = [&](auto JP) {
        constexpr int jp = decltype(JP)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          quad_epilogue<EP_CONV, 0>(acc, jp, q, un, vec_s + RR_V_BE, nobits, vmax, stg_e, e_E, Nh, nullptr, 0, dot3, li, hh);
        if (a.e) {
          stg_flush(stg, lane, a.e + (size_t)m0 * W, W, 32 * jp);
          rg.count(4);
        }
      };
      static_for<0, 8>([&](auto J) {
        constexpr int j = decltype(J)::value;
        rg.template begin<(j & 3)>(lds);
      const char* p = lds + (j & 3) * RR_SLOT + lane * 16;
        if constexpr (j > 0) {
          if (lag) tile_epi(std::integral_constant<int, (j > 0 ? j - 1 : 0)>{});
        }
        acc_clear(acc);
        kpart<16, false>(acc, p, Bh, nofill);
        if (!lag) tile_epi(J);
      });
      if (lag) tile_epi(std::integral_constant<int, 7>{});
      amax_e = wave_max_rr(vmax);
      track_lds(mx_s, D, amax_e, lane);
#pragma unroll
      for (int s = 0; s < 16; ++s) Bh[s] = Nh[s];
    }
    // ---- colour head (folded first layer, nerf.py:95 + 102-109; rgb_share_layer.2 + sigmoid, nerf.py:56-61)
    if (last_stage > D && use_rgb) {
      h8 Ah[5];  // [PE(dir) | appearance | 0] of this sample's ray as operand k-blocks
#pragma unroll
      for (int s = 0; s < 5; ++s) Ah[s] = row_op(rows_s + rs * C::ROWF, s, hh, e_E);
      const float un = pow2r(-(e_E + wexp[11]));
      char* stg_r = train ? stg : nullptr;
      float* __restrict__ rdst = train ? a.r1 + (size_t)m0 * W2 : nullptr;
      unsigned int bits[4] = {0u, 0u, 0u, 0u};
      float vmax = 0.0f;
      dot3[0] = dot3[1] = dot3[2] = 0.0f;
      static_for<0, 4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        rg.template begin<(j & 3)>(lds);
      const char* p = lds + (j & 3) * RR_SLOT + lane * 16;
        f32x16 acc;
        acc_clear(acc);
        kpart<16, false>(acc, p, Bh, nofill);
        kpart<5, false>(acc, p + 16 * 1024, Ah, nofill);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          quad_epilogue<EP_RELU | EP_MASK, 3>(acc, j, q, un, vec_s + RR_V_BR1, bits, vmax, stg_r, 0, Nh, vec_s + RR_V_WR2, W2, dot3, li, hh);
        if (train) {
          stg_flush(stg, lane, rdst, W2, 32 * j);
          rg.count(4);
        }
      });
      if (train) {
        NT_STORE((u32x4_t*)((char*)a.hmask + (((size_t)(D + 2) * nt32 + t32) * 64 + lane) * 16), (u32x4_t{bits[0], bits[1], 0u, 0u}));
        rg.count(1);
      }
      track_lds(mx_s, D + 3, wave_max_rr(vmax), lane);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = dot3[c] + __shfl_xor(dot3[c], 32);
        if (hh == 0 && valid) a.rgb[(size_t)m * 3 + c] = sigmoid_f(d + br2[c]);
      }
    }
    // ---- candidate head (nerf.py:97-100)
    if (last_stage > D && use_cand) {
      h8 Ch[1];
      Ch[0] = row_op(rows_s + rs * C::ROWF + 96, 0, hh, e_E);
      int e_G;
      {
        const float un = pow2r(-(e_E + wexp[9]));
        e_G = scale_exp(wnorm[D + 1] * fmaxf(amax_e, sidemax) + vec_s[RR_V_BMAX + 10]);
        char* stg_g = train ? stg : nullptr;
        float* __restrict__ gdst = train ? a.g1 + (size_t)m0 * W2 : nullptr;
        unsigned int bits[4] = {0u, 0u, 0u, 0u};
        float vmax = 0.0f;
        static_for<0, 4>([&](auto J) {
          constexpr int j = decltype(J)::value;
          rg.template begin<(j & 3)>(lds);
      const char* p = lds + (j & 3) * RR_SLOT + lane * 16;
          f32x16 acc;
          acc_clear(acc);
          kpart<16, false>(acc, p, Bh, nofill);
          kpart<1, false>(acc, p + 16 * 1024, Ch, nofill);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            quad_epilogue<EP_RELU | EP_MASK | EP_CONV, 0>(acc, j, q, un, vec_s + RR_V_BC1, bits, vmax, stg_g, e_G, Nh, nullptr, 0, dot3, li, hh);
          if (train) {
            stg_flush(stg, lane, gdst, W2, 32 * j);
            rg.count(4);
          }
        });
        if (train) {
          NT_STORE((u32x4_t*)((char*)a.hmask + (((size_t)D * nt32 + t32) * 64 + lane) * 16), (u32x4_t{bits[0], bits[1], 0u, 0u}));
          rg.count(1);
        }
        track_lds(mx_s, D + 1, wave_max_rr(vmax), lane);
      }
      {
        const float un = pow2r(-(e_G + wexp[10]));
        float* __restrict__ gdst = a.g2 + (size_t)m0 * W2;  // compositing reads g2 in inference too
        unsigned int bits[4] = {0u, 0u, 0u, 0u};
        float vmax = 0.0f;
        dot3[0] = 0.0f;
        static_for<0, 4>([&](auto J) {
          constexpr int j = decltype(J)::value;
          rg.template begin<(j & 3)>(lds);
      const char* p = lds + (j & 3) * RR_SLOT + lane * 16;
          f32x16 acc;
          acc_clear(acc);
          kpart<8, false>(acc, p, Nh, nofill);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            quad_epilogue<EP_RELU | EP_MASK, 1>(acc, j, q, un, vec_s + RR_V_BC2, bits, vmax, stg, 0, Bh, vec_s + RR_V_WCSIG, W2, dot3, li, hh);
          stg_flush(stg, lane, gdst, W2, 32 * j);
          rg.count(4);
        });
        if (train) {
          NT_STORE((u32x4_t*)((char*)a.hmask + (((size_t)(D + 1) * nt32 + t32) * 64 + lane) * 16), (u32x4_t{bits[0], bits[1], 0u, 0u}));
          rg.count(1);
        }
        const float d = dot3[0] + __shfl_xor(dot3[0], 32);
        if (hh == 0 && valid) a.sigma_c[m] = softplus_f(d + bcsig);
      }
    }
  }
  // running maxima -> global table (scales of the weight-gradient contraction), once per workgroup
  __syncthreads();
  if (a.amax && tid < 16) {
    const unsigned int v = mx_s[tid];
    if (v && tid != 15) atomicMax((unsigned int*)a.amax + tid, v);
  }
}


// ================================================================================================================================
// backward: data-gradient chain (autograd of nerf.py:80-124), stage for stage as field16.hip:field16_bwd_kernel
// ================================================================================================================================
// LDS of the backward kernel: [ring | transposers | vectors (w_sigma 256, w_csigma 128, w_rgb2 3 x 128) | per-ray gradient rows
// (g_E_s 256 + g_G_c 128 per ray slot) | sign-bit buffers (2 x 1 KiB per wave, filled by LDS-DMA) | tables]
#define RB_V_WSIG 0
#define RB_V_WCSIG 256
#define RB_V_WR2 384
#define RB_V_TOTAL 768
#define RB_NSTAGE 24
template <int NW>
struct RBCfg {
  static constexpr int THREADS = 64 * NW, TILE = 32 * NW;
  static constexpr int MAXR = TILE / 32 + 1;
  static constexpr int RING = RR_NSLOT * RR_SLOT;
  static constexpr int STG0 = RING;
  static constexpr int VEC0 = STG0 + NW * RR_STG;
  static constexpr int ROW0 = VEC0 + RB_V_TOTAL * 4;
  static constexpr int ROWF = 256 + 128;
  static constexpr int MSK0 = ROW0 + MAXR * ROWF * 4;
  static constexpr int INT0 = MSK0 + NW * 2048;
  static constexpr int LDS = INT0 + (4 * RB_NSTAGE + 16) * 4;
  static_assert(NW * 32 * 64 * 4 <= RING, "d x0 exchange scratch lives in the ring");
};

// the sign bit of accumulator element (tile j, register 4 q + u) as an all-ones / all-zeros word
__device__ __forceinline__ unsigned int mask_word(const unsigned int (&bits)[4], int j, int q, int u) {
  return (unsigned int)(((int)(bits[(j >> 1) & 3] << (31 - (16 * (j & 1) + 4 * q + u)))) >> 31);
}

enum { EB_MASK = 1, EB_CONV = 2 };
// Epilogue of register quad q of tile j of a backward stage: v = acc * un + add, masked by the forward pass's sign bits;
// running maximum, fp32 copy into the transposer (stg != nullptr), conversion into the next operand's fragments
// nx[blk0 + 2 j], nx[blk0 + 2 j + 1] at exponent eo.
template <int FLAGS, int NB>
__device__ __forceinline__ void quad_epilogue_b(const f32x16& acc, int j, int q, float un, const f32x4& add, const unsigned int (&bits)[4],
                                                float& vmax, char* stg, int eo, h8 (&nx)[NB], int blk0, int li, int hh) {
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    v[u] = fmaf(acc[4 * q + u], un, add[u]);
    if (FLAGS & EB_MASK) v[u] = __uint_as_float(__float_as_uint(v[u]) & mask_word(bits, j, q, u));
  }
  vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  if (stg) stg_put(stg, li, hh, q, f32x4{v[0], v[1], v[2], v[3]});
  if constexpr ((FLAGS & EB_CONV) != 0) {
    h4 hi, lo;
    split_quad<1>(ldexpf(v[0], eo), ldexpf(v[1], eo), ldexpf(v[2], eo), ldexpf(v[3], eo), hi, lo);
    const int blk = blk0 + 2 * j + (q >> 1);
    if (q & 1) nx[blk] = __builtin_shufflevector(nx[blk], __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 12, 13, 14, 15);
    else nx[blk] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3);
  }
}

// NT tiles of one stage: slab j sits in ring slot j % 4; the two waves of a SIMD run half a tile apart (see the forward kernel)
template <int NW, int NT, int KB, int NOP, class EPI>
__device__ __forceinline__ void run_stage(Ring<NW>& rg, char* lds, int lane, bool lag, f32x16& acc, const h8 (&op)[NOP], EPI epi) {
  auto nofill = [](int) {};
  static_for<0, NT>([&](auto J) {
    constexpr int j = decltype(J)::value;
    rg.template begin<(j & 3)>(lds);
    const char* p = lds + (j & 3) * RR_SLOT + lane * 16;
    if constexpr (j > 0) {
      if (lag) epi(std::integral_constant<int, (j > 0 ? j - 1 : 0)>{});
    }
    acc_clear(acc);
    kpart<KB, false>(acc, p, op, nofill);
    if (!lag) epi(J);
  });
  if (lag) epi(std::integral_constant<int, NT - 1>{});
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void rr16_bwd_kernel(upnerf_layout L, upnerf_field_bwd_args a) {
  constexpr int W = 256, W2 = 128;
  using C = RBCfg<NW>;
  __shared__ __attribute__((aligned(16))) char lds[C::LDS];
  float* vec_s = (float*)(lds + C::VEC0);
  float* rows_s = (float*)(lds + C::ROW0);
  int* int_s = (int*)(lds + C::INT0);  // [0,24) stage offsets (bytes), [24,48) k-blocks, [48,72) tiles, [72,96) tile wrap, [96,112) maxima
  unsigned int* mx_s = (unsigned int*)(int_s + 4 * RB_NSTAGE);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  char* stg = lds + C::STG0 + wave * RR_STG;
  char* msk = lds + C::MSK0 + wave * 2048;
  const int S = a.S, M = a.R * a.S, D = L.D;
  const int m0 = blockIdx.x * C::TILE + 32 * wave;
  const int m = m0 + li;
  const int mc = m < M ? m : M - 1;
  const bool valid = m < M;
  const int ray = mc / S, ray0 = (blockIdx.x * C::TILE) / S;
  const int rs = ray - ray0;
  const size_t t32 = (size_t)blockIdx.x * NW + wave;
  const size_t nt32 = (size_t)gridDim.x * NW;
  const float* __restrict__ P = a.P;
  const int* __restrict__ wexp = a.wexp;
  const bool use_rgb = a.use_rgb != 0, use_cand = a.use_cand != 0, heads = use_rgb || use_cand;
  const bool need_dxyz = a.need_dxyz != 0;
  const int hs = L.skip > 0 ? 1 : 0;                 // the skip layer has two transposed descriptors (row norms: descriptor order)
  const float* __restrict__ wnt = a.wnorm + 32;      // row norms of the transposed set
  const int gld = a.gz_rg_ld > 0 ? a.gz_rg_ld : W2;  // row stride of gz_r1 / gz_g1
  const bool lag = wave >= NW / 2;

  // ---- stage tables.  Order of consumption: [t_wc2] [t_head] t_we, then for l = D-1 .. 1: [t_skipx at l == skip] t_w[l], then
  // [t_w[0]].  Stage ids: 0 wc2, 1 head, 2 we, 3 + 2 i (skipx) / 4 + 2 i (trunk) for l = D-1-i, 3 + 2 (D-1) = layer 0.
  if (tid < RB_NSTAGE) {
    int off = 0, kb = W / 16, tiles = 0, wrap = 8;
    const int st = tid;
    if (st == 0) {
      off = L.t_wc2, kb = W2 / 16, tiles = use_cand ? 4 : 0, wrap = 4;
    } else if (st == 1) {
      off = L.t_head, tiles = heads ? 8 : 0;
    } else if (st == 2) {
      off = L.t_we, tiles = 8;
    } else if (st < 3 + 2 * (D - 1)) {
      const int i = (st - 3) >> 1, l = D - 1 - i;
      if ((st - 3) & 1) {
        int w = L.t_w[0];  // (no runtime index into the by-value struct: hipcc would copy it to scratch)
#pragma unroll
        for (int k = 1; k < UPNERF_MAX_D; ++k) w = l == k ? L.t_w[k] : w;
        off = w, tiles = 8;
      } else {
        off = L.t_skipx, tiles = (need_dxyz && L.skip > 0 && l == L.skip) ? 4 : 0, wrap = 2;  // 64 outputs = 2 tiles, issued twice
      }
    } else if (st == 3 + 2 * (D - 1)) {
      off = L.t_w[0], tiles = need_dxyz ? 4 : 0, wrap = 2;
    }
    int_s[st] = 4 * off;
    int_s[RB_NSTAGE + st] = kb;
    int_s[2 * RB_NSTAGE + st] = tiles;
    int_s[3 * RB_NSTAGE + st] = wrap;
    if (st < 16) mx_s[st] = 0u;
  }
  for (int c = tid; c < W; c += C::THREADS) vec_s[RB_V_WSIG + c] = P[L.wsig + c];
  for (int c = tid; c < W2; c += C::THREADS) vec_s[RB_V_WCSIG + c] = use_cand ? P[L.wcsig + c] : 0.0f;
  for (int c = tid; c < 3 * W2; c += C::THREADS) vec_s[RB_V_WR2 + c] = use_rgb ? P[L.wr2 + c] : 0.0f;
  // upstream gradients of the per-ray sums (rank-1 terms of d e and d g2), one row per ray slot; absent = zero
  float gEmax = 0.0f, gGmax = 0.0f;
  {
    const int mlast = (blockIdx.x * C::TILE + C::TILE < M ? blockIdx.x * C::TILE + C::TILE : M) - 1;
    const int nr = mlast / S - ray0 + 1;
    for (int i = tid; i < nr * W; i += C::THREADS) {
      const float v = a.g_E_s ? a.g_E_s[(size_t)ray0 * W + i] : 0.0f;
      rows_s[(i >> 8) * C::ROWF + (i & 255)] = v;
      gEmax = fmaxf(gEmax, fabsf(v));
    }
    for (int i = tid; i < nr * W2; i += C::THREADS) {
      const float v = (use_cand && a.g_G_c) ? a.g_G_c[(size_t)ray0 * W2 + i] : 0.0f;
      rows_s[(i >> 7) * C::ROWF + 256 + (i & 127)] = v;
      gGmax = fmaxf(gGmax, fabsf(v));
    }
  }
  // ---- per-row scalars: softplus'(x) = 1 - exp(-softplus(x)), sigmoid' = y (1 - y); rows past M carry zeros
  float dps = 0.0f, dpc = 0.0f, wf = 0.0f, cwj = 0.0f;
  f32x4 dprgb = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
    dps = a.d_sigma_s[m] * (1.0f - expf(-a.sigma_s[m]));
    if (a.g_E_s) wf = a.w_feat_s[m];
    if (use_cand) {
      dpc = a.d_sigma_c[m] * (1.0f - expf(-a.sigma_c[m]));
      if (a.g_G_c) cwj = a.w_cj[m];
    }
    if (use_rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float y = a.rgb[(size_t)m * 3 + c];
        dprgb[c] = a.d_rgb[(size_t)m * 3 + c] * (y * (1.0f - y));
      }
    }
    if (hh == 0) {
      a.dpre_sig_s[m] = dps;
      if (use_cand) a.dpre_sig_c[m] = dpc;
      if (use_rgb) *(f32x4*)&a.dpre_rgb[(size_t)m * 4] = dprgb;
    }
  }
  // sign bits of g2 / r1 (forward slots D + 1, D + 2): ordinary loads, nothing is in flight yet
  unsigned int bits_g2[4] = {0u, 0u, 0u, 0u}, bits_r1[4] = {0u, 0u, 0u, 0u};
  if (use_cand) {
    const u32x4_t w = *(const u32x4_t*)((const char*)a.hmask + (((size_t)(D + 1) * nt32 + t32) * 64 + lane) * 16);
    bits_g2[0] = w[0], bits_g2[1] = w[1];
  }
  if (use_rgb) {
    const u32x4_t w = *(const u32x4_t*)((const char*)a.hmask + (((size_t)(D + 2) * nt32 + t32) * 64 + lane) * 16);
    bits_r1[0] = w[0], bits_r1[1] = w[1];
  }
  gEmax = wave_max_rr(gEmax);
  gGmax = wave_max_rr(gGmax);
  __syncthreads();  // tables zeroed
  if (lane == 0) {
    atomicMax(&mx_s[14], __float_as_uint(gEmax));
    atomicMax(&mx_s[15], __float_as_uint(gGmax));
  }
  __syncthreads();  // tables, vectors, rows, row maxima visible
  gEmax = __uint_as_float(mx_s[14]);
  gGmax = __uint_as_float(mx_s[15]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every ordinary load / store of the prologue has retired: the counter starts at 0

  Ring<NW> rg;
  rg.src = (const char*)a.PT16;
  rg.sq_off = int_s;
  rg.sq_kb = int_s + RB_NSTAGE;
  rg.sq_tiles = int_s + 2 * RB_NSTAGE;
  rg.sq_wrap = int_s + 3 * RB_NSTAGE;
  rg.nstage = 3 + 2 * (D - 1) + 1;
  rg.wave = wave;
  rg.lane = lane;
  rg.pre.stage = 0;
  rg.pre.tile = 0;
  rg.vmc = 0;
  // the first stage with tiles (the forward kernel's sequence always starts at stage 0)
  while (rg.pre.stage < rg.nstage && int_s[2 * RB_NSTAGE + rg.pre.stage] == 0) ++rg.pre.stage;
  rg.template issue_next<0>(lds);
  rg.template issue_next<1>(lds);
  rg.template issue_next<2>(lds);

  // ---- sign bits of the stages ahead: forward slots [D (g1)], D-1, ..., 0, each a 1 KiB LDS-DMA into buffer (slot & 1) of this
  // wave, requested two uses ahead; counted like every other vector-memory instruction of the loop
  int mmark0 = 0, mmark1 = 0;
  auto mask_request = [&](int slot) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // the reads of the buffer's previous content have returned
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)((const char*)a.hmask + (((size_t)slot * nt32 + t32) * 64 + lane) * 16),
        (__attribute__((address_space(3))) void*)(msk + (slot & 1) * 1024), 16, 0, 0);
    asm volatile("" ::: "memory");
    rg.count(1);
    if (slot & 1) mmark1 = rg.vmc;
    else mmark0 = rg.vmc;
  };
  auto mask_acquire = [&](int slot, unsigned int (&bits)[4]) {
    wait_vmcnt(rg.vmc - ((slot & 1) ? mmark1 : mmark0));
    asm volatile("" ::: "memory");
    const u32x4_t w = *(const u32x4_t*)(msk + (slot & 1) * 1024 + lane * 16);
    bits[0] = w[0], bits[1] = w[1], bits[2] = w[2], bits[3] = w[3];
    if (slot >= 2) mask_request(slot - 2);
  };
  if (use_cand) {
    mask_request(D);
    mask_request(D - 1);
  } else {
    mask_request(D - 1);
    if (D >= 2) mask_request(D - 2);
  }

  h8 Bh[16];  // operand of the running stage
  h8 Nh[16];  // operand of the next stage
  f32x16 acc;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float amax_in = 0.0f;
  int e_in = 0;

  if (heads) {
    // ---- elementwise head stages, in operand order.  d g2 = relu'(g2) (w_csig dpre_c + w_cj g_G_c[ray]) (candidate_sigma /
    // feat_candidate_layer, nerf.py:99-100); d r1 = relu'(r1) W_r2^T (d rgb * rgb (1 - rgb)) (rgb_share_layer.2 + sigmoid)
    h8 Gh[8];
    float amax_g2 = 0.0f, amax_r1 = 0.0f;
    int e_g2 = 0;
    if (use_cand) {
      float vals[4][4][4];  // [tile][quad][u]
      float vmax = 0.0f;
      const float* gG = rows_s + rs * C::ROWF + 256;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * jt + 8 * q + 4 * hh;
          const f32x4 wv = *(const f32x4*)&vec_s[RB_V_WCSIG + col], gg = *(const f32x4*)&gG[col];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float t = fmaf(wv[u], dpc, cwj * gg[u]);
            const float v = __uint_as_float(__float_as_uint(t) & mask_word(bits_g2, jt, q, u));
            vals[jt][q][u] = v;
            vmax = fmaxf(vmax, fabsf(v));
          }
        }
      amax_g2 = wave_max_rr(vmax);
      track_lds(mx_s, D + 2, amax_g2, lane);
      e_g2 = scale_exp(amax_g2);
      float* __restrict__ dst = a.gz_g2 + (size_t)m0 * W2;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          stg_put(stg, li, hh, q, f32x4{vals[jt][q][0], vals[jt][q][1], vals[jt][q][2], vals[jt][q][3]});
          h4 hi, lo;
          split_quad<1>(ldexpf(vals[jt][q][0], e_g2), ldexpf(vals[jt][q][1], e_g2), ldexpf(vals[jt][q][2], e_g2), ldexpf(vals[jt][q][3], e_g2), hi, lo);
          const int blk = 2 * jt + (q >> 1);
          if (q & 1) Gh[blk] = __builtin_shufflevector(Gh[blk], __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 12, 13, 14, 15);
          else Gh[blk] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3);
        }
        stg_flush(stg, lane, dst, W2, 32 * jt);
        rg.count(4);
      }
    }
    float r1v[4][4][4];
    if (use_rgb) {
      float vmax = 0.0f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * jt + 8 * q + 4 * hh;
          const f32x4 w0 = *(const f32x4*)&vec_s[RB_V_WR2 + col], w1 = *(const f32x4*)&vec_s[RB_V_WR2 + W2 + col],
                      w2 = *(const f32x4*)&vec_s[RB_V_WR2 + 2 * W2 + col];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float t = w0[u] * dprgb[0] + w1[u] * dprgb[1] + w2[u] * dprgb[2];
            const float v = __uint_as_float(__float_as_uint(t) & mask_word(bits_r1, jt, q, u));
            r1v[jt][q][u] = v;
            vmax = fmaxf(vmax, fabsf(v));
          }
        }
      amax_r1 = wave_max_rr(vmax);
      track_lds(mx_s, D + 3, amax_r1, lane);
    }
    // exponent of the joint operand [gz_r1 | gz_g1] of the d e stage
    const float bound_g1 = use_cand ? wnt[D + hs + 3] * amax_g2 : 0.0f;
    const int erg = scale_exp(fmaxf(bound_g1, amax_r1));
    if (use_rgb) {
      float* __restrict__ dst = a.gz_r1 + (size_t)m0 * gld;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          stg_put(stg, li, hh, q, f32x4{r1v[jt][q][0], r1v[jt][q][1], r1v[jt][q][2], r1v[jt][q][3]});
          h4 hi, lo;
          split_quad<1>(ldexpf(r1v[jt][q][0], erg), ldexpf(r1v[jt][q][1], erg), ldexpf(r1v[jt][q][2], erg), ldexpf(r1v[jt][q][3], erg), hi, lo);
          const int blk = 2 * jt + (q >> 1);
          if (q & 1) Bh[blk] = __builtin_shufflevector(Bh[blk], __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 12, 13, 14, 15);
          else Bh[blk] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3);
        }
        stg_flush(stg, lane, dst, gld, 32 * jt);
        rg.count(4);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) Bh[s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    float amax_g1 = 0.0f;
    if (use_cand) {
      // ---- gz_g1 = relu'(g1) (gz_g2 . W_c2)   (candidate_encoding.2, nerf.py:98)
      unsigned int bits[4];
      mask_acquire(D, bits);
      const float un = pow2r(-(e_g2 + wexp[10]));
      float vmax = 0.0f;
      float* __restrict__ dst = a.gz_g1 + (size_t)m0 * gld;
      run_stage<NW, 4, 8>(rg, lds, lane, lag, acc, Gh, [&](auto JP) {
        constexpr int jp = decltype(JP)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) quad_epilogue_b<EB_MASK | EB_CONV>(acc, jp, q, un, zero4, bits, vmax, stg, erg, Bh, 8, li, hh);
        stg_flush(stg, lane, dst, gld, 32 * jp);
        rg.count(4);
      });
      amax_g1 = wave_max_rr(vmax);
      track_lds(mx_s, D + 1, amax_g1, lane);
    } else {
#pragma unroll
      for (int s = 8; s < 16; ++s) Bh[s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    if (use_cand && use_rgb) track_lds(mx_s, D + 4, fmaxf(amax_g1, amax_r1), lane);  // of [gz_r1 | gz_g1] as one tensor (gz_rg_ld)
    // ---- d e = [gz_r1 | gz_g1] . [W_fold | W_c1e] + w_feat g_E_s[ray]   (e has no activation)
    {
      const float un = pow2r(-(erg + wexp[12]));
      const float wfmax = wave_max_rr(fabsf(wf));
      const float bound = wnt[D + hs + 1] * amax_r1 + wnt[D + hs + 2] * amax_g1 + wfmax * gEmax;
      const int eo = scale_exp(bound);
      const float* gE = rows_s + rs * C::ROWF;
      unsigned int nobits[4] = {0u, 0u, 0u, 0u};
      float vmax = 0.0f;
      float* __restrict__ dst = a.gz_e + (size_t)m0 * W;
      run_stage<NW, 8, 16>(rg, lds, lane, lag, acc, Bh, [&](auto JP) {
        constexpr int jp = decltype(JP)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 g = *(const f32x4*)&gE[32 * jp + 8 * q + 4 * hh];
          quad_epilogue_b<EB_CONV>(acc, jp, q, un, g * wf, nobits, vmax, stg, eo, Nh, 0, li, hh);
        }
        stg_flush(stg, lane, dst, W, 32 * jp);
        rg.count(4);
      });
      amax_in = wave_max_rr(vmax);
      track_lds(mx_s, D, amax_in, lane);
      e_in = eo;
#pragma unroll
      for (int s = 0; s < 16; ++s) Bh[s] = Nh[s];
    }
  } else {
    // no head consumed e (density-only evaluation): d e = 0, written because the weight gradient of the final layer reads it
#pragma unroll
    for (int s = 0; s < 16; ++s) Bh[s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
    float* __restrict__ dst = a.gz_e + (size_t)m0 * W;
#pragma unroll
    for (int q = 0; q < 4; ++q) stg_put(stg, li, hh, q, zero4);
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
      stg_flush(stg, lane, dst, W, 32 * jt);
      rg.count(4);
    }
  }
  // ---- d h_{D-1} = relu'(h_{D-1}) (gz_e . W_e + w_sig dpre_s)
  {
    unsigned int bits[4];
    mask_acquire(D - 1, bits);
    const float un = pow2r(-(e_in + wexp[8]));
    float wsmax = 0.0f;
    for (int c = lane; c < W; c += 64) wsmax = fmaxf(wsmax, fabsf(vec_s[RB_V_WSIG + c]));
    wsmax = wave_max_rr(wsmax);
    const float bound = wnt[D + hs] * amax_in + wsmax * wave_max_rr(fabsf(dps));
    const int eo = scale_exp(bound);
    float vmax = 0.0f;
    uint16_t* __restrict__ gzl = a.gz16 + (size_t)(D - 1) * nt32 * 16 * 512;
    run_stage<NW, 8, 16>(rg, lds, lane, lag, acc, Bh, [&](auto JP) {
      constexpr int jp = decltype(JP)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ws = *(const f32x4*)&vec_s[RB_V_WSIG + 32 * jp + 8 * q + 4 * hh];
        quad_epilogue_b<EB_MASK | EB_CONV>(acc, jp, q, un, ws * dps, bits, vmax, nullptr, eo, Nh, 0, li, hh);
      }
      frag_store(gzl, t32, 2 * jp, lane, Nh[2 * jp]);
      frag_store(gzl, t32, 2 * jp + 1, lane, Nh[2 * jp + 1]);
      rg.count(2);
    });
    if (lane == 0) a.gzexp[(size_t)(D - 1) * nt32 + t32] = eo;
    amax_in = wave_max_rr(vmax);
    track_lds(mx_s, D - 1, amax_in, lane);
    e_in = eo;
#pragma unroll
    for (int s = 0; s < 16; ++s) Bh[s] = Nh[s];
  }
  // ---- trunk, last layer to first: gz_{l-1} = relu'(h_{l-1}) (gz_l . W_l); the skip layer also feeds d x0
  f32x16 accx[2];
  acc_clear(accx[0]);
  acc_clear(accx[1]);
#pragma unroll 1
  for (int l = D - 1; l >= 1; --l) {
    const int wel = wexp[l];
    if (need_dxyz && L.skip > 0 && l == L.skip) {
      // d x0 += gz_skip . W_skip[:, :64]: two tiles of 32 encoding features (the stage is issued twice: slots stay aligned)
      const float unx = pow2r(-(e_in + wel));
      static_for<0, 4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        rg.template begin<(j & 3)>(lds);
        if constexpr (j < 2) {
          const char* p = lds + (j & 3) * RR_SLOT + lane * 16;
          acc_clear(acc);
          kpart<16, false>(acc, p, Bh, [](int) {});
#pragma unroll
          for (int r = 0; r < 16; ++r) accx[j][r] = acc[r] * unx;
        }
      });
    }
    unsigned int bits[4];
    mask_acquire(l - 1, bits);
    const float un = pow2r(-(e_in + wel));
    const int eo = scale_exp(wnt[l + ((hs && l >= L.skip) ? 1 : 0)] * amax_in);
    float vmax = 0.0f;
    uint16_t* __restrict__ gzl = a.gz16 + (size_t)(l - 1) * nt32 * 16 * 512;
    run_stage<NW, 8, 16>(rg, lds, lane, lag, acc, Bh, [&](auto JP) {
      constexpr int jp = decltype(JP)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q) quad_epilogue_b<EB_MASK | EB_CONV>(acc, jp, q, un, zero4, bits, vmax, nullptr, eo, Nh, 0, li, hh);
      frag_store(gzl, t32, 2 * jp, lane, Nh[2 * jp]);
      frag_store(gzl, t32, 2 * jp + 1, lane, Nh[2 * jp + 1]);
      rg.count(2);
    });
    if (lane == 0) a.gzexp[(size_t)(l - 1) * nt32 + t32] = eo;
    amax_in = wave_max_rr(vmax);
    track_lds(mx_s, l - 1, amax_in, lane);
    e_in = eo;
#pragma unroll
    for (int s = 0; s < 16; ++s) Bh[s] = Nh[s];
  }
  if (need_dxyz) {
    // ---- d x0 += gz_0 . W_0, then d xyz through the encoding
    const float unx = pow2r(-(e_in + wexp[0]));
    static_for<0, 4>([&](auto J) {
      constexpr int j = decltype(J)::value;
      rg.template begin<(j & 3)>(lds);
      if constexpr (j < 2) {
        const char* p = lds + (j & 3) * RR_SLOT + lane * 16;
        acc_clear(acc);
        kpart<16, false>(acc, p, Bh, [](int) {});
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[j][r] = fmaf(acc[r], unx, accx[j][r]);
      }
    });
    __syncthreads();  // every wave is done with the ring: it becomes the exchange scratch of d x0 (64 floats per row)
    float* Gs = (float*)lds + (32 * wave + li) * UPNERF_X0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4*)&Gs[32 * t + 8 * q + 4 * hh] = f32x4{accx[t][4 * q], accx[t][4 * q + 1], accx[t][4 * q + 2], accx[t][4 * q + 3]};
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // lane half 0: coordinates 0 and 1, lane half 1: coordinate 2
    if (valid) {
      const float* __restrict__ x0r = a.x0 + (size_t)m * UPNERF_X0;
      for (int n = hh ? 2 : 0; n < (hh ? 3 : 2); ++n) {
        float xs[10], xc[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          xs[k] = x0r[3 + 20 * n + k];
          xc[k] = x0r[3 + 20 * n + 10 + k];
        }
        float g = Gs[n];
#pragma unroll
        for (int k = 0; k < 10; ++k) g += ldexpf(PI_F, k) * (xc[k] * Gs[3 + 20 * n + k] - xs[k] * Gs[3 + 20 * n + 10 + k]);
        a.dxyz[(size_t)m * 3 + n] = g;
      }
    }
  }
  // running maxima -> global table, once per workgroup
  __syncthreads();
  if (a.gmax && tid < 14) {
    const unsigned int v = mx_s[tid];
    if (v) atomicMax((unsigned int*)a.gmax + tid, v);
  }
}

}  // namespace

// Entry points behind upnerf_field_fwd_f16x3 / upnerf_field_bwd_f16x3 (csrc/field16.hip validates the arguments): planes = 1 and
// tile_rows = 256.  Needs a->P16 / PT16 written by upnerf_frag16 with perm = 1 and a->wnorm from the same call.
int upnerf_rr16_bwd_launch(const upnerf_layout* L, const upnerf_field_bwd_args* a, void* stream) {
  const long long M = (long long)a->R * a->S;
  const int grid = (int)((M + 255) / 256);
  hipLaunchKernelGGL((rr16_bwd_kernel<8>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *L, *a);
  return (int)hipGetLastError();
}

int upnerf_rr16_fwd_launch(const upnerf_layout* L, const upnerf_field_fwd_args* a, void* stream) {
  const long long M = (long long)a->R * a->S;
  const int grid = (int)((M + 255) / 256);
  hipLaunchKernelGGL((rr16_fwd_kernel<8>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *L, *a);
  return (int)hipGetLastError();
}
