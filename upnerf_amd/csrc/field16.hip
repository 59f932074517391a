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
#include <type_traits>

// Tiling (template parameters TILE, NW): 64 samples x 4 waves, two workgroups per CU: a wave owns 64 columns x all 64 rows
// (MT = 2, NT = 2); every workgroup pulls a layer's whole weight matrix from L2 (4 KB per sample and layer in the f16x3 mode).
// Round 3's 128-sample software-pipelined variant left the library (tools/repro/pipe16); the fp16 mode's register-resident
// kernels with LDS-staged weights are csrc/field16rr.hip.
#define ACT_STORE(p, v) NT_STORE(p, v)  // activations / gradients for a later kernel (common.cuh: streaming accesses)
#define F16_TILE 64
#ifndef F16_WAVES
#define F16_WAVES 4  // waves of a 64-sample workgroup (experiment: 8 = one 32-column tile per wave, four waves per SIMD)
#endif
// two workgroups per CU = 2 waves per SIMD; hipcc takes the second __launch_bounds__ argument as the minimum number of
// waves per SIMD
#define F16_WAVES_PER_EU 2
// f16 mode (one plane: 42-48 KB of LDS per 64-sample workgroup): a third workgroup per CU fits if a wave stays within 168 registers
#ifndef F16_WAVES_PER_EU_F16
#define F16_WAVES_PER_EU_F16 2
#endif
#define F16_EU(NP, TILE) ((TILE) == 64 && F16_WAVES == 8 ? 4 : ((NP) == 1 && (TILE) == 64 ? F16_WAVES_PER_EU_F16 : F16_WAVES_PER_EU))
// Weight fragments are requested this many 16-deep k-blocks ahead of the MFMAs that consume them.  Measured with the stamps
// build (per wave and trunk layer): f16 mode 10.4k cycles per K loop one block ahead = 650 cycles per k-block = the L2
// latency, 8.1k three ahead and 8.3k seven ahead -- from there on the loop is bound by the bytes the CU can pull from L2
// (~32 B/clk: 128 KB of fp16 weights per 64-row tile and layer).  f16x3 mode: TWO ahead (a ring of three fragment sets, the K
// loop fully unrolled: common16.cuh) -- 2.57 / 2.50 ms per forward / backward launch against 2.66 / 2.58 one ahead and
// 2.60 / 2.52 three ahead (round 3, same box, alternating runs): one block of twelve MFMAs does not cover the L2 latency
// when the partner workgroup is in its epilogue, three blocks cost registers the epilogue needs.
#ifndef F16_AHEAD_X3
#define F16_AHEAD_X3 2
#endif
#ifndef F16_STORE_IN_LOOP
// Activation stores as one 1 KiB piece per k-block INSIDE the next layer's K loop (PlaneStore).  Built, parity-green,
// measured slower (round 5, alternating runs on one box: forward launch 2.61 / 2.58 ms against 2.52 / 2.54 behind the loop): the K loop
// already keeps the CU's vector-memory path at the rate it sustains under this mix, a store per k-block lengthens it by what the
// burst behind the loop costs and the epilogue still waits for its barriers.  0 = behind the loop (shipped); 1 = the experiment;
// 2 (round 6) = the experiment with the piece's LDS words read one k-block ahead of their store.  In the trunk probe -- whose
// workgroups are exact copies of each other and stay in phase -- 2 beats the burst (2.465 against 2.585 ms); in this kernel, whose
// two workgroups per CU drift apart through their encoding and head stages and so already overlap one's burst with the other's K
// loop, it does not: forward launch 2.431 (2) / 2.449 (1) against 2.374 ms, alternating on one box (profiles/r06_ab_stores_in_loop.txt).
#define F16_STORE_IN_LOOP 0
#endif
// the backward head stage's sums of the per-row scalars: 1 = one wave, a row per lane, butterfly (round 6); 0 = four threads walking
// the tile's 64 rows while the other 252 wait at the barrier behind them
#ifndef F16_BH_SCALAR_SUMS_WAVE
#define F16_BH_SCALAR_SUMS_WAVE 1
#endif
#ifndef F16_JOIN_HEADS
#define F16_JOIN_HEADS 1  // the first layers of the colour and the candidate head in one K loop when both run (0: one after the other, for A/B runs)
#endif
#ifndef F16_AHEAD_F16
#define F16_AHEAD_F16 5  // (with non-temporal stores: 19.63 ms per Trevi step against 19.78 at three ahead; round 3)
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
// make stamps EXP=-DUPNERF_STAMPS_HEADS: slots 8..15 hold eight pieces of the FORWARD kernel's head stage instead of the backward
// kernel's stages (tools/stamps_field16.py --heads)
// ... EXP=-DUPNERF_STAMPS_BHEADS: slots 8..15 hold seven pieces of the BACKWARD kernel's head stage (--bheads)
#if defined(UPNERF_STAMPS) && defined(UPNERF_STAMPS_BHEADS)
#define BHSTAMP(i) STAMP(i)
#define BH_ONLY 1
#else
#define BHSTAMP(i)
#define BH_ONLY 0
#endif
#if defined(UPNERF_STAMPS) && defined(UPNERF_STAMPS_HEADS)
#define HSTAMP(i) STAMP(i)
#define HSTAMP_RESET                               \
  do {                                             \
    for (int _i = 0; _i < 8; ++_i) _t_acc[_i] = 0; \
    _t_prev = __builtin_amdgcn_s_memtime();        \
  } while (0)
#define BSTAMP_FLUSH_AT(base)
#else
#define HSTAMP(i)
#define HSTAMP_RESET
#if BH_ONLY
#define BSTAMP_FLUSH_AT(base)
#else
#define BSTAMP_FLUSH_AT(base) STAMP_FLUSH_AT(base)
#endif
#endif

namespace {

// How the 4 waves of a workgroup share a [TILE x N] output tile (32 x 32 MFMA tiles); same rules as WaveTile in
// common.cuh: surplus waves of a narrow layer recompute a piece another wave owns.
template <int N, int TILE, int NWAVES>
struct WaveTile16 {
  static constexpr int NW = NWAVES;
  static constexpr int MG = TILE / 32;                       // 32-row m-tiles of the workgroup's tile
  static constexpr int WN = (N / 32) >= NW ? NW : (N / 32);  // waves side by side along N
  static constexpr int NT = (N / 32) / WN;                   // n-tiles per wave
  static constexpr int WM = (NW / WN) < MG ? (NW / WN) : MG; // row groups
  static constexpr int MT = MG / WM;                         // m-tiles per wave
  __device__ static __forceinline__ int n0(int wave) { return (wave % WN) * 32 * NT; }
  __device__ static __forceinline__ int row0(int wave) { return ((wave / WN) % WM) * 32 * MT; }
};

template <int NW>
__device__ __forceinline__ float wg_max(const float* smax) {
  float m = smax[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) m = fmaxf(m, smax[w]);
  return m;
}

__device__ __forceinline__ float pow2f(int n) { return ldexpf(1.0f, n); }

// LDS planes -> row-major fp32 global tensor in WHOLE LINES: a thread converts four columns (8 bytes of each plane) and
// stores 16 bytes; consecutive threads take consecutive chunks of a row, so one wave-wide store is 1 KiB contiguous (a
// whole 256-column row, two 128-column rows, four 64-column rows).  Stores shaped by the MFMA layouts instead (a lane = a
// row, 16 or 32 bytes of it) touch 32 lines per instruction: the vector memory path then moves ~14 B/clk/CU and HBM takes
// partial lines at ~3.5 TB/s; whole lines were measured at about twice that (round 3).  The LDS reads of BATCH
// iterations are issued before the first conversion.
template <int NP, int W, int TILE, int THREADS, int NCOLS, int BATCH = 4, bool STREAM = true>
__device__ __forceinline__ void tile_store16(const char* Ph, const char* Pl, int c0, float unscale, float unscale2,
                                             float* __restrict__ dst, int ldg, int m0, int M, int tid) {
  constexpr int GPR = NCOLS >> 2, ITER = TILE * GPR / THREADS;
  static_assert(TILE * GPR % THREADS == 0, "whole passes of the workgroup");
  constexpr int B = (BATCH == 0 || BATCH > ITER) ? ITER : BATCH;
  static_assert(ITER % B == 0, "whole batches");
#if F16_WAVES == 8 || defined(F16_OPAQUE_STORE)
  asm volatile("" : "+v"(tid));  // (128-register build: keep hipcc from hoisting every row address out of the caller's layer loop)
#endif
  const bool whole = m0 + TILE <= M;
#pragma unroll 1
  for (int it0 = 0; it0 < ITER; it0 += B) {
    u32x2_t wh[B], wl[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int idx = tid + (it0 + j) * THREADS, row = idx / GPR, g = idx % GPR;
      const int o = poff<W>(row, c0 + 4 * g);
      wh[j] = *(const u32x2_t*)(Ph + o);
      if constexpr (NP == 2) wl[j] = *(const u32x2_t*)(Pl + o);
    }
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int idx = tid + (it0 + j) * THREADS, row = idx / GPR, g = idx % GPR;
      const float un = row < TILE / 2 ? unscale : unscale2;  // the two row halves may carry different exponents
      f32x4 v;
      if constexpr (NP == 2)
        v = f32x4{mix16<0>(wh[j][0], un, mix16<0>(wl[j][0], un, 0.f)), mix16<1>(wh[j][0], un, mix16<1>(wl[j][0], un, 0.f)),
                  mix16<0>(wh[j][1], un, mix16<0>(wl[j][1], un, 0.f)), mix16<1>(wh[j][1], un, mix16<1>(wl[j][1], un, 0.f))};
      else
        v = f32x4{mix16<0>(wh[j][0], un, 0.f), mix16<1>(wh[j][0], un, 0.f), mix16<0>(wh[j][1], un, 0.f), mix16<1>(wh[j][1], un, 0.f)};
#ifdef UPNERF_EXP_HALFROW
      if (NCOLS == 256 && g >= UPNERF_EXP_HALFROW) continue;
#endif
      if (whole || m0 + row < M) {
        if constexpr (STREAM) ACT_STORE((f32x4*)&dst[(size_t)(m0 + row) * ldg + 4 * g], v);
        else *(f32x4*)&dst[(size_t)(m0 + row) * ldg + 4 * g] = v;  // re-read by this workgroup soon (x0 at the skip layer)
      }
    }
  }
}
template <int NP, int W, int TILE, int THREADS, int NCOLS, int BATCH = 4, bool STREAM = true>
__device__ __forceinline__ void tile_store16(const char* Ph, const char* Pl, int c0, float unscale, float* __restrict__ dst,
                                             int ldg, int m0, int M, int tid) {
  tile_store16<NP, W, TILE, THREADS, NCOLS, BATCH, STREAM>(Ph, Pl, c0, unscale, unscale, dst, ldg, m0, M, tid);
}

// tile_store16 of a full-width (W-column) tensor cut into its 1 KiB pieces -- piece t = the workgroup's store pass t of
// tile_store16: rows 4 t .. 4 t + 3 at 256 threads -- for issue INSIDE the K loop of the next contraction, one piece per k-block
// (common16.cuh:mma16_lds `piece`): that loop reads the same planes, nothing writes them before its closing barrier, and the wave
// that is held by a store's issue (~260 cycles) leaves the matrix pipe to its SIMD partner instead of idling it
// together with all eight waves of the CU in a burst of sixteen stores behind the loop.  Whole tiles only (no row guard: no control flow in the K loop --
// a branch there makes hipcc's vmcnt bookkeeping conservative and drains the weight ring).
template <int NP, int W, int TILE, int THREADS>
struct PlaneStore {
  static constexpr int GPR = W >> 2, ITER = TILE * GPR / THREADS;
  const char *Ph, *Pl;
  float* __restrict__ dst;  // row m0 of the tensor
  float un0, un1;
  int tid;
  u32x2_t ch, cl;
  __device__ __forceinline__ void read(int it, u32x2_t& h, u32x2_t& l) const {
    const int idx = tid + it * THREADS, row = idx / GPR, g = idx % GPR;
    const int o = poff<W>(row, 4 * g);
    h = *(const u32x2_t*)(Ph + o);
    if constexpr (NP == 2) l = *(const u32x2_t*)(Pl + o);
  }
  __device__ __forceinline__ PlaneStore(const char* ph, const char* pl, float* d, float u0, float u1, int t)
      : Ph(ph), Pl(pl), dst(d), un0(u0), un1(u1), tid(t) {
    // (opaque per layer: the pieces' row offsets are layer-invariant, hipcc hoisted all sixteen address pairs out of the layer
    // loop, spilled them and reloaded them inside the K loop -- scratch loads, i.e. vmcnt waits behind the stores)
    asm volatile("" : "+v"(tid));
  }
#if F16_STORE_IN_LOOP == 2
  // Round 6: the piece's LDS words are read ONE K-BLOCK AHEAD of their conversion and store (operator()(t) stores piece t - 1 and reads
  // piece t; flush(ITER - 1) behind the loop): the store's LDS round trip passes under the previous k-block's MFMAs instead of standing
  // in front of this one's.  In the trunk probe (tools/repro/pair_trunk_probe.hip, mode 12) that turns the in-loop stores from a loss
  // (2.70 against 2.585 ms) into a gain (2.465).
  __device__ __forceinline__ void flush(int it) const {
    const int idx = tid + it * THREADS, row = idx / GPR, g = idx % GPR;
    const float un = row < TILE / 2 ? un0 : un1;
    f32x4 v;
    if constexpr (NP == 2)
      v = f32x4{mix16<0>(ch[0], un, mix16<0>(cl[0], un, 0.f)), mix16<1>(ch[0], un, mix16<1>(cl[0], un, 0.f)),
                mix16<0>(ch[1], un, mix16<0>(cl[1], un, 0.f)), mix16<1>(ch[1], un, mix16<1>(cl[1], un, 0.f))};
    else
      v = f32x4{mix16<0>(ch[0], un, 0.f), mix16<1>(ch[0], un, 0.f), mix16<0>(ch[1], un, 0.f), mix16<1>(ch[1], un, 0.f)};
    ACT_STORE((f32x4*)&dst[(size_t)row * W + 4 * g], v);
  }
  __device__ __forceinline__ void operator()(int it) {
    if (it > 0) flush(it - 1);
    read(it, ch, cl);
  }
#else
  __device__ __forceinline__ void flush(int) const {}
  __device__ __forceinline__ void operator()(int it) {
    // (read, convert and store in one go: carrying the next piece's LDS words across the MFMAs cost the forward kernel, which
    // sits at 256 registers, 47 spilled registers)
    read(it, ch, cl);
    const int idx = tid + it * THREADS, row = idx / GPR, g = idx % GPR;
    const float un = row < TILE / 2 ? un0 : un1;
    f32x4 v;
    if constexpr (NP == 2)
      v = f32x4{mix16<0>(ch[0], un, mix16<0>(cl[0], un, 0.f)), mix16<1>(ch[0], un, mix16<1>(cl[0], un, 0.f)),
                mix16<0>(ch[1], un, mix16<0>(cl[1], un, 0.f)), mix16<1>(ch[1], un, mix16<1>(cl[1], un, 0.f))};
    else
      v = f32x4{mix16<0>(ch[0], un, 0.f), mix16<1>(ch[0], un, 0.f), mix16<0>(ch[1], un, 0.f), mix16<1>(ch[1], un, 0.f)};
    ACT_STORE((f32x4*)&dst[(size_t)row * W + 4 * g], v);
  }
#endif
};

// tile_store16 for a half-width tensor that also adds every stored row to the accumulator of the row's ray slot (sums[slot],
// this thread's four columns; slot_s[row] < NS): the per-tile part of upnerf_ray_sum (see upnerf_field_bwd_args.tile_part).
template <int NP, int W, int TILE, int THREADS, int NCOLS, int NS>
__device__ __forceinline__ void tile_store16_sum(const char* Ph, const char* Pl, int c0, float un, float* __restrict__ dst,
                                                 int ldg, int m0, int M, int tid, const int* slot_s, f32x4 (&sums)[NS]) {
  constexpr int GPR = NCOLS >> 2, ITER = TILE * GPR / THREADS, B = 4;
  static_assert(TILE * GPR % THREADS == 0 && ITER % B == 0, "whole batches of whole passes");
#pragma unroll 1
  for (int it0 = 0; it0 < ITER; it0 += B) {
    u32x2_t wh[B], wl[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int idx = tid + (it0 + j) * THREADS, row = idx / GPR, g = idx % GPR;
      const int o = poff<W>(row, c0 + 4 * g);
      wh[j] = *(const u32x2_t*)(Ph + o);
      if constexpr (NP == 2) wl[j] = *(const u32x2_t*)(Pl + o);
    }
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int idx = tid + (it0 + j) * THREADS, row = idx / GPR, g = idx % GPR;
      f32x4 v;
      if constexpr (NP == 2)
        v = f32x4{mix16<0>(wh[j][0], un, mix16<0>(wl[j][0], un, 0.f)), mix16<1>(wh[j][0], un, mix16<1>(wl[j][0], un, 0.f)),
                  mix16<0>(wh[j][1], un, mix16<0>(wl[j][1], un, 0.f)), mix16<1>(wh[j][1], un, mix16<1>(wl[j][1], un, 0.f))};
      else
        v = f32x4{mix16<0>(wh[j][0], un, 0.f), mix16<1>(wh[j][0], un, 0.f), mix16<0>(wh[j][1], un, 0.f), mix16<1>(wh[j][1], un, 0.f)};
      const bool in = m0 + row < M;
      if (in) ACT_STORE((f32x4*)&dst[(size_t)(m0 + row) * ldg + 4 * g], v);
      const int sl = slot_s[row];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const float k = (in && sl == q) ? 1.0f : 0.0f;
        sums[q] += v * k;
      }
    }
  }
}

// The hi plane of the tile (columns [0, W)) -> row-major fp16 global tensor, as it stands (16 bytes per thread and step);
// the tile's exponent goes to its slot(s) of the exponent table, which has one entry per 64 rows whatever the kernel's
// tile (the weight-gradient kernel reads it per 64-row block).  f16 storage only.
template <int W, int TILE, int THREADS>
__device__ __forceinline__ void tile_copy16(const char* Ph, int e, int e2, uint16_t* __restrict__ dst, int32_t* __restrict__ dexp,
                                            int m0, int M, int tid) {
  constexpr int GPR = W >> 3, ITER = TILE * GPR / THREADS;
  if (tid < TILE / 64 && m0 + 64 * tid < M) dexp[m0 / 64 + tid] = tid == 0 ? e : e2;
#if F16_WAVES == 8  // (128-register build; in the default build the hoisted, partly spilled offsets measured FASTER than recomputing them)
  asm volatile("" : "+v"(tid));
#endif
  if (dexp == nullptr) asm volatile("" : "+v"(tid));
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * THREADS, row = idx / GPR, g = idx % GPR;
    const f32x4 v = *(const f32x4*)(Ph + poff<W>(row, 8 * g));
    if (m0 + row < M) ACT_STORE((f32x4*)((char*)dst + ((size_t)(m0 + row) * W + 8 * g) * 2), v);
  }
}
// The lo plane of the tile -> row-major byte tensor: the rounding residual of every element in 1/32 of the tile's scaled unit,
// byte = round(32 lo) + 128 (|lo| <= half an ulp of a value below 2^14: |32 lo| <= 128, clamped to 127).  With the hi plane
// (tile_copy16) that is the "24-bit" storage of the weight-gradient operands (upnerf_wgrad_f24p): hi + lo to 2^-20 of the
// tile's maximum in 3 bytes.  The fp16 magic-number add rounds to nearest and leaves the byte in the low mantissa bits.
template <int W, int TILE, int THREADS>
__device__ __forceinline__ void tile_copy8(const char* Pl, uint8_t* __restrict__ dst, int m0, int M, int tid) {
  constexpr int GPR = W >> 4, ITER = TILE * GPR / THREADS;  // 16 elements = 16 output bytes per thread and step
  const h8 k32 = {32, 32, 32, 32, 32, 32, 32, 32}, lim = {127, 127, 127, 127, 127, 127, 127, 127}, magic = {1408, 1408, 1408, 1408, 1408, 1408, 1408, 1408};
  // (offsets re-derived per call: hoisted out of the layer loop they spill, and every scratch reload is a vmcnt(0) wait behind
  // the stores in flight)
  asm volatile("" : "+v"(tid));
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * THREADS, row = idx / GPR, g = idx % GPR;
    u32x4_t out;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      h8 v = *(const h8*)(Pl + poff<W>(row, 16 * g + 8 * hf)) * k32;
      v = __builtin_elementwise_min(__builtin_elementwise_max(v, -lim), lim) + magic;  // 1024 + 256 + 128 + q: low byte = q + 128
      const u32x4_t w = __builtin_bit_cast(u32x4_t, v);
      out[2 * hf] = __builtin_amdgcn_perm(w[1], w[0], 0x06040200u);
      out[2 * hf + 1] = __builtin_amdgcn_perm(w[3], w[2], 0x06040200u);
    }
    if (m0 + row < M) ACT_STORE((u32x4_t*)(dst + (size_t)(m0 + row) * W + 16 * g), out);
  }
}
template <int W, int TILE, int THREADS>
__device__ __forceinline__ void tile_copy16(const char* Ph, int e, uint16_t* __restrict__ dst, int32_t* __restrict__ dexp, int m0,
                                            int M, int tid) {
  tile_copy16<W, TILE, THREADS>(Ph, e, e, dst, dexp, m0, M, tid);
}

__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  return m;
}

// Running maxima (scales of the weight-gradient contraction) are collected per workgroup in an LDS table and folded into the
// global table ONCE, by the last instructions of the kernel: a global atomic counts in the issuing wave's vmcnt until it has
// been performed at its L2 channel -- behind every other workgroup's atomic on the same address -- and the wave's next wait
// for a weight fragment waits for it too (vmcnt retires in order).  In the pipelined trunk that cost 20k cycles per phase.
__device__ __forceinline__ void track(unsigned int* mx_s, int slot, float mx, int tid) {
  if (tid == 0) mx_s[slot] = max(mx_s[slot], __float_as_uint(mx));  // non-negative floats order like their bit patterns
}
__device__ __forceinline__ void track_wave(unsigned int* mx_s, int slot, float mx, int lane) {
  if (lane == 0) atomicMax(&mx_s[slot], __float_as_uint(mx));  // LDS atomic (several waves at once)
}
__device__ __forceinline__ void track_flush(const unsigned int* mx_s, float* __restrict__ dst, int tid) {
  __syncthreads();
  if (dst && tid < 16) {
    const unsigned int v = mx_s[tid];
    if (v) atomicMax((unsigned int*)dst + tid, v);
  }
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
template <int NP, int TILE, int NW>
__global__ __launch_bounds__(64 * NW, F16_EU(NP, TILE)) void field16_fwd_kernel(upnerf_layout L, upnerf_field_fwd_args a) {
  constexpr int W = 256, W2 = 128, THREADS = 64 * NW;
  constexpr int AH = NP == 2 ? F16_AHEAD_X3 : F16_AHEAD_F16;  // weight k-blocks in flight (common16.cuh:mma16_lds)
  __shared__ __attribute__((aligned(16))) char planes[NP * TILE * W * 2];
  __shared__ float smax[NW], smaxb[NW];
  __shared__ unsigned int mx_s[16];  // running maxima of this workgroup (track / track_flush)
  __shared__ float xyz_s[TILE * 3];
  // per-layer offsets of the layout: read from LDS inside the layer loop.  Indexing the by-value struct with the runtime
  // layer makes hipcc copy it to scratch and fetch the entry with a VMEM load + s_waitcnt vmcnt(0) -- a full drain of the
  // previous layer's activation stores at the top of every layer.
  __shared__ int loff_s[2 * UPNERF_MAX_D];
  // weight exponents (wexp, WEXP_SLOTS ints): an epilogue that read its exponent from global memory opened with an L2 round trip
  // whose wait -- vmcnt retires in order -- also drained every store the previous stage had just issued (round 5: the layer
  // loop's `s_waitcnt vmcnt(0)` in front of the first weight request)
  __shared__ int wexp_s[WEXP_SLOTS];
  __shared__ float wk_s[10];
  // trunk biases [D][W]: the epilogues read them from LDS
  __shared__ __attribute__((aligned(16))) float bias_s[UPNERF_MAX_D * W];
  // the 1- and 3-wide heads' weights as scaled fp16 (hi, lo), split once per workgroup (common16.cuh: head16_mfma)
  __shared__ __attribute__((aligned(16))) _Float16 hw_hi[HEAD_STAGE_N], hw_lo[HEAD_STAGE_N];
  __shared__ float hw_max[NW];
  static_assert(HEAD_STAGE_N == W + 4 * W2 && HEAD_STAGE_N == 3 * THREADS && TILE == 16 * NW, "head staging: three weights per thread, 16 rows per wave");
  char* Ph = planes;
  char* Pl = planes + (NP - 1) * TILE * W * 2;  // NP == 1: never dereferenced
  using TW = WaveTile16<W, TILE, NW>;
  using TH = WaveTile16<W2, TILE, NW>;
#if F16_WAVES == 8 || defined(F16_SCALAR_WAVE)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: no 64-bit per-lane bases)
#else
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#endif
  const int li = lane & 31, hh = lane >> 5;
  const int S = a.S, M = a.R * a.S, m0 = blockIdx.x * TILE;
  const float* __restrict__ P = a.P;
  const char* __restrict__ P16 = (const char*)a.P16;
  const int* wexp = wexp_s;  // (filled below; the first reader sits behind several barriers)
  const int n0 = TW::n0(wave), row0 = TW::row0(wave);
  const int hn0 = TH::n0(wave), hrow0 = TH::row0(wave);
  const int D = L.D;
  STAMP_DECL;
#ifdef UPNERF_EXP_ASYMPRIO
  // experiment: ONE of the two waves that share a SIMD (odd hardware wave slot) runs at a raised priority for the whole kernel --
  // does an asymmetric pair settle into complementary phases (one workgroup's K loop beside the other's epilogue)?
  if (__builtin_amdgcn_s_getreg(6148 /* HW_REG_HW_ID, wave_id[3:0] */) & 1) __builtin_amdgcn_s_setprio(UPNERF_EXP_ASYMPRIO);
#endif
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < UPNERF_MAX_D; ++l) {
      loff_s[l] = L.w[l];
      loff_s[UPNERF_MAX_D + l] = L.b[l];
    }
  }
  if (tid < 16) mx_s[tid] = 0u;
  // ---- every load of the prologue, requested back to back and UNCONDITIONALLY (clamped indices, values masked afterwards).
  // Behind `if (tid in range)` / `if (l < D)` hipcc branches around each load and waits for it before the next one: the ISA of
  // round 5 opened this kernel with eleven `global_load_dword ; s_waitcnt vmcnt(0)` pairs in a row (weight exponents, band
  // weights, eight bias rows) and three more waits for the sample positions and the side-input maxima -- ~9k of the prologue's
  // 21.7k cycles per tile were L2 / HBM round trips taken one at a time.
  static_assert(W == THREADS && TILE <= THREADS && WEXP_SLOTS <= THREADS, "one bias column, one sample row per thread");
  const int wexp_v = a.wexp[tid & (WEXP_SLOTS - 1)];
  // the ten band weights of this step (device table under graph replay): one request here instead of one L2 round trip per trip
  // of the encoding loop below (round 5 stamps: the stage took 24.9k cycles per tile with five of ten bands skipped, as with none)
  const float* __restrict__ wk_src = a.wk_xyz_dev ? a.wk_xyz_dev : P;
  float wk_v = wk_src[tid < 10 ? tid : 9];
  float bias_v[UPNERF_MAX_D];
#pragma unroll
  for (int l = 0; l < UPNERF_MAX_D; ++l) bias_v[l] = P[(l < D ? L.b[l] : L.b[0]) + tid];
  // the head weights of this thread (staging index tid, tid + 256, tid + 512): split behind the barrier below
  float hwv[3];
  hwv[0] = P[L.wsig + tid];
  hwv[1] = P[L.wr2 + tid];                                             // W_r2 rows 0, 1 (256 of its 384 entries)
  hwv[2] = P[tid < W2 ? L.wr2 + 2 * W2 + tid : L.wcsig + tid - W2];     // W_r2 row 2 | w_csigma
  // sample positions (rendering.py:251 / 308): row tid % TILE, clamped to the last sample
  const int ms = m0 + (tid & (TILE - 1)), msc = ms < M ? ms : M - 1, rs = msc / S;
  const float zz = a.z[msc];
  const float ox = a.rays_o[3 * rs + 0], oy = a.rays_o[3 * rs + 1], oz = a.rays_o[3 * rs + 2];
  const float dx = a.rays_d[3 * rs + 0], dy = a.rays_d[3 * rs + 1], dz = a.rays_d[3 * rs + 2];
  // the maxima that bound the side inputs of this tile: the rows of its rays (one pass of the workgroup unless S < 32)
  const int mlast = (m0 + TILE < M ? m0 + TILE : M) - 1;
  const int ray0 = m0 / S, nr = mlast / S - ray0 + 1;
  const float* __restrict__ aux_src = a.use_rgb ? a.aux + (size_t)ray0 * UPNERF_AUXK : P;   // (a pointer select, not a branch around the load)
  const float* __restrict__ crow_src = a.use_cand ? a.c_rows + (size_t)ray0 * UPNERF_CK : P;
  const float av = aux_src[a.use_rgb && tid < nr * UPNERF_AUXK ? tid : 0];
  const float cv = crow_src[a.use_cand && tid < nr * UPNERF_CK ? tid : 0];
  if (!a.wk_xyz_dev) wk_v = a.wk_xyz[tid < 10 ? tid : 9];
  if (tid < WEXP_SLOTS) wexp_s[tid] = wexp_v;
  if (tid < 10) wk_s[tid] = wk_v;
#pragma unroll
  for (int l = 0; l < UPNERF_MAX_D; ++l) bias_s[l * W + tid] = bias_v[l];  // (rows >= D: copies of row 0, never read)
  {
    float xm = 0.0f;
    if (tid < TILE) {
      float x = 0.f, y = 0.f, zc = 0.f;
      if (ms < M) {
        x = mul_then_add(ox, dx, zz);
        y = mul_then_add(oy, dy, zz);
        zc = mul_then_add(oz, dz, zz);
      }
      xyz_s[tid * 3 + 0] = x;
      xyz_s[tid * 3 + 1] = y;
      xyz_s[tid * 3 + 2] = zc;
      xm = fmaxf(fmaxf(fabsf(x), fabsf(y)), fmaxf(fabsf(zc), 1.0f));  // |sin|, |cos| <= 1
    }
    float sm = fmaxf(a.use_rgb && tid < nr * UPNERF_AUXK ? fabsf(av) : 0.0f, a.use_cand && tid < nr * UPNERF_CK ? fabsf(cv) : 0.0f);
    if (a.use_rgb)
      for (int idx = tid + THREADS; idx < nr * UPNERF_AUXK; idx += THREADS) sm = fmaxf(sm, fabsf(a.aux[(size_t)ray0 * UPNERF_AUXK + idx]));
    if (a.use_cand)
      for (int idx = tid + THREADS; idx < nr * UPNERF_CK; idx += THREADS) sm = fmaxf(sm, fabsf(a.c_rows[(size_t)ray0 * UPNERF_CK + idx]));
    xm = wave_max(xm);
    sm = wave_max(sm);
    const float hm = wave_max(fmaxf(fmaxf(fabsf(hwv[0]), fabsf(hwv[1])), fabsf(hwv[2])));
    if (lane == 0) {
      smax[wave] = xm;
      smaxb[wave] = sm;
      hw_max[wave] = hm;
    }
  }
  __syncthreads();
  const float x0max = wg_max<NW>(smax), sidemax = wg_max<NW>(smaxb);
  const int hwexp = scale_exp(wg_max<NW>(hw_max));  // one exponent for the three heads: largest |w| -> [2^13, 2^14)
  {
    const float sc = pow2f(hwexp);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      _Float16 h, l;
      split16(hwv[j] * sc, h, l);
      hw_hi[tid + j * THREADS] = h;
      hw_lo[tid + j * THREADS] = l;
    }
  }  // (visible behind the barrier that closes the encoding)
  track(mx_s, D + 4, x0max, tid);
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
    for (int it = tid; it < TILE * 3; it += THREADS) {
      const int row = it / 3, n = it - row * 3;
      put(row, n, xyz_s[it]);
      if (n == 0) put(row, 63, 0.0f);
    }
    // One (band, row, coordinate) item per thread and trip, band-major: TILE * 3 = 192 items per band = three whole waves, so a
    // wave's band is uniform and a band whose weight is exactly zero (all ten below progress 0.1, five of ten at the headline's
    // 0.3: nerf.py:136-141) costs it two plane writes, not the fp64 sincos -- sin * 0 and cos * 0 are the zeros written here
    // (up to the sign of a zero, which no contraction sees).  All four waves work (192 threads x 10 bands in a row before).
    static_assert((TILE * 3) % 64 == 0, "a wave's items share one band");
#pragma unroll 1
    for (int i0 = 64 * __builtin_amdgcn_readfirstlane(tid >> 6); i0 < TILE * 3 * 10; i0 += THREADS) {
      const int k = i0 / (TILE * 3);
      const int it = i0 + lane - k * (TILE * 3), row = it / 3, n = it - row * 3;
      const float wk = wk_s[k];
      float sv = 0.0f, cv = 0.0f;
      if (__builtin_amdgcn_readfirstlane(__float_as_uint(wk)) != 0u) {
        sincos_f32_via_f64(xyz_s[it] * ldexpf(PI_F, k), sv, cv);
        sv *= wk;
        cv *= wk;
      }
      put(row, 3 + 20 * n + k, sv);
      put(row, 3 + 20 * n + 10 + k, cv);
    }
  }
  __syncthreads();
  tile_store16<NP, W, TILE, THREADS, UPNERF_X0, 4, false>(Ph, Pl, 0, pow2f(-ecur), a.x0, UPNERF_X0, m0, M, tid);

  // fp32 copy of trunk activation h_lidx for the weight gradients, from the planes (with fp16 storage only the last layer)
  auto store_h32 = [&](int lidx, float un0, float un1) {
#ifndef UPNERF_EXP_NOSTORE
    if (a.h && (!a.h_last_only || lidx == D - 1))
      tile_store16<NP, W, TILE, THREADS, W>(Ph, Pl, 0, un0, un1, a.h + (a.h_last_only ? 0 : (size_t)lidx * M * W), W, m0, M, tid);
#endif
  };
  // fp32 activation stores inside the K loops (PlaneStore): whole tiles of a training pass that stores every layer
#ifdef UPNERF_EXP_NOSTORE
  const bool h_in_loop = false;
#else
  const bool h_in_loop = F16_STORE_IN_LOOP && a.h && !a.h_last_only && (M % TILE) == 0 && NW == 4;
#endif
  // ---- trunk (nerf.py:84-87)
  STAMP(7);  // sample positions + encoding + x0 store
  // exponents of the two row halves of the planes (rows [0, TILE/2) and [TILE/2, TILE)): the lockstep stages keep them equal
  int ehalf[2] = {ecur, ecur};
  int lfirst = 0;  // first trunk layer the lockstep loop below still has to run
  for (int l = lfirst; l < D; ++l) {
    STAMP(0);
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    const int wel = __builtin_amdgcn_readfirstlane(wexp[l]);
    const int wl = __builtin_amdgcn_readfirstlane(loff_s[l]);
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
      mma16_glb<NP, UPNERF_X0>(acc, ap, ecur, P16 + 4 * (size_t)wl, (UPNERF_X0 + W) / 16, n0, 0, lane);
      mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, P16 + 4 * (size_t)wl, (UPNERF_X0 + W) / 16, n0, UPNERF_X0, lane);
    } else if (h_in_loop) {
      // h_{l-1} leaves from the planes this K loop reads, one 1 KiB piece per k-block (PlaneStore)
      PlaneStore<NP, W, TILE, THREADS> ps(Ph, Pl, a.h + ((size_t)(l - 1) * M + m0) * W, pow2f(-ehalf[0]), pow2f(-ehalf[1]), tid);
      static_assert(PlaneStore<NP, W, TILE, THREADS>::ITER == W / 16, "one store piece per k-block");
      mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, P16 + 4 * (size_t)wl, W / 16, n0, 0, lane, ps);
      ps.flush(W / 16 - 1);
    } else {
      mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, P16 + 4 * (size_t)wl, W / 16, n0, 0, lane);
    }
    STAMP(1);
    // h_{l-1} leaves from the planes this K loop has just read, in whole lines, behind the loop's last wait for a weight
    // fragment: the epilogue, two barriers and the plane write pass before the wave waits for a load again (the skip layer,
    // ragged last tiles and fp16-stored operands; everything else went out inside the K loop)
    if (l >= 1 && !(h_in_loop && l != L.skip)) store_h32(l - 1, pow2f(-ehalf[0]), pow2f(-ehalf[1]));
    // the bias row comes from the LDS copy made at kernel start: a global load here would open every epilogue with an L2
    // round trip, and its wait would also wait for the stores just issued (vmcnt retires in order)
    f32x4 bl[TW::NT][4];
    load_cols(bl, bias_s + l * W, n0, hh);
    const unsigned long long bits = acc_fma_relu_pack(acc, pow2f(-(ecur + wel)), bl);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    if (a.hmask) NT_STORE(&((unsigned long long*)a.hmask)[((size_t)l * gridDim.x + blockIdx.x) * THREADS + tid], bits);
    STAMP(2);
    __syncthreads();
    STAMP(3);
    float mx = wg_max<NW>(smax);
    track(mx_s, l, mx, tid);
    if (l + 1 == L.skip) mx = fmaxf(mx, x0max);  // the skip layer feeds x0 rows through the same accumulators
    ecur = scale_exp(mx);
    ehalf[0] = ehalf[1] = ecur;
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    STAMP(4);
    __syncthreads();
    STAMP(5);
    // fp16 storage: the (hi) plane IS the stored tile -- the next write to it is a barrier away.  (f16x3 mode: an option;
    // the weight-gradient operand then is the activation rounded to fp16, the forward chain keeps hi + lo.)
    if (a.h16) tile_copy16<W, TILE, THREADS>(Ph, ecur, a.h16 + (size_t)l * M * W, a.hexp + (size_t)l * ((M + 63) >> 6), m0, M, tid);
    if constexpr (NP == 2) {
      if (a.h16 && a.h_lo8) tile_copy8<W, TILE, THREADS>(Pl, a.h_lo8 + (size_t)l * M * W, m0, M, tid);
    }
    STAMP(6);
  }

  STAMP_FLUSH;
#ifdef UPNERF_STAMPS
  for (int _i = 0; _i < 8; ++_i) _t_acc[_i] = 0;
  _t_prev = __builtin_amdgcn_s_memtime();
#endif
  const int hrow16 = 16 * wave, hm = m0 + hrow16 + (lane & 15);  // the 1- / 3-wide heads: lanes 0..15 of a wave own its 16 rows
  // ---- shared density head (nerf.py:89): softplus(w . h + b)
  {
    const f32x4 pre = head16_mfma<NP, W, W, 1>(Ph, Pl, hrow16, 0, hw_hi, hw_lo, lane);
    const float us = pow2f(-((hrow16 >= TILE / 2 ? ehalf[1] : ehalf[0]) + hwexp));
    if (lane < 16 && hm < M) a.sigma_s[hm] = softplus_f(pre[0] * us + P[L.bsig]);
  }
  // density-only pass (nerf.py:90-91 `sigma_only`): nobody consumes e, so the pass ends here
  if (!a.e && !a.use_rgb && !a.use_cand) {
    store_h32(D - 1, pow2f(-ehalf[0]), pow2f(-ehalf[1]));
    if (!BH_ONLY) STAMP_FLUSH_AT(8);
    track_flush(mx_s, a.amax, tid);
    return;
  }
  // ---- xyz_encoding_final (nerf.py:93), no activation
  {
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    f32x4 be[TW::NT][4];
    load_cols(be, P + L.be, n0, hh);
    if (h_in_loop) {
      PlaneStore<NP, W, TILE, THREADS> ps(Ph, Pl, a.h + ((size_t)(D - 1) * M + m0) * W, pow2f(-ehalf[0]), pow2f(-ehalf[1]), tid);
      mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, P16 + 4 * (size_t)L.we, W / 16, n0, 0, lane, ps);
      ps.flush(W / 16 - 1);
    } else {
      mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, P16 + 4 * (size_t)L.we, W / 16, n0, 0, lane);
      asm volatile("" ::: "memory");  // the bias loads above stay above the stores below
      store_h32(D - 1, pow2f(-ehalf[0]), pow2f(-ehalf[1]));
    }
    acc_fma_bias_h<false>(acc, pow2f(-(ehalf[0] + wexp[8])), pow2f(-(ehalf[1] + wexp[8])), be);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    const float mx = wg_max<NW>(smax);
    track(mx_s, D, mx, tid);
    ecur = scale_exp(fmaxf(mx, sidemax));  // the heads feed per-ray rows through the same accumulators
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    __syncthreads();
  }
  STAMP(5);  // density head + xyz_encoding_final
  // e (no activation) leaves from the planes too: behind the K loop of the first head that reads it, or here
  bool e_pending = a.e != nullptr;
  auto store_e = [&]() {
    if (e_pending) tile_store16<NP, W, TILE, THREADS, W>(Ph, Pl, 0, pow2f(-ecur), a.e, W, m0, M, tid);
    e_pending = false;
  };
  if (!a.use_rgb && !a.use_cand) {
    store_e();
    if (!BH_ONLY) STAMP_FLUSH_AT(8);
    track_flush(mx_s, a.amax, tid);
    return;
  }

  // ---- first layer of the colour head (folded, nerf.py:95+102-109) and of the candidate head (nerf.py:97-98)
  HSTAMP_RESET;
  f32x16 accr[TH::MT][TH::NT], accc[TH::MT][TH::NT];
  int rayrow[TH::MT];
#pragma unroll
  for (int mt = 0; mt < TH::MT; ++mt) {
    int m = m0 + hrow0 + 32 * mt + li;
    m = m < M ? m : M - 1;
    rayrow[mt] = m / S;
  }
  float mr = 0.0f, mc = 0.0f;
  // both heads on (the schedule's middle phase): their first layers contract the same 256 columns of e -- ONE K loop (common16.cuh:
  // mma16_lds_pair; per accumulator bitwise the two loops it replaces)
  const bool pair = F16_JOIN_HEADS && a.use_rgb && a.use_cand && TH::NT == 1;
  if (pair) {
    f32x16 acc2[TH::MT][2];
    acc_zero(acc2);
    mma16_lds_pair<NP, W, W / 16, AH>(acc2, Ph, Pl, hrow0, 0, P16 + 4 * (size_t)L.wr1, (W + UPNERF_AUXK) / 16, P16 + 4 * (size_t)L.wc1,
                                      (W + UPNERF_CK) / 16, hn0, lane);
#pragma unroll
    for (int mt = 0; mt < TH::MT; ++mt) {
      accr[mt][0] = acc2[mt][0];
      accc[mt][0] = acc2[mt][1];
    }
  }
  if (a.use_rgb) {
    f32x4 br[TH::NT][4];
    load_cols(br, P + L.br1, hn0, hh);
    if (!pair) {
      acc_zero(accr);
      mma16_lds<NP, W, W / 16, AH>(accr, Ph, Pl, hrow0, 0, P16 + 4 * (size_t)L.wr1, (W + UPNERF_AUXK) / 16, hn0, 0, lane);
    }
    store_e();
    HSTAMP(0);  // colour: 256-deep K loop + the store of e
    const float* ap[TH::MT];
#pragma unroll
    for (int mt = 0; mt < TH::MT; ++mt) ap[mt] = a.aux + (size_t)rayrow[mt] * UPNERF_AUXK + 8 * hh;
    mma16_glb<NP, UPNERF_AUXK>(accr, ap, ecur, P16 + 4 * (size_t)L.wr1, (W + UPNERF_AUXK) / 16, hn0, W, lane);
    acc_fma_bias<true>(accr, pow2f(-(ecur + wexp[11])), br);
    mr = acc_absmax(accr);
    HSTAMP(1);  // colour: side-input contraction (rows from global) + bias / relu / max
  }
  if (a.use_cand) {
    f32x4 bc[TH::NT][4];
    load_cols(bc, P + L.bc1, hn0, hh);
    if (!pair) {
      acc_zero(accc);
      mma16_lds<NP, W, W / 16, AH>(accc, Ph, Pl, hrow0, 0, P16 + 4 * (size_t)L.wc1, (W + UPNERF_CK) / 16, hn0, 0, lane);
    }
    store_e();
    HSTAMP(2);  // candidate: 256-deep K loop
    const float* ap[TH::MT];
#pragma unroll
    for (int mt = 0; mt < TH::MT; ++mt) ap[mt] = a.c_rows + (size_t)rayrow[mt] * UPNERF_CK + 8 * hh;
    mma16_glb<NP, UPNERF_CK>(accc, ap, ecur, P16 + 4 * (size_t)L.wc1, (W + UPNERF_CK) / 16, hn0, W, lane);
    const unsigned long long bits = acc_fma_relu_pack(accc, pow2f(-(ecur + wexp[9])), bc);
    if (a.hmask) NT_STORE(&((unsigned long long*)a.hmask)[((size_t)D * gridDim.x + blockIdx.x) * THREADS + tid], bits);
    mc = acc_absmax(accc);
    HSTAMP(3);  // candidate: side-input contraction + relu / sign bits / max
  }
  if (lane == 0) {
    smax[wave] = mr;
    smaxb[wave] = mc;
  }
  __syncthreads();
  {
    const float mxr = wg_max<NW>(smax), mxc = wg_max<NW>(smaxb);
    if (a.use_rgb) track(mx_s, D + 3, mxr, tid);
    if (a.use_cand) track(mx_s, D + 1, mxc, tid);
    ecur = scale_exp(fmaxf(mxr, mxc));
  }
  if (a.use_rgb) acc_to_planes<NP, W>(accr, Ph, Pl, hrow0, hn0, 0, ecur, lane);
  if (a.use_cand) acc_to_planes<NP, W>(accc, Ph, Pl, hrow0, hn0, W2, ecur, lane);
  __syncthreads();
  HSTAMP(4);  // barrier, exponent, both plane writes, barrier
  if (a.use_rgb) {
    if (a.r1) tile_store16<NP, W, TILE, THREADS, W2>(Ph, Pl, 0, pow2f(-ecur), a.r1, W2, m0, M, tid);
    // rgb_share_layer.2 + sigmoid (nerf.py:56-61)
    {
      const f32x4 pre = head16_mfma<NP, W, W2, 3>(Ph, Pl, hrow16, 0, hw_hi + W, hw_lo + W, lane);
      const float us = pow2f(-(ecur + hwexp));
      if (lane < 16 && hm < M) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.rgb[(size_t)hm * 3 + c] = sigmoid_f(pre[c] * us + P[L.br2 + c]);
      }
    }
    HSTAMP(5);  // r1 store + the three colour outputs
  }
  if (a.use_cand) {
    if (a.g1) tile_store16<NP, W, TILE, THREADS, W2>(Ph, Pl, W2, pow2f(-ecur), a.g1, W2, m0, M, tid);
    f32x16 acc[TH::MT][TH::NT];
    acc_zero(acc);
    f32x4 b2[TH::NT][4];
    load_cols(b2, P + L.bc2, hn0, hh);
    mma16_lds<NP, W, W2 / 16, AH>(acc, Ph, Pl, hrow0, W2, P16 + 4 * (size_t)L.wc2, W2 / 16, hn0, 0, lane);
    acc_fma_bias<true>(acc, pow2f(-(ecur + wexp[10])), b2);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    ecur = scale_exp(wg_max<NW>(smax));
    acc_to_planes<NP, W>(acc, Ph, Pl, hrow0, hn0, W2, ecur, lane);
    __syncthreads();
    HSTAMP(6);  // g1 store, candidate_encoding.2 (128-deep K loop), epilogue, barrier, plane write, barrier
    if (a.g2) tile_store16<NP, W, TILE, THREADS, W2>(Ph, Pl, W2, pow2f(-ecur), a.g2, W2, m0, M, tid);
    const f32x4 pre = head16_mfma<NP, W, W2, 1>(Ph, Pl, hrow16, W2, hw_hi + W + 3 * W2, hw_lo + W + 3 * W2, lane);
    if (lane < 16 && hm < M) a.sigma_c[hm] = softplus_f(pre[0] * pow2f(-(ecur + hwexp)) + P[L.bcsig]);
    HSTAMP(7);  // g2 store + candidate density
  }
#if !(defined(UPNERF_STAMPS) && defined(UPNERF_STAMPS_HEADS))
  STAMP(6);  // colour / candidate heads
#endif
  if (!BH_ONLY) STAMP_FLUSH_AT(8);
  track_flush(mx_s, a.amax, tid);
}


// ------------------------------------------------------------------------------------------------------------------
// Backward data-gradient chain (autograd of nerf.py:80-124), stage for stage as field.hip:field_bwd_kernel.
template <int NP, int TILE, int NW>
__global__ __launch_bounds__(64 * NW, F16_EU(NP, TILE)) void field16_bwd_kernel(upnerf_layout L, upnerf_field_bwd_args a) {
  constexpr int W = 256, W2 = 128, THREADS = 64 * NW;
  constexpr int AH = NP == 2 ? F16_AHEAD_X3 : F16_AHEAD_F16;
  constexpr int MAXRAYS = 3;            // rays a 64-sample tile can touch when S >= 32
  constexpr int GPR = W2 / 4;           // 16-byte groups per half-width row
  constexpr int EPT = TILE * GPR / THREADS;  // groups per thread in the elementwise stages
  constexpr int PLANE_BYTES = NP * TILE * W * 2;
  constexpr int LDS_BYTES = PLANE_BYTES < TILE * UPNERF_X0 * 4 ? TILE * UPNERF_X0 * 4 : PLANE_BYTES;
  __shared__ __attribute__((aligned(16))) char planes[LDS_BYTES];
  __shared__ float smax[NW], smaxb[NW];
  __shared__ unsigned int mx_s[16];  // running maxima of this workgroup (track / track_flush)
  __shared__ float pre_s[TILE];
  __shared__ __attribute__((aligned(16))) float wfj[MAXRAYS][TILE];  // w_feat[row] on the row's ray slot, else 0
  // per-row scalars of the head stages, computed once per row (not once per 16-byte column group): d pre-activation of
  // the candidate density, its compositing weight, the three d pre-activations of the colour output
  __shared__ float dpc_s[TILE], cwj_s[TILE];
  __shared__ __attribute__((aligned(16))) float dprgb_s[TILE][4];
  __shared__ int loff_s[UPNERF_MAX_D];  // t_w[l] (see the forward kernel: no runtime index into the by-value struct)
  __shared__ int wexp_s[WEXP_SLOTS];    // weight exponents (see the forward kernel: no global load in front of a contraction)
  // per-tile partial sums (a.tile_part, 64-sample tiles only): ray slot of every row; cross-wave reduction scratch
  constexpr int TPW = (TILE == F16_TILE && NW == 4) ? NW : 1;  // (the 8-wave experiment of the 64-sample tile: no partial sums)
  __shared__ int slot_s[TILE];
  __shared__ __attribute__((aligned(16))) float red_s[TPW][32][24];
  const bool tp = TILE == F16_TILE && NW == 4 && a.tile_part != nullptr;
  const int gld = a.gz_rg_ld > 0 ? a.gz_rg_ld : W2;  // row stride of gz_r1 / gz_g1
  f32x4 tp_c = {0.f, 0.f, 0.f, 0.f}, tp_r[3], tp_sg[MAXRAYS], tp_sr[MAXRAYS];
#pragma unroll
  for (int q = 0; q < 3; ++q) tp_r[q] = tp_sg[q] = tp_sr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  char* Ph = planes;
  char* Pl = planes + (NP - 1) * TILE * W * 2;
  using TW = WaveTile16<W, TILE, NW>;
  using TH = WaveTile16<W2, TILE, NW>;
  using TX = WaveTile16<UPNERF_X0, TILE, NW>;
#if F16_WAVES == 8 || defined(F16_SCALAR_WAVE)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: no 64-bit per-lane bases)
#else
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#endif
  const int li = lane & 31, hh = lane >> 5;
  const int S = a.S, M = a.R * a.S, m0 = blockIdx.x * TILE, D = L.D;
  const float* __restrict__ P = a.P;
  const char* __restrict__ PT16 = (const char*)a.PT16;
  const int* wexp = wexp_s;
  const int n0 = TW::n0(wave), row0 = TW::row0(wave);
  const int hn0 = TH::n0(wave), hrow0 = TH::row0(wave);
  const int xn0 = TX::n0(wave), xrow0 = TX::row0(wave);
  const int ray0 = m0 / S;
  const unsigned long long* __restrict__ hm = (const unsigned long long*)a.hmask + (size_t)blockIdx.x * THREADS + tid;
  const size_t hm_stride = (size_t)gridDim.x * THREADS;
#ifdef UPNERF_EXP_ASYMPRIO
  // experiment: ONE of the two waves that share a SIMD (odd hardware wave slot) runs at a raised priority for the whole kernel --
  // does an asymmetric pair settle into complementary phases (one workgroup's K loop beside the other's epilogue)?
  if (__builtin_amdgcn_s_getreg(6148 /* HW_REG_HW_ID, wave_id[3:0] */) & 1) __builtin_amdgcn_s_setprio(UPNERF_EXP_ASYMPRIO);
#endif

  if (tid == 64) {
#pragma unroll
    for (int l = 0; l < UPNERF_MAX_D; ++l) loff_s[l] = L.t_w[l];
  }
  if (tid >= 64 && tid < 80) mx_s[tid - 64] = 0u;
  // ---- every load of the prologue AND the row loads of the head stages, requested back to back and unconditionally (clamped
  // rows, pointer selects for absent inputs, values masked afterwards).  Behind `if (m < M)`, `if (a.use_cand)`, `ptr ? load : 0`
  // hipcc branches around each load and waits for it before the next: this kernel opened with five to six HBM round trips in a
  // row for the per-row scalars, and its candidate stage carried a `s_waitcnt vmcnt(0)` in the middle of its sixteen row loads
  // (the g_G_c rows behind a null check) -- 21.8k cycles per tile for a stage that moves 1.5 KB per sample (round 5 stamps).
  static_assert(TILE <= THREADS && WEXP_SLOTS <= THREADS, "one sample row per thread");
  const int wexp_v = a.wexp[tid & (WEXP_SLOTS - 1)];
  const int ms = m0 + (tid & (TILE - 1)), msc = ms < M ? ms : M - 1;
  const bool has_gc = a.use_cand && a.g_G_c != nullptr;
  const float dss_v = a.d_sigma_s[msc], ss_v = a.sigma_s[msc];
  const float wf_v = (a.g_E_s ? a.w_feat_s : P)[a.g_E_s ? msc : 0];
  const float dsc_v = (a.use_cand ? a.d_sigma_c : P)[a.use_cand ? msc : 0], sc_v = (a.use_cand ? a.sigma_c : P)[a.use_cand ? msc : 0];
  const float cw_v = (has_gc ? a.w_cj : P)[has_gc ? msc : 0];
  float y_v[3], dy_v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    y_v[c] = (a.use_rgb ? a.rgb : P)[a.use_rgb ? (size_t)msc * 3 + c : 0];
    dy_v[c] = (a.use_rgb ? a.d_rgb : P)[a.use_rgb ? (size_t)msc * 3 + c : 0];
  }
  // column group and first row of this thread in the elementwise head stages (row advances by THREADS / GPR per step)
  const int eg = tid % GPR, er0 = tid / GPR;
  constexpr int ERS = THREADS / GPR;
  // the r1 / g2 rows of the head stages and the per-ray candidate-feature gradient rows: they travel under the prologue
  f32x4 rv[EPT], gv[EPT], gg[EPT];
  {
    const float* __restrict__ r1p = a.use_rgb ? a.r1 : P;
    const float* __restrict__ g2p = a.use_cand ? a.g2 : P;
    const float* __restrict__ ggp = has_gc ? a.g_G_c : P;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      int m = m0 + er0 + ERS * q;
      m = m < M ? m : M - 1;
      rv[q] = NT_LOAD((const f32x4*)&r1p[a.use_rgb ? (size_t)m * W2 + 4 * eg : 0]);
      gv[q] = NT_LOAD((const f32x4*)&g2p[a.use_cand ? (size_t)m * W2 + 4 * eg : 0]);
      gg[q] = *(const f32x4*)&ggp[has_gc ? (size_t)(m / S) * W2 + 4 * eg : 0];
    }
  }
  if (tid < WEXP_SLOTS) wexp_s[tid] = wexp_v;
  // softplus'(x) = 1 - exp(-softplus(x)); per-row feature weight on its ray slot; per-row scalars of the head stages
  if (tid < TILE) {
    const int m = ms;
    float v = 0.0f, wf = 0.0f, dpc = 0.0f, cwj = 0.0f;
    f32x4 dprgb = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    if (m < M) {
      v = dss_v * (1.0f - expf(-ss_v));
      a.dpre_sig_s[m] = v;
      if (a.g_E_s) wf = wf_v;
      j = m / S - ray0;
      if (a.use_cand) {
        dpc = dsc_v * (1.0f - expf(-sc_v));
        a.dpre_sig_c[m] = dpc;
        if (has_gc) cwj = cw_v;
      }
      if (a.use_rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dprgb[c] = dy_v[c] * (y_v[c] * (1.0f - y_v[c]));
        *(f32x4*)&a.dpre_rgb[(size_t)m * 4] = dprgb;
      }
    }
    pre_s[tid] = v;
    dpc_s[tid] = dpc;
    cwj_s[tid] = cwj;
    *(f32x4*)&dprgb_s[tid][0] = dprgb;
    slot_s[tid] = j;
#pragma unroll
    for (int q = 0; q < MAXRAYS; ++q) wfj[q][tid] = (q == j) ? wf : 0.0f;
  }
  __syncthreads();

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
        // (the g2 / g_G_c rows were requested in the prologue; without a feature gradient cw is zero and gg finite: P's first words)
        const f32x4 wv = *(const f32x4*)&P[L.wcsig + 4 * eg];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
          const int row = er0 + ERS * q, m = m0 + row;
          const float dp = dpc_s[row], cw = cwj_s[row];
          f32x4 out;
#pragma unroll
          for (int c = 0; c < 4; ++c) out[c] = (m < M && gv[q][c] > 0.f) ? wv[c] * dp + cw * gg[q][c] : 0.f;
          if (m < M) ACT_STORE((f32x4*)&a.gz_g2[(size_t)m * W2 + 4 * eg], out);
          tp_c += gv[q] * dp;  // d w_csig: dp is zero for rows past M (the clamped row is then ignored)
          vals[q] = out;
          lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(out[0]), fabsf(out[1])), fmaxf(fabsf(out[2]), fabsf(out[3]))));
        }
      }
      lmax = wave_max(lmax);
      if (lane == 0) smax[wave] = lmax;
      BHSTAMP(0);  // candidate: row loads (g2, g_G_c), d g2, store, column sums
      __syncthreads();
      const float mx = wg_max<NW>(smax);
      track(mx_s, D + 2, mx, tid);
      const int eg2 = scale_exp(mx);
#pragma unroll
      for (int q = 0; q < EPT; ++q) put_quad<NP, W>(Ph, Pl, er0 + ERS * q, W2 + 4 * eg, vals[q], eg2);
      __syncthreads();
      BHSTAMP(1);  // barrier, plane write, barrier
      const unsigned long long cbits = NT_LOAD(&hm[(size_t)D * hm_stride]);  // arrives under the contraction below
      mma16_lds<NP, W, W2 / 16, AH>(accg, Ph, Pl, hrow0, W2, PT16 + 4 * (size_t)L.t_wc2, W2 / 16, hn0, 0, lane);
      acc_scale(accg, pow2f(-(eg2 + wexp[10])));
      acc_apply_mask(accg, cbits);
      mg1 = acc_absmax(accg);
      BHSTAMP(2);  // candidate_encoding.2^T: 128-deep contraction, scale, mask, max
    }
    f32x4 valr[EPT];
    float mr1 = 0.0f;
    if (a.use_rgb) {
      // d r1 = W_r2^T (d rgb * rgb (1-rgb))   (rgb_share_layer.2 + sigmoid); loads first, as above
      f32x4 wr[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) wr[c] = *(const f32x4*)&P[L.wr2 + c * W2 + 4 * eg];
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
        if (m < M) ACT_STORE((f32x4*)&a.gz_r1[(size_t)m * gld + 4 * eg], out);
        if (tp) {
          const int sl = slot_s[row];
#pragma unroll
          for (int c = 0; c < 3; ++c) tp_r[c] += rv[q] * dp[c];
#pragma unroll
          for (int c = 0; c < MAXRAYS; ++c) tp_sr[c] += out * (sl == c ? 1.0f : 0.0f);
        }
        valr[q] = out;
        mr1 = fmaxf(mr1, fmaxf(fmaxf(fabsf(out[0]), fabsf(out[1])), fmaxf(fabsf(out[2]), fabsf(out[3]))));
      }
      mr1 = wave_max(mr1);
      BHSTAMP(3);  // colour: d r1 (rank 3), store, column / ray sums
    }
    if (lane == 0) {
      smax[wave] = mg1;
      smaxb[wave] = mr1;
    }
    __syncthreads();  // also: every wave is done reading the gz_g2 planes
    {
      const float mxg = wg_max<NW>(smax), mxr = wg_max<NW>(smaxb);
      if (a.use_cand) track(mx_s, D + 1, mxg, tid);
      if (a.use_rgb) track(mx_s, D + 3, mxr, tid);
      if (a.use_cand && a.use_rgb) track(mx_s, D + 4, fmaxf(mxg, mxr), tid);  // of [gz_r1 | gz_g1] as one tensor (gz_rg_ld)
      erg = scale_exp(fmaxf(mxg, mxr));
    }
    if (a.use_cand) acc_to_planes<NP, W>(accg, Ph, Pl, hrow0, hn0, W2, erg, lane);
    if (a.use_rgb) {
#pragma unroll
      for (int q = 0; q < EPT; ++q) put_quad<NP, W>(Ph, Pl, er0 + ERS * q, 4 * eg, valr[q], erg);
    }
    __syncthreads();
    BHSTAMP(4);  // barrier, exponent, two plane writes, barrier
    if (a.use_cand) {
      if (tp) tile_store16_sum<NP, W, TILE, THREADS, W2, MAXRAYS>(Ph, Pl, W2, pow2f(-erg), a.gz_g1, gld, m0, M, tid, slot_s, tp_sg);
      else tile_store16<NP, W, TILE, THREADS, W2>(Ph, Pl, W2, pow2f(-erg), a.gz_g1, gld, m0, M, tid);
    }
    BHSTAMP(5);  // gz_g1 store (+ its per-ray sums)
    if (tp) {
      // A thread holds four columns (eg) of 2 * EPT rows' worth of sums; the two halves of a wave fold first, then the
      // waves through LDS, in a fixed order (bitwise reproducible).  Two rounds of at most 24 floats per column group.
      float* __restrict__ part = a.tile_part + (size_t)blockIdx.x * UPNERF_TILE_PART_STRIDE;
      auto fold = [&](f32x4 v) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += __shfl_xor(v[c], 32);
        return v;
      };
      const int g = tid & 31, k = tid >> 5;  // column group, vector index of the read-out
      auto readout = [&](int kk) {
        f32x4 sacc = *(const f32x4*)&red_s[0][g][4 * kk];
#pragma unroll
        for (int w = 1; w < TPW; ++w) sacc += *(const f32x4*)&red_s[w][g][4 * kk];
        return sacc;
      };
      tp_c = fold(tp_c);
#pragma unroll
      for (int c = 0; c < 3; ++c) tp_r[c] = fold(tp_r[c]);
      if (lane < 32) {
        *(f32x4*)&red_s[wave % TPW][lane][0] = tp_c;
#pragma unroll
        for (int c = 0; c < 3; ++c) *(f32x4*)&red_s[wave % TPW][lane][4 + 4 * c] = tp_r[c];
      }
      __syncthreads();
      if (k < 4) *(f32x4*)&part[(k == 0 ? 0 : W2 * k) + 4 * g] = readout(k);
#if F16_BH_SCALAR_SUMS_WAVE
      if (wave == NW - 1) {  // sums of the per-row scalars (d b_csig, d b_r2): one row per lane, a butterfly over the wave (fixed order)
        static_assert(TILE == 64 || TPW == 1, "one row per lane");
        float v[4] = {dpc_s[lane], dprgb_s[lane][0], dprgb_s[lane][1], dprgb_s[lane][2]};
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += __shfl_xor(v[c], d);
        if (lane < 4) {
          part[4 * W2 + lane] = lane == 0 ? v[0] : (lane == 1 ? v[1] : (lane == 2 ? v[2] : v[3]));
          part[4 * W2 + 4 + lane] = 0.0f;  // pad
        }
      }
#else
      if (tid >= THREADS - 4) {  // sums of the per-row scalars: d b_csig, d b_r2 (four threads walking the 64 rows)
        const int c = tid - (THREADS - 4);
        float sacc = 0.0f;
        for (int r = 0; r < TILE; ++r) sacc += c == 0 ? dpc_s[r] : dprgb_s[r][c - 1];
        part[4 * W2 + c] = sacc;
        part[4 * W2 + 4 + c] = 0.0f;  // pad
      }
#endif
      __syncthreads();
#pragma unroll
      for (int c = 0; c < MAXRAYS; ++c) {
        tp_sr[c] = fold(tp_sr[c]);
        tp_sg[c] = fold(tp_sg[c]);
      }
      if (lane < 32) {
#pragma unroll
        for (int c = 0; c < MAXRAYS; ++c) {
          *(f32x4*)&red_s[wave % TPW][lane][4 * c] = tp_sg[c];
          *(f32x4*)&red_s[wave % TPW][lane][12 + 4 * c] = tp_sr[c];
        }
      }
      __syncthreads();
      if (k < 6) *(f32x4*)&part[4 * W2 + 8 + W2 * k + 4 * g] = readout(k);
    }
    BHSTAMP(6);  // partial sums: folds through LDS, three barriers
  }
#if BH_ONLY
  STAMP_FLUSH_AT(8);
  for (int _i = 0; _i < 8; ++_i) _t_acc[_i] = 0;
#endif

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
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    const float mx = wg_max<NW>(smax);
    track(mx_s, D, mx, tid);
    ecur = scale_exp(mx);
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    __syncthreads();
  }
  STAMP(1);  // d e
  // ---- d h_{D-1} = gz_e . W_e + w_sig * dpre_s, masked by relu (sign bits from the forward, in this lane's layout)
  {
    const unsigned long long bits = NT_LOAD(&hm[(size_t)(D - 1) * hm_stride]);
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    f32x4 ws[TW::NT][4];
    load_cols(ws, P + L.wsig, n0, hh);
    mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, PT16 + 4 * (size_t)L.t_we, W / 16, n0, 0, lane);
    // gz_e leaves from the planes this K loop has just read, in whole lines (see the forward kernel); every load requested
    // so far (ws, the sign bits) stays in front of the stores
    asm volatile("" ::: "memory");
    tile_store16<NP, W, TILE, THREADS, W>(Ph, Pl, 0, pow2f(-ecur), a.gz_e, W, m0, M, tid);
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
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    const float mx = wg_max<NW>(smax);
    track(mx_s, D - 1, mx, tid);
    ecur = scale_exp(mx);
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    __syncthreads();
    if (a.gz16) tile_copy16<W, TILE, THREADS>(Ph, ecur, a.gz16 + (size_t)(D - 1) * M * W, a.gzexp + (size_t)(D - 1) * ((M + 63) >> 6), m0, M, tid);
    if constexpr (NP == 2) {
      if (a.gz16 && a.gz_lo8) tile_copy8<W, TILE, THREADS>(Pl, a.gz_lo8 + (size_t)(D - 1) * M * W, m0, M, tid);
    }
  }
  STAMP(2);  // d h_{D-1}
  // ---- trunk, last layer to first
  f32x16 accx[TX::MT][TX::NT];
  acc_zero(accx);
  int ehalf[2] = {ecur, ecur};
  auto store_gz32 = [&](int lidx, float un0, float un1) {
#ifndef UPNERF_EXP_NOSTORE
    if (a.gz_h) tile_store16<NP, W, TILE, THREADS, W>(Ph, Pl, 0, un0, un1, a.gz_h + (size_t)lidx * M * W, W, m0, M, tid);
#endif
  };
  if (D == 1) store_gz32(0, pow2f(-ecur), pow2f(-ecur));
  {
  for (int l = D - 1; l >= 1; --l) {
    const unsigned long long bits = NT_LOAD(&hm[(size_t)(l - 1) * hm_stride]);  // arrives under the contraction below
    if (a.need_dxyz && l == L.skip) {
      mma16_lds<NP, W, W / 16, AH>(accx, Ph, Pl, xrow0, 0, PT16 + 4 * (size_t)L.t_skipx, W / 16, xn0, 0, lane);
      acc_scale(accx, pow2f(-(ecur + wexp[l])));  // natural units: the layer-0 term arrives at another exponent
    }
    f32x16 acc[TW::MT][TW::NT];
    acc_zero(acc);
    const int wel = __builtin_amdgcn_readfirstlane(wexp[l]);  // wave-uniform; asked for before the contraction
    mma16_lds<NP, W, W / 16, AH>(acc, Ph, Pl, row0, 0, PT16 + 4 * (size_t)__builtin_amdgcn_readfirstlane(loff_s[l]), W / 16, n0, 0, lane);
    asm volatile("" ::: "memory");  // the sign-bit load stays in front of the stores
    store_gz32(l, pow2f(-ecur), pow2f(-ecur));  // gz_l: the planes this K loop has just read
    acc_scale(acc, pow2f(-(ecur + wel)));
    acc_apply_mask(acc, bits);
    const float wm = acc_absmax(acc);
    if (lane == 0) smax[wave] = wm;
    __syncthreads();
    const float mx = wg_max<NW>(smax);
    track(mx_s, l - 1, mx, tid);
    ecur = scale_exp(mx);
    ehalf[0] = ehalf[1] = ecur;
    acc_to_planes<NP, W>(acc, Ph, Pl, row0, n0, 0, ecur, lane);
    __syncthreads();
    if (a.gz16) tile_copy16<W, TILE, THREADS>(Ph, ecur, a.gz16 + (size_t)(l - 1) * M * W, a.gzexp + (size_t)(l - 1) * ((M + 63) >> 6), m0, M, tid);
    if constexpr (NP == 2) {
      if (a.gz16 && a.gz_lo8) tile_copy8<W, TILE, THREADS>(Pl, a.gz_lo8 + (size_t)(l - 1) * M * W, m0, M, tid);
    }
  }
  }
  STAMP(3);  // D-1 trunk layers
  if (D > 1) store_gz32(0, pow2f(-ecur), pow2f(-ecur));  // gz_0: still in the planes
  if (!a.need_dxyz) {
    BSTAMP_FLUSH_AT(8);
    track_flush(mx_s, a.gmax, tid);
    return;
  }
  // ---- d x0 (first layer + skip) -> d xyz through the encoding
  {
    f32x16 acc0[TX::MT][TX::NT];
    acc_zero(acc0);
    mma16_lds<NP, W, W / 16, AH>(acc0, Ph, Pl, xrow0, 0, PT16 + 4 * (size_t)L.t_w[0], W / 16, xn0, 0, lane);
    const float un = pow2f(-((xrow0 >= TILE / 2 ? ehalf[1] : ehalf[0]) + wexp[0]));
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[0][0][r] = fmaf(acc0[0][0][r], un, accx[0][0][r]);
  }
  static_assert(TX::MT == 1 && TX::NT == 1, "d x0 tiling");
  __syncthreads();
  float* Gs = (float*)planes;  // fp32 [TILE][64] scratch over the (now dead) planes
  acc_to_lds_t(accx, Gs, UPNERF_X0, xrow0, xn0, lane);
  __syncthreads();
  for (int it = tid; it < TILE * 3; it += THREADS) {
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
  BSTAMP_FLUSH_AT(8);
  track_flush(mx_s, a.gmax, tid);
}

// Rows per workgroup: 64 (`want` = 0 or 64), or 128 (needs S >= 64: a 128-row tile then touches at most three rays, and the
// backward kernel keeps three ray slots).  The forward and the backward pass of one field evaluation must use the same
// value: the ReLU sign bits (hmask) are laid out per workgroup tile.  The 128-sample pipelined kernels are correct but, as
// measured in round 3, not yet faster than two 64-sample workgroups per CU: opt-in.
int tile_rows16(int want, int S) {
  (void)S;
  if (want == 0 || want == F16_TILE) return F16_TILE;
  return -1;  // (128: round 3's pipelined kernels left the library -- tools/repro/pipe16; 256: csrc/field16rr.hip, dispatched earlier)
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

int upnerf_rr16_fwd_launch(const upnerf_layout* L, const upnerf_field_fwd_args* a, void* stream);  // csrc/field16rr.hip
int upnerf_rr16_bwd_launch(const upnerf_layout* L, const upnerf_field_bwd_args* a, void* stream);

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
  if (a->h16 && !a->hexp) return UPNERF_EINVAL;
  if (a->tile_rows == 256) {  // register-resident kernels (csrc/field16rr.hip): fp16 mode, weights staged in LDS once per 256 samples
    if (a->planes != 1) return UPNERF_EUNSUP;
    if (a->S < 32) return UPNERF_EUNSUP;  // at most 9 rays per 256-sample tile
    if (!a->wnorm) return UPNERF_EINVAL;
    if (a->rows_capacity != 0 && a->rows_capacity < (M + 255) / 256 * 256) return UPNERF_EINVAL;  // whole tiles are written
    if ((a->e16 && !a->eexp) || (a->g2_16 && !a->g2exp) || (a->r1_16 && !a->r1exp) || (a->g1_16 && !a->g1exp)) return UPNERF_EINVAL;
    if (a->h16 && !a->hmask) return UPNERF_EINVAL;
    if (a->use_cand && !a->g2 && !a->g2_16) return UPNERF_EINVAL;
    if (a->h16 && ((a->use_cand && !a->g1 && !a->g1_16) || (a->use_rgb && !a->r1 && !a->r1_16))) return UPNERF_EINVAL;
    return upnerf_rr16_fwd_launch(L, a, stream);
  }
  if (a->wnorm || a->e16 || a->g2_16 || a->r1_16 || a->g1_16) return UPNERF_EUNSUP;  // only the register-resident kernels read the row norms / write e as fragments
  const int tile = tile_rows16(a->tile_rows, a->S);
  if (tile < 0) return UPNERF_EINVAL;
  const int grid = (int)((M + tile - 1) / tile);
  const hipStream_t st = (hipStream_t)stream;
  if (a->planes == 1)
    hipLaunchKernelGGL((field16_fwd_kernel<1, F16_TILE, F16_WAVES>), dim3(grid), dim3(64 * F16_WAVES), 0, st, *L, *a);
  else
    hipLaunchKernelGGL((field16_fwd_kernel<2, F16_TILE, F16_WAVES>), dim3(grid), dim3(64 * F16_WAVES), 0, st, *L, *a);
  return (int)hipGetLastError();
}

extern "C" int upnerf_field_bwd_f16x3(const upnerf_layout* L, const upnerf_field_bwd_args* a, void* stream) {
  int rc = check_layout16(L);
  if (rc) return rc;
  if (!a || a->R <= 0 || a->S <= 0 || !a->P || !a->PT16 || !a->wexp || !a->d_sigma_s || !a->sigma_s || (!a->gz_h && !a->gz16) ||
      (!a->gz_e && a->tile_rows != 256) || !a->dpre_sig_s || !a->hmask)
    return UPNERF_EINVAL;
  if (a->S < 32) return UPNERF_EUNSUP;  // at most 3 rays per 64-sample tile
  const bool rg16 = a->tile_rows == 256 && a->gz_rg16 && a->gzrgexp && a->use_cand && a->use_rgb;  // fp16 fragments instead of gz_r1 / gz_g1
  if (a->gz_rg16 && !rg16) return UPNERF_EINVAL;
  if (a->use_cand && (!a->d_sigma_c || !a->sigma_c || (!a->g2 && a->tile_rows != 256) || (!a->gz_g1 && !rg16) || (!a->gz_g2 && !(a->tile_rows == 256 && a->gz_g2_16 && a->gzg2exp)) || !a->dpre_sig_c))
    return UPNERF_EINVAL;
  if (a->use_cand && a->g_G_c && !a->w_cj) return UPNERF_EINVAL;
  if (a->use_rgb && (!a->d_rgb || !a->rgb || (!a->r1 && a->tile_rows != 256) || (!a->gz_r1 && !rg16) || !a->dpre_rgb)) return UPNERF_EINVAL;
  if (a->g_E_s && !a->w_feat_s) return UPNERF_EINVAL;
  if (a->need_dxyz && (!a->dxyz || !a->x0)) return UPNERF_EINVAL;
  if (a->planes != 0 && a->planes != 1 && a->planes != 2) return UPNERF_EINVAL;
  if (a->gz16 && !a->gzexp) return UPNERF_EINVAL;
  const long long M = (long long)a->R * a->S;
  if (M > 0x7fffffffLL) return UPNERF_EINVAL;
  if (a->tile_rows == 256) {  // register-resident kernels (csrc/field16rr.hip)
    if (a->planes != 1) return UPNERF_EUNSUP;
    if (!a->wnorm || !a->gz16) return UPNERF_EINVAL;
    if (a->rows_capacity != 0 && a->rows_capacity < (M + 255) / 256 * 256) return UPNERF_EINVAL;  // whole tiles are written
    return upnerf_rr16_bwd_launch(L, a, stream);
  }
  if (a->wnorm) return UPNERF_EUNSUP;
  const int tile = tile_rows16(a->tile_rows, a->S);
  if (tile < 0) return UPNERF_EINVAL;
  const int grid = (int)((M + tile - 1) / tile);
  const hipStream_t st = (hipStream_t)stream;
  if (a->planes == 1)
    hipLaunchKernelGGL((field16_bwd_kernel<1, F16_TILE, F16_WAVES>), dim3(grid), dim3(64 * F16_WAVES), 0, st, *L, *a);
  else
    hipLaunchKernelGGL((field16_bwd_kernel<2, F16_TILE, F16_WAVES>), dim3(grid), dim3(64 * F16_WAVES), 0, st, *L, *a);
  return (int)hipGetLastError();
}
