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
#define RR_AHEAD 2                 // slabs travel in pairs: the next pair is in flight while a pair is contracted
#define RR_MAXKB 21                // k-blocks of the widest matrix row (colour head: 256 + 80)
#define RR_SLOT (RR_MAXKB * 1024)  // bytes per ring slot
#define RR_STG 4096                // per-wave transposer: 32 rows x 32 fp32
#ifndef RR_PF
#define RR_PF 3                    // weight fragments requested from LDS this many k-blocks ahead
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
  static constexpr int BSW0 = ROW0 + MAXR * ROWF * 4;         // per-wave copy of the running layer's bias, times 2^eo (1 KiB each)
  static constexpr int INT0 = BSW0 + NW * 1024;               // small integer tables
  static constexpr int SLB0 = INT0 + 96 * 4;                  // slab list of the pass (<= 96 entries of 8 bytes)
  static constexpr int LDS = SLB0 + 96 * 8;
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

// One LDS-DMA instruction (1 KiB: lane l's 16 bytes from `g` land at LDS byte address lds + 16 l), issued as INLINE ASM on purpose
// (round 6).  Behind __builtin_amdgcn_global_load_lds hipcc's wait-count pass knows that a vector-memory operation is writing LDS and,
// having no alias information, puts `s_waitcnt vmcnt(0)` in front of every later LDS read it cannot tell apart from the destination --
// in these kernels the read of the slab table inside every request (twice per slab pair) and the transposer's reads: each one a wait
// for ALL of the wave's outstanding stores and requests, which is what the counted waits below exist to avoid (found in the ISA of
// round 5's build: two `vmcnt(0)` per trip of the slab loop).  The kernels synchronise their DMA by hand -- counted vmcnt + barrier
// before any read of a slot -- so nothing is lost with the compiler's view of it.
// ASM: per kernel.  Measured (alternating runs on one box, 8192 rays, `profiles/r06_ab_rr_dma.txt`): the BACKWARD kernel gains 4.4 %
// (2.304 -> 2.202 ms; slab loop 300k -> 277k cycles per wave), the FORWARD kernel loses 1.7 % (1.989 -> 2.023 ms: its contraction gets
// faster, 1081 -> 892 cycles per slab, and its barrier waits longer) -- so the forward kernel keeps the builtin.
// RR_EXP_DMA_BUILTIN / RR_EXP_DMA_ASM force one form in both kernels (A/B builds).
template <bool ASM>
__device__ __forceinline__ void lds_dma16(const void* g, const void* lds) {
#if defined(RR_EXP_DMA_BUILTIN)
  constexpr bool use_asm = false;
#elif defined(RR_EXP_DMA_ASM)
  constexpr bool use_asm = true;
#else
  constexpr bool use_asm = ASM;
#endif
  if constexpr (!use_asm) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  } else {
    const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) const char*)lds));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(d), "v"(g) : "memory", "m0");
  }
}

// s_waitcnt vmcnt(n) for a wave-uniform RUN-TIME n (the instruction takes an immediate): a computed jump into a table of 32
// {s_waitcnt vmcnt(k); s_branch end} pairs.  n above 31 waits for 31 -- a stronger wait, always safe.  expcnt / lgkmcnt untouched
// (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]).  A switch statement here compiled into a chain of
// ~10 taken branches per call (round 4 stamps: ~400 cycles per slab).
__device__ __forceinline__ void wait_vmcnt(int n) {
#ifdef RR_SAFE_WAIT  // diagnostic: drain everything (A/B against the counted waits)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  return;
#endif
  n = __builtin_amdgcn_readfirstlane(n > 31 ? 31 : (n < 0 ? 0 : n));
#define RR_WE(N) "s_waitcnt vmcnt(" #N ")\n\ts_branch 1f\n\t"
  asm volatile(
      "s_getpc_b64 s[92:93]\n\t"      // address of the next instruction
      "s_lshl_b32 s94, %0, 3\n\t"     // 8 bytes per table entry
      "s_add_u32 s92, s92, s94\n\t"
      "s_addc_u32 s93, s93, 0\n\t"
      "s_add_u32 s92, s92, 24\n\t"    // the six 4-byte instructions between the s_getpc and the table
      "s_addc_u32 s93, s93, 0\n\t"
      "s_setpc_b64 s[92:93]\n\t" RR_WE(0) RR_WE(1) RR_WE(2) RR_WE(3) RR_WE(4) RR_WE(5) RR_WE(6) RR_WE(7) RR_WE(8) RR_WE(9) RR_WE(10)
          RR_WE(11) RR_WE(12) RR_WE(13) RR_WE(14) RR_WE(15) RR_WE(16) RR_WE(17) RR_WE(18) RR_WE(19) RR_WE(20) RR_WE(21) RR_WE(22)
              RR_WE(23) RR_WE(24) RR_WE(25) RR_WE(26) RR_WE(27) RR_WE(28) RR_WE(29) RR_WE(30) RR_WE(31) "1:\n\t"
      :
      : "s"(n)
      : "memory", "scc", "s92", "s93", "s94");
#undef RR_WE
}

// ---- slab sequence -------------------------------------------------------------------------------------------------------
// A pass is a list of STAGES (one weight matrix each) of `tiles` 32-feature slabs; every stage has a multiple of RR_NSLOT
// tiles, so slab i of a stage always sits in ring slot i % RR_NSLOT (compile-time in the unrolled tile loops).  The tables
// live in LDS (sq_*): byte offset of the matrix in the fragment buffer, k-blocks per row, tiles.
#define RR_MAXSTAGE 16

// Diagnostic build only (-DUPNERF_STAMPS): shader-clock sums per phase of the slab loop (tools/stamps_rr16.py):
// 0 wait for the slab's DMA, 1 barrier, 2 DMA issue, 3 everything between two slab tops (MFMAs, epilogues, stores)
#ifdef UPNERF_STAMPS
__device__ unsigned long long upnerf_stamp_acc_rr[24];  // [0..7] forward, [8..15] backward, [16..23] forward epilogue detail
#define RR_STAMP(i)                                              \
  do {                                                           \
    const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);                          \
    st_acc[i] += _t - st_prev;                                   \
    st_prev = _t;                                                \
  } while (0)
#define RR_STAMP_RG(rg, i)                                       \
  do {                                                           \
    const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);                          \
    (rg).st_acc[i] += _t - (rg).st_prev;                         \
    (rg).st_prev = _t;                                           \
  } while (0)
#else
#define RR_STAMP(i)
#define RR_STAMP_RG(rg, i)
#endif

// The slab list of a pass, built once per workgroup from its stage tables (offsets in bytes, k-blocks, tiles, wrap: a 2-tile stage
// issued twice keeps the ring slots aligned): thread i writes entry i.  Returns the number of slabs.
__device__ __forceinline__ int build_slab_table(int2* slab_s, const int* st_off, const int* st_kb, const int* st_tiles, const int* st_wrap,
                                                int nstage, int tid) {
  int total = 0, mine = -1, mt = 0;
  for (int st = 0; st < nstage; ++st) {
    const int t = st_tiles[st];
    if (mine < 0 && tid < total + t) {
      mine = st;
      mt = tid - total;
    }
    total += t;
  }
  if (mine >= 0) {
    const int kb = st_kb[mine], wr = st_wrap ? st_wrap[mine] : st_tiles[mine];
    const int gt = mt >= wr ? mt - wr : mt;
    slab_s[tid] = int2{st_off[mine] + gt * kb * 2048, kb};
  }
  return total;
}

// ---- per-wave ring state -------------------------------------------------------------------------------------------------
// vmc: vector-memory instructions this wave has issued in the slab loop so far (DMA + counted stores); mark of slot s: its
// value right after the DMA into slot s was issued.  vmc - mark instructions are younger than that DMA.  The tile loops are
// RUN-TIME loops (the unrolled form of round 4's first build was 60 KB of code per trunk layer against a 64 KB instruction
// cache shared by two CUs: every phase of it ran at a third of its speed), so the slot index is a run-time scalar: four
// scalars and selects, never an indexed array (hipcc would put it in scratch).
template <int NW, bool ASM_DMA = false>
struct Ring {
  const char* src;     // fragment buffer (P16 or PT16)
  const int2* slab_s;  // LDS table of the pass's slabs in consumption order: {byte offset of the tile's first k-block, k-blocks}
  int nslab, nreq;     // slabs of the pass; next slab to request
  int wave, lane;
  int vmc;
  int mk0, mk1, mk2, mk3;
#ifdef UPNERF_STAMPS
  unsigned long long st_acc[10], st_prev;
#endif

  __device__ __forceinline__ int mark(int slot) const { return slot == 0 ? mk0 : (slot == 1 ? mk1 : (slot == 2 ? mk2 : mk3)); }

  // request the next slab of the sequence into ring slot `slot` (`ring`: the kernel's LDS array, passed in so that the compiler
  // keeps the address space).  Everything about the slab comes from ONE table entry: no stage logic in the tile loops (round 4:
  // with the stage bookkeeping inlined at every request site a tile iteration was 750 instructions, and at two waves per
  // SIMD a wave issues an instruction about every four cycles -- the loop was bound by its instruction COUNT).
  __device__ __forceinline__ void issue(char* ring, int slot) {
    // Only the LAGGING half of the waves (w >= NW/2) requests slabs: a 1 KiB request blocks its wave until the CU's one
    // vector-memory path takes it (~60 cycles each when eight waves issue together), and a blocked wave issues no MFMA.  The
    // lagging waves start a tile with the previous tile's epilogue anyway; their SIMD partners go straight to the matrix
    // work.  A leading wave waits for nothing of its own: the barrier behind the lagging waves' counted wait covers it.
#ifndef RR_DMA_ALL
#define RR_DMA_ALL 0  // 1 (experiment): all eight waves request (two pieces each per 16 KB slab) instead of the lagging four (four each)
#endif
    if (nreq < nslab && (RR_DMA_ALL || wave >= NW / 2)) {
      constexpr int ND = RR_DMA_ALL ? NW : NW / 2;
      const int2 e = slab_s[nreq];
      const int off = __builtin_amdgcn_readfirstlane(e.x), kb = __builtin_amdgcn_readfirstlane(e.y);
      const int per = (kb + ND - 1) / ND;
      const char* g = src + (size_t)off + lane * 16;
      char* d = ring + slot * RR_SLOT;
      asm volatile("" ::: "memory");
#ifndef RR_EXP_NODMA
#pragma unroll 1
      for (int q = 0; q < per; ++q) {
        int ch = (RR_DMA_ALL ? wave : wave - ND) + ND * q;
        ch = ch < kb ? ch : kb - 1;  // surplus waves repeat the last chunk (same bytes to the same place)
#ifdef RR_EXP_PLAINLOAD  // timing experiment (wrong results): the same bytes by ordinary loads whose results are dropped
        const f32x4 v = *(const volatile f32x4*)(g + (size_t)ch * 2048);
        asm volatile("" ::"v"(v));
#else
        lds_dma16<ASM_DMA>(g + (size_t)ch * 2048, d + ch * 1024);
#endif
      }
      vmc += per;
#else
      (void)g;
      (void)d;
#endif
      asm volatile("" ::: "memory");
    }
    if (nreq < nslab) ++nreq;
    mk0 = slot == 0 ? vmc : mk0;
    mk1 = slot == 1 ? vmc : mk1;
    mk2 = slot == 2 ? vmc : mk2;
    mk3 = slot == 3 ? vmc : mk3;
  }

  // first pair of slabs of the sequence
  __device__ __forceinline__ void start(char* ring) {
    vmc = 0;
    nreq = 0;
    mk0 = mk1 = mk2 = mk3 = 0;
    static_assert(RR_NSLOT == 4, "slot of slab i = i % 4; slabs travel in pairs");
    issue(ring, 0);
    issue(ring, 1);
  }

  // Slabs are consumed in PAIRS (every stage has a multiple of four tiles): at the top of an even tile -- data in `slot`, slot + 1
  // -- wait for the pair's DMA (the lagging waves' own parts), meet the other waves (everybody's parts have landed, and everybody
  // is done with the previous pair's slots), request the next pair into those slots.  An odd tile needs nothing: ONE barrier per
  // two tiles (round 4 stamps: with one per tile a wave spent a third of a tile in the wait + barrier + request sequence --
  // the two halves of the waves run half a tile apart by design, and the barrier re-aligned them every tile).
  __device__ __forceinline__ void begin(char* ring, int slot) {
    if (slot & 1) return;
    __builtin_amdgcn_sched_barrier(0);
    RR_STAMP(3);  // everything since the previous pair's top that has no stamp of its own
    wait_vmcnt(vmc - mark(slot + 1));
    RR_STAMP(0);  // wait for the pair's DMA (and, in order, for every older store)
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS traffic (transposer, previous fragments) is done
#ifndef RR_EXP_NOBARRIER
    __builtin_amdgcn_s_barrier();
#endif
    RR_STAMP(1);  // barrier
    asm volatile("" ::: "memory");
    issue(ring, (slot + 2) & 3);
    issue(ring, (slot + 3) & 3);
    RR_STAMP(2);  // DMA issue
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void count(int n) {
#ifdef RR_EXP_NOSTORE
    (void)n;  // (the experiment issues no stores; the sign-bit DMAs count themselves through count_dma)
#else
    vmc += n;
#endif
  }
  __device__ __forceinline__ void count_dma(int n) { vmc += n; }
};

// acc (32 features x 32 samples, transposed) += slab k-blocks [p, p + T KiB) . o[0 .. T)
// Weight fragments come from the LDS slot through a ring PF k-blocks ahead of their MFMAs; the order is pinned (hipcc otherwise
// sinks the reads to just before their use and waits lgkmcnt(0) every other MFMA).
template <int T, int N, bool FIRST = false, int PF = RR_PF>
__device__ __forceinline__ void kpart(f32x16& acc, const char* p, const h8 (&o)[N]) {
  static_assert(T <= N, "operand array");
  constexpr int SETS = PF + 1;
  h8 af[SETS];
#ifdef RR_EXP_NOLDS  // timing experiment: no weight-fragment reads
#pragma unroll
  for (int t = 0; t < SETS; ++t) af[t] = o[0];
#define RR_LDSREAD(dst, addr) asm volatile("" ::"v"(addr))
#else
#define RR_LDSREAD(dst, addr) dst = *(const h8*)(addr)
#endif
#pragma unroll
  for (int t = 0; t < PF && t < T; ++t) RR_LDSREAD(af[t], p + t * 1024);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t + PF < T) RR_LDSREAD(af[(t + PF) % SETS], p + (t + PF) * 1024);
    __builtin_amdgcn_sched_barrier(0);
#ifndef RR_EXP_NOMMA
    if (FIRST && t == 0) {  // C = 0 as an inline constant: no 16-register clear in front of the tile
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t % SETS], o[t], z, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t % SETS], o[t], acc, 0, 0, 0);
    }
#else
    asm volatile("" ::"v"(af[t % SETS]), "v"(o[t]));
#endif
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
// rows [0, 32) of the tile -> dst[(row0 + r) * ld + col0 + 0..31]; four store instructions of 8 rows x 128 B.
// part != nullptr: the tile's column sums over rows [0, rb) and [rb, 32) (the wave's first and second ray; rb >= 32: one ray)
// also go to part[col0 + ..] and part[128 + col0 + ..] -- this wave's share of upnerf_ray_sum, in a fixed order.  Those two
// stores are NOT counted by the caller (an under-count only makes a later wait stricter).
__device__ __forceinline__ void stg_flush(const char* stg, int lane, float* __restrict__ dst, size_t ld, int col0,
                                          float* __restrict__ part = nullptr, int rb = 32) {
  const int c = lane & 7;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3);
    const f32x4 v = *(const f32x4*)(stg + row * 128 + ((c ^ (row & 7)) << 4));
#ifndef RR_EXP_NOSTORE
    if (dst) NT_STORE((f32x4*)(dst + (size_t)row * ld + col0 + 4 * c), v);  // (dst == nullptr: the sums only; the caller counts no stores)
#else
    asm volatile("" ::"v"(v));
#endif
    if (part) {
      const float k = row < rb ? 1.0f : 0.0f;
      s0 += v * k;
      s1 += v * (1.0f - k);
    }
  }
  if (part) {
#pragma unroll
    for (int sh = 8; sh < 64; sh <<= 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s0[u] += __shfl_xor(s0[u], sh);
        s1[u] += __shfl_xor(s1[u], sh);
      }
    }
    if (lane < 8) {
      *(f32x4*)(part + col0 + 4 * c) = s0;
      if (rb < 32) *(f32x4*)(part + 128 + col0 + 4 * c) = s1;
    }
  }
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

// ---- epilogue of one 32-feature tile, forward and backward.  What an epilogue costs decides the kernel: with two waves per SIMD
// a wave's vector instructions take ~4 cycles each, the 16 MFMAs of a tile 512 -- an epilogue of 200 instructions (round 4's
// first builds) ran 2.7x longer than the contraction it follows and the two never overlapped (elimination runs: 0.8 ms per
// launch without epilogues, 2.2 ms without MFMAs, 2.85 ms with both).  So everything is done on register PAIRS (two consecutive
// features of the lane's sample), in as few instructions as the ISA allows:
//   FAST form (SCALED = true; the trunk): the addend arrives pre-multiplied by 2^eo (a per-wave, per-layer copy of the bias in LDS)
//     s = acc * (un 2^eo) + add'             v_pk_fma_f32                 h = fp16(s)   v_cvt_pk_f16_f32
//     [ReLU]  v_pk_max_f16                   [sign bits out]  v_pk_min_u16 + v_lshl_or_b32
//     [sign-bit mask in]  4 integer ops      running maximum  v_pk_max_f16 (in units of 2^-eo: vmax2)
//   EXACT form (SCALED = false; stages whose fp32 rows leave through the transposer): v = acc * un + add in natural units
//     (v_pk_fma_f32), ReLU / sign-bit mask in fp32, rows to the transposer, then h = fp16(v 2^eo).
// A pair IS one register of the next operand's fragment: blk[q / 2] register 2 (q % 2) + t.  Sign bits of a tile = one word `tw`:
// element r of the tile at bit 16 (r % 2) + r / 2 (the halves of a pair 16 bits apart); tiles 2 w and 2 w + 1 share word w of the
// lane's four words per stage (tile 2 w + 1 shifted left by 8).  A positive pre-activation that underflows fp16 (below 2^-38 of
// the stage's bound) counts as inactive.
template <bool SCALED, int ADD /*0 none, 1 add_s[col], 2 add_s[col] * add_sc*/, int MASKIN /*0, 1*/, bool RELU, bool MASKOUT>
__device__ __forceinline__ void tile_epilogue(const f32x16& acc, int jp, float un, float pe, const float* add_s, float add_sc, unsigned int tw_in,
                                              unsigned int& tw_out, float& vmax, h2& vmax2, char* stg, u32x4_t (&blk)[2], int li, int hh) {
  f32x4 aq[4];
  if constexpr (ADD != 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) aq[q] = *(const f32x4*)&add_s[32 * jp + 8 * q + 4 * hh];  // requested together: one LDS wait per tile
  }
  const float s1 = SCALED ? un * pe : un;
  const f32x2 un2 = {s1, s1}, pe2 = {pe, pe}, sc2 = {add_sc, add_sc};
  tw_out = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x2 vv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x2 av = {acc[4 * q + 2 * t], acc[4 * q + 2 * t + 1]};
      const int bit = 2 * q + t;
      f32x2 v;
      if constexpr (ADD == 1) v = __builtin_elementwise_fma(av, un2, f32x2{aq[q][2 * t], aq[q][2 * t + 1]});
      else if constexpr (ADD == 2) v = __builtin_elementwise_fma(f32x2{aq[q][2 * t], aq[q][2 * t + 1]}, sc2, av * un2);
      else v = av * un2;
      h2 h;
      if constexpr (SCALED) {
        h = __builtin_convertvector(v, h2);
        if constexpr (RELU) h = __builtin_elementwise_max(h, h2{(_Float16)0, (_Float16)0});
      } else {
        if constexpr (RELU) v = f32x2{fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
        if constexpr (MASKIN == 1) {
          v[0] = __uint_as_float(__float_as_uint(v[0]) & (unsigned int)(((int)(tw_in << (31 - bit))) >> 31));
          v[1] = __uint_as_float(__float_as_uint(v[1]) & (unsigned int)(((int)(tw_in << (15 - bit))) >> 31));
        }
        vmax = fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1])));
        h = __builtin_convertvector(v * pe2, h2);
      }
      if constexpr (SCALED && MASKIN == 1) {
        const unsigned int m = (tw_in >> bit) & 0x00010001u;
        const u16x2_t full = u16x2_t{0, 0} - __builtin_bit_cast(u16x2_t, m);  // 0xffff where the bit is set
        h = __builtin_bit_cast(h2, __builtin_bit_cast(unsigned int, h) & __builtin_bit_cast(unsigned int, full));
      }
      if constexpr (SCALED) vmax2 = __builtin_elementwise_max(vmax2, MASKIN == 1 ? __builtin_elementwise_abs(h) : h);
      if constexpr (MASKOUT) {
        // (in C -- min(bits, 1) per half -- hipcc "knows" the operand is a float and emits ~30 compare / select instructions
        // per pair; the operand is a conversion result, never an MFMA destination: DESIGN 4.4)
        unsigned int one;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(one) : "v"(__builtin_bit_cast(unsigned int, h)), "v"(0x00010001u));
        tw_out |= one << bit;
      }
      blk[q >> 1][2 * (q & 1) + t] = __builtin_bit_cast(unsigned int, h);
      vv[t] = v;
    }
    if constexpr (!SCALED) {
      if (stg) stg_put(stg, li, hh, q, f32x4{vv[0][0], vv[0][1], vv[1][0], vv[1][1]});
    }
  }
}
__device__ __forceinline__ float pk_hmax(h2 v) { return fmaxf((float)v[0], (float)v[1]); }
// NV dot products of this lane's sample row, held as KB operand fragments at exponent e, with LDS-staged fp32 vectors w_s[c * ld ..]:
// the 1- and 3-wide heads (density, colour).  Outside the slab loop: inside it, three running sums per tile made hipcc spill
// 200 registers around the colour head.
template <int KB, int NV, int N>
__device__ __forceinline__ void operand_dots(const h8 (&op)[N], int e, const float* w_s, int ld, int hh, float (&out)[3]) {
  float acc[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) acc[c] = 0.0f;
#pragma unroll
  for (int s = 0; s < KB; ++s) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = (float)op[s][u];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const f32x4 w0 = *(const f32x4*)&w_s[c * ld + 16 * s + 4 * hh], w1 = *(const f32x4*)&w_s[c * ld + 16 * s + 8 + 4 * hh];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[c] = fmaf(x[u], w0[u], fmaf(x[4 + u], w1[u], acc[c]));
    }
  }
  const float un = ldexpf(1.0f, -e);
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    acc[c] += __shfl_xor(acc[c], 32);
    out[c] = acc[c] * un;
  }
}

// one operand fragment (k-block blk of this wave's 32-row tile) -> the fragment-ordered fp16 tensor: 1 KiB per instruction
// (fb: k-blocks per 32-row tile of the tensor -- 16 for a 256-wide one, 8 for a 128-wide one)
__device__ __forceinline__ void frag_store(uint16_t* __restrict__ base, size_t tile32, int blk, int lane, const h8& v, int fb = 16) {
#ifdef RR_EXP_NOSTORE  // timing experiment (wrong results): no activation / gradient stores at all
  return;
#endif
  NT_STORE((f32x4*)((char*)base + (tile32 * fb + blk) * 1024 + lane * 16), __builtin_bit_cast(f32x4, v));
}

__device__ __forceinline__ void track_lds(unsigned int* mx_s, int slot, float wave_mx, int lane) {
  if (lane == 0) atomicMax(&mx_s[slot], __float_as_uint(wave_mx));
}

// ---- what a stage does with a finished tile besides the arithmetic: its operand fragments leave (frag != nullptr) and enter the
// next operand, its fp32 rows leave through the transposer (rows != nullptr)
struct TileOut {
  uint16_t* frag;   // fragment-ordered fp16 tensor of this stage, or nullptr
  float* rows;      // row-major fp32 tensor (this wave's first row), or nullptr
  int ld;           // its row stride
  int fb = 16;            // k-blocks per tile of `frag` (8: a 128-wide tensor)
  float* part = nullptr;  // per-ray column sums of the rows (stg_flush), or nullptr
  int rb = 32;            // first row of the wave's second ray
};
template <int NW, int BLK0, int JP, int NB, class RG>
__device__ __forceinline__ void tile_out(RG& rg, const TileOut& o, const u32x4_t (&blk)[2], h8 (&nx)[NB], char* stg, size_t t32, int lane) {
  nx[BLK0 + 2 * JP] = __builtin_bit_cast(h8, blk[0]);
  nx[BLK0 + 2 * JP + 1] = __builtin_bit_cast(h8, blk[1]);
  if (o.frag) {
    frag_store(o.frag, t32, 2 * JP, lane, nx[BLK0 + 2 * JP], o.fb);
    frag_store(o.frag, t32, 2 * JP + 1, lane, nx[BLK0 + 2 * JP + 1], o.fb);
    rg.count(2);
  }
  if (o.rows || o.part) {
    stg_flush(stg, lane, o.rows, o.ld, 32 * JP, o.part, o.rb);
    if (o.rows) rg.count(4);
  }
}
// sign-bit words of a stage: tiles 2 w, 2 w + 1 -> word w
template <int JP>
__device__ __forceinline__ void mask_add(u32x4_t& words, unsigned int tw) {
  if constexpr ((JP & 1) == 0) words[JP >> 1] = tw;
  else words[JP >> 1] |= tw << 8;
}
template <int JP>
__device__ __forceinline__ unsigned int mask_tile(const u32x4_t& words) {
  return words[JP >> 1] >> (8 * (JP & 1));
}

// ---- the tile loop of a stage, unrolled (NT tiles: every register index of the operand being built is a compile-time constant;
// a run-time loop over dynamically indexed 32-register vectors was built and measured slower: hipcc copies the whole vector
// around every insertion).  Slab j sits in ring slot j % 4 (every stage has a multiple of four tiles).  The two waves of a
// SIMD (w and w + NW/2) run it half a tile apart: the leading half contracts tile j and then runs its epilogue, the lagging
// half runs the epilogue of tile j - 1 and then contracts tile j -- one wave's vector work beside the other's matrix work on
// every SIMD, one accumulator per wave (MI355X_MICROARCH.md, "try a stagger").  mma(p): the contraction of the slab at LDS
// address p; epi(J): epilogue + outputs of tile J.
template <int NW, int NT, class MMA, class EPI, class RG>
__device__ __forceinline__ void run_tiles(RG& rg, char* lds, bool lag, MMA mma, EPI epi) {
  static_for<0, NT>([&](auto J) {
    constexpr int j = decltype(J)::value, slot = j & 3;
    rg.begin(lds, slot);
#ifndef RR_EXP_NOEPI
    if constexpr (j > 0) {
      if (lag) epi(std::integral_constant<int, (j > 0 ? j - 1 : 0)>{});
    }
#endif
    RR_STAMP_RG(rg, 5);
    mma(lds + slot * RR_SLOT + rg.lane * 16);
    RR_STAMP_RG(rg, 4);
#ifndef RR_EXP_NOEPI
    if (!lag) epi(J);
#endif
    RR_STAMP_RG(rg, 5);
  });
#ifndef RR_EXP_NOEPI
  if (lag) epi(std::integral_constant<int, NT - 1>{});
#endif
}

// ================================================================================================================================
// forward
// ================================================================================================================================
template <int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void rr16_fwd_kernel(upnerf_layout L, upnerf_field_fwd_args a) {
  constexpr int W = 256, W2 = 128;
  using C = RRCfg<NW>;
  __shared__ __attribute__((aligned(16))) char lds[C::LDS];  // ONE object: [ring | transposers | vectors | ray rows | tables]
  __shared__ float wk_w[NW][16];                             // the ten band weights, one row per wave (encoding loop)
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
  // exponents and row norms of the matrices: copied to LDS once (read through the pointers inside the slab loop they are
  // vector-memory loads -- the kernel stores, so hipcc cannot prove them read-only -- and every one drains the DMA queue)
  const int* wexp = int_s + 64;
  const float* wnorm = (const float*)(int_s + 80);
  const bool use_rgb = a.use_rgb != 0, use_cand = a.use_cand != 0;
  const bool train = a.h16 != nullptr;
  const int last_stage = (!a.e && !a.e16 && !use_rgb && !use_cand) ? D - 1 : ((!use_rgb && !use_cand) ? D : D + 3);

  // ---- every ordinary load of the prologue, requested back to back and UNCONDITIONALLY (clamped indices, pointer selects for
  // absent inputs; the values are masked when they are stored).  One workgroup owns a CU here, so nothing hides a prologue that
  // takes its loads one at a time -- and behind `l < D ? P[..] : 0`, `use_rgb ? P[..] : 0`, `if (tid < n)` hipcc branches around
  // every load and waits for it before the next: the ISA of round 5 had twenty `global_load ; s_waitcnt vmcnt(0)` pairs in a row
  // in front of this kernel's first barrier, and three more per trip of the encoding loop (the band weights).
  static_assert(C::THREADS >= 3 * W2 && C::THREADS >= W, "one thread per vector element");
  const int tw = tid & (W - 1), th = tid & (W2 - 1);
  float bias_v[UPNERF_MAX_D];
#pragma unroll
  for (int l = 0; l < UPNERF_MAX_D; ++l) bias_v[l] = P[(l < D ? L.b[l] : L.b[0]) + tw];
  const float be_v = P[L.be + tw], wsig_v = P[L.wsig + tw];
  const float br1_v = P[L.br1 + th], bc1_v = P[L.bc1 + th], bc2_v = P[L.bc2 + th], wcsig_v = P[L.wcsig + th];
  const float wr2_v = P[L.wr2 + (tid < 3 * W2 ? tid : 0)];
  const float bsig_v = P[L.bsig], bcsig_v = P[L.bcsig];               // (head biases: the layout has them whatever heads run)
  const float br2_v[3] = {P[L.br2], P[L.br2 + 1], P[L.br2 + 2]};
  const int wexp_v = a.wexp[tid < RR_MAXSTAGE ? tid : 0];
  const float wnorm_v = a.wnorm[tid < RR_MAXSTAGE ? tid : 0];
  const int mlast = (blockIdx.x * C::TILE + C::TILE < M ? blockIdx.x * C::TILE + C::TILE : M) - 1;
  const int nr = mlast / S - ray0 + 1;
  const float* __restrict__ aux_src = use_rgb ? a.aux + (size_t)ray0 * UPNERF_AUXK : P;
  const float* __restrict__ crow_src = use_cand ? a.c_rows + (size_t)ray0 * UPNERF_CK : P;
  const float aux_v = aux_src[use_rgb && tid < nr * UPNERF_AUXK ? tid : 0];
  const float crow_v = crow_src[use_cand && tid < nr * UPNERF_CK ? tid : 0];
  const float zz = a.z[mc];
  float ro_v[3], rd_v[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    ro_v[n] = a.rays_o[3 * ray + n];
    rd_v[n] = a.rays_d[3 * ray + n];
  }
  // the ten band weights (device table under graph replay): one request per wave here, read from a wave-private LDS row inside the
  // encoding loop
  const float* __restrict__ wk_src = a.wk_xyz_dev ? a.wk_xyz_dev : P;
  float wk_v = wk_src[lane < 10 ? lane : 9];
  if (!a.wk_xyz_dev) wk_v = a.wk_xyz[lane < 10 ? lane : 9];

  // ---- stage tables, vectors, per-ray rows
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
    int_s[64 + st] = wexp_v;
    ((float*)int_s)[80 + st] = wnorm_v;
  }
  if (tid < W) {
#pragma unroll
    for (int l = 0; l < UPNERF_MAX_D; ++l) vec_s[256 * l + tid] = l < D ? bias_v[l] : 0.0f;
    vec_s[RR_V_BE + tid] = be_v;
    vec_s[RR_V_WSIG + tid] = wsig_v;
  }
  if (tid < W2) {
    vec_s[RR_V_BR1 + tid] = use_rgb ? br1_v : 0.0f;
    vec_s[RR_V_BC1 + tid] = use_cand ? bc1_v : 0.0f;
    vec_s[RR_V_BC2 + tid] = use_cand ? bc2_v : 0.0f;
    vec_s[RR_V_WCSIG + tid] = use_cand ? wcsig_v : 0.0f;
  }
  if (tid < 3 * W2) vec_s[RR_V_WR2 + tid] = use_rgb ? wr2_v : 0.0f;
  if (lane < 10) wk_w[wave][lane] = wk_v;
  // per-ray side inputs of the heads: [aux 80 | candidate row 16] per ray slot; their largest magnitude bounds the exponent of e
  float sidemax = 0.0f;
  {
    if (use_rgb && tid < nr * UPNERF_AUXK) {
      rows_s[(tid / UPNERF_AUXK) * C::ROWF + tid % UPNERF_AUXK] = aux_v;
      sidemax = fabsf(aux_v);
    }
    if (use_cand && tid < nr * UPNERF_CK) {
      rows_s[(tid / UPNERF_CK) * C::ROWF + 96 + tid % UPNERF_CK] = crow_v;
      sidemax = fmaxf(sidemax, fabsf(crow_v));
    }
    if (use_rgb)  // (more rows than threads: S < 64 only)
      for (int i = tid + C::THREADS; i < nr * UPNERF_AUXK; i += C::THREADS) {
        const float v = a.aux[(size_t)ray0 * UPNERF_AUXK + i];
        rows_s[(i / UPNERF_AUXK) * C::ROWF + i % UPNERF_AUXK] = v;
        sidemax = fmaxf(sidemax, fabsf(v));
      }
    if (use_cand)
      for (int i = tid + C::THREADS; i < nr * UPNERF_CK; i += C::THREADS) {
        const float v = a.c_rows[(size_t)ray0 * UPNERF_CK + i];
        rows_s[(i / UPNERF_CK) * C::ROWF + 96 + i % UPNERF_CK] = v;
        sidemax = fmaxf(sidemax, fabsf(v));
      }
  }
  // ---- sample position (rendering.py:251 / 308) and its encoding (nerf.py:126-147): each lane half evaluates 15 of the 30
  // (coordinate, band) pairs of ITS sample once; the halves meet in an LDS scratch (the ring, not yet in use)
  float xm;
  {
    float xyz[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) xyz[n] = mul_then_add(ro_v[n], rd_v[n], zz);
    xm = fmaxf(fmaxf(fabsf(xyz[0]), fabsf(xyz[1])), fmaxf(fabsf(xyz[2]), 1.0f));  // |sin|, |cos| <= 1
    float* pe = (float*)ring + (32 * wave + li) * RR_PE_LD;
    const float* wkw = wk_w[wave];  // (written by this wave above: LDS operations of a wave execute in order)
    asm volatile("" ::: "memory");
    if (hh == 0) {
      pe[0] = xyz[0];
      pe[1] = xyz[1];
      pe[2] = xyz[2];
      pe[63] = 0.0f;
    }
    // A sample's 30 (coordinate, band) items are shared by its two lanes so that BOTH lanes are in the same band or in two
    // neighbouring ones on every trip: trips 0-9 = band p of x (lane half 0) and of y (half 1), trips 10-14 = bands 2q and
    // 2q + 1 of z.  A trip whose band weights are exactly zero (all below progress 0.1, half of them at 0.3: nerf.py:136-141)
    // then skips the fp64 sincos for the whole wave and writes the zeros sin * 0, cos * 0 would have been.
#pragma unroll 1
    for (int p = 0; p < 15; ++p) {
      const int n = p < 10 ? hh : 2, k = p < 10 ? p : 2 * (p - 10) + hh;
      const int ku = p < 10 ? p : 2 * (p - 10);  // wave-uniform: the (first) band of this trip
      const float wk = wkw[k];
      const float wu0 = wkw[ku], wu1 = wkw[ku + (p >= 10)];
      float sv = 0.0f, cv = 0.0f;
      if ((__builtin_amdgcn_readfirstlane(__float_as_uint(wu0)) | __builtin_amdgcn_readfirstlane(__float_as_uint(wu1))) != 0u) {
        const float xv = n == 0 ? xyz[0] : (n == 1 ? xyz[1] : xyz[2]);
        sincos_f32_via_f64(xv * ldexpf(PI_F, k), sv, cv);
        sv *= wk;
        cv *= wk;
      }
      pe[3 + 20 * n + k] = sv;
      pe[3 + 20 * n + 10 + k] = cv;
    }
  }
  __syncthreads();
  int2* slab_s = (int2*)(lds + C::SLB0);
  const int nslab = build_slab_table(slab_s, int_s, int_s + 16, int_s + 32, nullptr, D + 4, tid);  // (visible after the next barrier)
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

  const float bsig = bsig_v;
  float br2[3] = {use_rgb ? br2_v[0] : 0.f, use_rgb ? br2_v[1] : 0.f, use_rgb ? br2_v[2] : 0.f};
  const float bcsig = use_cand ? bcsig_v : 0.0f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every ordinary load / store of the prologue has retired: the counter starts at 0

  Ring<NW> rg;
  rg.src = (const char*)a.P16;
  rg.slab_s = slab_s;
  rg.nslab = nslab;
  rg.wave = wave;
  rg.lane = lane;
#ifdef UPNERF_STAMPS
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 10; ++i) rg.st_acc[i] = 0;
  rg.st_prev = st_t0;
#endif
  rg.start(lds);

  h8 Bh[16];  // operand of the running stage (previous stage's outputs)
  h8 Nh[16];  // operand of the next stage, filled tile by tile
  float dot3[3] = {0.f, 0.f, 0.f};
  int e_in = e0;          // exponent of the operand the running stage reads
  float amax_in = x0max;  // its exact largest magnitude in this wave
  const bool lag = wave >= NW / 2;
  f32x16 acc;
  float* bsw = (float*)(lds + C::BSW0 + wave * 1024);

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
    const float pe = pow2r(eo);
    // this wave's copy of the layer's bias in units of 2^-eo: the epilogue then is ONE packed fma per pair (LDS traffic of a
    // wave is ordered: no barrier)
    *(f32x4*)&bsw[4 * lane] = *(const f32x4*)&vec_s[256 * l + 4 * lane] * pe;
    const bool rows32 = train && a.h != nullptr && l == D - 1;  // fp32 copy of the last trunk layer (density-head / final-layer gradients)
    const int xkb = (has_x && has_h) ? UPNERF_X0 / 16 : 0;  // k-blocks of the encoding part in front of the h part
    TileOut to;
    to.frag = train ? a.h16 + (size_t)l * nt32 * 16 * 512 : nullptr;
    to.rows = rows32 ? a.h + (size_t)m0 * W : nullptr;
    to.ld = W;
    u32x4_t words = {0u, 0u, 0u, 0u};
    float vmax = 0.0f;
    h2 vmax2 = {(_Float16)0, (_Float16)0};
    const float inv_pe = pow2r(-eo);
    run_tiles<NW, 8>(
        rg, lds, lag,
        [&](const char* p) {
          if (has_x) {
            kpart<4, 4, true>(acc, p, Xh);
            if (has_h) kpart<16>(acc, p + xkb * 1024, Bh);
          } else {
            kpart<16, 16, true>(acc, p, Bh);
          }
        },
        [&](auto JP) {
          constexpr int jp = decltype(JP)::value;
          u32x4_t blk[2];
          unsigned int tw;
          // (last layer: the exact form, whose fp32 rows -- the operand of the density head's and the final layer's weight
          // gradients -- leave through the transposer; every other layer: the scaled form, one packed fma per pair)
          RR_STAMP_RG(rg, 6);  // (whatever precedes the epilogue since the last stamp)
          if (rows32) tile_epilogue<false, 1, 0, true, true>(acc, jp, un, pe, vec_s + 256 * l, 1.0f, 0u, tw, vmax, vmax2, stg, blk, li, hh);
          else tile_epilogue<true, 1, 0, true, true>(acc, jp, un, pe, bsw, 1.0f, 0u, tw, vmax, vmax2, nullptr, blk, li, hh);
          mask_add<jp>(words, tw);
          asm volatile("" ::"v"(blk[0]), "v"(blk[1]));
          RR_STAMP_RG(rg, 7);  // epilogue arithmetic (incl. the wait for the tile's last MFMA and for the bias quads)
          tile_out<NW, 0, jp>(rg, to, blk, Nh, stg, t32, lane);
          RR_STAMP_RG(rg, 8);  // fragment / row stores
        });
    if (train) {
#ifndef RR_EXP_NOSTORE
      NT_STORE((u32x4_t*)((char*)a.hmask + (((size_t)l * nt32 + t32) * 64 + lane) * 16), words);
#endif
      rg.count(1);
      if (lane == 0) a.hexp[(size_t)l * nt32 + t32] = eo;  // (one lane: not counted)
    }
    amax_in = wave_max_rr(fmaxf(vmax, pk_hmax(vmax2) * inv_pe));
    track_lds(mx_s, l, amax_in, lane);
    e_in = eo;
#pragma unroll
    for (int s = 0; s < 16; ++s) Bh[s] = Nh[s];
  }

  // ---- shared density head (nerf.py:89): softplus(w . h + b), from the operand fragments
  {
    operand_dots<16, 1>(Bh, e_in, vec_s + RR_V_WSIG, W, hh, dot3);
    if (hh == 0 && valid) a.sigma_s[m] = softplus_f(dot3[0] + bsig);
  }
  h2 novmax2 = {(_Float16)0, (_Float16)0};
  if (last_stage >= D) {
    // ---- xyz_encoding_final (nerf.py:93), no activation; its operand form E feeds both heads
    float amax_e;
    int e_E;
    {
      const float un = pow2r(-(e_in + wexp[8]));
      const float bound = fmaxf(wnorm[D] * amax_in + vec_s[RR_V_BMAX + 8], sidemax);  // the heads add per-ray rows at E's exponent
      e_E = scale_exp(bound);
      const float pe = pow2r(e_E);
      TileOut to;
      to.frag = a.e16;  // (nullptr, or e as operand fragments: compositing and the heads' weight gradients read those instead)
      to.rows = a.e ? a.e + (size_t)m0 * W : nullptr;
      to.ld = W;
      char* stg_e = a.e ? stg : nullptr;
      float vmax = 0.0f;
      run_tiles<NW, 8>(
          rg, lds, lag,
          [&](const char* p) {
            kpart<16, 16, true>(acc, p, Bh);
          },
          [&](auto JP) {
            constexpr int jp = decltype(JP)::value;
            u32x4_t blk[2];
            unsigned int tw;
            tile_epilogue<false, 1, 0, false, false>(acc, jp, un, pe, vec_s + RR_V_BE, 1.0f, 0u, tw, vmax, novmax2, stg_e, blk, li, hh);
            tile_out<NW, 0, jp>(rg, to, blk, Nh, stg, t32, lane);
          });
      if (a.e16 && lane == 0) a.eexp[t32] = e_E;  // (one lane: not counted)
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
      const int e_R = scale_exp(wnorm[D + 3] * fmaxf(amax_e, sidemax) + vec_s[RR_V_BMAX + 9]);  // r1 as an operand: feeds the colour dots
      const float pe = pow2r(e_R);
      TileOut to;
      to.frag = train ? a.r1_16 : nullptr;  // (nullptr, or r1 as operand fragments for the colour output layer's weight gradient)
      to.fb = 8;
      to.rows = (train && a.r1) ? a.r1 + (size_t)m0 * W2 : nullptr;
      to.ld = W2;
      char* stg_r = to.rows ? stg : nullptr;
      u32x4_t words = {0u, 0u, 0u, 0u};
      float vmax = 0.0f;
      run_tiles<NW, 4>(
          rg, lds, lag,
          [&](const char* p) {
            kpart<16, 16, true>(acc, p, Bh);
            kpart<5>(acc, p + 16 * 1024, Ah);
          },
          [&](auto JP) {
            constexpr int jp = decltype(JP)::value;
            u32x4_t blk[2];
            unsigned int tw;
            tile_epilogue<false, 1, 0, true, true>(acc, jp, un, pe, vec_s + RR_V_BR1, 1.0f, 0u, tw, vmax, novmax2, stg_r, blk, li, hh);
            mask_add<jp>(words, tw);
            tile_out<NW, 0, jp>(rg, to, blk, Nh, stg, t32, lane);
          });
      if (train) {
#ifndef RR_EXP_NOSTORE
        NT_STORE((u32x4_t*)((char*)a.hmask + (((size_t)(D + 2) * nt32 + t32) * 64 + lane) * 16), words);
#endif
        rg.count(1);
      }
      if (train && a.r1_16 && lane == 0) a.r1exp[t32] = e_R;
      track_lds(mx_s, D + 3, wave_max_rr(vmax), lane);
      operand_dots<8, 3>(Nh, e_R, vec_s + RR_V_WR2, W2, hh, dot3);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (hh == 0 && valid) a.rgb[(size_t)m * 3 + c] = sigmoid_f(dot3[c] + br2[c]);
    }
    // ---- candidate head (nerf.py:97-100)
    if (last_stage > D && use_cand) {
      h8 Ch[1];
      Ch[0] = row_op(rows_s + rs * C::ROWF + 96, 0, hh, e_E);
      int e_G;
      float amax_g1;
      {
        const float un = pow2r(-(e_E + wexp[9]));
        e_G = scale_exp(wnorm[D + 1] * fmaxf(amax_e, sidemax) + vec_s[RR_V_BMAX + 10]);
        const float pe = pow2r(e_G);
        TileOut to;
        to.frag = train ? a.g1_16 : nullptr;  // (nullptr, or g1 as operand fragments for candidate_encoding.2's weight gradient)
        to.fb = 8;
        to.rows = (train && a.g1) ? a.g1 + (size_t)m0 * W2 : nullptr;
        to.ld = W2;
        char* stg_g = to.rows ? stg : nullptr;
        u32x4_t words = {0u, 0u, 0u, 0u};
        float vmax = 0.0f;
        run_tiles<NW, 4>(
            rg, lds, lag,
            [&](const char* p) {
              kpart<16, 16, true>(acc, p, Bh);
              kpart<1>(acc, p + 16 * 1024, Ch);
            },
            [&](auto JP) {
              constexpr int jp = decltype(JP)::value;
              u32x4_t blk[2];
              unsigned int tw;
              tile_epilogue<false, 1, 0, true, true>(acc, jp, un, pe, vec_s + RR_V_BC1, 1.0f, 0u, tw, vmax, novmax2, stg_g, blk, li, hh);
              mask_add<jp>(words, tw);
              tile_out<NW, 0, jp>(rg, to, blk, Nh, stg, t32, lane);
            });
        if (train) {
#ifndef RR_EXP_NOSTORE
          NT_STORE((u32x4_t*)((char*)a.hmask + (((size_t)D * nt32 + t32) * 64 + lane) * 16), words);
#endif
          rg.count(1);
        }
        if (train && a.g1_16 && lane == 0) a.g1exp[t32] = e_G;
        amax_g1 = wave_max_rr(vmax);
        track_lds(mx_s, D + 1, amax_g1, lane);
      }
      {
        const float un = pow2r(-(e_G + wexp[10]));
        const int e_G2 = scale_exp(wnorm[D + 2] * amax_g1 + vec_s[RR_V_BMAX + 11]);  // g2 as an operand: feeds the candidate density dot
        const float pe = pow2r(e_G2);
        TileOut to;
        to.frag = a.g2_16;  // (nullptr, or g2 as operand fragments: compositing and the candidate density's weight gradient read those)
        to.fb = 8;
        to.rows = a.g2 ? a.g2 + (size_t)m0 * W2 : nullptr;  // compositing reads g2 in inference too
        to.ld = W2;
        char* stg_g2 = a.g2 ? stg : nullptr;
        u32x4_t words = {0u, 0u, 0u, 0u};
        float vmax = 0.0f;
        // (operand: g1 in Nh[0..8); result: g2 into Bh[0..8) -- E is no longer needed)
        run_tiles<NW, 4>(
            rg, lds, lag,
            [&](const char* p) {
              kpart<8, 16, true>(acc, p, Nh);
            },
            [&](auto JP) {
              constexpr int jp = decltype(JP)::value;
              u32x4_t blk[2];
              unsigned int tw;
              tile_epilogue<false, 1, 0, true, true>(acc, jp, un, pe, vec_s + RR_V_BC2, 1.0f, 0u, tw, vmax, novmax2, stg_g2, blk, li, hh);
              mask_add<jp>(words, tw);
              tile_out<NW, 0, jp>(rg, to, blk, Bh, stg, t32, lane);
            });
        if (train) {
#ifndef RR_EXP_NOSTORE
          NT_STORE((u32x4_t*)((char*)a.hmask + (((size_t)(D + 1) * nt32 + t32) * 64 + lane) * 16), words);
#endif
          rg.count(1);
        }
        if (a.g2_16 && lane == 0) a.g2exp[t32] = e_G2;
        operand_dots<8, 1>(Bh, e_G2, vec_s + RR_V_WCSIG, W2, hh, dot3);
        if (hh == 0 && valid) a.sigma_c[m] = softplus_f(dot3[0] + bcsig);
      }
    }
  }
#ifdef UPNERF_STAMPS
  if (lane == 0 && (blockIdx.x & 15) == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&upnerf_stamp_acc_rr[i], rg.st_acc[i]);
    atomicAdd(&upnerf_stamp_acc_rr[4], __builtin_amdgcn_s_memtime() - st_t0);  // slab loop, whole
    atomicAdd(&upnerf_stamp_acc_rr[5], 1ull);                                   // waves sampled
    atomicAdd(&upnerf_stamp_acc_rr[6], rg.st_acc[4]);                           // trunk: contraction
    atomicAdd(&upnerf_stamp_acc_rr[7], rg.st_acc[5]);                           // trunk: epilogue + stores
    for (int i = 6; i < 10; ++i) atomicAdd(&upnerf_stamp_acc_rr[16 + i - 6], rg.st_acc[i]);
  }
#endif
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
  static constexpr int SLB0 = INT0 + (4 * RB_NSTAGE + 48) * 4;  // slab list of the pass (<= 96 entries of 8 bytes)
  static constexpr int LDS = SLB0 + 96 * 8;
  static_assert(NW * 32 * 64 * 4 <= RING, "d x0 exchange scratch lives in the ring");
};

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
  const int* wexp = int_s + 4 * RB_NSTAGE + 16;                      // LDS copies (see the forward kernel)
  const float* wnt = (const float*)(int_s + 4 * RB_NSTAGE + 32);     // row norms of the transposed set, in descriptor order
  const bool use_rgb = a.use_rgb != 0, use_cand = a.use_cand != 0, heads = use_rgb || use_cand;
  const bool need_dxyz = a.need_dxyz != 0;
  const int hs = L.skip > 0 ? 1 : 0;                 // the skip layer has two transposed descriptors (row norms: descriptor order)
  const int gld = a.gz_rg_ld > 0 ? a.gz_rg_ld : W2;  // row stride of gz_r1 / gz_g1
  const bool lag = wave >= NW / 2;
  // per-ray sums of gz_r1 / gz_g1, this wave's part (a.tile_part: [M/32][UPNERF_RR_PART_STRIDE] = [gz_r1, gz_g1][ray slot 2][128])
  float* __restrict__ rpart = a.tile_part ? a.tile_part + t32 * UPNERF_RR_PART_STRIDE : nullptr;
  const int rb = __builtin_amdgcn_readfirstlane((m0 / S + 1) * S - m0);  // first row of this wave's second ray (>= 32: none)

  // ---- every ordinary load of the prologue, requested back to back and unconditionally (see the forward kernel: one workgroup
  // per CU, nothing hides a prologue that takes its ~25 loads one round trip at a time)
  static_assert(C::THREADS >= 3 * W2 && C::THREADS >= W, "one thread per vector element");
  const bool has_ge = a.g_E_s != nullptr, has_gc = use_cand && a.g_G_c != nullptr;
  const int wexp_v = a.wexp[tid < 16 ? tid : 0];
  const float wnorm_v = a.wnorm[32 + (tid < 16 ? tid : 0)];
  const float wsig_v = P[L.wsig + (tid & (W - 1))], wcsig_v = P[L.wcsig + (tid & (W2 - 1))], wr2_v = P[L.wr2 + (tid < 3 * W2 ? tid : 0)];
  const int mlast = (blockIdx.x * C::TILE + C::TILE < M ? blockIdx.x * C::TILE + C::TILE : M) - 1;
  const int nr = mlast / S - ray0 + 1;
  const float* __restrict__ ge_src = has_ge ? a.g_E_s + (size_t)ray0 * W : P;
  const float* __restrict__ gc_src = has_gc ? a.g_G_c + (size_t)ray0 * W2 : P;
  float ge_v[3], gc_v[2];
#pragma unroll
  for (int j = 0; j < 3; ++j) ge_v[j] = ge_src[has_ge && tid + j * C::THREADS < nr * W ? tid + j * C::THREADS : 0];
#pragma unroll
  for (int j = 0; j < 2; ++j) gc_v[j] = gc_src[has_gc && tid + j * C::THREADS < nr * W2 ? tid + j * C::THREADS : 0];
  const float dss_v = a.d_sigma_s[mc], ss_v = a.sigma_s[mc];
  const float wf_v = (has_ge ? a.w_feat_s : P)[has_ge ? mc : 0];
  const float dsc_v = (use_cand ? a.d_sigma_c : P)[use_cand ? mc : 0], sc_v = (use_cand ? a.sigma_c : P)[use_cand ? mc : 0];
  const float cw_v = (has_gc ? a.w_cj : P)[has_gc ? mc : 0];
  float y_v[3], dy_v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    y_v[c] = (use_rgb ? a.rgb : P)[use_rgb ? (size_t)mc * 3 + c : 0];
    dy_v[c] = (use_rgb ? a.d_rgb : P)[use_rgb ? (size_t)mc * 3 + c : 0];
  }
  // sign bits of g2 / r1 (forward slots D + 1, D + 2): ordinary loads, nothing is in flight yet
  const char* __restrict__ hm_src = (use_cand || use_rgb) ? (const char*)a.hmask : (const char*)P;
  u32x4_t bits_g2 = *(const u32x4_t*)(hm_src + (use_cand ? (((size_t)(D + 1) * nt32 + t32) * 64 + lane) * 16 : 0));
  u32x4_t bits_r1 = *(const u32x4_t*)(hm_src + (use_rgb ? (((size_t)(D + 2) * nt32 + t32) * 64 + lane) * 16 : 0));
  if (!use_cand) bits_g2 = u32x4_t{0u, 0u, 0u, 0u};
  if (!use_rgb) bits_r1 = u32x4_t{0u, 0u, 0u, 0u};

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
    if (st < 16) {
      mx_s[st] = 0u;
      int_s[4 * RB_NSTAGE + 16 + st] = wexp_v;
      ((float*)int_s)[4 * RB_NSTAGE + 32 + st] = wnorm_v;
    }
  }
  if (tid < W) vec_s[RB_V_WSIG + tid] = wsig_v;
  if (tid < W2) vec_s[RB_V_WCSIG + tid] = use_cand ? wcsig_v : 0.0f;
  if (tid < 3 * W2) vec_s[RB_V_WR2 + tid] = use_rgb ? wr2_v : 0.0f;
  // upstream gradients of the per-ray sums (rank-1 terms of d e and d g2), one row per ray slot; absent = zero
  float gEmax = 0.0f, gGmax = 0.0f;
  {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + j * C::THREADS;
      if (i < nr * W) {
        const float v = has_ge ? ge_v[j] : 0.0f;
        rows_s[(i >> 8) * C::ROWF + (i & 255)] = v;
        gEmax = fmaxf(gEmax, fabsf(v));
      }
    }
    for (int i = tid + 3 * C::THREADS; i < nr * W; i += C::THREADS) {  // (more than six rays per tile: S < 64 only)
      const float v = has_ge ? a.g_E_s[(size_t)ray0 * W + i] : 0.0f;
      rows_s[(i >> 8) * C::ROWF + (i & 255)] = v;
      gEmax = fmaxf(gEmax, fabsf(v));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = tid + j * C::THREADS;
      if (i < nr * W2) {
        const float v = has_gc ? gc_v[j] : 0.0f;
        rows_s[(i >> 7) * C::ROWF + 256 + (i & 127)] = v;
        gGmax = fmaxf(gGmax, fabsf(v));
      }
    }
    for (int i = tid + 2 * C::THREADS; i < nr * W2; i += C::THREADS) {
      const float v = has_gc ? a.g_G_c[(size_t)ray0 * W2 + i] : 0.0f;
      rows_s[(i >> 7) * C::ROWF + 256 + (i & 127)] = v;
      gGmax = fmaxf(gGmax, fabsf(v));
    }
  }
  // ---- per-row scalars: softplus'(x) = 1 - exp(-softplus(x)), sigmoid' = y (1 - y); rows past M carry zeros
  float dps = 0.0f, dpc = 0.0f, wf = 0.0f, cwj = 0.0f;
  f32x4 dprgb = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
    dps = dss_v * (1.0f - expf(-ss_v));
    if (has_ge) wf = wf_v;
    if (use_cand) {
      dpc = dsc_v * (1.0f - expf(-sc_v));
      if (has_gc) cwj = cw_v;
    }
    if (use_rgb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dprgb[c] = dy_v[c] * (y_v[c] * (1.0f - y_v[c]));
    }
    if (hh == 0) {
      a.dpre_sig_s[m] = dps;
      if (use_cand) a.dpre_sig_c[m] = dpc;
      if (use_rgb) *(f32x4*)&a.dpre_rgb[(size_t)m * 4] = dprgb;
    }
  }
  gEmax = wave_max_rr(gEmax);
  gGmax = wave_max_rr(gGmax);
  __syncthreads();  // tables written
  int2* slab_s = (int2*)(lds + C::SLB0);
  const int nslab = build_slab_table(slab_s, int_s, int_s + RB_NSTAGE, int_s + 2 * RB_NSTAGE, int_s + 3 * RB_NSTAGE, 3 + 2 * (D - 1) + 1, tid);
  if (lane == 0) {
    atomicMax(&mx_s[14], __float_as_uint(gEmax));
    atomicMax(&mx_s[15], __float_as_uint(gGmax));
  }
  __syncthreads();  // tables, vectors, rows, row maxima visible
  gEmax = __uint_as_float(mx_s[14]);
  gGmax = __uint_as_float(mx_s[15]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every ordinary load / store of the prologue has retired: the counter starts at 0

  Ring<NW, true> rg;  // (requests as inline asm: lds_dma16)
  rg.src = (const char*)a.PT16;
  rg.slab_s = slab_s;
  rg.nslab = nslab;
  rg.wave = wave;
  rg.lane = lane;
#ifdef UPNERF_STAMPS
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 10; ++i) rg.st_acc[i] = 0;
  rg.st_prev = st_t0;
#endif
  rg.start(lds);

  // ---- sign bits of the stages ahead: forward slots [D (g1)], D-1, ..., 0, each a 1 KiB LDS-DMA into buffer (slot & 1) of this
  // wave, requested two uses ahead; counted like every other vector-memory instruction of the loop
  int mmark0 = 0, mmark1 = 0;
  auto mask_request = [&](int slot) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // the reads of the buffer's previous content have returned
    lds_dma16<true>((const char*)a.hmask + (((size_t)slot * nt32 + t32) * 64 + lane) * 16, msk + (slot & 1) * 1024);
    asm volatile("" ::: "memory");
    rg.count_dma(1);
    if (slot & 1) mmark1 = rg.vmc;
    else mmark0 = rg.vmc;
  };
  auto mask_acquire = [&](int slot) -> u32x4_t {
    wait_vmcnt(rg.vmc - ((slot & 1) ? mmark1 : mmark0));
    asm volatile("" ::: "memory");
    const u32x4_t w = *(const u32x4_t*)(msk + (slot & 1) * 1024 + lane * 16);
    if (slot >= 2) mask_request(slot - 2);
    return w;
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
  float amax_in = 0.0f;
  int e_in = 0;
  h2 novmax2 = {(_Float16)0, (_Float16)0};
  // a 256-deep stage: the contraction of one slab with the DMA request at this wave's quarter of the k loop
  auto mma16 = [&](const char* p) {
    kpart<16, 16, true>(acc, p, Bh);
  };

  if (heads) {
    // ---- elementwise head stages, in operand order.  d g2 = relu'(g2) (w_csig dpre_c + w_cj g_G_c[ray]) (candidate_sigma /
    // feat_candidate_layer, nerf.py:99-100); d r1 = relu'(r1) W_r2^T (d rgb * rgb (1 - rgb)) (rgb_share_layer.2 + sigmoid)
    h8 Gh[8];
    float amax_g2 = 0.0f, amax_r1 = 0.0f;
    int e_g2 = 0;
    // value of element u of quad q of a tile, masked by the tile's sign-bit word
    auto masked = [](float t, unsigned int tw, int q, int u) {
      const int bit = 16 * (u & 1) + 2 * q + (u >> 1);
      return __uint_as_float(__float_as_uint(t) & (unsigned int)(((int)(tw << (31 - bit))) >> 31));
    };
    auto tile_word = [](const u32x4_t& words, int jt) { return words[jt >> 1] >> (8 * (jt & 1)); };
    if (use_cand) {
      const float* gG = rows_s + rs * C::ROWF + 256;
      float vmax = 0.0f;
      // two passes over the 128 features (maximum first: the exponent of the operand), nothing kept in between
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
          amax_g2 = wave_max_rr(vmax);
          track_lds(mx_s, D + 2, amax_g2, lane);
          e_g2 = scale_exp(amax_g2);
          if (a.gz_g2_16 && lane == 0) a.gzg2exp[t32] = e_g2;
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const unsigned int tw = tile_word(bits_g2, jt);
          h8 blk[2];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 32 * jt + 8 * q + 4 * hh;
            const f32x4 wv = *(const f32x4*)&vec_s[RB_V_WCSIG + col], gg = *(const f32x4*)&gG[col];
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              v[u] = masked(fmaf(wv[u], dpc, cwj * gg[u]), tw, q, u);
              vmax = fmaxf(vmax, fabsf(v[u]));
            }
            if (pass == 1) {
              if (a.gz_g2) stg_put(stg, li, hh, q, f32x4{v[0], v[1], v[2], v[3]});
              h4 hi, lo;
              split_quad<1>(ldexpf(v[0], e_g2), ldexpf(v[1], e_g2), ldexpf(v[2], e_g2), ldexpf(v[3], e_g2), hi, lo);
              if (q & 1) blk[q >> 1] = __builtin_shufflevector(blk[q >> 1], __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 12, 13, 14, 15);
              else blk[q >> 1] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3);
            }
          }
          if (pass == 1) {
            Gh[2 * jt] = blk[0];
            Gh[2 * jt + 1] = blk[1];
            if (a.gz_g2) {
              stg_flush(stg, lane, a.gz_g2 + (size_t)m0 * W2, W2, 32 * jt);
              rg.count(4);
            }
            if (a.gz_g2_16) {  // the operand fragments themselves (candidate_encoding.2's weight gradient reads them)
              frag_store(a.gz_g2_16, t32, 2 * jt, lane, blk[0], 8);
              frag_store(a.gz_g2_16, t32, 2 * jt + 1, lane, blk[1], 8);
              rg.count(2);
            }
          }
        }
      }
    }
    int erg = 0;
    if (use_rgb) {
      float vmax = 0.0f;
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
          amax_r1 = wave_max_rr(vmax);
          track_lds(mx_s, D + 3, amax_r1, lane);
          // exponent of the joint operand [gz_r1 | gz_g1] of the d e stage
          erg = scale_exp(fmaxf(use_cand ? wnt[D + hs + 3] * amax_g2 : 0.0f, amax_r1));
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const unsigned int tw = tile_word(bits_r1, jt);
          h8 blk[2];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 32 * jt + 8 * q + 4 * hh;
            const f32x4 w0 = *(const f32x4*)&vec_s[RB_V_WR2 + col], w1 = *(const f32x4*)&vec_s[RB_V_WR2 + W2 + col],
                        w2 = *(const f32x4*)&vec_s[RB_V_WR2 + 2 * W2 + col];
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              v[u] = masked(w0[u] * dprgb[0] + w1[u] * dprgb[1] + w2[u] * dprgb[2], tw, q, u);
              vmax = fmaxf(vmax, fabsf(v[u]));
            }
            if (pass == 1) {
              stg_put(stg, li, hh, q, f32x4{v[0], v[1], v[2], v[3]});
              h4 hi, lo;
              split_quad<1>(ldexpf(v[0], erg), ldexpf(v[1], erg), ldexpf(v[2], erg), ldexpf(v[3], erg), hi, lo);
              if (q & 1) blk[q >> 1] = __builtin_shufflevector(blk[q >> 1], __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 12, 13, 14, 15);
              else blk[q >> 1] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3);
            }
          }
          if (pass == 1) {
            Bh[2 * jt] = blk[0];
            Bh[2 * jt + 1] = blk[1];
            stg_flush(stg, lane, a.gz_r1 ? a.gz_r1 + (size_t)m0 * gld : nullptr, gld, 32 * jt, rpart, rb);
            if (a.gz_r1) rg.count(4);
          }
        }
      }
    } else {
      erg = scale_exp(use_cand ? wnt[D + hs + 3] * amax_g2 : 0.0f);
#pragma unroll
      for (int s = 0; s < 8; ++s) Bh[s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    float amax_g1 = 0.0f;
    if (use_cand) {
      // ---- gz_g1 = relu'(g1) (gz_g2 . W_c2)   (candidate_encoding.2, nerf.py:98)
      const u32x4_t words = mask_acquire(D);
      const float un = pow2r(-(e_g2 + wexp[10]));
      const float pe = pow2r(erg);
      TileOut to;
      to.frag = nullptr;
      to.rows = a.gz_g1 ? a.gz_g1 + (size_t)m0 * gld : nullptr;
      to.ld = gld;
      to.part = rpart ? rpart + 2 * W2 : nullptr;
      to.rb = rb;
      float vmax = 0.0f;
      run_tiles<NW, 4>(
          rg, lds, lag,
          [&](const char* p) {
            kpart<8, 8, true>(acc, p, Gh);
          },
          [&](auto JP) {
            constexpr int jp = decltype(JP)::value;
            u32x4_t blk[2];
            unsigned int tw;
            tile_epilogue<false, 0, 1, false, false>(acc, jp, un, pe, nullptr, 0.0f, mask_tile<jp>(words), tw, vmax, novmax2, stg, blk, li, hh);
            tile_out<NW, 8, jp>(rg, to, blk, Bh, stg, t32, lane);
          });
      amax_g1 = wave_max_rr(vmax);
      track_lds(mx_s, D + 1, amax_g1, lane);
    } else {
#pragma unroll
      for (int s = 8; s < 16; ++s) Bh[s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    if (use_cand && use_rgb) track_lds(mx_s, D + 4, fmaxf(amax_g1, amax_r1), lane);  // of [gz_r1 | gz_g1] as one tensor (gz_rg_ld)
    if (a.gz_rg16) {
      // [gz_r1 | gz_g1] as it stands in the operand registers: the fp16 fragments of the d e stage, one exponent per wave
#pragma unroll
      for (int b = 0; b < 16; ++b) frag_store(a.gz_rg16, t32, b, lane, Bh[b]);
      rg.count(16);
      if (lane == 0) a.gzrgexp[t32] = erg;
    }
    // ---- d e = [gz_r1 | gz_g1] . [W_fold | W_c1e] + w_feat g_E_s[ray]   (e has no activation)
    {
      const float un = pow2r(-(erg + wexp[12]));
      const float wfmax = wave_max_rr(fabsf(wf));
      const float bound = wnt[D + hs + 1] * amax_r1 + wnt[D + hs + 2] * amax_g1 + wfmax * gEmax;
      const int eo = scale_exp(bound);
      const float pe = pow2r(eo);
      const float* gE = rows_s + rs * C::ROWF;
      // a.gz_e == NULL: d e leaves as operand fragments, layer D of gz16 / gzexp (the final layer's weight gradient then reads
      // 512 bytes per sample instead of 1024, and this stage stores two 1 KiB pieces per tile instead of four)
      const bool efrag = a.gz_e == nullptr;
      TileOut to;
      to.frag = efrag ? a.gz16 + (size_t)D * nt32 * 16 * 512 : nullptr;
      to.rows = efrag ? nullptr : a.gz_e + (size_t)m0 * W;
      to.ld = W;
      float vmax = 0.0f;
      h2 vmax2 = {(_Float16)0, (_Float16)0};
      const float wf_pe = wf * pe;
      run_tiles<NW, 8>(rg, lds, lag, mma16, [&](auto JP) {
        constexpr int jp = decltype(JP)::value;
        u32x4_t blk[2];
        unsigned int tw;
        if (efrag) tile_epilogue<true, 2, 0, false, false>(acc, jp, un, pe, gE, wf_pe, 0u, tw, vmax, vmax2, nullptr, blk, li, hh);
        else tile_epilogue<false, 2, 0, false, false>(acc, jp, un, pe, gE, wf, 0u, tw, vmax, novmax2, stg, blk, li, hh);
        tile_out<NW, 0, jp>(rg, to, blk, Nh, stg, t32, lane);
      });
      if (efrag && lane == 0) a.gzexp[(size_t)D * nt32 + t32] = eo;
      amax_in = wave_max_rr(fmaxf(vmax, pk_hmax(vmax2) * pow2r(-eo)));
      track_lds(mx_s, D, amax_in, lane);
      e_in = eo;
#pragma unroll
      for (int s = 0; s < 16; ++s) Bh[s] = Nh[s];
    }
  } else {
    // no head consumed e: d e is the rank-1 feature term alone, w_feat g_E_s[ray] (a field without candidate encoding before the
    // colour head switches on: rendering.py:134-150) -- zero in a density-only evaluation; written either way because the weight
    // gradient of the final layer reads it.  Elementwise: the exponent comes from the bound max|w_feat| max|g_E_s|.
    const float wfmax = wave_max_rr(fabsf(wf));
    const int eo = scale_exp(wfmax * gEmax);
    const float* gE = rows_s + rs * C::ROWF;
    float vmax = 0.0f;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
      h8 blk[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * jt + 8 * q + 4 * hh;
        const f32x4 gg = *(const f32x4*)&gE[col];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u] = wf * gg[u];
          vmax = fmaxf(vmax, fabsf(v[u]));
        }
        if (a.gz_e) stg_put(stg, li, hh, q, f32x4{v[0], v[1], v[2], v[3]});
        h4 hi, lo;
        split_quad<1>(ldexpf(v[0], eo), ldexpf(v[1], eo), ldexpf(v[2], eo), ldexpf(v[3], eo), hi, lo);
        if (q & 1) blk[q >> 1] = __builtin_shufflevector(blk[q >> 1], __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 12, 13, 14, 15);
        else blk[q >> 1] = __builtin_shufflevector(hi, hi, 0, 1, 2, 3, 0, 1, 2, 3);
      }
      Bh[2 * jt] = blk[0];
      Bh[2 * jt + 1] = blk[1];
      if (a.gz_e) {
        stg_flush(stg, lane, a.gz_e + (size_t)m0 * W, W, 32 * jt);
        rg.count(4);
      } else {
        frag_store(a.gz16 + (size_t)D * nt32 * 16 * 512, t32, 2 * jt, lane, blk[0]);
        frag_store(a.gz16 + (size_t)D * nt32 * 16 * 512, t32, 2 * jt + 1, lane, blk[1]);
        rg.count(2);
      }
    }
    if (!a.gz_e && lane == 0) a.gzexp[(size_t)D * nt32 + t32] = eo;
    amax_in = wave_max_rr(vmax);
    track_lds(mx_s, D, amax_in, lane);
    e_in = eo;
  }
  // ---- d h_{D-1} = relu'(h_{D-1}) (gz_e . W_e + w_sig dpre_s)
  {
    const u32x4_t words = mask_acquire(D - 1);
    const float un = pow2r(-(e_in + wexp[8]));
    float wsmax = 0.0f;
    for (int c = lane; c < W; c += 64) wsmax = fmaxf(wsmax, fabsf(vec_s[RB_V_WSIG + c]));
    wsmax = wave_max_rr(wsmax);
    const float bound = wnt[D + hs] * amax_in + wsmax * wave_max_rr(fabsf(dps));
    const int eo = scale_exp(bound);
    const float pe = pow2r(eo);
    TileOut to;
    to.frag = a.gz16 + (size_t)(D - 1) * nt32 * 16 * 512;
    to.rows = nullptr;
    to.ld = 0;
    float novmax = 0.0f;
    h2 vmax2 = {(_Float16)0, (_Float16)0};
    const float dps_pe = dps * pe;  // the rank-1 term in units of 2^-eo
    run_tiles<NW, 8>(rg, lds, lag, mma16, [&](auto JP) {
      constexpr int jp = decltype(JP)::value;
      u32x4_t blk[2];
      unsigned int tw;
      tile_epilogue<true, 2, 1, false, false>(acc, jp, un, pe, vec_s + RB_V_WSIG, dps_pe, mask_tile<jp>(words), tw, novmax, vmax2, nullptr, blk, li, hh);
      tile_out<NW, 0, jp>(rg, to, blk, Nh, stg, t32, lane);
    });
    if (lane == 0) a.gzexp[(size_t)(D - 1) * nt32 + t32] = eo;
    amax_in = wave_max_rr(pk_hmax(vmax2) * pow2r(-eo));
    track_lds(mx_s, D - 1, amax_in, lane);
    e_in = eo;
#pragma unroll
    for (int s = 0; s < 16; ++s) Bh[s] = Nh[s];
  }
  // ---- trunk, last layer to first: gz_{l-1} = relu'(h_{l-1}) (gz_l . W_l); the skip layer also feeds d x0
  f32x16 accx[2];
  acc_clear(accx[0]);
  acc_clear(accx[1]);
  // d x0 (+)= gz . W[:, :64]: a stage of two tiles of 32 encoding features, issued twice so that the ring slots stay aligned
  auto dx0_stage = [&](float unx) {
    static_for<0, 4>([&](auto J) {
      constexpr int j = decltype(J)::value, slot = j & 3;
      rg.begin(lds, slot);
      if constexpr (j < 2) {
        acc_clear(acc);
        kpart<16>(acc, lds + slot * RR_SLOT + lane * 16, Bh);
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[j][r] = fmaf(acc[r], unx, accx[j][r]);
      }
    });
  };
#pragma unroll 1
  for (int l = D - 1; l >= 1; --l) {
    const int wel = wexp[l];
    if (need_dxyz && L.skip > 0 && l == L.skip) dx0_stage(pow2r(-(e_in + wel)));
    const u32x4_t words = mask_acquire(l - 1);
    const float un = pow2r(-(e_in + wel));
    const int eo = scale_exp(wnt[l + ((hs && l >= L.skip) ? 1 : 0)] * amax_in);
    const float pe = pow2r(eo);
    TileOut to;
    to.frag = a.gz16 + (size_t)(l - 1) * nt32 * 16 * 512;
    to.rows = nullptr;
    to.ld = 0;
    float novmax = 0.0f;
    h2 vmax2 = {(_Float16)0, (_Float16)0};
    run_tiles<NW, 8>(rg, lds, lag, mma16, [&](auto JP) {
      constexpr int jp = decltype(JP)::value;
      u32x4_t blk[2];
      unsigned int tw;
      tile_epilogue<true, 0, 1, false, false>(acc, jp, un, pe, nullptr, 0.0f, mask_tile<jp>(words), tw, novmax, vmax2, nullptr, blk, li, hh);
      tile_out<NW, 0, jp>(rg, to, blk, Nh, stg, t32, lane);
    });
    if (lane == 0) a.gzexp[(size_t)(l - 1) * nt32 + t32] = eo;
    amax_in = wave_max_rr(pk_hmax(vmax2) * pow2r(-eo));
    track_lds(mx_s, l - 1, amax_in, lane);
    e_in = eo;
#pragma unroll
    for (int s = 0; s < 16; ++s) Bh[s] = Nh[s];
  }
  if (need_dxyz) {
    // ---- d x0 += gz_0 . W_0, then d xyz through the encoding
    dx0_stage(pow2r(-(e_in + wexp[0])));
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
#ifdef UPNERF_STAMPS
  if (lane == 0 && (blockIdx.x & 15) == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&upnerf_stamp_acc_rr[8 + i], rg.st_acc[i]);
    atomicAdd(&upnerf_stamp_acc_rr[12], __builtin_amdgcn_s_memtime() - st_t0);
    atomicAdd(&upnerf_stamp_acc_rr[13], 1ull);
  }
#endif
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
#ifdef UPNERF_STAMPS
extern "C" int upnerf_stamps_read_rr(unsigned long long* out16, int reset) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(upnerf_stamp_acc_rr), 24 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[24] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(upnerf_stamp_acc_rr), z, sizeof(z)));
  }
  return 0;
}
#endif

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
