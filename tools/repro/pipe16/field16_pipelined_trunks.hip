// Round 3's 128-sample, eight-wave, software-pipelined trunks of the f16x3 / f16 field kernels: parity-green,
// slower than two 64-sample workgroups per CU (3.2 / 2.6 ms against 2.48 / 2.47), never the default.  Left the library in round 4
// (VERDICT r3 item 8); kept here with pipe16.cuh as the measured basis of DESIGN 4.6.  Not compiled: the two functions were
// called from field16_fwd_kernel / field16_bwd_kernel<NP, 128, 8> (git history: round 3, csrc/field16.hip).
#include "pipe16.cuh"

// ---- pipelined trunk of the forward pass, 128-sample tile (csrc/pipe16.cuh) -----------------------------------------------
// Runs trunk layers 0 .. D-1 for the tile in the planes (layer 0 in lockstep: its K is 64), leaves h_{D-1} in the planes
// with one exponent per row half (ehalf), and has stored what the backward pass needs: h_0 .. h_{D-1} (fp32 rows or fp16
// rows + exponents), the ReLU sign bits, the running maxima.  Returns D (no trunk layer left for the caller).
template <int NP, int W, int NW>
__device__ __forceinline__ int fwd_trunk_pipelined(const upnerf_layout& L, const upnerf_field_fwd_args& a, char* Ph, char* Pl,
                                                   float* smax, float* smaxb, unsigned int* mx_s, const int* loff_s,
                                                   const float* bias_s, int (&ehalf)[2], int m0, int M, int tid) {
  constexpr int TILE = F16_TILE_BIG, THREADS = 64 * NW;
  static_assert(NW == 8 && W == 256, "one 32-column n-tile per wave");
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, hh = lane >> 5;
  const int D = L.D, n0 = 32 * wave;
  const char* __restrict__ P16 = (const char*)a.P16;
  const int* __restrict__ wexp = a.wexp;
  const int rbyteA = li * (W * 2), rsw = li & 15;
  const bool train = a.h != nullptr || a.h16 != nullptr;
  const int n64 = (M + 63) >> 6;
  const size_t hm_stride32 = (size_t)gridDim.x * THREADS * 2;

  // ---- the encoding planes as operand fragments in global memory (skip connection, read back at layer L.skip):
  // x0f[tile][t = 0..3][mt = 0..3][plane][lane] x 16 bytes
  char* x0f = a.x0f ? (char*)a.x0f + (size_t)blockIdx.x * (UPNERF_X0 / 16 * 4 * 2 * 1024) : nullptr;
  if (x0f && L.skip > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = wave * 4 + j, p = f & 1, mt = (f >> 1) & 3, t = f >> 3;
      if (NP == 2 || p == 0) {
        const int o = (32 * mt + li) * (W * 2) + (((2 * t + hh) ^ rsw) << 4);
        *(h8*)(x0f + (size_t)f * 1024 + lane * 16) = *(const h8*)((p ? Pl : Ph) + o);
      }
    }
  }

  f32x16 acc[4];
  int eA = ehalf[0], eB = ehalf[1];
  const int e_x0 = eA;
  STAMP_DECL;
  // weight fragment base of layer l for this wave's n-tile, first k-block kb0
  // (every helper takes the lane id as an argument: inside the layer loop it is rebuilt per stage, see there)
  auto wbase = [&](int l, int kp16, int kb0, int lane) -> const char* {
    const int wl = __builtin_amdgcn_readfirstlane(loff_s[l]);
    return P16 + 4 * (size_t)wl + ((size_t)wave * kp16 + kb0) * 2048 + lane * 16;
  };
  // max |bias| over this wave's 32 columns of layer l
  auto bias_absmax = [&](int l, int lane) { return wave_max_nn(fabsf(bias_s[l * W + n0 + (lane & 31)])); };
  // bound of a half's outputs from the maximum of its accumulators, shared through LDS at the next barrier
  auto publish = [&](float* slot, float accmax, int e_in, int wel, float bmax, int lane) {
    if (lane == 0) slot[wave] = fmaf(accmax, pow2f(-(e_in + wel)), bmax);
  };
  auto finish_half = [&](int l, int half, unsigned int bits, float vmax, int e_out, int lane) {
    if (a.hmask)
      NT_STORE(&((unsigned int*)a.hmask)[(size_t)l * hm_stride32 + ((size_t)blockIdx.x * THREADS + wave * 64 + lane) * 2 + half], bits);
    track_wave(mx_s, l, ldexpf(wave_max_nn(vmax), -e_out), lane);
  };

  // ---- layer 0 (K = 64) in lockstep; its epilogue of half B already rides in layer 1's phase 1
  int wel_prev, eB_in_prev;
  {
    const char* wp = wbase(0, UPNERF_X0 / 16, 0, lane);
    h8 wh[4], wl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) pl_ldw<NP>(wh[t], wl[t], wp, t);
    pl_zero<0>(acc);
    pl_zero<1>(acc);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      XFrag<NP> xa, xb;
      pl_ldx<NP, W>(xa, Ph, Pl, rbyteA, rsw, t, hh);
      pl_ldx<NP, W>(xb, Ph, Pl, rbyteA + 64 * W * 2, rsw, t, hh);
      pl_mma<NP, 0>(acc, xa, wh[t], wl[t]);
      pl_mma<NP, 1>(acc, xb, wh[t], wl[t]);
    }
    const int wel = __builtin_amdgcn_readfirstlane(wexp[0]);
    const float bmax = bias_absmax(0, lane);
    publish(smax, pl_absmax<0>(acc), eA, wel, bmax, lane);
    publish(smaxb, pl_absmax<1>(acc), eB, wel, bmax, lane);
    __syncthreads();  // every wave has read the encoding planes; the bounds are in LDS; the x0f stores have completed
    const int eAo = bound_exp(wg_max<NW>(smax)), eBo = bound_exp(wg_max<NW>(smaxb));
    unsigned int bits = 0u;
    float vmax = 0.0f;
    const float s = pow2f(-(eA + wel)), pe = pow2f(eAo);
#pragma unroll
    for (int c = 0; c < 8; ++c) pl_epi_quad<NP, W, 0, true>(acc, c, bias_s, s, pe, Ph, Pl, rbyteA, rsw, n0, hh, bits, vmax);
    finish_half(0, 0, bits, vmax, eAo, lane);
    wel_prev = wel;
    eB_in_prev = eB;
    eA = eAo;
    eB = eBo;  // exponent the planes of B will carry once layer 0's epilogue of B has run (phase 1 of layer 1)
    pl_barrier();  // planes A hold h_0
  }

  STAMP(6);  // layer 0
  // ---- layers 1 .. D-1, software-pipelined
  WPre<NP> wpre;
  if (D > 1) {
    const bool sk = 1 == L.skip;
    pl_preload<NP>(wpre, wbase(1, sk ? (UPNERF_X0 + W) / 16 : W / 16, sk ? UPNERF_X0 / 16 : 0, lane));
  }

  unsigned int bitsB = 0u;
  float vmaxB = 0.0f;
#pragma unroll 1
  for (int l = 1; l < D; ++l) {
    // Every per-lane constant of the stage is rebuilt here from v_mbcnt (two instructions, no input register): values that
    // merely pass through the loop would be spilled in front of it and reloaded inside -- and a scratch reload is a vector
    // memory load whose wait also waits for every older activation store of the wave.
    int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane));
    const int li = lane & 31, hh = lane >> 5, rbyteA = li * (W * 2), rsw = li & 15;
    const bool sk = l == L.skip, skn = l + 1 == L.skip;
    const char* wp = wbase(l, sk ? (UPNERF_X0 + W) / 16 : W / 16, sk ? UPNERF_X0 / 16 : 0, lane);
    const char* wpn = l + 1 < D ? wbase(l + 1, skn ? (UPNERF_X0 + W) / 16 : W / 16, skn ? UPNERF_X0 / 16 : 0, lane) : nullptr;
    const int wel = __builtin_amdgcn_readfirstlane(wexp[l]);
    const int eA_in = eA, eB_in = eB;  // exponents of h_{l-1} in the planes (B: after phase 1)
    // the encoding part of the skip layer: 4 k-blocks from the fragment copy in global memory, then the accumulators move
    // to the exponent of the planes this half is about to read
    auto skip_part = [&](auto half_c, int e_in) {
      constexpr int H = decltype(half_c)::value;
      const char* wx = wbase(l, (UPNERF_X0 + W) / 16, 0, lane);
#pragma unroll
      for (int t = 0; t < UPNERF_X0 / 16; ++t) {
        h8 wh, wl;
        pl_ldw<NP>(wh, wl, wx, t);
        XFrag<NP> x;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const char* f = x0f + (size_t)(((t * 4 + 2 * H + mt) * 2) * 1024) + lane * 16;
          x.h[mt] = *(const h8*)f;
          if constexpr (NP == 2) x.l[mt] = *(const h8*)(f + 1024);
        }
        pl_mma<NP, H>(acc, x, wh, wl);
      }
      pl_scale<H>(acc, pow2f(e_in - e_x0));
    };
    // destinations of h_{l-1}: written from the planes in 16-row pieces of whole lines, four pieces per row half and stage
    auto piece = [&](int half, int j, float un) {
      const int r0 = m0 + 64 * half;
#ifdef PL_EXP_STOREL2  // timing experiment: every store lands in one L2-resident megabyte per XCD (no HBM write stream)
      float* b32 = a.h ? a.h + (size_t)(r0 % (64 * TILE)) * W : nullptr;
      uint16_t* b16 = nullptr;
#else
      float* b32 = (a.h && !a.h16) ? a.h + ((size_t)(l - 1) * M + r0) * W : nullptr;
      uint16_t* b16 = a.h16 ? a.h16 + ((size_t)(l - 1) * M + r0) * W : nullptr;
#endif
      pl_store_piece<NP, W>(Ph, Pl, 64 * half, j, lane, wave, b32, b16, M - r0, un);
    };
    if (a.h16 && wave == 0 && lane == 0) {
      a.hexp[(size_t)(l - 1) * n64 + (m0 >> 6)] = eA_in;
      if (m0 + 64 < M) a.hexp[(size_t)(l - 1) * n64 + (m0 >> 6) + 1] = eB_in;
    }
    const float unA = pow2f(-eA_in), unB = pow2f(-eB_in);
    // epilogue state
    const float sB = pow2f(-(eB_in_prev + wel_prev)), peB = pow2f(eB_in);  // half B of layer l-1 -> planes B at exponent eB_in
    const float* bias_prev = bias_s + (l - 1) * W;
    const float* bias_cur = bias_s + l * W;
    unsigned int bitsA = 0u;
    float vmaxA = 0.0f, sA = 0.0f, peA = 0.0f;
    int eAo = 0, eBo = 0;
    pl_zero<0>(acc);
    if (sk) skip_part(std::integral_constant<int, 0>{}, eA_in);
    pl_stage<NP, W>(
        acc, Ph, Pl, rbyteA, rsw, hh, wp, wpre, wpn,
        /* epiB */ [&](int c) { pl_epi_quad<NP, W, 1, true>(acc, c, bias_prev, sB, peB, Ph, Pl, rbyteA + 64 * W * 2, rsw, n0, hh, bitsB, vmaxB); },
        /* epiA */ [&](int c) { pl_epi_quad<NP, W, 0, true>(acc, c, bias_cur, sA, peA, Ph, Pl, rbyteA, rsw, n0, hh, bitsA, vmaxA); },
        /* pieceA */ [&](int j) { if (train) piece(0, j, unA); },
        /* pieceB */ [&](int j) { if (train) piece(1, j, unB); },
        /* preB */ [&]() {
          pl_zero<1>(acc);
          if (sk) skip_part(std::integral_constant<int, 1>{}, eB_in);
        },
        /* endP1 */ [&]() {
          STAMP_C(0);
          finish_half(l - 1, 1, bitsB, vmaxB, eB_in, lane);
          bitsB = 0u;
          vmaxB = 0.0f;
          pl_barrier();  // planes B hold h_{l-1}
          STAMP_C(1);
        },
        /* endP2 */ [&]() {
          STAMP_C(2);
          publish(smax, pl_absmax<0>(acc), eA_in, wel, bias_absmax(l, lane), lane);
          pl_barrier();  // every wave is done reading planes A; the bound of A is in LDS
          STAMP_C(3);
          eAo = bound_exp(wg_max<NW>(smax));
          sA = pow2f(-(eA_in + wel));
          peA = pow2f(eAo);
        },
        /* endP3 */ [&]() {
          STAMP_C(4);
          finish_half(l, 0, bitsA, vmaxA, eAo, lane);
          publish(smaxb, pl_absmax<1>(acc), eB_in, wel, bias_absmax(l, lane), lane);
          pl_barrier();  // planes A hold h_l; every wave is done reading planes B; the bound of B is in LDS
          STAMP_C(5);
          eBo = bound_exp(wg_max<NW>(smaxb));
        });
    wel_prev = wel;
    eB_in_prev = eB_in;
    eA = eAo;
    eB = eBo;
  }
  // ---- drain: epilogue of half B of the last layer
  {
    const float sB = pow2f(-(eB_in_prev + wel_prev)), peB = pow2f(eB);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      pl_epi_quad<NP, W, 1, true>(acc, c, bias_s + (D - 1) * W, sB, peB, Ph, Pl, rbyteA + 64 * W * 2, rsw, n0, hh, bitsB, vmaxB);
    finish_half(D - 1, 1, bitsB, vmaxB, eB, lane);
    pl_barrier();
  }
  ehalf[0] = eA;
  ehalf[1] = eB;
  STAMP(7);  // drain
  {
    const int lane = tid & 63;
    (void)lane;
    STAMP_FLUSH;
  }
  // h_{D-1} has no later K loop of this function to ride in: planes -> global, whole rows
  if (train) {
    if (a.h16) tile_copy16<W, TILE, THREADS>(Ph, eA, eB, a.h16 + (size_t)(D - 1) * M * W, a.hexp + (size_t)(D - 1) * n64, m0, M, tid);
  }
  return D;
}


// ---- pipelined trunk of the backward pass, 128-sample tile (csrc/pipe16.cuh) ----------------------------------------------
// In: the planes hold gz_{D-1} with the exponents ehalf.  Runs the data-gradient stages
// l = D-1 .. 1 (gz_{l-1} = mask_{l-1} (gz_l . W_l)), stores gz_{D-1} .. gz_0 (fp32 rows or fp16 rows + exponents) and the
// running maxima, parks the skip layer's encoding term (d x0 += gz_skip . W_skip,x) in `xs` (this tile's 32 KiB of the
// forward pass's x0f scratch, in accumulator order), and leaves gz_0 in the planes with the exponents ehalf.
template <int NP, int W, int NW>
__device__ __forceinline__ void bwd_trunk_pipelined(const upnerf_layout& L, const upnerf_field_bwd_args& a, char* Ph, char* Pl,
                                                    float* smax, float* smaxb, unsigned int* mx_s, const int* loff_s,
                                                    int (&ehalf)[2], int m0, int M, int tid) {
  constexpr int TILE = F16_TILE_BIG, THREADS = 64 * NW;
  static_assert(NW == 8 && W == 256, "one 32-column n-tile per wave");
  using TX = WaveTile16<UPNERF_X0, TILE, NW>;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = L.D, n0 = 32 * wave;
  const char* __restrict__ PT16 = (const char*)a.PT16;
  const int* __restrict__ wexp = a.wexp;
  const int n64 = (M + 63) >> 6;
  const size_t hm_stride32 = (size_t)gridDim.x * THREADS * 2;
  const bool store32 = a.gz_h != nullptr, store16 = a.gz16 != nullptr;
  f32x16 acc[4];
  int eA = ehalf[0], eB = ehalf[1];
  auto wbase = [&](int l, int lane) -> const char* {
    const int wl = __builtin_amdgcn_readfirstlane(loff_s[l]);
    return PT16 + 4 * (size_t)wl + (size_t)wave * (W / 16) * 2048 + lane * 16;
  };
  auto mask32 = [&](int l, int half, int lane) -> unsigned int {
    return NT_LOAD(&((const unsigned int*)a.hmask)[(size_t)l * hm_stride32 + ((size_t)blockIdx.x * THREADS + wave * 64 + lane) * 2 + half]);
  };
  auto publish = [&](float* slot, float accmax, int e_in, int wel, int lane) {
    if (lane == 0) slot[wave] = accmax * pow2f(-(e_in + wel));
  };
  auto finish_half = [&](int l, float vmax, int e_out, int lane) {  // l: index of the gradient just written (gz_l)
    track_wave(mx_s, l, ldexpf(wave_max_nn(vmax), -e_out), lane);
  };
  WPre<NP> wpre;
  if (D > 1) pl_preload<NP>(wpre, wbase(D - 1, tid & 63));

  float vmaxB = 0.0f, spB = 0.0f;
  unsigned int bitsB = 0u;
  bool pendB = false;  // half B of the previous stage still has its epilogue to run
#pragma unroll 1
  for (int l = D - 1; l >= 1; --l) {
    int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));  // per-stage rebuild, see the forward pass
    asm volatile("" : "+v"(lane));
    const int li = lane & 31, hh = lane >> 5, rbyteA = li * (W * 2), rsw = li & 15;
    const char* wp = wbase(l, lane);
    const char* wpn = l > 1 ? wbase(l - 1, lane) : nullptr;
    const int wel = __builtin_amdgcn_readfirstlane(wexp[l]);
    const int eA_in = eA, eB_in = eB;  // exponents of gz_l in the planes (B: once phase 1 has written it)
    const unsigned int bitsA = mask32(l - 1, 0, lane);
    const unsigned int bitsBn = mask32(l - 1, 1, lane);
    // gz_l rides out in this stage's K loop
    const bool st = store32 || (store16 && l < D - 1);  // fp16 rows of gz_{D-1}: copied out by the caller
    auto piece = [&](int half, int j, float un) {
      const int r0 = m0 + 64 * half;
      float* b32 = store16 ? nullptr : a.gz_h + ((size_t)l * M + r0) * W;
      uint16_t* b16 = store16 ? a.gz16 + ((size_t)l * M + r0) * W : nullptr;
      pl_store_piece<NP, W>(Ph, Pl, 64 * half, j, lane, wave, b32, b16, M - r0, un);
    };
    if (st && store16 && wave == 0 && lane == 0) {
      a.gzexp[(size_t)l * n64 + (m0 >> 6)] = eA_in;
      if (m0 + 64 < M) a.gzexp[(size_t)l * n64 + (m0 >> 6) + 1] = eB_in;
    }
    const float unA = pow2f(-eA_in), unB = pow2f(-eB_in);
    float vmaxA = 0.0f, spA = 0.0f;
    int eAo = 0, eBo = 0;
    pl_zero<0>(acc);
    pl_stage<NP, W>(
        acc, Ph, Pl, rbyteA, rsw, hh, wp, wpre, wpn,
        /* epiB */ [&](int c) { if (pendB) pl_epi_bwd_quad<NP, W, 1>(acc, c, spB, bitsB, Ph, Pl, rbyteA + 64 * W * 2, rsw, n0, hh, vmaxB); },
        /* epiA */ [&](int c) { pl_epi_bwd_quad<NP, W, 0>(acc, c, spA, bitsA, Ph, Pl, rbyteA, rsw, n0, hh, vmaxA); },
        /* pieceA */ [&](int j) { if (st) piece(0, j, unA); },
        /* pieceB */ [&](int j) { if (st) piece(1, j, unB); },
        /* preB */ [&]() { pl_zero<1>(acc); },
        /* endP1 */ [&]() {
          if (pendB) finish_half(l, vmaxB, eB_in, lane);
          vmaxB = 0.0f;
          pl_barrier();  // planes B hold gz_l
          // the encoding term of the skip layer: both row halves of gz_skip are in the planes during this phase only
          if (a.need_dxyz && l == L.skip) {
            f32x16 accx[TX::MT][TX::NT];
            acc_zero(accx);
            const int xn0 = TX::n0(wave), xrow0 = TX::row0(wave);
            mma16_lds<NP, W, W / 16, 1>(accx, Ph, Pl, xrow0, 0, PT16 + 4 * (size_t)L.t_skipx, W / 16, xn0, 0, lane);
            const float un = pow2f(-((xrow0 >= TILE / 2 ? eB_in : eA_in) + wel));
            float* xs = (float*)((char*)a.xs + (size_t)blockIdx.x * (UPNERF_X0 / 16 * 4 * 2 * 1024)) + (size_t)wave * 1024 + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *(f32x4*)(xs + q * 256) = f32x4{accx[0][0][4 * q] * un, accx[0][0][4 * q + 1] * un, accx[0][0][4 * q + 2] * un, accx[0][0][4 * q + 3] * un};
          }
        },
        /* endP2 */ [&]() {
          publish(smax, pl_absmax<0>(acc), eA_in, wel, lane);
          pl_barrier();  // every wave is done reading planes A; the bound of A is in LDS
          eAo = bound_exp(wg_max<NW>(smax));
          spA = pow2f(eAo - eA_in - wel);
        },
        /* endP3 */ [&]() {
          finish_half(l - 1, vmaxA, eAo, lane);
          publish(smaxb, pl_absmax<1>(acc), eB_in, wel, lane);
          pl_barrier();  // planes A hold gz_{l-1}; every wave is done reading planes B; the bound of B is in LDS
          eBo = bound_exp(wg_max<NW>(smaxb));
        });
    spB = pow2f(eBo - eB_in - wel);
    bitsB = bitsBn;
    pendB = true;
    eA = eAo;
    eB = eBo;
  }
  // ---- drain: epilogue of half B of the last stage
  if (pendB) {
    const int lane = tid & 63, li = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      pl_epi_bwd_quad<NP, W, 1>(acc, c, spB, bitsB, Ph, Pl, li * (W * 2) + 64 * W * 2, li & 15, n0, hh, vmaxB);
    finish_half(0, vmaxB, eB, lane);
    pl_barrier();
  }
  ehalf[0] = eA;
  ehalf[1] = eB;
  if (D > 1) {  // gz_0 has no later K loop of this function to ride in
    if (store16) tile_copy16<W, TILE, THREADS>(Ph, eA, eB, a.gz16, a.gzexp, m0, M, tid);
    else if (store32) tile_store16<NP, W, TILE, THREADS, W>(Ph, Pl, 0, pow2f(-eA), pow2f(-eB), a.gz_h, W, m0, M, tid);
  }
}

