// 3x3 / stride-1 / pad-1 convolution (forward and data gradient), bf16 / fp16, gfx950 — PING-PONG version of
// conv3x3_halo.hpp: one 512-thread workgroup per CU whose two halves (waves 0-3 / 4-7, i.e. the two waves that share each
// SIMD) alternate between a LOAD phase and a MATRIX phase, half a step apart.
//
// Why: in conv3x3_halo_rw_kernel two INDEPENDENT 4-wave workgroups share a CU and their phases meet by chance — the SQ counters
// put the matrix pipe at 54 % busy, and the LDS-DMA instructions (60-180 cycles of issue each, MI355X_MICROARCH.md) sit inside
// the MFMA stream.  Here the pairing is deterministic (guide, "Two waves per SIMD"): while half A runs its 48 MFMAs of a step
// (768 pipe cycles, nothing else in the stream), half B reads its fragments of the same step from LDS and issues the DMA of
// the stages two steps ahead; a workgroup barrier swaps the roles.
//
//   tile        : 16 x 32 pixels (two 8 x 32 half tiles, one per half) x 64 output channels; the halves share the weight
//                 stages (read once from L2 per 512 pixels instead of per 256) and one 18 x 34 halo patch per 32-channel slab
//                 (re-fetch 1.20x instead of 1.33x of the tile's pixels)
//   LDS         : 3 patch buffers x 39 KiB + 3 weight stages x 12 KiB = 153 KiB -> one workgroup per CU
//   phases      : p = 0, 1, ...: half 0 runs R(s) at p = 2s and M(s) at p = 2s+1, half 1 runs R(s) at p = 2s+1 and M(s) at
//                 p = 2s+2 (R = fragment reads + DMA issue of step s, M = its MFMAs); one s_barrier per phase
//   DMA         : in R(s) a wave issues its share of the weight stage of step s+2 (ring slot (s+2) % 3, last read in R(s-1) of
//                 half 1: one phase earlier) and 2 / 2 / 1 of its five pieces of the patch of the slab AFTER NEXT (buffer
//                 (c+2) % 3, last read during slab c-1): at most four DMA issues per load phase — with all five patch pieces in
//                 the first step of a slab that phase ran 1.5x the matrix phase and the kernel lost a quarter of its rate
//                 (no-DMA ceiling 1885 TFLOP/s on 32x32x1024->512, 1482 with the unbalanced schedule).  Counted vmcnt waits at
//                 the end of every wave's (2s+1)-phase keep exactly the youngest issues in flight: this step's stage and patch
//                 pieces and the previous step's patch pieces (stage pieces are issued first)
//   epilogue    : as conv3x3_halo.hpp (a lane owns four consecutive channels of a pixel; LDS-staged 16-byte row stores), one
//                 C tile per half; half 0 stages its tile while half 1 runs its last matrix phase
// Everything else (patch / slab images, swizzles, fragment maps, tap mirroring for the data gradient, x2 up-sampling in the
// gather, ReLU / statistics / 2x2-sum epilogues) is that kernel's.
#pragma once
#include <type_traits>

#include "common.hpp"

struct HaloPpCfg {
  static constexpr int TH = 16, TW = 32, HTH = 8, BN = 64;
  static constexpr int NPIX = (TH + 2) * (TW + 2);
  static constexpr int P_INSTR = (NPIX + 15) / 16;
  static constexpr int PATCH_BYTES = P_INSTR * 1024;
  static constexpr int STAGE_BYTES = 3 * BN * 64;
  static constexpr int NS = 3;
  static constexpr int NPB = 3;                                          // patch buffers: slab c lives in buffer c % 3
  static constexpr int RING = NPB * PATCH_BYTES + NS * STAGE_BYTES;
  static constexpr int C_BYTES = HTH * TW * (BN * 2 + 16);             // C tile of one half
  static constexpr int EPI_BYTES = 2 * C_BYTES + 2 * 2 * 2 * BN * 4;    // + statistics scratch [half][wm][q][BN]
  static constexpr int LDS_BYTES = RING > EPI_BYTES ? RING : EPI_BYTES;
};

template <typename T>
__global__ __launch_bounds__(512) void conv3x3_halo_pp_kernel(const ConvArgs a) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  typedef HaloPpCfg Cfg;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, HTH = Cfg::HTH;
  constexpr int BN = 64, BK = 32, EPC = 8, HBM = HTH * TW;            // 256 pixels per half
  constexpr int PW = TW + 2, NPIX = Cfg::NPIX;
  constexpr int PIXB = BK * 2;
  constexpr int P_INSTR = Cfg::P_INSTR;
  constexpr int P_IT = (P_INSTR + 7) / 8;            // patch pieces per wave (8 waves)
  constexpr int PATCH_BYTES = Cfg::PATCH_BYTES;
  constexpr int SLAB = BN * PIXB, STAGE = Cfg::STAGE_BYTES;
  constexpr int WN = 2, WTM = HBM / 2, WTN = BN / WN;
  constexpr int RW = WTM / TW, XB = TW / 16;         // 4 tile rows per wave, two 16-pixel blocks per row
  constexpr int MB = RW * XB, NB = WTN / 16;         // 8 x 2 MFMA blocks of 16x16 per wave
  constexpr int NPR = RW + 2;
  constexpr int C_PITCH = BN * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, w4 = wave & 3;
  const int l16 = lane & 15, c4 = lane >> 4;
  const int wm = w4 / WN, wn = w4 % WN;
  const int NT = a.Co / BN, TXN = a.Wo / TW, TYN = a.Ho / TH;
  const int bid = xcd_tile(blockIdx.x, gridDim.x);
  int t = bid;
  const int nt = t % NT; t /= NT;
  const int tx = t % TXN; t /= TXN;
  const int ty = t % TYN;
  const int n = t / TYN;
  const int y0 = ty * TH, x0 = tx * TW, n0 = nt * BN;
  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wk = reinterpret_cast<const T*>(a.wk);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const bool flip = a.kmul < 0;

  // ---- DMA lane geometry (a 1-KiB instruction = 16 patch pixels x 64 B; wave w issues patch pieces w, w+8, ...) ----------
  const int lrow = lane >> 2, slot = lane & 3;
  // per-lane source of every patch piece as a 32-bit BYTE offset inside image n (0xffffffff: padding -> zero page); the image
  // base and the piece's LDS destination are wave-uniform (scalar registers)
  const char* const img = reinterpret_cast<const char*>(in + (size_t)n * a.Hi * a.Wi * a.ldi);
  unsigned p_off[P_IT];
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int piece = min(wave + 8 * i, P_INSTR - 1);
    const int q = piece * 16 + lrow;
    const int py = q / PW, px = q - py * PW;
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    const bool ok = q < NPIX && (unsigned)yy < (unsigned)a.Hlog && (unsigned)xx < (unsigned)a.Wlog;
    p_off[i] = ok ? (unsigned)((((yy >> a.up) * a.Wi + (xx >> a.up)) * a.ldi + (slot ^ (((q >> 2) & 1) << 1)) * EPC) * 2) : 0xffffffffu;
  }
  unsigned char* const patch0 = lds;
  unsigned char* const bring = lds + Cfg::NPB * PATCH_BYTES;
  auto issue_patch_piece = [&](int buf, int c0, int i) __attribute__((always_inline)) {
    unsigned o = p_off[i];
    asm volatile("" : "+v"(o));          // keep the 32-bit offset: a hoisted 64-bit address per piece costs 10 VGPRs (spills)
    const char* p = o != 0xffffffffu ? img + o + c0 * 2 : zero + slot * 16;
    dma16(p, lds_addr(patch0 + buf * PATCH_BYTES + min(wave + 8 * i, P_INSTR - 1) * 1024));
  };
  // weight stage = the three taps (ph = 0, 1, 2) of patch column pw, 4 KiB each = four 1-KiB pieces of 16 rows.  Half 0's wave
  // w4 brings rows [16 w4, +16) of slabs 0 and 2, half 1's wave w4 the same rows of slab 1.
  const size_t wrow = (size_t)9 * a.Ci;
  const int brow = w4 * 16 + lrow;
  const T* const b_src = wk + (size_t)(n0 + brow) * wrow + (slot ^ (((brow >> 2) & 1) << 1)) * EPC;
  auto issue_stage_piece = [&](int stage, int pw, int c0, int ph) __attribute__((always_inline)) {      // c0 < 0: nothing left to fetch (zeros into a dead slot)
    const int tap = flip ? (2 - ph) * 3 + (2 - pw) : ph * 3 + pw;
    const char* p = c0 >= 0 ? reinterpret_cast<const char*>(b_src + (size_t)tap * a.Ci + c0) : zero + slot * 16;
    dma16(p, lds_addr(bring + stage * STAGE + w4 * 1024 + ph * SLAB));
  };
  auto issue_stage = [&](auto half_tag, int stage, int pw, int c0) __attribute__((always_inline)) {
    if constexpr (decltype(half_tag)::value == 0) {
      issue_stage_piece(stage, pw, c0, 0);
      issue_stage_piece(stage, pw, c0, 2);
    } else {
      issue_stage_piece(stage, pw, c0, 1);
    }
  };

  const int nC = a.Ci / BK;

  // ---- fragment geometry -----------------------------------------------------------------------------------------------
  const int q00 = (half * HTH + wm * RW) * PW + l16;
  const int brow0 = wn * WTN + l16;
  const int boff0 = brow0 * PIXB + ((c4 ^ (((brow0 >> 2) & 1) << 1)) << 4);
  // (the 4-wave kernel keeps the 18 fragment offsets [patch column][patch row] in registers; here the R phase has VALU slots to
  // spare and the registers do not: the offsets are rebuilt from q00 — five VALU operations per patch row and phase)
  f32x4 acc[MB][NB];
  f32x4 bias4[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    bias4[nb] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n0 + wn * WTN + nb * 16 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: patch of slab 0, the stages of steps 0 and 1 (stage index == patch column), then the patch of slab 1 (slab c + 2 is
  // fetched during slab c, two or one pieces per wave and step: no load phase carries more than four DMA issues)
  constexpr int NP0 = 2, NP1 = 2, NP2 = P_IT - 4;       // patch pieces a wave issues in the steps pw = 0, 1, 2 of a slab
  static_assert(P_IT == 5, "piece schedule below is written for five patch pieces per wave and slab");
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using No = std::false_type;
  using Yes = std::true_type;
#pragma unroll
  for (int i = 0; i < P_IT; ++i) issue_patch_piece(0, 0, i);
  if (half == 0) {
    issue_stage(I0{}, 0, 0, 0);
    issue_stage(I0{}, 1, 1, 0);
  } else {
    issue_stage(I1{}, 0, 0, 0);
    issue_stage(I1{}, 1, 1, 0);
  }
#pragma unroll
  for (int i = 0; i < P_IT; ++i) issue_patch_piece(1, nC > 1 ? BK : 0, i);       // (one slab only: a dead buffer)
  if (half == 0) wait_vmcnt<2 + P_IT>(); else wait_vmcnt<1 + P_IT>();            // all but stage 1 and the patch of slab 1
  __builtin_amdgcn_s_barrier();

  bf16x8 bfr[3][NB];
  bf16x8 afr[NPR][XB];
  // R(step): fragment reads of (slab chunk, patch column pw) + this wave's DMA issues: stage of step + 2, then (pw == 0) the next
  // slab's patch.
  auto phase_r = [&](auto half_tag, int chunk, auto pw_tag, auto pb_tag) __attribute__((always_inline)) {
    constexpr int pw = decltype(pw_tag)::value;
    constexpr int pb = decltype(pb_tag)::value;        // chunk % 3: the patch buffer is a compile-time LDS offset
    constexpr int pw2 = (pw + 2) % 3;
    const int c2 = (pw == 0 ? chunk : chunk + 1);
    const int c0_stage = c2 < nC ? c2 * BK : -1;
    const unsigned char* pa = patch0 + pb * PATCH_BYTES;
    const unsigned char* pb_ = bring + pw * STAGE;
#pragma unroll
    for (int ph = 0; ph < 3; ++ph)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bfr[ph][nb] = *reinterpret_cast<const bf16x8*>(pb_ + ph * SLAB + nb * 16 * PIXB + boff0);
    int qb = q00;
    asm volatile("" : "+v"(qb));                     // (opaque: the offsets must not be hoisted back into 18 loop-invariant registers)
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr) {
      const int q = qb + pr * PW + pw;
      const int ao = q * PIXB + ((c4 ^ (((q >> 2) & 1) << 1)) << 4);
#pragma unroll
      for (int xb = 0; xb < XB; ++xb) afr[pr][xb] = *reinterpret_cast<const bf16x8*>(pa + xb * 16 * PIXB + ao);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_stage(half_tag, pw2, pw2, c0_stage);
    // this step's share of the patch of slab chunk + 2 (buffer (pb + 2) % 3: last read during slab chunk - 1)
    constexpr int i0 = pw == 0 ? 0 : (pw == 1 ? NP0 : NP0 + NP1), i1 = pw == 0 ? NP0 : (pw == 1 ? NP0 + NP1 : P_IT);
    const int cp = chunk + 2 < nC ? (chunk + 2) * BK : chunk * BK;        // (past the last slab: into a dead buffer)
#pragma unroll
    for (int i = i0; i < i1; ++i) issue_patch_piece((pb + 2) % 3, cp, i);
    __builtin_amdgcn_sched_barrier(0);
  };
  // M: the 48 MFMAs of the step whose fragments sit in afr / bfr — nothing else in the stream
  auto phase_m = [&](auto first_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr) {
#pragma unroll
      for (int xb = 0; xb < XB; ++xb)
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
          const int orow = pr - ph;
          if (orow >= 0 && orow < RW) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[orow * XB + xb][nb] = mfma_16x16x32<T>(bfr[ph][nb], afr[pr][xb],
                                                         (FIRST && ph == 0) ? bias4[nb] : acc[orow * XB + xb][nb]);
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // end of a wave's (2s+1)-phase: the stage issued one step earlier has landed.  In flight may stay (issue order: stage pieces,
  // then patch pieces): the previous step's patch pieces, this step's stage and patch pieces
  auto wait_w = [&](auto half_tag, auto pw_tag, auto first_tag) __attribute__((always_inline)) {
    constexpr int pw = decltype(pw_tag)::value;
    constexpr int nS = decltype(half_tag)::value == 0 ? 2 : 1;
    constexpr int np = pw == 0 ? NP0 : (pw == 1 ? NP1 : NP2), np_prev = pw == 0 ? NP2 : (pw == 1 ? NP0 : NP1);
    if constexpr (decltype(first_tag)::value) wait_vmcnt<nS + np + P_IT>();        // step 0: behind stage 1 sits the prologue's patch 1
    else wait_vmcnt<nS + np + np_prev>();
  };
  // one iteration = the two phases 2s+1 and 2s+2 of step s = (chunk, pw):   half 0: M(s) | R(s+1)      half 1: R(s) | M(s)
  auto iter = [&](auto half_tag, int chunk, auto pw_tag, auto pb_tag, auto first_tag) __attribute__((always_inline)) {
    constexpr int H = decltype(half_tag)::value;
    constexpr int pw = decltype(pw_tag)::value;
    constexpr int pb = decltype(pb_tag)::value;
    constexpr int pwn = (pw + 1) % 3;
    constexpr int pbn = pw == 2 ? (pb + 1) % 3 : pb;
    const int chunkn = pw == 2 ? chunk + 1 : chunk;
    const bool last = pw == 2 && chunk + 1 >= nC;
    if constexpr (H == 0) {
      phase_m(first_tag);
      if (last) wait_vmcnt<0>(); else wait_w(half_tag, pw_tag, first_tag);
      __builtin_amdgcn_s_barrier();
      if (!last) {
        phase_r(half_tag, chunkn, std::integral_constant<int, pwn>{}, std::integral_constant<int, pbn>{});
        __builtin_amdgcn_s_barrier();
      }
    } else {
      phase_r(half_tag, chunk, pw_tag, pb_tag);
      if (last) {
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // half 0 stages its C tile over the ring right after this barrier
      } else {
        wait_w(half_tag, pw_tag, first_tag);
      }
      __builtin_amdgcn_s_barrier();
      phase_m(first_tag);
      if (!last) __builtin_amdgcn_s_barrier();
    }
  };
  // the two halves run two separate instruction streams (same barrier count): roles are compile-time constants in each
  auto run = [&](auto half_tag) __attribute__((always_inline)) {
    constexpr int H = decltype(half_tag)::value;
    if constexpr (H == 0) phase_r(half_tag, 0, I0{}, I0{});          // phase 0: half 0 reads step 0, half 1 waits
    __builtin_amdgcn_s_barrier();
    iter(half_tag, 0, I0{}, I0{}, Yes{});
    for (int chunk = 0;; chunk += 3) {           // (starts at column 1 so that only ONE extra step body exists: the FIRST one)
      iter(half_tag, chunk, I1{}, I0{}, No{});
      iter(half_tag, chunk, I2{}, I0{}, No{});
      if (chunk + 1 >= nC) break;
      iter(half_tag, chunk + 1, I0{}, I1{}, No{});
      iter(half_tag, chunk + 1, I1{}, I1{}, No{});
      iter(half_tag, chunk + 1, I2{}, I1{}, No{});
      if (chunk + 2 >= nC) break;
      iter(half_tag, chunk + 2, I0{}, I2{}, No{});
      iter(half_tag, chunk + 2, I1{}, I2{}, No{});
      iter(half_tag, chunk + 2, I2{}, I2{}, No{});
      if (chunk + 3 >= nC) break;
      iter(half_tag, chunk + 3, I0{}, I0{}, No{});
    }
  };
  if (half == 0) run(I0{}); else run(I1{});
  // here: every DMA of the workgroup has landed (vmcnt(0) before the last barrier every wave passed); half 0 has finished its
  // last matrix phase one phase ago, half 1 is about to run / has just run its own.  The ring is dead: C tiles go on top of it.

  // ---- epilogue ---------------------------------------------------------------------------------------------------------
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  struct alignas(8) Pack4 { T v[4]; };
  unsigned char* const ctile = lds + half * Cfg::C_BYTES;
  float* const red = reinterpret_cast<float*>(lds + 2 * Cfg::C_BYTES);      // [half][wm][2][BN]
  auto finish = [&](auto relu_tag, auto stats_tag) __attribute__((always_inline)) {
    constexpr bool RELU = decltype(relu_tag)::value, STATS = decltype(stats_tag)::value;
    f32x2 sm[NB][2], sq[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 2; ++j) { sm[nb][j] = f32x2{0.f, 0.f}; sq[nb][j] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int row = wm * WTM + mb * 16 + l16;                // pixel of this lane inside its half tile
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        Pack4 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pk.v[r] = from_f32<T>(RELU ? __builtin_amdgcn_fmed3f(acc[mb][nb][r], 0.f, INFINITY) : acc[mb][nb][r]);
        if constexpr (STATS) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x2 v = {to_f32<T>(pk.v[2 * j]), to_f32<T>(pk.v[2 * j + 1])};
            sm[nb][j] += v;
            sq[nb][j] += v * v;
          }
        }
        *reinterpret_cast<Pack4*>(ctile + row * C_PITCH + (wn * WTN + nb * 16 + 4 * c4) * 2) = pk;
      }
    }
    if constexpr (STATS) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = row16_sum(sm[nb][r >> 1][r & 1]), s2 = row16_sum(sq[nb][r >> 1][r & 1]);
          if (l16 == 0) {
            red[((half * 2 + wm) * 2 + 0) * BN + wn * WTN + nb * 16 + 4 * c4 + r] = s1;
            red[((half * 2 + wm) * 2 + 1) * BN + wn * WTN + nb * 16 + 4 * c4 + r] = s2;
          }
        }
    }
  };
  auto finish_any = [&]() __attribute__((always_inline)) {
    if (a.stats) {
      if (a.relu) finish(Yes{}, Yes{}); else finish(No{}, Yes{});
    } else {
      if (a.relu) finish(Yes{}, No{}); else finish(No{}, No{});
    }
  };
  // half 0 stages its tile while half 1 runs its last matrix phase (which iter() left without a trailing barrier)
  if (half == 0) finish_any();
  __builtin_amdgcn_s_barrier();
  if (half == 1) finish_any();
  __syncthreads();
  if (a.stats && tid < 2 * BN) {
    const int q = tid / BN, c = tid - q * BN;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[(w * 2 + q) * BN + c];
    a.stats[((size_t)(bid / NT) * 2 + q) * a.Co + n0 + c] = v;
  }
  constexpr int CPRC = BN / EPC;
  constexpr int BM = TH * TW;                 // 512 tile pixels: tile row r lives in C tile r / 256 at row r % 256
  if (a.pool2) {
    const int Ho2 = a.Ho >> 1, Wo2 = a.Wo >> 1;
    for (int id = tid; id < (BM / 4) * CPRC; id += 512) {
      const int g = id / CPRC, c = id - g * CPRC;
      const int gy = g / (TW / 2), gx = g - gy * (TW / 2);
      float sum[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) sum[e] = 0.f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int row = (2 * gy + dy) * TW + 2 * gx + dx;
          const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(lds + (row / HBM) * Cfg::C_BYTES + (row % HBM) * C_PITCH + c * 16);
#pragma unroll
          for (int e = 0; e < EPC; ++e) sum[e] += to_f32<T>(v.v[e]);
        }
      T* p = out + ((size_t)(n * Ho2 + (y0 >> 1) + gy) * Wo2 + (x0 >> 1) + gx) * a.ldo + n0 + c * EPC;
      Vec16<T> v;
      if (a.accumulate) {
        const Vec16<T> o = ld16<T>(p);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(sum[e] + to_f32<T>(o.v[e]));
      } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(sum[e]);
      }
      st16<T>(p, v);
    }
    return;
  }
  for (int id = tid; id < BM * CPRC; id += 512) {
    const int row = id / CPRC, c = id - row * CPRC;
    const int py = row / TW, px = row - py * TW;
    T* p = out + ((size_t)(n * a.Ho + y0 + py) * a.Wo + x0 + px) * a.ldo + n0 + c * EPC;
    Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(lds + (row / HBM) * Cfg::C_BYTES + (row % HBM) * C_PITCH + c * 16);
    if (a.accumulate) {
      const Vec16<T> o = ld16<T>(p);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) + to_f32<T>(o.v[e]));
    }
    st16<T>(p, v);
  }
}

template <typename T>
static int launch_halo_pp(const ConvArgs& a, hipStream_t s) {
  const int grid = a.N * (a.Ho / HaloPpCfg::TH) * (a.Wo / HaloPpCfg::TW) * (a.Co / 64);
  constexpr int lds_bytes = HaloPpCfg::LDS_BYTES;
  static const hipError_t configured =
      hipFuncSetAttribute((const void*)conv3x3_halo_pp_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (configured != hipSuccess)
    MI355_FAIL((int)configured, "conv3x3_halo_pp: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(configured));
  hipLaunchKernelGGL((conv3x3_halo_pp_kernel<T>), dim3(grid), dim3(512), lds_bytes, s, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
