// 3x3 / stride-1 / pad-1 convolution (forward and data gradient), bf16 / fp16, gfx950 — the 128-CHANNEL ping-pong kernel:
// conv3x3_halo_pp.hpp's two phase-shifted halves (LOAD phase | MATRIX phase, half a step apart) with a 16 x 32-pixel x
// 128-channel workgroup tile, so that a wave owns 128 pixels x 64 channels (128 accumulator registers).
//
// Why: the 64-channel kernels move one operand byte L2 -> LDS per 160-250 FLOP, and that transfer (6-9 TB/s chip-wide), not
// the matrix pipe, bounds them.  Doubling BOTH tile edges halves the bytes per FLOP (weights per 512
// pixels instead of 256, patch per 128 channels instead of 64: 37 KiB per step of 3072 pipe cycles = 12 B/clk/CU at full
// rate against 25) and raises the MFMAs per LDS fragment read from 2.7 to 4 (24 ds_read_b128 per 96 MFMAs).  A matrix phase
// is 96 MFMAs = 1536 pipe cycles, twice the 64-channel kernel's, while a load phase carries 24 fragment reads and 3-6 DMA
// issues — it now fits under the other half's matrix phase.
//
//   LDS         : 2 patch buffers x 41 KiB (36-pixel pitch) + 3 weight stages x 24 KiB = 154 KiB -> one workgroup per CU
//   DMA         : every wave brings rows [16 wave, +16) of the three slabs of the stage two steps ahead (3 pieces per step) and
//                 2 / 2 / 1 of its five pieces of the NEXT slab's patch in the steps pw = 0 / 1 / 2 of a slab (one in R(s, 0), the
//                 other with the stage in R(s, 1): PA / PB below; buffer (c+1) % 2, last read during slab c-1).  The counted wait at
//                 the end of a step leaves in flight exactly what was issued after the youngest piece the next load phase needs:
//                 5 / 6 / 3 instructions for pw = 0 / 1 / 2 (wait_w)
//   registers   : 128 accumulators + 12 + 12 fragments (96) -> launch bound 512 threads = 256 registers per lane
// Selected for Co % 128 == 0 and deep reductions only (one workgroup per CU exposes each tile's prologue and epilogue).
#pragma once
#include <type_traits>

#include "common.hpp"

struct HaloPp128Cfg {
  static constexpr int TH = 16, TW = 32, HTH = 8, BN = 128;
  static constexpr int PWL = 36;                                         // patch row pitch in LDS (pixels): a multiple of four, so that
                                                                         // the swizzle bit of a pixel is (row + (x >> 2)) & 1 — see the
                                                                         // fragment addresses in the kernel; columns 34, 35 are never read
  static constexpr int NPIX = (TH + 2) * PWL;
  static constexpr int P_INSTR = (NPIX + 15) / 16;                       // 41 DMA instructions of 16 pixels per slab
  static constexpr int PATCH_BYTES = P_INSTR * 1024;
  static constexpr int STAGE_BYTES = 3 * BN * 64;
  static constexpr int NS = 3;
  static constexpr int NPB = 2;                                          // patch buffers: slab c lives in buffer c % 2
  static constexpr int RING = NPB * PATCH_BYTES + NS * STAGE_BYTES;
  static constexpr int C_BYTES = HTH * TW * (BN * 2 + 16);             // C tile of one half
  static constexpr int EPI_BYTES = 2 * C_BYTES + 2 * 2 * 2 * BN * 4;    // + statistics scratch [half][wm][q][BN]
  static constexpr int LDS_BYTES = RING > EPI_BYTES ? RING : EPI_BYTES;
};

template <typename T>
__global__ __launch_bounds__(512) void conv3x3_halo_pp128_kernel(const ConvArgs a) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  typedef HaloPp128Cfg Cfg;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, HTH = Cfg::HTH;
  constexpr int BN = Cfg::BN, BK = 32, EPC = 8, HBM = HTH * TW;            // 256 pixels per half
  constexpr int PW = TW + 2, PWL = Cfg::PWL, NPIX = Cfg::NPIX;
  constexpr int PIXB = BK * 2;
  constexpr int P_INSTR = Cfg::P_INSTR;
  constexpr int P_IT = P_INSTR / 8;                  // patch pieces every wave brings (pieces w, w + 8, ...); piece 40 is wave 0's sixth
  static_assert(P_INSTR == 8 * P_IT + 1, "one left-over piece");
  constexpr int PATCH_BYTES = Cfg::PATCH_BYTES;
  constexpr int SLAB = BN * PIXB, STAGE = Cfg::STAGE_BYTES;
  constexpr int WN = 2, WTM = HBM / 2, WTN = BN / WN;
  // a wave owns a 16-pixel COLUMN of its half tile: all 8 rows x one 16-pixel block (10 patch-row fragments serve 8 x 3 row taps;
  // 4 rows x two blocks would need 12) x 64 channels
  constexpr int RW = HTH, XB = 1;
  constexpr int MB = RW * XB, NB = WTN / 16;         // 8 x 4 MFMA blocks of 16x16 per wave
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
  // per-lane source of every patch piece as a 32-bit BYTE offset inside image n (DMA_PAD: padding); the image base and the
  // piece's LDS destination are wave-uniform (scalar registers)
  const char* const img = reinterpret_cast<const char*>(in + (size_t)n * a.Hi * a.Wi * a.ldi);
  unsigned p_off[P_IT + 1];                          // ([P_IT]: the left-over piece, used by wave 0 only)
#pragma unroll
  for (int i = 0; i <= P_IT; ++i) {
    const int piece = i < P_IT ? wave + 8 * i : P_INSTR - 1;
    const int q = piece * 16 + lrow;
    const int py = q / PWL, px = q - py * PWL;
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    const bool ok = q < NPIX && px < PW && (unsigned)yy < (unsigned)a.Hlog && (unsigned)xx < (unsigned)a.Wlog;
    p_off[i] = ok ? (unsigned)((((yy >> a.up) * a.Wi + (xx >> a.up)) * a.ldi + (slot ^ (((q >> 2) & 1) << 1)) * EPC) * 2) : DMA_PAD;
  }
  const unsigned lds0 = lds_addr(lds);               // LDS byte address of the ring (DMA destinations are integers)
  unsigned char* const patch0 = lds;
  unsigned char* const bring = lds + Cfg::NPB * PATCH_BYTES;
  // Patch pieces go through a buffer descriptor on the image (dma.hpp: dma16_buf): a padding lane carries the always-out-of-range
  // offset and the hardware's range check writes its zeros — no zero page, no EXEC masking, no 64-bit address in vector registers.
  const bufdesc_t desc_in = make_buf(img);
  auto issue_patch_piece = [&](int buf, int c0, int i) __attribute__((always_inline)) {
    const int piece = i < P_IT ? wave + 8 * i : P_INSTR - 1;
    dma16_buf(desc_in, p_off[i], (unsigned)c0 * 2u, lds0 + buf * PATCH_BYTES + piece * 1024);
  };
  // weight stage = the three taps (ph = 0, 1, 2) of patch column pw, 8 KiB each = eight 1-KiB pieces of 16 rows: wave w brings
  // rows [16 w, +16) of all three slabs.
  const size_t wrow = (size_t)9 * a.Ci;
  const int brow = wave * 16 + lrow;
  const T* const wk0 = wk + (size_t)n0 * wrow;                                            // wave-uniform
  const unsigned b_off = (unsigned)(((size_t)brow * wrow + (slot ^ (((brow >> 2) & 1) << 1)) * EPC) * 2);
  auto issue_stage_piece = [&](int stage, int pw, int c0, int ph) __attribute__((always_inline)) {
    const int tap = flip ? (2 - ph) * 3 + (2 - pw) : ph * 3 + pw;
    dma16_sv_m0(wk0 + (size_t)tap * a.Ci + c0, b_off, lds0 + Cfg::NPB * PATCH_BYTES + stage * STAGE + wave * 1024 + ph * SLAB);
  };
  auto issue_stage = [&](int stage, int pw, int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) issue_stage_piece(stage, pw, c0, ph);
  };

  const int nC = a.Ci / BK;

  // ---- fragment geometry -----------------------------------------------------------------------------------------------
  // Pixel fragment of (patch row pr, column shift pw): pixel q = (8 half + pr) * 36 + x, x = 16 wm + l16 + pw.  With a pitch
  // that is a multiple of four the swizzle bit (q >> 2) & 1 is (pr + (x >> 2)) & 1: the ten row addresses of a column shift are
  // ONE lane register (even rows; odd rows: the same XOR 32) plus the immediate pr * 2304 — three registers and one XOR per load
  // phase instead of six VALU operations per row (the 34-pixel pitch of the other halo kernels needs those).
  int a_even[3];
#pragma unroll
  for (int pw = 0; pw < 3; ++pw) {
    const int x = wm * 16 + l16 + pw;
    a_even[pw] = (half * HTH * PWL + x) * PIXB + ((c4 ^ (((x >> 2) & 1) << 1)) << 4);
  }
  const int brow0 = wn * WTN + l16;
  const int boff0 = brow0 * PIXB + ((c4 ^ (((brow0 >> 2) & 1) << 1)) << 4);
  // (the first MFMA of a block takes a literal zero as its C operand; the bias joins in the epilogue — 16 registers the main
  // loop does not have to carry)
  f32x4 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: patch of slab 0 and the stages of steps 0 and 1 (stage index == patch column).  Slab c + 1's patch is fetched during
  // slab c: three pieces per wave in its first step, two in the second, none in the third (they must have landed when it ends)
  // Patch pieces of the NEXT slab a wave issues in the sub-steps (pw, h): one in every R(., 0) (which already carries sixteen
  // fragment reads and their address arithmetic), one more in R(0, 1) and R(1, 1) next to the three stage pieces — so that both
  // kinds of load phase stay shorter than the other half's matrix phase (measured: without any DMA in the loop the kernel runs
  // 17-25 % faster; what the DMA costs is its issue time inside the load phases)
  constexpr int PA[3] = {1, 1, 1}, PB[3] = {1, 1, 0};
  static_assert(P_IT == 5, "piece schedule below is written for five patch pieces per wave and slab");
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using No = std::false_type;
  using Yes = std::true_type;
#pragma unroll
  for (int i = 0; i < P_IT; ++i) issue_patch_piece(0, 0, i);
  if (wave == 0) issue_patch_piece(0, 0, P_IT);
  issue_stage(0, 0, 0);
  issue_stage(1, 1, 0);
  wait_vmcnt<3>();                                                                // all but stage 1
  __builtin_amdgcn_s_barrier();

  // A step s = (slab chunk, patch column pw) runs as TWO sub-steps h = 0, 1 over the channel blocks {2h, 2h+1} of the wave's four:
  // the ten pixel fragments are read once (h = 0) and stay in registers for both, the weight fragments come six at a time —
  // 128 accumulators + 40 + 24 fragment registers (with all twelve weight fragments resident the allocator spilled).
  //   R(s, 0): 10 pixel + 6 weight fragment reads, this step's share of the next slab's patch (3 / 2 / 0 DMA pieces)
  //   R(s, 1): 6 weight fragment reads, the three pieces of the stage two steps ahead
  //   M(s, h): 48 MFMAs, nothing else in the stream
  constexpr int NBH = NB / 2;
  bf16x8 bfr[3][NBH];
  bf16x8 afr[NPR][XB];
  auto phase_r = [&](int chunk, auto pw_tag, auto pb_tag, auto h_tag) __attribute__((always_inline)) {
    constexpr int pw = decltype(pw_tag)::value;
    constexpr int pb = decltype(pb_tag)::value;        // chunk % 2: the patch buffer is a compile-time LDS offset
    constexpr int h = decltype(h_tag)::value;
    const unsigned char* pb_ = bring + pw * STAGE;
    const int cp = chunk + 1 < nC ? (chunk + 1) * BK : chunk * BK;        // next slab's patch (past the last slab: into a dead buffer)
#pragma unroll
    for (int ph = 0; ph < 3; ++ph)
#pragma unroll
      for (int nb = 0; nb < NBH; ++nb)
        bfr[ph][nb] = *reinterpret_cast<const bf16x8*>(pb_ + ph * SLAB + (h * NBH + nb) * 16 * PIXB + boff0);
    if constexpr (h == 0) {
      const unsigned char* pa = patch0 + pb * PATCH_BYTES;
      int ae = a_even[pw];
      asm volatile("" : "+v"(ae));                     // (opaque: one base register per parity, the rows are immediates)
      int ao = ae ^ 32;
      asm volatile("" : "+v"(ao));
#pragma unroll
      for (int pr = 0; pr < NPR; ++pr)
        afr[pr][0] = *reinterpret_cast<const bf16x8*>(pa + ((pr & 1) ? ao : ae) + pr * PWL * PIXB);
      static_assert(XB == 1, "one 16-pixel block per wave row");
      __builtin_amdgcn_sched_barrier(0);
      // this sub-step's share of the patch of slab chunk + 1 (buffer pb ^ 1: last read during slab chunk - 1)
      constexpr int i0 = pw == 0 ? 0 : (pw == 1 ? PA[0] + PB[0] : PA[0] + PB[0] + PA[1] + PB[1]), i1 = i0 + PA[pw];
#pragma unroll
      for (int i = i0; i < i1; ++i) {
#ifndef PP128_T_NODMA
        issue_patch_piece(pb ^ 1, cp, i);
#endif
      }
      // wave 0's left-over piece travels with the slab's last regular one: in front of this step's stage, so that the pw == 2 wait
      // (everything but that stage) covers it and no other count changes
      if constexpr (pw == 2) {
        if (wave == 0) issue_patch_piece(pb ^ 1, cp, P_IT);
      }
    } else {
      __builtin_amdgcn_sched_barrier(0);
      // the stage two steps ahead (ring slot of step s - 1, last read in half 1's R(s - 1, 1), several phases ago)
      constexpr int pw2 = (pw + 2) % 3;
      const int c2 = (pw == 0 ? chunk : chunk + 1);
#ifndef PP128_T_NODMA        // timing only: no DMA inside the loop (stale operands)
      issue_stage(pw2, pw2, min(c2, nC - 1) * BK);       // (past the last step: the last slab again, into a dead slot)
      constexpr int i0 = (pw == 0 ? 0 : (pw == 1 ? PA[0] + PB[0] : PA[0] + PB[0] + PA[1] + PB[1])) + PA[pw], i1 = i0 + PB[pw];
#pragma unroll
      for (int i = i0; i < i1; ++i) issue_patch_piece(pb ^ 1, cp, i);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase_m = [&](auto h_tag) __attribute__((always_inline)) {
    constexpr int h = decltype(h_tag)::value;
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
            for (int nb = 0; nb < NBH; ++nb) mfma_16x16x32_acc<T>(bfr[ph][nb], afr[pr][xb], acc[orow * XB + xb][h * NBH + nb]);
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // The one counted wait of a step sits in front of the barrier that ends it (behind M(s, 1) in half 0, behind R(s, 1) in half 1):
  // stage s + 1 — issued in R(s - 1, 1) — and, when the next step opens a slab, that slab's patch (issued before it) have landed;
  // what was issued after them may stay in flight: this step's patch pieces and its stage
  // Issue order of a wave: ... R(s-1, 1): stage s+1 (3), PB[pw_(s-1)] | R(s, 0): PA[pw_s] | R(s, 1): stage s+2 (3), PB[pw_s].
  auto wait_w = [&](auto pw_tag) __attribute__((always_inline)) {
    constexpr int pw = decltype(pw_tag)::value;
    if constexpr (pw == 2) wait_vmcnt<3 + PB[2]>();                       // the next slab's patch (last piece: R(s, 0)) as well
    else wait_vmcnt<PB[(pw + 2) % 3] + PA[pw] + 3 + PB[pw]>();
  };
  // sub-step (s, h): R | M with a barrier behind each, the SAME shape in both halves (half 1 one barrier behind half 0)
  auto sub = [&](auto half_tag, int chunk, auto pw_tag, auto pb_tag, auto h_tag) __attribute__((always_inline)) {
    constexpr int H = decltype(half_tag)::value;
    constexpr int h = decltype(h_tag)::value;
    phase_r(chunk, pw_tag, pb_tag, h_tag);
    if constexpr (H == 1 && h == 1) wait_w(pw_tag);
    __builtin_amdgcn_s_barrier();
    phase_m(h_tag);
    if constexpr (H == 0 && h == 1) wait_w(pw_tag);
    __builtin_amdgcn_s_barrier();
  };
  auto iter = [&](auto half_tag, int chunk, auto pw_tag, auto pb_tag) __attribute__((always_inline)) {
    sub(half_tag, chunk, pw_tag, pb_tag, I0{});
    sub(half_tag, chunk, pw_tag, pb_tag, I1{});
  };
  // the two halves run two separate instruction streams (same barrier count): roles are compile-time constants in each.  One loop
  // body = two slabs (Ci % 64 == 0 is a launch condition), no special first / last step: the accumulators start at zero and the
  // ring drains behind the loop.
  auto run = [&](auto half_tag) __attribute__((always_inline)) {
    constexpr int H = decltype(half_tag)::value;
    if constexpr (H == 1) __builtin_amdgcn_s_barrier();          // phase 0: half 0 reads sub-step 0, half 1 waits
    for (int chunk = 0; chunk < nC; chunk += 2) {
      iter(half_tag, chunk, I0{}, I0{});
      iter(half_tag, chunk, I1{}, I0{});
      iter(half_tag, chunk, I2{}, I0{});
      iter(half_tag, chunk + 1, I0{}, I1{});
      iter(half_tag, chunk + 1, I1{}, I1{});
      iter(half_tag, chunk + 1, I2{}, I1{});
    }
    if constexpr (H == 0) __builtin_amdgcn_s_barrier();          // (half 1's last matrix phase)
  };
  if (half == 0) run(I0{}); else run(I1{});
  wait_vmcnt<0>();                                     // (the zero-fill pieces of the last two steps)
  __builtin_amdgcn_s_barrier();                        // every wave's DMA has landed, every fragment read is done: the ring is dead
  mfma_results_ready();                              // (in-place asm MFMAs: the wait states in front of the epilogue's reads are ours)
  // here: every DMA of the workgroup has landed (vmcnt(0) before the last barrier every wave passed); half 0 has finished its
  // last matrix phase one phase ago, half 1 is about to run / has just run its own.  The ring is dead: C tiles go on top of it.

  // ---- epilogue ---------------------------------------------------------------------------------------------------------
#ifdef PP128_NOEPI
  {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) t += acc[mb][nb];
    reinterpret_cast<f32x4*>(a.out)[tid + 512 * blockIdx.x] = t;
    return;
  }
#endif
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  struct alignas(8) Pack4 { T v[4]; };
  unsigned char* const ctile = lds + half * Cfg::C_BYTES;
  float* const red = reinterpret_cast<float*>(lds + 2 * Cfg::C_BYTES);      // [half][wm][2][BN]
  auto finish = [&](auto relu_tag, auto stats_tag) __attribute__((always_inline)) {
    constexpr bool RELU = decltype(relu_tag)::value, STATS = decltype(stats_tag)::value;
    f32x2 sm[NB][2], sq[NB][2];
    f32x4 bias4[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      bias4[nb] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n0 + wn * WTN + nb * 16 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) { sm[nb][j] = f32x2{0.f, 0.f}; sq[nb][j] = f32x2{0.f, 0.f}; }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int row = mb * TW + wm * 16 + l16;                 // pixel of this lane inside its half tile (mb = tile row)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        Pack4 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[mb][nb][r] + bias4[nb][r];
          pk.v[r] = from_f32<T>(RELU ? __builtin_amdgcn_fmed3f(v, 0.f, INFINITY) : v);
        }
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
  finish_any();
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
#ifdef PP128_ACC_BATCH      // (A/B: all old values of the accumulate epilogue before the first store, as in the four-wave kernel)
  constexpr int NST = BM * CPRC / 512;
  static_assert(NST * 512 == BM * CPRC, "whole store rounds");
  if (a.accumulate) {
    Vec16<T> old[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int id = tid + it * 512, row = id / CPRC, c = id - row * CPRC;
      const int py = row / TW, px = row - py * TW;
      old[it] = ld16<T>(out + ((size_t)(n * a.Ho + y0 + py) * a.Wo + x0 + px) * a.ldo + n0 + c * EPC);
    }
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int id = tid + it * 512, row = id / CPRC, c = id - row * CPRC;
      const int py = row / TW, px = row - py * TW;
      T* p = out + ((size_t)(n * a.Ho + y0 + py) * a.Wo + x0 + px) * a.ldo + n0 + c * EPC;
      Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(lds + (row / HBM) * Cfg::C_BYTES + (row % HBM) * C_PITCH + c * 16);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) + to_f32<T>(old[it].v[e]));
      st16<T>(p, v);
    }
    return;
  }
#endif
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
static int launch_halo_pp128(const ConvArgs& a, hipStream_t s) {
  const int grid = a.N * (a.Ho / HaloPp128Cfg::TH) * (a.Wo / HaloPp128Cfg::TW) * (a.Co / HaloPp128Cfg::BN);
  constexpr int lds_bytes = HaloPp128Cfg::LDS_BYTES;
  static const hipError_t configured =
      hipFuncSetAttribute((const void*)conv3x3_halo_pp128_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (configured != hipSuccess)
    MI355_FAIL((int)configured, "conv3x3_halo_pp128: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(configured));
  hipLaunchKernelGGL((conv3x3_halo_pp128_kernel<T>), dim3(grid), dim3(512), lds_bytes, s, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
