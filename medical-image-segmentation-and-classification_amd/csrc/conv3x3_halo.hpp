// 3x3 / stride-1 / pad-1 convolution (forward and data gradient) for bf16 / fp16 on gfx950:
// implicit GEMM with a HALO-PATCH A operand and a patch-row register window.
//
// A workgroup owns a TH x TW spatial tile (256 pixels) of one image times 64 output channels.
// For every 32-channel slab of Ci the (TH+2) x (TW+2) input patch is streamed into LDS ONCE by
// LDS-DMA and then serves all nine taps (the tap only shifts the fragment read address), so the
// activation is fetched ~1.3x instead of 9x per K sweep; the weight slabs of the three taps of one patch
// COLUMN travel together through a 3-stage ring (counted s_waitcnt vmcnt across raw s_barriers).
//   (nearest x2 up-sampling of the input is folded into the patch gather: source pixel = logical >> 1)
//   MFMA        : v_mfma_f32_16x16x32_bf16 / _f16 — one instruction consumes a whole 32-deep slab of a 16x16 block;
//                 at equal operand traffic it ran 8 % faster in situ than 32x32x16 (the chip holds a higher
//                 clock on this shape, MI355X_MICROARCH.md DVFS item 7)
//   patch image : pixel-linear, 64 B per pixel, 16-B chunk slot = chunk ^ (((pixel >> 2) & 1) << 1)
//                 (source-side swizzle; conflict-free ds_read_b128 for the 16x16x32 operand map — 16 consecutive
//                 pixels x 4 k-chunks per wave-instruction — at ANY pixel alignment, i.e. for all nine tap shifts)
//   weight slab : [64 rows][64 B], same slot rule on the row index
// History: a tap-by-tap version of this kernel (one weight slab and one barrier per tap, 128- or
// 64-wide) was the round's dominant kernel at 1100 TFLOP/s; the patch-row window below replaced it at +5-8 % on
// the same layers (+19 % on the 64-wide ones).
#pragma once
#include <type_traits>

#include "common.hpp"

// ---- patch-row register window -------------------------------------------------------------------------------------
// Wave tile = 128 pixels
// (RW tile rows) x 32 channels, 64 accumulator registers per lane.  The K loop runs over (32-channel chunk, patch
// COLUMN shift pw): for one pw the three taps of that column are resident together (three 4-KiB weight slabs per ring
// stage), and a patch-row fragment is read from LDS ONCE and multiplied into the up to three output rows it serves
// (patch row offsets ph = 0, 1, 2): (RW + 2) * XB pixel-fragment reads + 6 weight-fragment reads per 48 MFMAs and
// one workgroup barrier per 48 MFMAs per wave (tap by tap: 24 reads and three barriers for the same work).
template <int TH, int TW> struct HaloRwCfg {
  static constexpr int BN = 64;
  static constexpr int NPIX = (TH + 2) * (TW + 2);
  static constexpr int P_INSTR = (NPIX + 15) / 16;               // 1-KiB DMA instructions per patch; the buffers are packed
  static constexpr int PATCH_BYTES = P_INSTR * 1024;
  static constexpr int STAGE_BYTES = 3 * BN * 64;                // three taps of one patch column
  static constexpr int NS = 3;
  static constexpr int RING = 2 * PATCH_BYTES + NS * STAGE_BYTES;
  static constexpr int C_BYTES = TH * TW * (BN * 2 + 16);
  static constexpr int EPI_BYTES = C_BYTES + 2 * 2 * BN * 4;     // C tile + statistics scratch [WM = 2][2][BN]
  static constexpr int LDS_BYTES = RING > EPI_BYTES ? RING : EPI_BYTES;
};

template <typename T, int TH, int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_rw_kernel(const ConvArgs a) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  typedef HaloRwCfg<TH, TW> Cfg;
  constexpr int BN = 64, BK = 32, EPC = 8, BM = TH * TW;
  static_assert(BM == 256 && TW % 16 == 0, "tile must hold 256 pixels in rows of 16-pixel blocks");
  constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW;
  constexpr int PIXB = BK * 2;                       // 64 B per pixel / weight row
  constexpr int P_INSTR = Cfg::P_INSTR;
  constexpr int P_IT = (P_INSTR + 3) / 4;            // per wave (the last round may be short: see p_it)
  constexpr int PATCH_BYTES = Cfg::PATCH_BYTES;
  constexpr int SLAB = BN * PIXB, STAGE = Cfg::STAGE_BYTES, NS = Cfg::NS;
  constexpr int B_IT = 3;                            // DMA instructions per wave per stage: 3 slabs x 4 KiB / 4 waves
  constexpr int WM = 2, WN = 2, WTM = BM / WM, WTN = BN / WN;
  constexpr int RW = WTM / TW, XB = TW / 16;         // tile rows per wave, 16-pixel blocks per row
  constexpr int MB = RW * XB, NB = WTN / 16;         // 8 x 2 MFMA blocks of 16x16 per wave
  constexpr int C_PITCH = BN * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, c4 = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
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
  const bool flip = a.kmul < 0;                      // data gradient: taps mirrored

  // ---- DMA lane geometry: a 1-KiB instruction = 16 patch pixels x 64 B; wave w issues pieces w, w+4, ... (P_IT of them; a
  // piece index past the patch repeats the last piece: same bytes to the same place, so every wave issues the same count) ----
  // Sources go through buffer descriptors (dma.hpp: dma16_buf): base = this image / this channel tile's weight rows (wave-
  // uniform), a scalar offset for the slab (and tap), ONE 32-bit register per piece for the lane's offset — the always-out-of-
  // range offset where the lane is padding, so that the hardware's range check writes the zeros.
  const int lrow = lane >> 2, slot = lane & 3;
  const bufdesc_t desc_in = make_buf(in + (size_t)n * a.Hi * a.Wi * a.ldi);
  unsigned p_off[P_IT];
  int p_dst[P_IT];
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int piece = min(wave + 4 * i, P_INSTR - 1);
    const int q = piece * 16 + lrow;
    const int py = q / PW, px = q - py * PW;
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    const bool ok = q < NPIX && (unsigned)yy < (unsigned)a.Hlog && (unsigned)xx < (unsigned)a.Wlog;
    p_off[i] = ok ? (unsigned)((((yy >> a.up) * a.Wi + (xx >> a.up)) * a.ldi + (slot ^ (((q >> 2) & 1) << 1)) * EPC) * 2) : DMA_PAD;
    p_dst[i] = piece * 1024;
  }
  const unsigned lds0 = lds_addr(lds);               // LDS byte address of the ring (DMA destinations are integers)
  unsigned char* const patch0 = lds;
  unsigned char* const bring = lds + 2 * PATCH_BYTES;
  auto issue_patch_piece = [&](int buf, int c0, int i) {
    dma16_buf(desc_in, p_off[i], (unsigned)c0 * 2u, lds0 + buf * PATCH_BYTES + p_dst[i]);
  };
  // stage = the three taps of patch column pw: slab ph holds tap (ph, pw), mirrored for the data gradient; a wave
  // brings 16 rows x 64 B of each slab (one instruction per slab)
  const size_t wrow = (size_t)9 * a.Ci;
  const int brow = wave * 16 + lrow;
  const bufdesc_t desc_w = make_buf(wk + (size_t)n0 * wrow), desc_none = make_buf(wk, false);
  const unsigned b_off = (unsigned)(((size_t)brow * wrow + (slot ^ (((brow >> 2) & 1) << 1)) * EPC) * 2);
  auto issue_stage_piece = [&](int stage, int pw, int c0, int ph) {      // c0 < 0: nothing left to fetch (zeros into a dead slot)
    const int tap = flip ? (2 - ph) * 3 + (2 - pw) : ph * 3 + pw;
    dma16_buf(c0 >= 0 ? desc_w : desc_none, b_off, (unsigned)(tap * a.Ci + c0) * 2u, lds0 + 2 * PATCH_BYTES + stage * STAGE + wave * 1024 + ph * SLAB);
  };

  // ---- fragment geometry ------------------------------------------------------------------------------------------
  const int q00 = (wm * RW) * PW + l16;              // patch pixel of (first wave row, x = l16, column shift 0)
  // weight rows of block nb sit nb * 16 rows further on: the swizzle bit (row >> 2) & 1 does not change, so the block is an
  // immediate offset of the LDS read
  const int brow0 = wn * WTN + l16;
  const int boff0 = brow0 * PIXB + ((c4 ^ (((brow0 >> 2) & 1) << 1)) << 4);
  // byte offset of every pixel fragment this lane ever reads inside a patch buffer: [patch column][patch row] (18 registers;
  // the x block is an immediate + xb * 16 * PIXB for the same reason), so the main loop issues its LDS reads without any
  // address arithmetic in front of the MFMAs
  int aoff[3][RW + 2];
#pragma unroll
  for (int pw = 0; pw < 3; ++pw)
#pragma unroll
    for (int pr = 0; pr < RW + 2; ++pr) {
      const int q = q00 + pr * PW + pw;
      aoff[pw][pr] = q * PIXB + ((c4 ^ (((q >> 2) & 1) << 1)) << 4);
    }
  // first written by the very first step, whose MFMAs take the BIAS as their C operand: no 64 v_mov to clear the accumulators
  // and no 64 v_add in the epilogue (a lane's four values of a block are channels 4*c4 .. +3 of block nb)
  f32x4 acc[MB][NB];
  f32x4 bias4[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    bias4[nb] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n0 + wn * WTN + nb * 16 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};

  const int nC = a.Ci / BK;
  // prologue: patch of chunk 0 and the stages of steps 0 and 1 (stage index == patch column: three steps per chunk)
#pragma unroll
  for (int i = 0; i < P_IT; ++i) issue_patch_piece(0, 0, i);
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) issue_stage_piece(0, 0, 0, ph);
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) issue_stage_piece(1, 1, 0, ph);
  wait_vmcnt<B_IT>();
  __builtin_amdgcn_s_barrier();

  // One step = patch column pw of chunk `chunk` (48 MFMAs per wave).  Every DMA instruction of the step — the stage two steps
  // ahead, and in column 0 the next chunk's patch — is issued from INSIDE the MFMA stream, a piece after each patch row's
  // MFMAs, so that its address arithmetic and m0 traffic run in the shadow of the matrix pipe instead of in front of it.
  auto step = [&](int chunk, auto pw_tag, auto par_tag, auto first_tag) {
    constexpr int pw = decltype(pw_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;      // step (chunk 0, column 0): the ph == 0 MFMA of a block is its first
    constexpr int par = decltype(par_tag)::value;      // chunk & 1: the patch buffer is a compile-time LDS offset
    constexpr int NPIECE = B_IT + (pw == 0 ? P_IT : 0);
    constexpr int NPR = RW + 2;
    // the stage two steps ahead: (chunk, pw + 2) or (chunk + 1, pw - 1); nothing past the last step
    constexpr int pw2 = (pw + 2) % 3;
    const int c2 = (pw == 0 ? chunk : chunk + 1);
    const int c0_stage = c2 < nC ? c2 * BK : -1;
    const bool next_patch = chunk + 1 < nC;
    const unsigned char* pa = patch0 + par * PATCH_BYTES;
    const unsigned char* pb = bring + pw * STAGE;
    bf16x8 bfr[3][NB];
#pragma unroll
    for (int ph = 0; ph < 3; ++ph)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bfr[ph][nb] = *reinterpret_cast<const bf16x8*>(pb + ph * SLAB + nb * 16 * PIXB + boff0);
    bf16x8 afr[NPR][XB];
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
      for (int xb = 0; xb < XB; ++xb) afr[pr][xb] = *reinterpret_cast<const bf16x8*>(pa + xb * 16 * PIXB + aoff[pw][pr]);
    __builtin_amdgcn_sched_barrier(0);              // every fragment read of the step is in flight before the first MFMA
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr) {
#pragma unroll
      for (int xb = 0; xb < XB; ++xb)
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
          const int orow = pr - ph;                  // output row of this wave served through patch-row offset ph
          if (orow >= 0 && orow < RW) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[orow * XB + xb][nb] = mfma_16x16x32<T>(bfr[ph][nb], afr[pr][xb],      // D[channel][pixel]
                                                         (FIRST && ph == 0) ? bias4[nb] : acc[orow * XB + xb][nb]);
          }
        }
#pragma unroll
      for (int k = pr * NPIECE / NPR; k < (pr + 1) * NPIECE / NPR; ++k) {
        if (k < B_IT) issue_stage_piece(pw2, pw2, c0_stage, k);
        else issue_patch_piece(par ^ 1, next_patch ? (chunk + 1) * BK : chunk * BK, k - B_IT);      // (last chunk: a dead buffer)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // stage s+1 (and, in order before it, the patch issued in column 0) must have landed; what may stay in flight: the
    // stage issued this step, and the patch issued in this column-0 step or the one before
    if (pw <= 1) wait_vmcnt<B_IT + P_IT>(); else wait_vmcnt<B_IT>();
    __builtin_amdgcn_s_barrier();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using No = std::false_type;
  step(0, I0{}, I0{}, std::true_type{});
  for (int chunk = 0;; chunk += 2) {                   // (the loop body starts at column 1 so that only ONE extra step body exists)
    step(chunk, I1{}, I0{}, No{});
    step(chunk, I2{}, I0{}, No{});
    if (chunk + 1 >= nC) break;
    step(chunk + 1, I0{}, I1{}, No{});
    step(chunk + 1, I1{}, I1{}, No{});
    step(chunk + 1, I2{}, I1{}, No{});
    if (chunk + 2 >= nC) break;
    step(chunk + 2, I0{}, I0{}, No{});
  }
  wait_vmcnt<0>();                                   // (the zero-fill pieces of the last two steps)

  // ---- epilogue --------------------------------------------------------------------------------------------------------
  // The MFMAs ran with the weight fragment as the A operand, so a lane holds, per 16x16 block, FOUR CONSECUTIVE CHANNELS
  // (4*c4 .. +3) of ONE pixel (l16): bias / ReLU / rounding happen once per value in registers, a block is staged with ONE
  // 8-byte LDS write per lane (conflict-free at the 144-B row pitch), and the tile leaves as 16-byte row-contiguous stores
  // (whole 128-B lines; 8-byte stores straight from the registers were measured: 4x the line accesses, -10 %).
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  struct alignas(8) Pack4 { T v[4]; };
  float* const red = reinterpret_cast<float*>(lds + Cfg::C_BYTES);      // [WM][2][BN] behind the C tile
  // ReLU and the statistics are workgroup-uniform switches: four straight-line variants instead of 64 dead v_max / a test per
  // block.  Statistics are of the ROUNDED outputs, summed two channels at a time (packed fp32 adds / fmas), folded over
  // the 16 pixel lanes of a row with DPP adds (no LDS traffic), then over the two wave rows through LDS.
  auto finish = [&](auto relu_tag, auto stats_tag) {
    constexpr bool RELU = decltype(relu_tag)::value, STATS = decltype(stats_tag)::value;
    f32x2 sm[NB][2], sq[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 2; ++j) { sm[nb][j] = f32x2{0.f, 0.f}; sq[nb][j] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int row = wm * WTM + mb * 16 + l16;                // tile pixel of this lane in block mb
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
        *reinterpret_cast<Pack4*>(lds + row * C_PITCH + (wn * WTN + nb * 16 + 4 * c4) * 2) = pk;
      }
    }
    if constexpr (STATS) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = row16_sum(sm[nb][r >> 1][r & 1]), s2 = row16_sum(sq[nb][r >> 1][r & 1]);
          if (l16 == 0) {
            red[(wm * 2 + 0) * BN + wn * WTN + nb * 16 + 4 * c4 + r] = s1;
            red[(wm * 2 + 1) * BN + wn * WTN + nb * 16 + 4 * c4 + r] = s2;
          }
        }
    }
  };
  using Yes = std::true_type;
  if (a.stats) {
    if (a.relu) finish(Yes{}, Yes{}); else finish(No{}, Yes{});
  } else {
    if (a.relu) finish(Yes{}, No{}); else finish(No{}, No{});
  }
  __syncthreads();
  if (a.stats && tid < 2 * BN) {
    const int q = tid / BN, c = tid - q * BN;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) v += red[(w * 2 + q) * BN + c];
    a.stats[((size_t)(bid / NT) * 2 + q) * a.Co + n0 + c] = v;
  }
  constexpr int CPRC = BN / EPC;
  if (a.pool2) {       // gradient of a fused nearest x2 up-sampling: the four outputs of a 2x2 group are summed (fp32, from the
                       // ROUNDED tile values, as the separate mi355_upsample2_bwd pass did) into the half-resolution tensor
    const int Ho2 = a.Ho >> 1, Wo2 = a.Wo >> 1;
    for (int id = tid; id < (BM / 4) * CPRC; id += 256) {
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
          const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(lds + row * C_PITCH + c * 16);
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
  constexpr int NST = BM * CPRC / 256;
  static_assert(NST * 256 == BM * CPRC, "whole store rounds");
  auto addr = [&](int it) __attribute__((always_inline)) {
    const int id = tid + it * 256, row = id / CPRC, c = id - row * CPRC;
    const int py = row / TW, px = row - py * TW;
    return out + ((size_t)(n * a.Ho + y0 + py) * a.Wo + x0 + px) * a.ldo + n0 + c * EPC;
  };
  if (a.accumulate) {      // all old values first: read per round, every load sat behind the previous round's store to the same tensor
    Vec16<T> old[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) old[it] = ld16<T>(addr(it));
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int id = tid + it * 256, row = id / CPRC, c = id - row * CPRC;
      Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(lds + row * C_PITCH + c * 16);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) + to_f32<T>(old[it].v[e]));
      st16<T>(addr(it), v);
    }
  } else {
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int id = tid + it * 256, row = id / CPRC, c = id - row * CPRC;
      st16<T>(addr(it), *reinterpret_cast<const Vec16<T>*>(lds + row * C_PITCH + c * 16));
    }
  }
}

template <typename T, int TH, int TW>
static int launch_halo_rw(const ConvArgs& a, hipStream_t s) {
  const int grid = a.N * (a.Ho / TH) * (a.Wo / TW) * (a.Co / 64);
  constexpr int lds_bytes = HaloRwCfg<TH, TW>::LDS_BYTES;
  // once per process and kernel variant; a function-local static is initialised exactly once even when two threads launch
  // concurrently (forward on the main thread, backward on the autograd worker)
  static const hipError_t configured = hipFuncSetAttribute((const void*)conv3x3_halo_rw_kernel<T, TH, TW>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (configured != hipSuccess)
    MI355_FAIL((int)configured, "conv3x3_halo_rw: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(configured));
  hipLaunchKernelGGL((conv3x3_halo_rw_kernel<T, TH, TW>), dim3(grid), dim3(256), lds_bytes, s, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
