// Convolutions WITHOUT padding whose every tap is in bounds, as plain MFMA GEMMs, bf16 / fp16, gfx950:
//
//   mode 0   1x1 (stride 1 or 2) and 2x2 / stride-2 convolutions:  Y[m][co] = sum_{tap, ci} X[src(m, tap)][ci] * W[co][tap][ci]
//            (the bottlenecks' and shortcuts' 1x1 convolutions of the ResNet-50 encoder, torchvision layout behind
//             /root/reference/models/segmentation_models/ResnetUnet.py:32-43; the attention gates' W_g / W_x at 256-512 channels,
//             AttentionUNet.py:32-38; the data gradient of ConvTranspose2d(k = 2, s = 2), ResnetUnet.py:21)
//   mode 1   ConvTranspose2d(k = 2, s = 2) forward (ResnetUnet.py:21,51) as FOUR pointwise phases of one GEMM with 4 Co columns:
//            Y[n][2h + dh][2w + dw][co] = sum_ci X[n][h][w][ci] * W[co][2 dh + dw][ci]   (a column tile lies inside one phase)
//
// These were served by conv_igemm_dma_kernel (128 x 128 tile, four waves, two-stage ring): 8 DMA pieces per wave per 16 MFMAs and a
// ring one tile deep — 180-330 TFLOP/s on K = 256 ... 2048 (profiles/r04b_C2_ResNetUnet_bf16_kernel_table.txt: 3.0 of 12.1 ms).
// Here: a 512-thread workgroup owns 256 pixels x 128 channels (wave = 64 x 64: 16 accumulator blocks of 16 x 16), K steps of 64
// through a THREE-stage LDS-DMA ring with one barrier per step (tile k + 2 is requested from inside step k's MFMA stream, the
// counted vmcnt leaves tile k + 1 in flight), 6 pieces and 16 fragment reads per wave per 32 MFMAs, fragments of the second K
// half read behind the first half's MFMAs.  LDS rows are 128 bytes with the 16-byte chunk index XOR-ed with (row >> 1) & 7 on
// the SOURCE side (conflict-free ds_read_b128 for the 16 x 16 x 32 operand map); weights are the MFMA's A operand, so a lane
// ends up with four consecutive channels of a pixel: 8-byte staging writes, 16-byte row-contiguous stores.
#pragma once
#include "common.hpp"

struct Gemm256Cfg {
  static constexpr int BM = 256, BN = 128, BK = 64, NS = 3;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  static constexpr int C_PITCH = BN * 2 + 16;
  static constexpr int LDS_BYTES = NS * STAGE;                            // 144 KiB: one workgroup per CU
  static_assert(BM * C_PITCH + 4 * 2 * BN * 4 <= LDS_BYTES, "the C tile and the statistics scratch go on top of the dead ring");
};

template <typename T>
__global__ __launch_bounds__(512, 2) void conv_gemm256_kernel(const ConvArgs a, const int mode) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  typedef Gemm256Cfg Cfg;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, NS = Cfg::NS, EPC = 8;
  constexpr int A_BYTES = Cfg::A_BYTES, STAGE = Cfg::STAGE, C_PITCH = Cfg::C_PITCH;
  constexpr int A_IT = BM / 8 / 8, B_IT = BN / 8 / 8, PER_TILE = A_IT + B_IT;        // 1-KiB pieces (8 rows x 128 B) per wave and K tile: 4 + 2
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, c4 = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;                 // 4 pixel blocks x 2 channel blocks of 64
  const int taps = a.KH * a.KW;
  const int NCOL = mode == 1 ? taps * a.Co : a.Co;          // GEMM columns
  const int NT = NCOL / BN;
  const int bid = xcd_tile(blockIdx.x, gridDim.x);          // channel tiles of one pixel tile are neighbours: they meet in one L2
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int ptap = mode == 1 ? n0 / a.Co : 0, co0 = mode == 1 ? n0 - ptap * a.Co : n0;      // (mode 1: the tile's phase and channels)
  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wk = reinterpret_cast<const T*>(a.wk);
  auto fsw = [](int row) { return (row >> 1) & 7; };

  // ---- DMA lane geometry: piece p = rows [8 p, 8 p + 8) x 128 B; lane -> (row, 16-byte slot); the source chunk is swizzled -------
  const int lrow = lane >> 3, slot = lane & 7;
  const T* a_src[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int row = (wave + 8 * i) * 8 + lrow;
    const int m = m0 + row;                                 // (M % 256 == 0: every row exists)
    long long pix;
    if (mode == 1) {
      pix = m;                                              // the GEMM rows ARE the input pixels
    } else {
      const int n = m / a.HoWo, rem = m - n * a.HoWo;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      pix = ((long long)n * a.Hi + ho * a.mul) * a.Wi + wo * a.mul;
    }
    a_src[i] = in + pix * a.ldi + (slot ^ fsw(row)) * EPC;
  }
  const T* b_src[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int row = (wave + 8 * i) * 8 + lrow;
    b_src[i] = wk + ((size_t)(co0 + row) * taps + ptap) * a.Ci + (slot ^ fsw(row)) * EPC;
  }
  const unsigned lds0 = lds_addr(lds);
  const int KC = a.Ci / BK;                                 // K tiles per tap
  const int KT = (mode == 1 ? 1 : taps) * KC;
  // issue side: tile index -> (tap, channel offset); piece j of the tile (0 .. PER_TILE - 1)
  int i_tap = 0, i_c0 = 0, i_stage = 0;
  size_t i_aoff = 0, i_boff = 0;                            // element offsets of the tile the issue side stands on
  auto issue_piece = [&](int j) __attribute__((always_inline)) {
    const unsigned st = lds0 + i_stage * STAGE;
    if (j < A_IT) dma16(a_src[j] + i_aoff, st + (wave + 8 * j) * 1024);
    else dma16(b_src[j - A_IT] + i_boff, st + A_BYTES + (wave + 8 * (j - A_IT)) * 1024);
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    i_c0 += BK;
    if (i_c0 == a.Ci) {
      i_c0 = 0;
      ++i_tap;
    }
    const int kh = i_tap / a.KW, kw = i_tap - kh * a.KW;
    i_aoff = (mode == 1 ? (size_t)0 : ((size_t)kh * a.Wi + kw) * a.ldi) + i_c0;
    i_boff = (mode == 1 ? (size_t)0 : (size_t)i_tap * a.Ci) + i_c0;
    i_stage = i_stage + 1 == NS ? 0 : i_stage + 1;
  };

  f32x4 acc[4][4];                                          // [channel block][pixel block]
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a stage: row * 128 + ((chunk ^ f(row)) << 4); the K half adds 4 chunks before the XOR
  int fa_w[4], fa_p[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int rw = wn * 64 + b * 16 + l16, rp = wm * 64 + b * 16 + l16;
    fa_w[b] = A_BYTES + rw * 128 + fsw(rw) * 16;            // (chunk 0 position; the reader XORs (kh2 * 4 + c4) << 4 into bits 4-6)
    fa_p[b] = rp * 128 + fsw(rp) * 16;
  }
  auto frag = [&](const unsigned char* st, int base, int kh2) __attribute__((always_inline)) {
    // base holds f(row) << 4 in bits 4-6 and the row in the bits above: chunk (kh2 * 4 + c4) ^ f(row) = XOR on bits 4-6
    return *reinterpret_cast<const bf16x8*>(st + (base ^ ((kh2 * 4 + c4) << 4)));
  };

  // ---- prologue: two tiles in flight ------------------------------------------------------------------------------------
  int issued = 0;
#pragma unroll
  for (int t = 0; t < NS - 1; ++t) {
    if (issued < KT) {
#pragma unroll
      for (int j = 0; j < PER_TILE; ++j) issue_piece(j);
      issue_advance();
      ++issued;
    }
  }

  int stage = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // tile kt has landed for this wave (tile kt + 1 may still fly), then for every wave; behind the barrier nobody reads tile kt - 1 any more
    if (kt + 1 < KT) wait_vmcnt<PER_TILE>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const unsigned char* st = lds + stage * STAGE;
    const bool more = issued < KT;                          // tile kt + 2 goes into the stage tile kt - 1 has just left
    bf16x8 w0[4], p0[4], w1[4], p1[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      w0[b] = frag(st, fa_w[b], 0);
      p0[b] = frag(st, fa_p[b], 0);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {                           // the second K half's fragments load behind the first half's MFMAs
      w1[b] = frag(st, fa_w[b], 1);
      p1[b] = frag(st, fa_p[b], 1);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = mfma_16x16x32<T>(w0[cb], p0[pb], acc[cb][pb]);
      if (more) {                                           // one piece behind every fourth MFMA (three back to back stall the pipe)
        issue_piece(cb);
      }
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = mfma_16x16x32<T>(w1[cb], p1[pb], acc[cb][pb]);
      if (more && cb < PER_TILE - 4) issue_piece(4 + cb);
    }
    if (more) {
      issue_advance();
      ++issued;
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
  }
  __syncthreads();                                          // every wave has read the last tile: the ring is dead

  // ---- epilogue: bias, ReLU, rounding; C tile staged in LDS; fused BatchNorm statistics; 16-byte row-contiguous stores ----------
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  unsigned char* const cst = lds;
  float* const red = reinterpret_cast<float*>(lds + BM * C_PITCH);      // [wm][2][BN]
  struct alignas(8) Pack4 { T v[4]; };
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int ch = wn * 64 + cb * 16 + 4 * c4;              // this lane's four channels inside the tile
    const f32x4 bias4 = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + co0 + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sm = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      const int row = wm * 64 + pb * 16 + l16;
      Pack4 pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[cb][pb][r] + bias4[r];
        pk.v[r] = from_f32<T>(a.relu ? fmaxf(v, 0.f) : v);
        const float q = to_f32<T>(pk.v[r]);
        sm[r] += q;
        sq[r] += q * q;
      }
      *reinterpret_cast<Pack4*>(cst + row * C_PITCH + ch * 2) = pk;
    }
    if (a.stats) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = row16_sum(sm[r]), s2 = row16_sum(sq[r]);
        if (l16 == 0) {
          red[(wm * 2 + 0) * BN + ch + r] = s1;
          red[(wm * 2 + 1) * BN + ch + r] = s2;
        }
      }
    }
  }
  __syncthreads();
  if (a.stats && tid < 2 * BN) {
    const int q = tid / BN, c = tid - q * BN;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += red[(w * 2 + q) * BN + c];
    a.stats[((size_t)mt * 2 + q) * a.Co + n0 + c] = v;
  }
  constexpr int CPRC = BN / EPC;                            // 16 chunks per pixel row of the tile
  const int pdh = ptap / a.KW, pdw = ptap - pdh * a.KW;
  for (int id = tid; id < BM * CPRC; id += 512) {
    const int row = id / CPRC, c = id - row * CPRC;
    const int m = m0 + row;
    size_t opix;
    if (mode == 1) {                                        // input pixel (n, h, w) -> output pixel (n, 2 h + dh, 2 w + dw)
      const int hw = a.Hi * a.Wi;
      const int n = m / hw, rem = m - n * hw;
      const int h = rem / a.Wi, w = rem - h * a.Wi;
      opix = ((size_t)n * a.Ho + 2 * h + pdh) * a.Wo + 2 * w + pdw;
    } else {
      opix = (size_t)m;
    }
    T* p = out + opix * a.ldo + co0 + c * EPC;
    Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(cst + row * C_PITCH + c * 16);
    if (a.accumulate) {
      const Vec16<T> o = ld16<T>(p);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) + to_f32<T>(o.v[e]));
    }
    st16<T>(p, v);
  }
}

// 0: not served; 1: mode 0 (1x1 / 2x2 stride-2 gather), 2: mode 1 (ConvTranspose2d(2, 2) forward phases)
static int gemm256_mode(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int mul, int kmul, int off, int div, int up) {
  if (up || off != 0 || KH != KW || Ci % 64 != 0 || Co % 128 != 0) return 0;
  if (KH == 1 && div == 1 && (mul == 1 || mul == 2) && (long long)(Ho - 1) * mul < Hi && (long long)(Wo - 1) * mul < Wi)
    return ((long long)N * Ho * Wo) % 256 == 0 ? 1 : 0;
  if (KH == 2 && div == 1 && mul == 2 && kmul == 1 && 2 * Ho == Hi && 2 * Wo == Wi) return ((long long)N * Ho * Wo) % 256 == 0 ? 1 : 0;
  if (KH == 2 && div == 2 && mul == 1 && kmul == -1 && Ho == 2 * Hi && Wo == 2 * Wi) return ((long long)N * Hi * Wi) % 256 == 0 ? 2 : 0;
  return 0;
}
static long long gemm256_tiles(int mode, int N, int Hi, int Wi, int Ho, int Wo, int Co) {
  return mode == 2 ? (long long)N * Hi * Wi / 256 * (4 * Co / 128) : (long long)N * Ho * Wo / 256 * (Co / 128);
}

template <typename T>
static int launch_gemm256(const ConvArgs& a, int mode, hipStream_t s) {
  constexpr int lds_bytes = Gemm256Cfg::LDS_BYTES;
  static const hipError_t configured =
      hipFuncSetAttribute((const void*)conv_gemm256_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (configured != hipSuccess)
    MI355_FAIL((int)configured, "conv_gemm256: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(configured));
  const long long grid = gemm256_tiles(mode, a.N, a.Hi, a.Wi, a.Ho, a.Wo, a.Co);
  hipLaunchKernelGGL((conv_gemm256_kernel<T>), dim3((int)grid), dim3(512), lds_bytes, s, a, mode == 2 ? 1 : 0);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
