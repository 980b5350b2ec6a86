// 3x3 / stride-1 / pad-1 convolution (forward and data gradient) for Ci = 64, bf16 / fp16, gfx950 — the WEIGHT-STATIONARY kernel.
//
// The 64-channel layers at 256 x 256 / 128 x 128 are the largest tensors of every U-Net and sit at the HBM ridge: a 256-pixel
// tile is only 288 MFMAs per wave, and conv3x3_halo_rw_kernel pays for it 18 weight DMA pieces per wave (every tile of the
// layer fetches the same 72 KiB of weights again: more L2 -> LDS bytes than the activations), six counted waits + barriers, a
// prologue and an epilogue (0.41 of peak).  Here the weights never move again:
//
//   weights     : a wave owns 16 output channels for ALL of K = 9 x 64: 18 MFMA A-fragments = 72 registers per lane, loaded
//                 once per workgroup straight from global memory.  No weight bytes in LDS, no weight fragment reads.
//   workgroup   : 4 waves = 64 output channels, PERSISTENT: it walks a contiguous range of 8 x 32-pixel tiles of one channel
//                 tile (grid = 2 workgroups per CU, 80 KB of LDS and <= 256 registers each: the two run independently and
//                 fill each other's epilogues and waits)
//   wave tile   : ALL 256 pixels x 16 channels (64 accumulator registers); per (32-channel slab, patch column) step the ten
//                 patch-row fragments of a 16-pixel column are read once and feed up to three output rows: 20 ds_read_b128
//                 per 48 MFMAs, no weight reads
//   patch       : two 32-channel slabs [10 rows][36-pixel pitch][64 B] (conv3x3_halo_pp128.hpp's layout: the swizzle bit of a
//                 pixel is (row + (x >> 2)) & 1, so a fragment address is ONE lane register per column shift + immediates),
//                 each brought by 23 LDS-DMA pieces through a buffer descriptor based at the patch origin
//   schedule    : slab 0 of tile t + 1 lands (buffer A) while slab 1 of tile t is multiplied, slab 1 of tile t + 1 (buffer B) is
//                 requested right behind tile t's last MFMA and lands during its epilogue and the next tile's first half; the C
//                 tile has a staging area of its own (136-B pixel pitch: conflict-free 8-byte writes), so the epilogue's reads
//                 need no barrier behind them.  gfx950 counts stores in vmcnt, in issue order with the DMA pieces: the mid-tile
//                 wait for slab 1 is a COUNTED vmcnt that leaves exactly the epilogue's stores (younger than the pieces) in
//                 flight, the end-of-tile wait a vmcnt(0) half a tile after the youngest store — the coupling that sank round 2's
//                 persistent variants never waits for a fresh store.  Three barriers per tile (the 4-wave halo kernel: eight).
#pragma once
#include <type_traits>

#include "common.hpp"

struct Ws64Cfg {
  static constexpr int TH = 8, TW = 32, BN = 64, CI = 64;
  static constexpr int PWL = 36, PH = TH + 2;
  static constexpr int NPIX = PH * PWL;                                  // 360 pixel slots per slab
  static constexpr int P_INSTR = (NPIX + 15) / 16;                       // 23 DMA pieces of 16 pixels
  static constexpr int SLAB_BYTES = P_INSTR * 1024;                      // 23 KiB
  static constexpr int C_PITCH = BN * 2 + 8;                             // 34 dwords: the 16 pixel rows of an 8-byte staging write hit 32 distinct banks
  static constexpr int C_BYTES = TH * TW * C_PITCH;                      // 34 KiB staging area of its own
  static constexpr int LDS_BYTES = 2 * SLAB_BYTES + C_BYTES;             // 80 KiB: exactly two workgroups per CU
};

template <int I> using WsI = std::integral_constant<int, I>;
template <int B, int E, typename F> __device__ __forceinline__ void ws_static_for(F&& f) {
  if constexpr (B < E) {
    f(WsI<B>{});
    ws_static_for<B + 1, E>(f);
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void conv3x3_ws64_kernel(const ConvArgs a, const int groups) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  typedef Ws64Cfg Cfg;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, BN = Cfg::BN, EPC = 8, BM = TH * TW;
  constexpr int PWL = Cfg::PWL, PH = Cfg::PH, NPIX = Cfg::NPIX, PIXB = 64;
  constexpr int P_INSTR = Cfg::P_INSTR, P_IT = (P_INSTR + 3) / 4;        // six pieces per wave (a piece index past the slab repeats the last one)
  constexpr int SLAB = Cfg::SLAB_BYTES, ROWB = PWL * PIXB;              // 2304 B per patch row
  constexpr int C_PITCH = Cfg::C_PITCH;
  constexpr int XB = TW / 16, MB = TH * XB;                              // 16 accumulator blocks of 16 pixels x 16 channels
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, c4 = lane >> 4;
  const int NT = a.Co / BN, TXN = a.Wo / TW, TYN = a.Ho / TH;
  // ---- which tiles: workgroup b sits on XCD b % 8; the workgroups of one XCD are dealt to the channel tiles round-robin and
  // share a contiguous range of spatial tiles (the NT channel tiles of one patch meet in one L2) ---------------------------------
  const int S = a.N * TYN * TXN;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int nt = j % NT;
  const int kk = xcd * (groups >> 3) + j / NT;                           // spatial group of this workgroup, 0 .. groups - 1
  const int sp_begin = (int)((long long)kk * S / groups), sp_end = (int)((long long)(kk + 1) * S / groups);
  if (sp_begin >= sp_end) return;
  const int n0 = nt * BN;
  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wk = reinterpret_cast<const T*>(a.wk);
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  const bool flip = a.kmul < 0;

  // ---- the stationary operand: W[n0 + 16 wave + l16][tap][32 slab + 8 c4 .. + 7] as MFMA A fragments ------------------------
  bf16x8 wf[2][9];
  {
    const T* wrow = wk + ((size_t)(n0 + wave * 16 + l16) * 9) * Cfg::CI + c4 * EPC;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int tap = flip ? 8 - tp : tp;                              // data gradient: taps mirrored
        wf[s][tp] = *reinterpret_cast<const bf16x8*>(wrow + tap * Cfg::CI + s * 32);
      }
  }
  f32x4 bias4 = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n0 + wave * 16 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  // The loads above must be COMPLETE, in the compiler's own book-keeping, before the tile loop: hipcc waits for a load in front
  // of its first use — the first MFMA of the loop body — with s_waitcnt vmcnt(0), on every trip, and that wait would also drain
  // the previous tile's stores and the slab in flight (measured on the first version of this kernel).  Passing the registers
  // through an empty asm statement puts the wait here and makes the values asm results with nothing pending behind them.
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) asm volatile("" : "+v"(wf[s][tp]));
  asm volatile("" : "+v"(bias4));

  // ---- DMA lane geometry: piece p = LDS pixel slots [16 p, +16) x 64 B; lane -> (slot pixel, 16-byte chunk) ------------------
  // source offsets are relative to the PATCH ORIGIN (pixel (y0 - 1, x0 - 1) of the tile, halved when the x2 up-sampling is
  // folded in), which is the base of the per-tile buffer descriptor: always >= 0; validity (image border, the two pad columns,
  // the tail of the last piece) is per tile and turns the offset into the out-of-range value the hardware zero-fills
  const int lrow = lane >> 2, slot = lane & 3;
  unsigned p_rel[P_IT];
  int p_yx[P_IT];                                                        // py | px << 8 | statically valid << 16
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int piece = min(wave + 4 * i, P_INSTR - 1);
    const int q = piece * 16 + lrow;
    const int py = q / PWL, px = q - py * PWL;
    const int ry = a.up ? (py + 1) >> 1 : py, rx = a.up ? (px + 1) >> 1 : px;
    p_rel[i] = (unsigned)(((ry * a.Wi + rx) * a.ldi + (slot ^ (((q >> 2) & 1) << 1)) * EPC) * 2);
    p_yx[i] = py | (px << 8) | ((q < NPIX && px < TW + 2) ? 1 << 16 : 0);
  }
  const unsigned lds0 = lds_addr(lds);

  // ---- fragment addresses: patch pixel (pr, 16 xb + l16 + pw), chunk c4 -> one register per column shift (even rows; odd rows
  // toggle the swizzle bit = byte 32), the row and the block are instruction immediates --------------------------------------
  int fa[3];
#pragma unroll
  for (int pw = 0; pw < 3; ++pw) {
    const int x = l16 + pw;
    fa[pw] = x * PIXB + ((c4 ^ (((x >> 2) & 1) << 1)) << 4);
  }

  f32x4 acc[MB];
  // The plain epilogue's stores are DEFERRED: a tile's 8 x 16 bytes per thread stay in the staging area (rewritten only behind
  // the next tile's two barriers) and leave from inside the next tile's first-half MFMA stream, one (staging read, store) pair
  // per three row-visits, instead of as a burst in front of it (timing-only builds: the burst cost 18 % of the kernel).
  T* pend = nullptr;                                                     // this thread's first chunk of the tile waiting in staging
  bool has_pend = false;                                                 // (workgroup-uniform)
  constexpr int NSTORE = BM * (BN / EPC) / 256;                          // 8: store `it` = tile row `it`, pixel tid >> 3, chunk tid & 7
  const size_t row_stride = (size_t)a.Wo * a.ldo;
  const unsigned char* const cst_rd = lds + 2 * Cfg::SLAB_BYTES + (tid >> 3) * C_PITCH + (tid & 7) * 16;
  auto store_pending = [&](int it) __attribute__((always_inline)) {
    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(cst_rd + it * TW * C_PITCH);
#ifdef WS64_T_NOSTORE                                                    // timing-only build: results never leave (one guard store)
    if (v.v[0] == (T)12345.f)
#endif
    st16<T>(pend + it * row_stride, v);
  };
#ifdef WS64_SKEW                                                           // A/B: the second workgroup of a CU starts half a tile late
  if (blockIdx.x >= (gridDim.x >> 1)) {
#pragma unroll 1
    for (int i = 0; i < WS64_SKEW; ++i) __builtin_amdgcn_s_sleep(32);
  }
#endif
  for (int sp = sp_begin; sp < sp_end; ++sp) {
    int t = sp;
    const int tx = t % TXN; t /= TXN;
    const int ty = t % TYN;
    const int n = t / TYN;
    const int y0 = ty * TH, x0 = tx * TW;
    // descriptor + lane offsets of a tile's patch
    auto patch_of = [&](int sp2, bufdesc_t& desc, unsigned (&off)[P_IT]) __attribute__((always_inline)) {
      int t2 = sp2;
      const int tx2 = t2 % TXN; t2 /= TXN;
      const int ty2 = t2 % TYN;
      const int n2 = t2 / TYN;
      const int yy0 = ty2 * TH - 1, xx0 = tx2 * TW - 1;                  // logical coordinates of the patch origin
      const long long org = ((long long)n2 * a.Hi * a.Wi + (long long)(yy0 >> a.up) * a.Wi + (xx0 >> a.up)) * a.ldi;   // (floor shifts)
      desc = make_buf(in + org);
#pragma unroll
      for (int i = 0; i < P_IT; ++i) {
        const int py = p_yx[i] & 255, px = (p_yx[i] >> 8) & 255;
        const bool ok = (p_yx[i] >> 16) && (unsigned)(yy0 + py) < (unsigned)a.Hlog && (unsigned)(xx0 + px) < (unsigned)a.Wlog;
        off[i] = ok ? p_rel[i] : DMA_PAD;
      }
    };
    auto issue_piece = [&](const bufdesc_t& desc, const unsigned (&off)[P_IT], int slab, int i) __attribute__((always_inline)) {
      const int piece = min(wave + 4 * i, P_INSTR - 1);
#ifdef WS64_T_NODMA                                                      // timing-only build: stale operands after the first tile
      if (sp != sp_begin) return;
#endif
      dma16_buf(desc, off[i], (unsigned)slab * 64u, lds0 + slab * SLAB + piece * 1024);
    };
    bufdesc_t desc_cur;
    unsigned off_cur[P_IT];
    patch_of(sp, desc_cur, off_cur);
    const bool first = sp == sp_begin;
    if (first) {                                                         // the first tile's slabs (later ones arrive during the previous tile)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < P_IT; ++i) issue_piece(desc_cur, off_cur, s2, i);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    bufdesc_t desc_nxt = desc_cur;
    unsigned off_nxt[P_IT];
    const bool has_next = sp + 1 < sp_end;
    if (has_next) patch_of(sp + 1, desc_nxt, off_nxt);
    else {
#pragma unroll
      for (int i = 0; i < P_IT; ++i) off_nxt[i] = DMA_PAD;               // (zeros into buffer A: nobody reads them)
    }

    // ---- one slab = 3 column shifts x 10 patch rows; a row's two fragments are read two row-visits ahead of their MFMAs ------
    auto slab_phase = [&](auto slab_tag) __attribute__((always_inline)) {
      constexpr int s = decltype(slab_tag)::value;
      const unsigned char* pa = lds + s * SLAB;
      bf16x8 fr[3][XB];                                                  // rolling window over row-visits
      auto rd = [&](auto v_tag) __attribute__((always_inline)) {
        constexpr int v = decltype(v_tag)::value;                       // visit = pw * 10 + pr
        constexpr int pw = v / PH, pr = v % PH;
#pragma unroll
        for (int xb = 0; xb < XB; ++xb)
          fr[v % 3][xb] = *reinterpret_cast<const bf16x8*>(pa + ((fa[pw] ^ ((pr & 1) << 5)) + pr * ROWB + xb * 16 * PIXB));
      };
      rd(WsI<0>{});
      rd(WsI<1>{});
      ws_static_for<0, 3 * PH>([&](auto v_tag) __attribute__((always_inline)) {
        constexpr int v = decltype(v_tag)::value;
        constexpr int pw = v / PH, pr = v % PH;
        if constexpr (v + 2 < 3 * PH) rd(WsI<v + 2>{});
        // during slab 1: the next tile's slab 0 -> buffer A (free since the mid-tile barrier), one piece per five row-visits
        if constexpr (s == 1 && v % 5 == 0 && v / 5 < P_IT) issue_piece(desc_nxt, off_nxt, 0, v / 5);
        // during slab 0: the previous tile's deferred stores
        if constexpr (s == 0 && v % 3 == 1 && v / 3 < NSTORE) {
          if (has_pend) store_pending(v / 3);
        }
#pragma unroll
        for (int xb = 0; xb < XB; ++xb)
#pragma unroll
          for (int ph = 0; ph < 3; ++ph) {
            const int orow = pr - ph;
            if (orow >= 0 && orow < TH) {
              f32x4& c = acc[orow * XB + xb];
              // the very first MFMA of a block takes the bias as its C operand (slab 0, column 0, tap row 0)
              c = mfma_16x16x32<T>(wf[s][ph * 3 + pw], fr[v % 3][xb], (s == 0 && pw == 0 && ph == 0) ? bias4 : c);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    slab_phase(WsI<0>{});
    // slab 1 was requested behind the previous tile's last MFMA, in FRONT of that tile's stores: the counted wait leaves exactly
    // those stores in flight (8 x 16 B per thread, 2 in the 2x2-sum epilogue; a smaller count would only wait for stores too).
    // (first tile: both slabs landed in the prologue)
    if (!first) {
      if (a.pool2) wait_vmcnt<(BM / 4) * (BN / EPC) / 256>(); else wait_vmcnt<BM * (BN / EPC) / 256>();
    }
    __builtin_amdgcn_s_barrier();                                        // ... for every wave; buffer A is free
    slab_phase(WsI<1>{});
    wait_vmcnt<0>();                                                     // next tile's slab 0 has landed (the previous tile's stores are long gone)
    __builtin_amdgcn_s_barrier();                                        // buffer B is free

    // ---- epilogue: a lane holds, per block, FOUR CONSECUTIVE CHANNELS (16 wave + 4 c4 .. + 3) of pixel (row, 16 xb + l16) ----
    unsigned char* const cst = lds + 2 * SLAB;
    struct alignas(8) Pack4 { T v[4]; };
    auto finish = [&](auto relu_tag, auto stats_tag) __attribute__((always_inline)) {
      constexpr bool RELU = decltype(relu_tag)::value, STATS = decltype(stats_tag)::value;
      f32x2 sm[2], sq[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) { sm[jj] = f32x2{0.f, 0.f}; sq[jj] = f32x2{0.f, 0.f}; }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int row = mb * 16 + l16;                                   // tile pixel: block mb = (output row, xb)
        Pack4 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pk.v[r] = from_f32<T>(RELU ? __builtin_amdgcn_fmed3f(acc[mb][r], 0.f, INFINITY) : acc[mb][r]);
        if constexpr (STATS) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const f32x2 v = {to_f32<T>(pk.v[2 * jj]), to_f32<T>(pk.v[2 * jj + 1])};
            sm[jj] += v;
            sq[jj] += v * v;
          }
        }
        *reinterpret_cast<Pack4*>(cst + row * C_PITCH + (wave * 16 + 4 * c4) * 2) = pk;
      }
      if constexpr (STATS) {                                             // a wave has seen all 256 pixels of its 16 channels: no exchange
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = row16_sum(sm[r >> 1][r & 1]), s2 = row16_sum(sq[r >> 1][r & 1]);
          if (l16 == 0) {
            float* st = a.stats + (size_t)sp * 2 * a.Co + n0 + wave * 16 + 4 * c4 + r;
            st[0] = s1;
            st[a.Co] = s2;
          }
        }
      }
    };
    using Yes = std::true_type;
    using No = std::false_type;
    if (a.stats) {
      if (a.relu) finish(Yes{}, Yes{}); else finish(No{}, Yes{});
    } else {
      if (a.relu) finish(Yes{}, No{}); else finish(No{}, No{});
    }
    // the next tile's slab 1 -> buffer B: the ONLY vector-memory operations between here and the next mid-tile wait are the
    // tile's stores below (the statistics went out above)
#pragma unroll
    for (int i = 0; i < P_IT; ++i) issue_piece(desc_nxt, off_nxt, 1, i);
    __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0): the staging writes are done
    __builtin_amdgcn_s_barrier();
    constexpr int CPRC = BN / EPC;
    if (a.pool2) {       // gradient of a fused nearest x2 up-sampling: 2x2 output groups summed into the half-resolution tensor
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
            const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(cst + row * C_PITCH + c * 16);
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
    } else if (a.accumulate) {
#pragma unroll
      for (int it = 0; it < NSTORE; ++it) {
        const int id = tid + it * 256;
        const int row = id / CPRC, c = id - row * CPRC;
        const int py = row / TW, px = row - py * TW;
        T* p = out + ((size_t)(n * a.Ho + y0 + py) * a.Wo + x0 + px) * a.ldo + n0 + c * EPC;
        Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(cst + row * C_PITCH + c * 16);
        const Vec16<T> o = ld16<T>(p);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) + to_f32<T>(o.v[e]));
        st16<T>(p, v);
      }
    } else {             // plain: the stores ride in the next tile's MFMA stream (store_pending), or behind the loop
      pend = out + ((size_t)(n * a.Ho + y0) * a.Wo + x0 + (tid >> 3)) * a.ldo + n0 + (tid & 7) * EPC;
      has_pend = true;
    }
    // (no barrier: the staging area is written again only behind the next tile's two barriers)
  }
  if (has_pend) {
#pragma unroll
    for (int it = 0; it < NSTORE; ++it) store_pending(it);
  }
}

// 2 workgroups per CU, each a contiguous range of spatial tiles of one channel tile
static bool ws64_shape(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co) {
  return Ci == 64 && Co % 64 == 0 && Wo % 32 == 0 && Ho % 8 == 0;
}

template <typename T>
static int launch_ws64(const ConvArgs& a, hipStream_t s, int cus) {
  const int NT = a.Co / 64;
  const int S = a.N * (a.Ho / 8) * (a.Wo / 32);
  int groups = (2 * cus / NT) & ~7;                                      // spatial groups: a multiple of 8 (one share per XCD)
  if (groups > (S & ~7)) groups = S & ~7;
  if (groups < 8) MI355_FAIL(MI355_ERR_ARG, "conv3x3_ws64: %d spatial tiles are too few for the persistent kernel", S);
  constexpr int lds_bytes = Ws64Cfg::LDS_BYTES;
  static const hipError_t configured = hipFuncSetAttribute((const void*)conv3x3_ws64_kernel<T>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (configured != hipSuccess)
    MI355_FAIL((int)configured, "conv3x3_ws64: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(configured));
  hipLaunchKernelGGL((conv3x3_ws64_kernel<T>), dim3(groups * NT), dim3(256), lds_bytes, s, a, groups);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
