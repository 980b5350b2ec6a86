// 3x3 / stride-1 / pad-1 convolution (forward and data gradient) for Ci = 64 / 128, bf16 / fp16, gfx950 — the WEIGHT-STATIONARY kernel.
//
// The 64- and 128-channel layers at 256 x 256 / 128 x 128 are the largest tensors of every U-Net and sit at the HBM ridge: a
// 256-pixel tile of a 64 -> 64 layer is only 288 MFMAs per wave, and conv3x3_halo_rw_kernel pays for it 18 weight DMA pieces per
// wave (every tile of the layer fetches the same 72 KiB of weights again: more L2 -> LDS bytes than the activations), six counted
// waits + barriers, a prologue and an epilogue (0.41 of peak).  Here the weights never move again:
//
//   weights     : a wave owns 16 output channels for ALL of K = 9 x CI: 18 (CI = 64) or 36 (CI = 128) MFMA A-fragments = 72 / 144
//                 registers per lane, loaded once per workgroup straight from global memory.  No weight bytes in LDS, no weight
//                 fragment reads, no weight DMA.
//   workgroup   : 4 waves = 64 output channels, PERSISTENT: it walks a contiguous range of TH x 32-pixel tiles of one channel
//                 tile (grid = 2 workgroups per CU, <= 80 KB of LDS and <= 256 registers each: the two run independently and
//                 fill each other's epilogues and waits).  TH = 8 for CI = 64 (64 accumulator registers), TH = 4 for CI = 128
//                 (32: what is left beside 144 weight registers)
//   wave tile   : ALL TH x 32 pixels x 16 channels; per (32-channel slab, patch column) step the TH + 2 patch-row fragments of a
//                 16-pixel column are read once and feed up to three output rows, no weight reads
//   patch       : CI / 32 slabs [TH + 2 rows][36-pixel pitch][64 B] (conv3x3_halo_pp128.hpp's layout: the swizzle bit of a
//                 pixel is (row + (x >> 2)) & 1, so a fragment address is ONE lane register per column shift + immediates),
//                 each brought by LDS-DMA pieces through a buffer descriptor based at the patch origin
//   schedule    : slab k of tile t + 1 is requested while slab k + 1 of tile t is multiplied (its buffer is free behind the barrier
//                 that ends phase k), the last slab right behind tile t's last MFMA; the C tile has a staging area of its own
//                 (136-B pixel pitch: conflict-free 8-byte writes) and its stores are DEFERRED into the next tile's first-phase
//                 MFMA stream.  gfx950 counts stores in vmcnt, in issue order with the DMA pieces: every wait for a slab is a
//                 COUNTED vmcnt that leaves exactly the operations issued behind that slab's pieces in flight — the later slabs of
//                 the tile and the previous tile's stores — so that no wait ever drains a fresh store (the coupling that sank
//                 round 2's persistent variants).  CI / 32 + 1 barriers per tile (the 4-wave halo kernel: 3 CI / 32 + 2).
#pragma once
#include <type_traits>

#include "common.hpp"

template <int CI_, int TH_> struct WsCfg {
  static constexpr int TH = TH_, TW = 32, BN = 64, CI = CI_, NS = CI_ / 32;
  static constexpr int PWL = 36, PH = TH + 2;
  static constexpr int NPIX = PH * PWL;                                  // pixel slots per slab (360 / 216)
  static constexpr int P_INSTR = (NPIX + 15) / 16;                       // DMA pieces of 16 pixels (23 / 14)
  static constexpr int SLAB_BYTES = P_INSTR * 1024;
  static constexpr int C_PITCH = BN * 2 + 8;                             // 34 dwords: the 16 pixel rows of an 8-byte staging write hit 32 distinct banks
  static constexpr int C_BYTES = TH * TW * C_PITCH;                      // staging area of its own
  static constexpr int LDS_BYTES = NS * SLAB_BYTES + C_BYTES;            // 80 KiB (CI 64, TH 8) / 73 KiB (CI 128, TH 4): two workgroups per CU
};

template <int I> using WsI = std::integral_constant<int, I>;
template <int B, int E, typename F> __device__ __forceinline__ void ws_static_for(F&& f) {
  if constexpr (B < E) {
    f(WsI<B>{});
    ws_static_for<B + 1, E>(f);
  }
}

template <typename T, int CI, int TH_>
__global__ __launch_bounds__(256, 2) void conv3x3_ws_kernel(const ConvArgs a, const int groups) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  typedef WsCfg<CI, TH_> Cfg;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, BN = Cfg::BN, EPC = 8, BM = TH * TW, NS = Cfg::NS;
  constexpr int PWL = Cfg::PWL, PH = Cfg::PH, NPIX = Cfg::NPIX, PIXB = 64;
  constexpr int P_INSTR = Cfg::P_INSTR, P_IT = (P_INSTR + 3) / 4;        // pieces per wave and slab (a piece index past the slab repeats the last one)
  constexpr int SLAB = Cfg::SLAB_BYTES, ROWB = PWL * PIXB;              // 2304 B per patch row
  constexpr int C_PITCH = Cfg::C_PITCH;
  constexpr int XB = TW / 16, MB = TH * XB;                              // accumulator blocks of 16 pixels x 16 channels
  constexpr int FRD = CI == 64 ? 3 : 2;                                  // row-visits of fragments in registers (CI = 128: 144 weight registers leave room for two)
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
  bf16x8 wf[NS][9];
  {
    const T* wrow = wk + ((size_t)(n0 + wave * 16 + l16) * 9) * Cfg::CI + c4 * EPC;
#pragma unroll
    for (int s = 0; s < NS; ++s)
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
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) asm volatile("" : "+v"(wf[s][tp]));
  asm volatile("" : "+v"(bias4));

  // ---- DMA lane geometry: piece p = LDS pixel slots [16 p, +16) x 64 B; lane -> (slot pixel, 16-byte chunk) ------------------
  // source offsets are relative to the PATCH ORIGIN (pixel (y0 - 1, x0 - 1) of the tile, halved when the x2 up-sampling is
  // folded in), which is the base of the per-tile buffer descriptor: always >= 0; validity (image border, the two pad columns,
  // the tail of the last piece) is per tile and turns the offset into the out-of-range value the hardware zero-fills
  const int lrow = lane >> 2, slot = lane & 3;
  unsigned p_rel[P_IT];                                                  // (or DMA_PAD where the slot never holds a pixel: pad columns, the tail)
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int piece = min(wave + 4 * i, P_INSTR - 1);
    const int q = piece * 16 + lrow;
    const int py = q / PWL, px = q - py * PWL;
    const int ry = a.up ? (py + 1) >> 1 : py, rx = a.up ? (px + 1) >> 1 : px;
    p_rel[i] = (q < NPIX && px < TW + 2) ? (unsigned)(((ry * a.Wi + rx) * a.ldi + (slot ^ (((q >> 2) & 1) << 1)) * EPC) * 2) : DMA_PAD;
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
  // The plain epilogue's stores are DEFERRED: a tile's TH x 16 bytes per thread stay in the staging area (rewritten only behind
  // the next tile's barriers) and leave from inside the next tile's first-phase MFMA stream, one (staging read, store) pair
  // per three row-visits, instead of as a burst in front of it (timing-only builds: the burst cost 18 % of the kernel).
  T* pend = nullptr;                                                     // this thread's first chunk of the tile waiting in staging
  bool has_pend = false;                                                 // (workgroup-uniform)
  constexpr int NSTORE = BM * (BN / EPC) / 256;                          // TH: store `it` = tile row `it`, pixel tid >> 3, chunk tid & 7
  constexpr int NSTORE_POOL = (BM / 4) * (BN / EPC) / 256;               // 2x2-sum epilogue: 2 (TH = 8) or 1 (TH = 4) stores per thread
  static_assert(NSTORE == TH && NSTORE_POOL * 256 == (BM / 4) * (BN / EPC), "store counts the counted waits rely on");
  const size_t row_stride = (size_t)a.Wo * a.ldo;
  const unsigned char* const cst_rd = lds + NS * Cfg::SLAB_BYTES + (tid >> 3) * C_PITCH + (tid & 7) * 16;
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
  // fused BatchNorm statistics: per-lane running sums over ALL tiles of the workgroup (one channel tile, so a lane always meets the
  // same four channels) — ONE partial row per workgroup behind the loop (row kk: mi355_conv2d_igemm_stat_rows == groups), not one
  // per tile: 512 rows instead of 8192 at 256 x 256 x 32 images (no pre-fold launch, a 16 x smaller fold)
  f32x2 wsm[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}}, wsq[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
  for (int sp = sp_begin; sp < sp_end; ++sp) {
    // tiles in COLUMN order (ty fastest): consecutive tiles of a workgroup's range are vertical neighbours, so the two halo rows a
    // tile shares with its predecessor were fetched one tile ago and are still in the XCD's L2 (-DWS_ROW_ORDER: tx fastest, A/B)
    int t = sp;
#ifdef WS_ROW_ORDER
    const int tx = t % TXN; t /= TXN;
    const int ty = t % TYN;
    const int n = t / TYN;
#else
    const int ty = t % TYN; t /= TYN;
    const int tx = t % TXN;
    const int n = t / TXN;
#endif
    const int y0 = ty * TH, x0 = tx * TW;
    // descriptor + lane offsets of a tile's patch
    auto patch_of = [&](int sp2, bufdesc_t& desc, unsigned (&off)[P_IT]) __attribute__((always_inline)) {
      int t2 = sp2;
#ifdef WS_ROW_ORDER
      const int tx2 = t2 % TXN; t2 /= TXN;
      const int ty2 = t2 % TYN;
      const int n2 = t2 / TYN;
#else
      const int ty2 = t2 % TYN; t2 /= TYN;
      const int tx2 = t2 % TXN;
      const int n2 = t2 / TXN;
#endif
      const int yy0 = ty2 * TH - 1, xx0 = tx2 * TW - 1;                  // logical coordinates of the patch origin
      const long long org = ((long long)n2 * a.Hi * a.Wi + (long long)(yy0 >> a.up) * a.Wi + (xx0 >> a.up)) * a.ldi;   // (floor shifts)
      desc = make_buf(in + org);
#pragma unroll
      for (int i = 0; i < P_IT; ++i) {                                   // (py, px) recomputed: two registers per piece less across the tile loop
        const int q = min(wave + 4 * i, P_INSTR - 1) * 16 + lrow;
        const int py = q / PWL, px = q - py * PWL;
        const bool ok = (unsigned)(yy0 + py) < (unsigned)a.Hlog && (unsigned)(xx0 + px) < (unsigned)a.Wlog;
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
    const bool first = sp == sp_begin;
    if (first) {                                                         // the first tile's slabs (later ones arrive during the previous tile)
      bufdesc_t desc_cur;
      unsigned off_cur[P_IT];
      patch_of(sp, desc_cur, off_cur);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
        for (int i = 0; i < P_IT; ++i) issue_piece(desc_cur, off_cur, s2, i);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    bufdesc_t desc_nxt = make_buf(in, false);
    unsigned off_nxt[P_IT];
    const bool has_next = sp + 1 < sp_end;
    if (has_next) patch_of(sp + 1, desc_nxt, off_nxt);
    else {
#pragma unroll
      for (int i = 0; i < P_IT; ++i) off_nxt[i] = DMA_PAD;               // (zeros into the slab buffers: nobody reads them)
    }

    // ---- one slab = 3 column shifts x (TH + 2) patch rows; a row's two fragments are read two row-visits ahead of their MFMAs ----
    auto slab_phase = [&](auto slab_tag) __attribute__((always_inline)) {
      constexpr int s = decltype(slab_tag)::value;
      const unsigned char* pa = lds + s * SLAB;
      bf16x8 fr[FRD][XB];                                                // rolling window over row-visits
      auto rd = [&](auto v_tag) __attribute__((always_inline)) {
        constexpr int v = decltype(v_tag)::value;                       // visit = pw * PH + pr
        constexpr int pw = v / PH, pr = v % PH;
#ifdef WS_T_NOSHIFTREAD                                                  // timing-only build: the shifted columns reuse stale fragments (a third of the LDS reads)
        if constexpr (pw > 0) return;
#endif
#pragma unroll
        for (int xb = 0; xb < XB; ++xb)
          fr[v % FRD][xb] = *reinterpret_cast<const bf16x8*>(pa + ((fa[pw] ^ ((pr & 1) << 5)) + pr * ROWB + xb * 16 * PIXB));
      };
      ws_static_for<0, FRD - 1>([&](auto v_tag) __attribute__((always_inline)) { rd(v_tag); });
      ws_static_for<0, 3 * PH>([&](auto v_tag) __attribute__((always_inline)) {
        constexpr int v = decltype(v_tag)::value;
        constexpr int pw = v / PH, pr = v % PH;
        if constexpr (v + FRD - 1 < 3 * PH) rd(WsI<v + FRD - 1>{});
        // during slab s >= 1: the next tile's slab s - 1 into its buffer (free since the barrier that ended phase s - 1), the
        // pieces spread over the phase's row-visits
        constexpr int DV = 3 * PH / P_IT, SV = 3 * PH / (NSTORE + 1);
        if constexpr (s >= 1 && v % DV == 0 && v / DV < P_IT) issue_piece(desc_nxt, off_nxt, s - 1, v / DV);
        // during slab 0: the previous tile's deferred stores
        if constexpr (s == 0 && v % SV == 1 && v / SV < NSTORE) {
          if (has_pend) store_pending(v / SV);
        }
#pragma unroll
        for (int xb = 0; xb < XB; ++xb)
#pragma unroll
          for (int ph = 0; ph < 3; ++ph) {
            const int orow = pr - ph;
            if (orow >= 0 && orow < TH) {
              f32x4& c = acc[orow * XB + xb];
              // the very first MFMA of a block takes the bias as its C operand (slab 0, column 0, tap row 0)
              c = mfma_16x16x32<T>(wf[s][ph * 3 + pw], fr[v % FRD][xb], (s == 0 && pw == 0 && ph == 0) ? bias4 : c);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    slab_phase(WsI<0>{});
    // Slab s of this tile was requested during the previous tile (phase s + 1; the last one behind its last MFMA).  What this wave
    // has issued SINCE, in order: the later slabs of this tile, the previous tile's stores (deferred into phase 0 above, or issued
    // by its epilogue), the next tile's slabs 0 .. s - 2 — (NS - 2) P_IT pieces + the stores, whatever s: the counted wait leaves
    // exactly those in flight (a smaller count would only wait for stores too).  (First tile: every slab landed in the prologue.)
    ws_static_for<1, NS>([&](auto s_tag) __attribute__((always_inline)) {
      if (!first) {
        if (a.pool2) wait_vmcnt<(NS - 2) * P_IT + NSTORE_POOL>(); else wait_vmcnt<(NS - 2) * P_IT + NSTORE>();
      }
      __builtin_amdgcn_s_barrier();                                      // ... for every wave; the previous phase's buffer is free
      slab_phase(s_tag);
    });
    wait_vmcnt<(NS - 2) * P_IT>();                                       // the next tile's slab 0 has landed (behind it: its slabs 1 .. NS - 2)
    __builtin_amdgcn_s_barrier();                                        // the last slab's buffer is free

    // ---- epilogue: a lane holds, per block, FOUR CONSECUTIVE CHANNELS (16 wave + 4 c4 .. + 3) of pixel (row, 16 xb + l16) ----
    unsigned char* const cst = lds + NS * SLAB;
    struct alignas(8) Pack4 { T v[4]; };
    auto finish = [&](auto relu_tag, auto stats_tag) __attribute__((always_inline)) {
      constexpr bool RELU = decltype(relu_tag)::value, STATS = decltype(stats_tag)::value;
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
            wsm[jj] += v;
            wsq[jj] += v * v;
          }
        }
        *reinterpret_cast<Pack4*>(cst + row * C_PITCH + (wave * 16 + 4 * c4) * 2) = pk;
      }
    };
    using Yes = std::true_type;
    using No = std::false_type;
    if (a.stats) {
      if (a.relu) finish(Yes{}, Yes{}); else finish(No{}, Yes{});
    } else {
      if (a.relu) finish(Yes{}, No{}); else finish(No{}, No{});
    }
    // the next tile's last slab: the ONLY vector-memory operations between here and the next tile's phase-1 wait are the tile's
    // stores (the statistics went out above)
#pragma unroll
    for (int i = 0; i < P_IT; ++i) issue_piece(desc_nxt, off_nxt, NS - 1, i);
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
    } else {
      // this thread's first chunk of the tile: row `it` of the tile is `it` pixel rows further on (store_pending's addressing)
      T* const ptile = out + ((size_t)(n * a.Ho + y0) * a.Wo + x0 + (tid >> 3)) * a.ldo + n0 + (tid & 7) * EPC;
      if (a.accumulate) {
        // all the old values first (the accumulators are dead here: registers to spare) — read one row at a time, every load sat
        // behind the previous row's store to the same tensor: TH dependent round trips per tile (R2AttU_Net's first application
        // of a recurrent block: 0.104 ms against 0.070 for the same convolution without the accumulation)
        Vec16<T> old[NSTORE];
#pragma unroll
        for (int it = 0; it < NSTORE; ++it) old[it] = ld16<T>(ptile + it * row_stride);
#pragma unroll
        for (int it = 0; it < NSTORE; ++it) {
          Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(cst_rd + it * TW * C_PITCH);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) + to_f32<T>(old[it].v[e]));
          st16<T>(ptile + it * row_stride, v);
        }
      } else {           // plain: the stores ride in the next tile's MFMA stream (store_pending), or behind the loop
        pend = ptile;
        has_pend = true;
      }
    }
    // (no barrier: the staging area is written again only behind the next tile's phase barriers)
  }
  if (has_pend) {
#pragma unroll
    for (int it = 0; it < NSTORE; ++it) store_pending(it);
  }
  if (a.stats) {                                                         // a wave has seen every pixel of its 16 channels: no exchange between waves
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s1 = row16_sum(wsm[r >> 1][r & 1]), s2 = row16_sum(wsq[r >> 1][r & 1]);
      if (l16 == 0) {
        float* st = a.stats + (size_t)kk * 2 * a.Co + n0 + wave * 16 + 4 * c4 + r;
        st[0] = s1;
        st[a.Co] = s2;
      }
    }
  }
}

// 2 workgroups per CU, each a contiguous range of spatial tiles of one channel tile
static int ws_tile_rows(int Ci, int Ho, int Wo, int Co) {               // 8 / 4: the tile height of the instantiation that serves the shape; 0: none
  if (Co % 64 != 0 || Wo % 32 != 0) return 0;
  if (Ci == 64 && Ho % 8 == 0) return 8;
  if (Ci == 128 && Ho % 4 == 0) return 4;
  return 0;
}

// spatial groups of a launch (= its statistics rows): a multiple of 8 (one share per XCD), every group gets at least one tile
static int ws_groups(int N, int Ho, int Wo, int Co, int th, int cus) {
  const int NT = Co / 64;
  const int S = N * (Ho / th) * (Wo / 32);
  int groups = (2 * cus / NT) & ~7;
  if (groups > (S & ~7)) groups = S & ~7;
  return groups;
}

template <typename T, int CI, int TH>
static int launch_ws(const ConvArgs& a, hipStream_t s, int cus) {
  const int NT = a.Co / 64;
  const int S = a.N * (a.Ho / TH) * (a.Wo / 32);
  const int groups = ws_groups(a.N, a.Ho, a.Wo, a.Co, TH, cus);
  if (groups < 8) MI355_FAIL(MI355_ERR_ARG, "conv3x3_ws: %d spatial tiles are too few for the persistent kernel", S);
  constexpr int lds_bytes = WsCfg<CI, TH>::LDS_BYTES;
  static const hipError_t configured = hipFuncSetAttribute((const void*)conv3x3_ws_kernel<T, CI, TH>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (configured != hipSuccess)
    MI355_FAIL((int)configured, "conv3x3_ws: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(configured));
  hipLaunchKernelGGL((conv3x3_ws_kernel<T, CI, TH>), dim3(groups * NT), dim3(256), lds_bytes, s, a, groups);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
