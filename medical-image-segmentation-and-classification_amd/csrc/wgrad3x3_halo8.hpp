// Weight gradient of a 3x3 / stride-1 / pad-1 convolution, bf16 / fp16, gfx950 — all nine taps, 512-thread workgroups.
//
//   dW[co][tap][ci] = sum_{n,y,x} dY[n,y,x,co] * X[n, y+kh-1, x+kw-1, ci]
//
// The eight-wave form of wgrad3x3_halo_kernel (wgrad3x3_halo.hpp; reference call site: `convolution_backward` under
// /root/reference/utils/helpers.py:329 for the 3x3 layers of models/segmentation_models/AttentionUNet.py:4-13).  A workgroup still
// owns ONE 64(co) x 64(ci) tile of dW for all nine taps, but walks 64-pixel row segments: waves 0-3 take the left 32 pixels
// of a row as their MFMA K block, waves 4-7 the right 32, each wave with the 32 x 32 quadrant x 9 taps = 144 accumulator
// registers of the four-wave kernel and the same row-step program.  What that buys:
//   * two waves per SIMD: one wave's fragment reads, counted waits and the per-row barrier sit in the shadow of its
//     partner's MFMAs (the four-wave kernel ran one wave per SIMD: matrix pipes 62 % busy);
//   * ONE halo per 64 pixels: a row image is 72 pixels of X for 64 pixels of dY (the four-wave kernel: 40 for 32), so the
//     L2 -> LDS stream and the HBM fetch of X shrink by 10 %, and the 17-18 DMA pieces of a row are spread over eight
//     waves with no dummy pieces (a wave issues two or three per row and counts its own);
//   * the two halves' partial tiles are added inside the workgroup (through the dead row rings in LDS) before anything
//     is written: one fp32 slab per workgroup as before, but a workgroup now covers twice the pixels per unit time.
// W32: images 32 pixels wide.  A 64-pixel "row" is then row y of TWO images side by side (work item = image pair x row
// band): the X row image holds two 40-pixel segments with their own halos, waves 4-7 read the second one.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "wgrad3x3_halo.hpp"

struct Wgrad8Lds {                                   // dynamic LDS of wgrad3x3_halo8_kernel (shared with its launcher)
  static constexpr int NR = 6;
  static constexpr int XPX = 80, DPX = 64;           // pixels per X / dY row image (X: 72 used at W % 64 == 0, 2 x 40 at W == 32)
  static constexpr int BYTES = NR * (XPX + DPX) * 128 + 4096;
};

template <typename T, bool W32>
__global__ __launch_bounds__(512, 2) void wgrad3x3_halo8_kernel(const Wgrad3Args a) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  constexpr int XPX = Wgrad8Lds::XPX, XROW = XPX * 128, DROW = Wgrad8Lds::DPX * 128;
  constexpr int XPIECES = W32 ? 10 : 9;              // 1-KiB DMA pieces per X row: waves 0..7, then waves 0 .. XPIECES-9
  constexpr int XHALF = W32 ? 40 : 32;               // X-image pixel offset of the right half's K block
  constexpr int NR = Wgrad8Lds::NR, PF = 5;          // rings of NR rows, row r + PF fetched during step r (wgrad3x3_halo.hpp; six rows: every ring offset fits the
                                                     // 16-bit immediate of a ds_read — eight-row rings of 80 + 64 pixels spilled ten registers)
  constexpr int X_BYTES = NR * XROW, D_BYTES = NR * DROW;
  constexpr int ZERO_IMG = X_BYTES + D_BYTES;
  static_assert(Wgrad8Lds::BYTES == X_BYTES + D_BYTES + 4096, "launcher and kernel disagree on the LDS size");
  
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const xr = lds;
  unsigned char* const dr = lds + X_BYTES;
  unsigned char* const zimg = lds + ZERO_IMG;
  const unsigned lds_x = lds_addr(xr), lds_d = lds_addr(dr);      // DMA destinations: LDS byte addresses

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, w4 = wave & 3;
  const int l16 = lane & 15, c4 = lane >> 4;
  const int qo = w4 >> 1, qi = w4 & 1;
  const int ciTiles = (a.Ci + 63) / 64;
  const int bid = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bx = bid % gridDim.x, by = bid / gridDim.x;
  // Tile order inside one row band: an XCD owns Q = grid / 8 consecutive `bid`s (xcd_tile), i.e. Q tiles of one band when the
  // band has at least Q.  As a BH x BW block of the (Co / 64) x (Ci / 64) tile grid they fetch BH / coT of dY and BW / ciT of X
  // through that XCD's L2 — least for the squarest block (1024 -> 512 @32²: 4 x 4 instead of 1 x 16 = a quarter of X + half of dY
  // instead of all of X + an eighth of dY).  -DWG8_ROW_TILES: row-major tiles as before (A/B).
  int tco = bx / ciTiles, tci = bx % ciTiles;
#ifndef WG8_ROW_TILES
  {
    const int coT = gridDim.x / ciTiles, Q = (int)(gridDim.x * gridDim.y) >> 3;
    if (Q >= 4 && (int)gridDim.x % Q == 0 && ((gridDim.x * gridDim.y) & 7) == 0) {
      int BW = 0, best = 1 << 30;
      for (int w = 1; w <= Q; w <<= 1) {
        const int h = Q / w;
        if (w * h == Q && ciTiles % w == 0 && coT % h == 0 && w + h < best) { best = w + h; BW = w; }
      }
      if (BW) {
        const int BH = Q / BW, blk = bx / Q, j = bx % Q, bpr = ciTiles / BW;
        tco = (blk / bpr) * BH + j / BW;
        tci = (blk % bpr) * BW + j % BW;
      }
    }
  }
#endif
  const int co0 = tco * 64, ci0 = tci * 64;
  auto swz = [](int px) { return (((px >> 1) & 1) | (((px >> 3) & 1) << 1)) << 1; };     // XOR on the 16-B chunk index

  const int lpx = lane >> 3, slot = lane & 7;       // DMA: a 1-KiB piece = 8 pixels x 128 B; lane -> (pixel, 16-B slot)
  const int TXN = W32 ? 1 : a.W / 64, BANDS = a.H / a.RB;
  if (tid < 256) *reinterpret_cast<uint4*>(zimg + tid * 16) = make_uint4(0, 0, 0, 0);      // the all-zero dY row image (32 pixels)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       // (written before the first item's barrier)
  const bool three = 8 + wave < XPIECES;             // this wave moves a third piece per row (wave-uniform)

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transpose-read lane geometry: 16-lane block b = c4 reads pixels 8b + tq (+4), channels col0 + 4*tp .. +3
  const int tq = l16 >> 2, tp = l16 & 3;
  const int pl = 8 * c4 + tq;                      // this lane's first pixel inside its half's 32-pixel K block
  const int plx = XHALF * half + pl;               // ... and inside the X row image (minus the 4-pixel halo)
  const int dhalf = half * 32 * 128;               // byte offset of the half's K block inside a dY row image (32 | px: same swizzle)
  auto rd = [&](int img_off, int px, int col) {
    const int chunk = (col >> 3) ^ swz(px);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(lds + img_off + px * 128 + chunk * 16 + (col & 7) * 2));
  };
  auto frag = [&](int img_off, int px, int col) {
    const s16x4 v0 = rd(img_off, px, col), v1 = rd(img_off, px + 4, col);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  const int colA = qo * 32 + 4 * tp, colB = qi * 32 + 4 * tp;     // + 16 * block

  const int item0 = by * a.items_per_block;
  const int item1 = min(a.items, item0 + a.items_per_block);
  for (int item = item0; item < item1; ++item) {
    const int app = item / a.items_per_app;
    const T* __restrict__ x = reinterpret_cast<const T*>(a.xs[app]);
    const T* __restrict__ dy = reinterpret_cast<const T*>(a.dys[app]);
    int t = item - app * a.items_per_app;
    const int band = t % BANDS; t /= BANDS;
    const int tx = t % TXN;
    const int n = t / TXN;
    const int ya = band * a.RB, yb = ya + a.RB, x0 = tx * 64;
    const int nbase = W32 ? 2 * n : n;              // (W32: `n` counts image pairs)
    // per-lane pixel geometry of this item
    const int px_d = 8 * wave + lpx;                 // pixel of the 64-px dY row image
    const int c_d = co0 + 8 * (slot ^ swz(px_d));
    const bool lane_ok_d = c_d < a.Co;
    const int dpix = W32 ? (px_d >> 5) * a.H * a.W + (px_d & 31) : x0 + px_d;      // pixel offset from (image nbase, row r, x 0)
    int xpix[2], c_x[2], xpiece[2];
    bool lane_ok_x[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      xpiece[k] = k == 0 ? wave : 8 + wave;
      const int q = 8 * xpiece[k] + lpx;             // pixel of the X row image
      const int hq = W32 ? q / 40 : 0;
      const int xx = W32 ? q % 40 - 4 : x0 - 4 + q;
      c_x[k] = ci0 + 8 * (slot ^ swz(q));
      lane_ok_x[k] = xpiece[k] < XPIECES && (unsigned)xx < (unsigned)a.W && c_x[k] < a.Ci;
      xpix[k] = hq * a.Hi * a.Wi + (xx >> a.up);
    }

    // DMA sources: a buffer descriptor per tensor whose base is image `nbase` (wave-uniform), a scalar row offset, and ONE
    // 32-bit register per piece holding the lane's offset inside the row — or the always-out-of-range offset where the lane is
    // padding; a row outside the image (X) or the band (dY) takes a descriptor with num_records = 0 (wgrad3x3_halo.hpp).
    const bufdesc_t desc_d = make_buf(dy + (size_t)nbase * a.H * a.W * a.ldy);
    const bufdesc_t desc_x = make_buf(x + (size_t)nbase * a.Hi * a.Wi * a.ldx);
    const unsigned voff_d = lane_ok_d ? (unsigned)((dpix * a.ldy + c_d) * (int)sizeof(T)) : DMA_PAD;
    unsigned voff_x[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) voff_x[k] = lane_ok_x[k] ? (unsigned)((xpix[k] * a.ldx + c_x[k]) * (int)sizeof(T)) : DMA_PAD;
    const unsigned d_stride = (unsigned)(a.W * a.ldy) * (unsigned)sizeof(T), x_stride = (unsigned)(a.Wi * a.ldx) * (unsigned)sizeof(T);
    auto with_rows = [](bufdesc_t d, bool ok) { d[2] = ok ? (int)DMA_PAD : 0; return d; };
    // piece 0: this wave's KiB of dY row r; piece 1: its KiB of X row r; piece 2 (waves 0 .. XPIECES-9): the X row's tail
    auto issue_piece = [&](int piece, int r, unsigned d_soff, unsigned x_soff, int xs, int ds) {
      if (piece == 0) {
        dma16_buf(with_rows(desc_d, r >= ya && r < yb), voff_d, d_soff, lds_d + ds * DROW + wave * 1024);
      } else if (piece == 1 || three) {
        const int k = piece - 1;
        dma16_buf(with_rows(desc_x, (unsigned)r < (unsigned)a.H), voff_x[k], x_soff, lds_x + xs * XROW + xpiece[k] * 1024);
      }
    };
    auto issue_row = [&](int r, int xs, int ds) {
#pragma unroll
      for (int piece = 0; piece < 3; ++piece)
        issue_piece(piece, r, (unsigned)r * d_stride, (unsigned)(r >> a.up) * x_stride, xs, ds);      // (r = -1: a dead offset under num_records = 0)
    };
    // at most K rows of this wave's pieces may still be in flight (a wave counts its own two or three pieces per row)
    auto wait_rows = [&](auto ktag) {
      constexpr int K = decltype(ktag)::value;
      if (three) wait_vmcnt<3 * K>(); else wait_vmcnt<2 * K>();
    };
    // X fragments of tap column kw: [ci block]; dY fragments of a row: [co block]
    auto load_x = [&](int xs, int kw, bf16x8 (&bf)[2]) {
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) bf[bi] = frag(xs * XROW, plx + 3 + kw, colB + 16 * bi);
    };
    auto load_dy = [&](int off, bf16x8 (&af)[2]) {
#pragma unroll
      for (int ao = 0; ao < 2; ++ao) af[ao] = frag(off, pl, colA + 16 * ao);
    };
    bf16x8 dp[2], dc[2], dm[2], dn[2];              // dY rows r+1, r, r-1 (kh = 0, 1, 2) and the incoming r+2
    auto mfma12 = [&](int kw, const bf16x8 (&xk)[2], auto between) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int ao = 0; ao < 2; ++ao)
#pragma unroll
          for (int bi = 0; bi < 2; ++bi)
            mfma_16x16x32_acc<T>(kh == 0 ? dp[ao] : (kh == 1 ? dc[ao] : dm[ao]), xk[bi], acc[kh * 3 + kw][ao][bi]);
        between(kh);
      }
    };
    unsigned d_soff_next = (unsigned)(ya - 1 + PF) * d_stride, x_soff_next = (unsigned)((ya - 1 + PF) >> a.up) * x_stride;      // scalar registers
    int r_next = ya - 1 + PF;
    auto issue_next_piece = [&](int piece, int xs, int ds) {      // row r_next into ring slots xs / ds; the last piece advances
#ifndef WG3_T_NODMA                                            // (timing-only build: stale rows, the no-DMA ceiling of the loop)
      issue_piece(piece, r_next, d_soff_next, x_soff_next, xs, ds);
#endif
      if (piece == 2) {
        d_soff_next += d_stride;
        if (!a.up || (r_next & 1)) x_soff_next += x_stride;  // the source row of an up-sampled input advances every second row
        ++r_next;
      }
    };

    // ring slots: X row q -> (q - (ya-1)) mod NR, dY row q -> the same
#pragma unroll
    for (int k = 0; k < PF; ++k) issue_row(ya - 1 + k, k, k);
    wait_rows(std::integral_constant<int, PF - 3>{});          // rows ya - 1, ya, ya + 1 have landed
    __builtin_amdgcn_s_barrier();

    bf16x8 xa[2], xb[2];
    load_x(0, 0, xa);
    load_dy(X_BYTES + DROW + dhalf, dp);           // r = ya-1: dY row ya is the only one of the window inside the band
    load_dy(ZERO_IMG, dc);
    load_dy(ZERO_IMG, dm);
    // One row step; `xa` holds X(r)[kw = 0]; on return `xb` holds X(r+1)[kw = 0].  The ring slot S of row r is a compile-time
    // constant (six step bodies per trip): every LDS offset of the step is an instruction immediate.
    auto row_step = [&](int r, auto slot_tag, bf16x8 (&xa)[2], bf16x8 (&xb)[2]) {
      constexpr int S = decltype(slot_tag)::value;
      constexpr int SN = (S + PF) % NR;                       // slot of row r + PF: it held row r - 1
      auto none = [](int) {};
      load_x(S, 1, xb);                                       // the next tap column's fragments first, then this one's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      mfma12(0, xa, [&](int kh) { if (kh < 2) issue_next_piece(kh, SN, SN); });
      load_x(S, 2, xa);
      __builtin_amdgcn_sched_barrier(0);
      mfma12(1, xb, [&](int kh) { if (kh == 0) issue_next_piece(2, SN, SN); });
      __builtin_amdgcn_sched_barrier(0);
#ifndef WG3_T_NOBARRIER                                        // (timing-only build: what the per-row synchronisation costs)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of X row r has returned: its slot is reused by row r + NR
      wait_rows(std::integral_constant<int, PF - 2>{});       // rows <= r+2 have landed
      __builtin_amdgcn_s_barrier();
#endif
      const int on = r + 2 < yb ? X_BYTES + ((S + 2) % NR) * DROW + dhalf : ZERO_IMG;      // dY row r+2 (or the zero image)
      load_x((S + 1) % NR, 0, xb);                            // (past the last row: harmless reads, never used)
      load_dy(on, dn);
      __builtin_amdgcn_sched_barrier(0);
      mfma12(2, xa, none);
#pragma unroll
      for (int ao = 0; ao < 2; ++ao) {
        dm[ao] = dc[ao];
        dc[ao] = dp[ao];
        dp[ao] = dn[ao];
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    using S4 = std::integral_constant<int, 4>;
    using S5 = std::integral_constant<int, 5>;
    static_assert(NR == 6, "six step bodies per trip");
    int r = ya - 1;                                 // RB in {8, 16, 32}: RB + 2 = 10, 18, 34 row steps; the X register sets swap every step
    for (; r + 5 <= yb; r += 6) {
      row_step(r, S0{}, xa, xb);
      row_step(r + 1, S1{}, xb, xa);
      row_step(r + 2, S2{}, xa, xb);
      row_step(r + 3, S3{}, xb, xa);
      row_step(r + 4, S4{}, xa, xb);
      row_step(r + 5, S5{}, xb, xa);
    }
    if (r <= yb) {                                  // (RB + 2) % 6 is 4 or 0: the ring is back at slot 0 here
      row_step(r, S0{}, xa, xb);
      row_step(r + 1, S1{}, xb, xa);
      row_step(r + 2, S2{}, xa, xb);
      row_step(r + 3, S3{}, xb, xa);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<0>();                               // (the trailing rows issued past the band)
    __builtin_amdgcn_s_barrier();                  // the next item's DMA overwrites the slots read last
  }

  mfma_results_ready();                              // (in-place asm MFMAs: the wait states in front of the stores' reads are ours)
  // The right half's partial tile joins the left half's through the dead rings, five taps and then four: block (tap, ao, bi)
  // of quadrant w4 at [block][w4][lane] x 16 B — consecutive lanes, consecutive 16-byte slots — and ALWAYS left + right:
  // a fixed order, so the result does not depend on timing.
  static_assert(5 * 4 * 256 * 16 <= X_BYTES + D_BYTES, "five taps of the half-tile exchange fit in the dead rings");
  float4* const xch = reinterpret_cast<float4*>(lds) + w4 * 64 + lane;
  float* __restrict__ ws = a.ws + (size_t)by * a.Co * 9 * a.Ci;
  auto exchange = [&](auto t0_tag, auto t1_tag) {
    constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
    if (half == 1) {
#pragma unroll
      for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int ao = 0; ao < 2; ++ao)
#pragma unroll
          for (int bi = 0; bi < 2; ++bi) {
            const f32x4 v = acc[t][ao][bi];
            xch[(((t - T0) * 2 + ao) * 2 + bi) * 256] = make_float4(v[0], v[1], v[2], v[3]);
          }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int ao = 0; ao < 2; ++ao)
#pragma unroll
          for (int bi = 0; bi < 2; ++bi) {
            const float4 o = xch[(((t - T0) * 2 + ao) * 2 + bi) * 256];
            const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = co0 + qo * 32 + ao * 16 + 4 * c4 + r;      // C/D map: row = 4*(lane >> 4) + reg, col = lane & 15
              const int ci = ci0 + qi * 32 + bi * 16 + l16;
              if (co < a.Co && ci < a.Ci) ws[((size_t)co * 9 + t) * a.Ci + ci] = acc[t][ao][bi][r] + ov[r];
            }
          }
    }
  };
  exchange(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
  __syncthreads();                                   // (the left half has read the first five taps)
  exchange(std::integral_constant<int, 5>{}, std::integral_constant<int, 9>{});
}
