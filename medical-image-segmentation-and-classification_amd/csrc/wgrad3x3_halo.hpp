// Weight gradient of a 3x3 / stride-1 / pad-1 convolution, bf16 / fp16, gfx950 — all nine taps per workgroup.
//
//   dW[co][tap][ci] = sum_{n,y,x} dY[n,y,x,co] * X[n, y+kh-1, x+kw-1, ci]
//
// A workgroup owns a 64(co) x 64(ci) tile of dW for ALL nine taps (wave w = 32x32 quadrant = 2 x 2 MFMA blocks of
// 16x16, 144 accumulator registers per lane) and walks 32-pixel row segments of the image: per step it needs ONE
// new dY row segment and ONE new X row segment (a rolling window with a 1-pixel halo serves the nine shifted
// reads), so dY and X are streamed from L2 once per 36 MFMAs per wave — 9x less L2->LDS traffic than the per-tap
// split-K kernel (which is L2-bound for C <= 128).  Rows arrive by LDS-DMA (asm-issued, counted vmcnt, WG3_PF = 7 rows
// ahead into eight-row rings); fragments are fetched with the hardware transpose read ds_read_b64_tr_b16 from [pixel][64 ch] row images.
// (nearest x2 up-sampling of X is folded into the row gather.)
#pragma once
#include <type_traits>
#include "common.hpp"

struct Wgrad3Args {
  // up to six (x, dy) pairs of ONE shared convolution (the recurrent blocks apply a conv six times, R2AttU_Net.py:41-44):
  // their weight gradients are one sum, so the pairs are simply more work items of the same launch
  const void* xs[6];
  const void* dys[6];
  int items_per_app;             // work items of one pair
  float* ws;
  int N, Hi, Wi, Ci, ldx;        // physical X
  int H, W, Co, ldy;             // dY / logical X grid
  int up;
  int RB;                        // rows per work item (even)
  int items, items_per_block;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

template <bool W16> struct Wgrad3Lds {              // dynamic LDS of wgrad3x3_halo_kernel (shared with its launchers)
  static constexpr int NR = 8;
  static constexpr int BYTES = NR * ((W16 ? 48 : 40) * 128 + 32 * 128) + 4096;
};

// ---- v_mfma_f32_16x16x32, dY rows held in registers ----------------------------------------------------------------
// The MFMA consumes a whole 32-pixel row segment per instruction and
// the wave tile is 2 x 2 blocks of 16 x 16 (the chip holds a higher clock on this shape, MI355X_MICROARCH.md DVFS
// item 7).  Operand map of the 16x16x32 MFMA: row (channel) = lane & 15, K = 8*(lane >> 4) .. +7, so one transpose
// read covers pixels 8*b + 0..3 (b = lane >> 4) of a 16-channel block and a wave instruction touches pixels
// {s..s+3, s+8..s+11, s+16.., s+24..}: the 32-B channel slot is XOR-ed with f(px) = bit1(px) | bit3(px) << 1, which
// keeps the four same-parity pixels of every 32-lane group in four distinct 32-B slots of the 256-B bank period
// for ANY pixel shift s (the three kw taps).
// A dY row meets three consecutive X rows (kh = 0, 1, 2), so its two fragments are read from LDS ONCE and ride a
// three-row register window; per row step only the six X fragments and two new dY fragments are read: 8 fragment
// reads per 36 MFMAs (a dY-row-major loop on 32x32x16 needed 20 per 18 twice as large ones and was 13 % slower).  A row step runs in three phases
// by tap column kw (12 MFMAs each), the X fragments of the next phase loading while the current one computes; the
// workgroup barrier sits between phases 1 and 2, so that the first fragments of the next row load behind phase 2.
// W16: images 16 pixels wide.  A 32-pixel "row" is then row y of TWO images side by side (work item = image pair x row band):
// the X row image holds two 24-pixel segments (16 + 4 + 4 halo each, so that the kw shifts of one image never read the other),
// the dY row image the two 16-pixel rows back to back; lanes of the upper half of a K block (c4 >= 2) read the second segment.
template <typename T, bool W16 = false>
__global__ __launch_bounds__(256, 2) void wgrad3x3_halo_kernel(const Wgrad3Args a) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  constexpr int XPX = W16 ? 48 : 40, XROW = XPX * 128, DROW = 32 * 128;
  constexpr int XPIECES = XPX / 8;                   // 1-KiB DMA pieces per X row: waves 0..3, then waves 0 .. XPIECES-5
  // Rings of NR rows, row r + PF fetched during step r (its slot held row r - 1).  PF = 3 left the waves waiting at the
  // counted vmcnt of every step (a timing-only build without the per-step wait + barrier ran 10-18 % faster, and exactly as
  // fast as the no-DMA build once the DMA was gone too: the wait was for rows, not for waves) — a row step is ~0.4 us, an
  // LDS-DMA row under load takes longer than three of them.  Equal ring depths: X row q and dY row q share the slot index.
#ifndef WG3_PF
#define WG3_PF 7
#endif
  constexpr int NR = Wgrad3Lds<W16>::NR, PF = WG3_PF;
  static_assert(PF >= 3 && PF < NR, "row r + PF lands in the slot of a row <= r - 1");
  constexpr int NRX = NR, NRD = NR;
  constexpr int X_BYTES = NRX * XROW, D_BYTES = NRD * DROW;
  constexpr int ZERO_IMG = X_BYTES + D_BYTES;
  static_assert(Wgrad3Lds<W16>::BYTES == X_BYTES + D_BYTES + 4096, "launcher and kernel disagree on the LDS size");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const xr = lds;
  unsigned char* const dr = lds + X_BYTES;
  unsigned char* const dump = lds + ZERO_IMG;
  const unsigned lds_x = lds_addr(xr), lds_d = lds_addr(dr), lds_dump = lds_addr(dump);      // DMA destinations: LDS byte addresses

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, c4 = lane >> 4;
  const int qo = wave >> 1, qi = wave & 1;
  const int ciTiles = (a.Ci + 63) / 64;
  const int bid = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bx = bid % gridDim.x, by = bid / gridDim.x;
  const int co0 = (bx / ciTiles) * 64, ci0 = (bx % ciTiles) * 64;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  auto swz = [](int px) { return (((px >> 1) & 1) | (((px >> 3) & 1) << 1)) << 1; };     // XOR on the 16-B chunk index

  const int lpx = lane >> 3, slot = lane & 7;       // DMA: a 1-KiB piece = 8 pixels x 128 B; lane -> (pixel, 16-B slot)
  const int TXN = W16 ? 1 : a.W / 32, BANDS = a.H / a.RB;
  *reinterpret_cast<uint4*>(dump + tid * 16) = make_uint4(0, 0, 0, 0);      // the all-zero dY row image

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transpose-read lane geometry: 16-lane block b = c4 reads pixels 8b + tq (+4), channels col0 + 4*tp .. +3
  const int tq = l16 >> 2, tp = l16 & 3;
  const int pl = 8 * c4 + tq;                      // this lane's first pixel inside a 32-pixel K block
  const int plx = W16 ? (c4 >> 1) * 24 + 8 * (c4 & 1) + tq : pl;      // ... and inside the X row image (minus the 4-pixel halo)
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
    const int ya = band * a.RB, yb = ya + a.RB, x0 = tx * 32;
    const int nbase = W16 ? 2 * n : n;              // (W16: `n` counts image pairs)
    // per-lane pixel geometry of this item, shared by the prologue rows and the running pointers below
    const int px_d = 8 * wave + lpx;                 // pixel of the 32-px dY row image
    const int c_d = co0 + 8 * (slot ^ swz(px_d));
    const bool lane_ok_d = c_d < a.Co;
    const int dpix = W16 ? (px_d >> 4) * a.H * a.W + (px_d & 15) : x0 + px_d;      // pixel offset from (image nbase, row r, x 0)
    int xpix[2], c_x[2], xpiece[2];
    bool lane_ok_x[2], real_x[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      xpiece[k] = k == 0 ? wave : 4 + wave;
      real_x[k] = xpiece[k] < XPIECES;               // (the others write a scratch KiB: every wave issues the same count)
      const int q = 8 * xpiece[k] + lpx;             // pixel of the X row image
      const int half = W16 ? q / 24 : 0;
      const int xx = W16 ? q % 24 - 4 : x0 - 4 + q;
      c_x[k] = ci0 + 8 * (slot ^ swz(q));
      lane_ok_x[k] = real_x[k] && (unsigned)xx < (unsigned)a.W && c_x[k] < a.Ci;
      xpix[k] = half * a.Hi * a.Wi + (xx >> a.up);
    }

    // DMA sources: a buffer descriptor per tensor whose base is image `nbase` (wave-uniform), a scalar row offset, and ONE
    // 32-bit register per piece holding the lane's offset inside the row — or the always-out-of-range offset where the lane is
    // padding (image column / channel range), so that the hardware's range check writes the zeros; a row outside the image
    // (X) or the band (dY) takes a descriptor with num_records = 0.  Nothing per-lane is computed or selected per row.
    const bufdesc_t desc_d = make_buf(dy + (size_t)nbase * a.H * a.W * a.ldy);
    const bufdesc_t desc_x = make_buf(x + (size_t)nbase * a.Hi * a.Wi * a.ldx);
    const unsigned voff_d = lane_ok_d ? (unsigned)((dpix * a.ldy + c_d) * (int)sizeof(T)) : DMA_PAD;
    unsigned voff_x[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) voff_x[k] = lane_ok_x[k] ? (unsigned)((xpix[k] * a.ldx + c_x[k]) * (int)sizeof(T)) : DMA_PAD;
    const unsigned d_stride = (unsigned)(a.W * a.ldy) * (unsigned)sizeof(T), x_stride = (unsigned)(a.Wi * a.ldx) * (unsigned)sizeof(T);
    auto with_rows = [](bufdesc_t d, bool ok) { d[2] = ok ? (int)DMA_PAD : 0; return d; };
    // piece 0: this wave's KiB of dY row r; pieces 1, 2: its KiB(s) of X row r.  xs / ds: ring slots of X row r / dY row r
    auto issue_piece = [&](int piece, int r, unsigned d_soff, unsigned x_soff, int xs, int ds) {
      if (piece == 0) {
        dma16_buf(with_rows(desc_d, r >= ya && r < yb), voff_d, d_soff, lds_d + ds * DROW + wave * 1024);
      } else {
        const int k = piece - 1;
        dma16_buf(with_rows(desc_x, (unsigned)r < (unsigned)a.H), voff_x[k], x_soff,
                  real_x[k] ? lds_x + xs * XROW + xpiece[k] * 1024 : lds_dump + wave * 1024);
      }
    };
    auto issue_at = [&](int r, unsigned d_soff, unsigned x_soff, int xs, int ds) {
#pragma unroll
      for (int piece = 0; piece < 3; ++piece) issue_piece(piece, r, d_soff, x_soff, xs, ds);
    };
    auto issue_row = [&](int r, int xs, int ds) {
      issue_at(r, (unsigned)r * d_stride, (unsigned)(r >> a.up) * x_stride, xs, ds);      // (r = -1: a dead offset under num_records = 0)
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
    // twelve MFMAs of tap column kw; `between(kh)` runs behind the four MFMAs of tap row kh (a DMA piece rides there: in the
    // matrix pipe's shadow, one at a time — three back to back stall the pipe for their issue time)
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
    auto wrap = [](int v, int n) { return v >= n ? v - n : v; };
    // The rows the main loop fetches are CONSECUTIVE (ya + 2, ya + 3, ...): their per-lane source pointers advance by a row
    // stride instead of being rebuilt from (n, r, x) with 64-bit multiplies each time, and the lane part of the bounds test
    // (channel / image-column range) is taken once per item; only the row part, wave-uniform, is evaluated per row.
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

    // ring slots: X row q -> (q - (ya-1)) mod NRX, dY row q -> (q - (ya-1)) mod NRD
#pragma unroll
    for (int k = 0; k < PF; ++k) issue_row(ya - 1 + k, k, k);
    wait_vmcnt<3 * (PF - 3)>();                    // rows ya - 1, ya, ya + 1 have landed
    __builtin_amdgcn_s_barrier();

    bf16x8 xa[2], xb[2];
    load_x(0, 0, xa);
    load_dy(X_BYTES + DROW, dp);                   // r = ya-1: dY row ya is the only one of the window inside the band
    load_dy(ZERO_IMG, dc);
    load_dy(ZERO_IMG, dm);
    // One row step; `xa` holds X(r)[kw = 0]; on return `xb` holds X(r+1)[kw = 0].  The ring slot S of row r is a COMPILE-TIME
    // constant (four step bodies per trip), so every LDS offset of the step is an instruction immediate.
    auto row_step = [&](int r, auto slot_tag, bf16x8 (&xa)[2], bf16x8 (&xb)[2]) {
      constexpr int S = decltype(slot_tag)::value;
      // Row r + PF is fetched from inside the MFMA stream, always: past the band it brings zeros (a descriptor with no records:
      // no memory access) into ring slots that are dead by then, which keeps vmcnt uniform.  Its ring slot held row r - 1, whose
      // last reads returned before the barrier of step r - 1.
      constexpr int SN = (S + PF) % NR;
      auto none = [](int) {};
      load_x(S, 1, xb);                                       // the next tap column's fragments first, then this one's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      mfma12(0, xa, [&](int kh) { if (kh < 2) issue_next_piece(kh, SN, SN); });
      load_x(S, 2, xa);
      __builtin_amdgcn_sched_barrier(0);
      mfma12(1, xb, [&](int kh) { if (kh == 0) issue_next_piece(2, SN, SN); });
      __builtin_amdgcn_sched_barrier(0);
#ifndef WG3_T_NOBARRIER                                        // (timing-only build: what the per-row synchronisation costs)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of X row r has returned: its slot is reused by row r+4
      wait_vmcnt<3 * (PF - 2)>();                             // rows <= r+2 have landed
      __builtin_amdgcn_s_barrier();
#endif
      const int on = r + 2 < yb ? X_BYTES + ((S + 2) % NR) * DROW : ZERO_IMG;      // dY row r+2 (or the zero image)
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
    using S6 = std::integral_constant<int, 6>;
    using S7 = std::integral_constant<int, 7>;
    static_assert(NR == 8, "eight step bodies per trip");
    int r = ya - 1;                                 // RB in {8, 16, 32}: RB + 2 row steps; the X register sets swap every step
    for (; r + 7 <= yb; r += 8) {
      row_step(r, S0{}, xa, xb);
      row_step(r + 1, S1{}, xb, xa);
      row_step(r + 2, S2{}, xa, xb);
      row_step(r + 3, S3{}, xb, xa);
      row_step(r + 4, S4{}, xa, xb);
      row_step(r + 5, S5{}, xb, xa);
      row_step(r + 6, S6{}, xa, xb);
      row_step(r + 7, S7{}, xb, xa);
    }
    if (r <= yb) {                                  // (RB + 2) % 8 == 2: the ring is back at slot 0 here
      row_step(r, S0{}, xa, xb);
      row_step(r + 1, S1{}, xb, xa);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<0>();                               // (the trailing rows issued past the band)
    __builtin_amdgcn_s_barrier();                  // the next item's DMA overwrites the slots read last
  }

  mfma_results_ready();                              // (in-place asm MFMAs: the wait states in front of the stores' reads are ours)
  float* __restrict__ ws = a.ws + (size_t)by * a.Co * 9 * a.Ci;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ao = 0; ao < 2; ++ao)
#pragma unroll
      for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + qo * 32 + ao * 16 + 4 * c4 + r;      // C/D map: row = 4*(lane >> 4) + reg, col = lane & 15
          const int ci = ci0 + qi * 32 + bi * 16 + l16;
          if (co < a.Co && ci < a.Ci) ws[((size_t)co * 9 + t) * a.Ci + ci] = acc[t][ao][bi][r];
        }
}
