// Weight gradient of a 3x3 / stride-1 / pad-1 convolution, bf16, gfx950 — all nine taps per workgroup.
//
//   dW[co][tap][ci] = sum_{n,y,x} dY[n,y,x,co] * X[n, y+kh-1, x+kw-1, ci]
//
// A workgroup owns a 64(co) x 64(ci) tile of dW for ALL nine taps (wave w = 32x32 quadrant, nine
// f32x16 accumulators) and walks 32-pixel row segments of the image: per step it needs ONE new dY
// row segment and ONE new X row segment (a rolling 3-row window with a 1-pixel halo serves the nine
// shifted reads), so dY and X are streamed from L2 once per 18 MFMAs per wave instead of once per 2
// — 9x less L2->LDS traffic than the per-tap split-K kernel (which is L2-bound for C <= 128).
// The loop runs over X rows r, not dY rows: X[r] meets dY[r+1], dY[r], dY[r-1] (kh = 0, 1, 2), so the three
// kw-shifted X fragments of a row are read from LDS ONCE and feed nine MFMAs together with three dY
// fragments: 12 fragment reads per 18 MFMAs instead of 20 — the dY-row-major order was LDS-bandwidth bound
// (4 waves x 10 KiB per row step > 128 B/clk x 576 MFMA clocks).
// Rows arrive by LDS-DMA (asm-issued, counted vmcnt, 2 rows of prefetch distance); fragments are
// fetched with the hardware transpose read ds_read_b64_tr_b16 from [pixel][64 ch] row images whose
// 16-B chunks are XOR-swizzled by ((pixel >> 1) & 1) << 2 (conflict-free for any pixel shift).
// (nearest x2 up-sampling of X is folded into the row gather.)
#pragma once
#include "common.hpp"

struct Wgrad3Args {
  const void* x;
  const void* dy;
  float* ws;
  int N, Hi, Wi, Ci, ldx;        // physical X
  int H, W, Co, ldy;             // dY / logical X grid
  int up;
  int RB;                        // rows per work item
  int items, items_per_block;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad3x3_halo_kernel(const Wgrad3Args a) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  constexpr int XPX = 40, XROW = XPX * 128, DROW = 32 * 128;      // row images in bytes
  constexpr int NRX = 4, NRD = 5;                 // live rows: X r..r+3, dY r-1..r+3 (NRX must be a power of two)
  constexpr int X_BYTES = NRX * XROW, D_BYTES = NRD * DROW;
  __shared__ __attribute__((aligned(16))) unsigned char lds[X_BYTES + D_BYTES + 4096];
  unsigned char* const xr = lds;
  unsigned char* const dr = lds + X_BYTES;
  unsigned char* const dump = lds + X_BYTES + D_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r32 = lane & 31;
  const int qo = wave >> 1, qi = wave & 1;
  const int ciTiles = (a.Ci + 63) / 64;
  // logical order: the (co, ci) tiles of one split are neighbours — they stream the same pixel rows
  const int bid = xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bx = bid % gridDim.x, by = bid / gridDim.x;
  const int co0 = (bx / ciTiles) * 64, ci0 = (bx % ciTiles) * 64;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // DMA lane geometry: a 1-KiB piece = 8 pixels x 128 B; lane -> (pixel, 16-B slot)
  const int lpx = lane >> 3, slot = lane & 7;
  const bool ci_ok = ci0 + 8 * slot < a.Ci || true;     // chunk validity is decided after un-swizzling below
  (void)ci_ok;
  const int TXN = a.W / 32, BANDS = a.H / a.RB;

  // the dump image doubles as the all-zero dY row (its DMAs only ever bring zeros, but wave 0 never dumps)
  *reinterpret_cast<uint4*>(dump + tid * 16) = make_uint4(0, 0, 0, 0);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transpose-read lane geometry (see conv_wgrad.hip)
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const int colA = qo * 32 + 16 * tg + 4 * tp;      // channel inside the 64-wide dY row image
  const int colB = qi * 32 + 16 * tg + 4 * tp;
  auto rd = [&](const unsigned char* row_img, int px, int col) {
    const int chunk = (col >> 3) ^ (((px >> 1) & 1) << 2);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(row_img + px * 128 + chunk * 16 + (col & 7) * 2));
  };

  const int item0 = by * a.items_per_block;
  const int item1 = min(a.items, item0 + a.items_per_block);
  for (int item = item0; item < item1; ++item) {
    int t = item;
    const int band = t % BANDS; t /= BANDS;
    const int tx = t % TXN;
    const int n = t / TXN;
    const int ya = band * a.RB, yb = ya + a.RB, x0 = tx * 32;

    // L(r): dY row r piece `wave`, X row r piece `wave`, X row r piece 4 (wave 0) — 3 DMA per wave
    auto issue_row = [&](int r, int xs, int ds) {   // xs / ds: ring slots of X row r / dY row r
      {   // dY piece: pixels 8*wave .. +7 of row r
        const int px = 8 * wave + lpx;
        const int chunk = slot ^ (((px >> 1) & 1) << 2);
        const int c = co0 + 8 * chunk;
        const bool ok = r >= ya && r < yb && c < a.Co;
        const char* p = ok ? reinterpret_cast<const char*>(dy + ((size_t)(n * a.H + r) * a.W + x0 + px) * a.ldy + c) : zero + slot * 16;
        dma16(p, lds_addr(dr + ds * DROW + wave * 1024));
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int piece = k == 0 ? wave : 4;
        const bool real = k == 0 || wave == 0;
        const int px = 8 * piece + lpx;              // pixel of the 40-px row image; image x = x0 - 4 + px
        const int xx = x0 - 4 + px;
        const int chunk = slot ^ (((px >> 1) & 1) << 2);
        const int c = ci0 + 8 * chunk;
        const bool ok = real && (unsigned)r < (unsigned)a.H && (unsigned)xx < (unsigned)a.W && c < a.Ci;
        const char* p = ok ? reinterpret_cast<const char*>(x + ((size_t)(n * a.Hi + (r >> a.up)) * a.Wi + (xx >> a.up)) * a.ldx + c)
                           : zero + slot * 16;
        dma16(p, lds_addr(real ? xr + xs * XROW + piece * 1024 : dump + wave * 1024));
      }
    };

    // fragments of one 16-pixel K block (ss) of row step r: three kw-shifted X fragments, three dY rows.
    // dY rows outside the band read the all-zero dump image (keeps the loop free of branches so that the
    // LDS reads of the next K block are in flight while the nine MFMAs of the current one run).
    auto load_frags = [&](int xs, int o0, int o1, int o2, int ss, bf16x8 (&af)[3], bf16x8 (&bf)[3]) {
      const unsigned char* ximg = xr + xs * XROW;
      const int pbase = ss * 16 + 8 * h + tq;      // tile pixel of this lane's first transpose block
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int px = pbase + 3 + kw;             // image x = x0 + pixel + kw - 1  <=>  row-image pixel + 3 + kw
        const s16x4 b0 = rd(ximg, px, colB), b1 = rd(ximg, px + 4, colB);
        bf[kw] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const unsigned char* dimg = lds + (kh == 0 ? o0 : (kh == 1 ? o1 : o2));   // dY row r+1-kh (or the zero image)
        const s16x4 a0 = rd(dimg, pbase, colA), a1 = rd(dimg, pbase + 4, colA);
        af[kh] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    auto mfma9 = [&](const bf16x8 (&af)[3], const bf16x8 (&bf)[3]) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc[kh * 3 + kw] = mfma_32x32x16<T>(af[kh], bf[kw], acc[kh * 3 + kw]);
    };
    auto wrap = [](int v, int n) { return v >= n ? v - n : v; };

    // ring slots: X row q -> (q - (ya-1)) mod NRX, dY row q -> (q - (ya-1)) mod NRD
    // prologue: rows ya-1 .. ya+1 in flight, the first two landed
    issue_row(ya - 1, 0, 0);
    issue_row(ya, 1, 1);
    issue_row(ya + 1, 2, 2);
    wait_vmcnt<3>();
    __builtin_amdgcn_s_barrier();

    bf16x8 a0f[3], b0f[3], a1f[3], b1f[3];
    constexpr int ZERO_IMG = X_BYTES + D_BYTES;    // the dump image only ever receives zeros
    int xs = 0, d = 0;                             // slots of X row r / dY row r
    int o0 = X_BYTES + DROW, o1 = ZERO_IMG, o2 = ZERO_IMG;   // LDS offsets of dY rows r+1, r, r-1 (r = ya-1: only ya is in the band)
    load_frags(xs, o0, o1, o2, 0, a0f, b0f);
    for (int r = ya - 1; r <= yb; ++r) {           // X row r: needs rows <= r+1 landed
      const bool more = r + 3 <= yb;               // rows up to yb (the bottom halo) are ever needed
      if (more) issue_row(r + 3, (xs + 3) & (NRX - 1), wrap(d + 3, NRD));
      load_frags(xs, o0, o1, o2, 1, a1f, b1f);
      mfma9(a0f, b0f);
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);   // all twelve LDS reads of the next K block first ...
      __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);    // ... then the nine MFMAs of the current one
      // every LDS read of row step r has returned before the barrier: the DMA of row r+4 reuses X slot r
      __builtin_amdgcn_sched_barrier(0);           // (keep the MFMAs above in front of the waits)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (more) wait_vmcnt<3>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      xs = (xs + 1) & (NRX - 1);
      d = wrap(d + 1, NRD);
      o2 = o1;
      o1 = o0;
      o0 = r + 2 < yb ? X_BYTES + wrap(d + 1, NRD) * DROW : ZERO_IMG;
      load_frags(xs, o0, o1, o2, 0, a0f, b0f);     // (past the last row: harmless reads, never used)
      mfma9(a1f, b1f);
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, 9, 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // the next item's DMA overwrites the slots read last
  }

  float* __restrict__ ws = a.ws + (size_t)by * a.Co * 9 * a.Ci;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + qo * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ci = ci0 + qi * 32 + r32;
      if (co < a.Co && ci < a.Ci) ws[((size_t)co * 9 + t) * a.Ci + ci] = acc[t][r];
    }
}
