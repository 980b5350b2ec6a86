// Weight gradient of a convolution as a split-K MFMA GEMM (gfx950).
//
//   dW[co][tap][ci] = sum_m dy[m][co] * x[src(m,tap)][ci]          m = (n,ho,wo)
//
// GEMM view per tap: rows = co, cols = ci, reduction = output pixels.  Both operands are
// pixel-major in HBM (NHWC), i.e. the reduction index is the slow one, so each tile is staged
// [pixel][channel] in LDS exactly as it is read (coalesced 16-B pieces) and the MFMA operand
// fragments are fetched with the hardware transpose read ds_read_b64_tr_b16 (bf16) or plain
// conflict-free ds_read_b32 (fp32, one k per lane half).  The pixel range is split over
// gridDim.z; every split writes its own fp32 slab (deterministic), summed by
// mi355_conv2d_wgrad_reduce into the parameter-gradient layout.
#include "common.hpp"
#include "dma.hpp"
#include "wgrad3x3_halo.hpp"
#include "wgrad3x3_halo8.hpp"
#include <stdlib.h>

struct WgradArgs {
  const void* x;
  const void* dy;
  float* ws;
  int N, Hi, Wi, Ci, ldx;
  int Ho, Wo, Co, ldy;
  int KH, KW, stride, pad, up;
  int M, Hlog, Wlog;
  int chunk;   // pixels per split (multiple of 32)
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
// MFMA element type of the 2-byte branch (the fp32 instantiation never executes it, but must compile)
template <typename T> struct TwoByte { typedef T type; };
template <> struct TwoByte<float> { typedef bf16_t type; };

template <typename T, int BCO, int BCI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int BP = 32;                         // pixels per K step
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr bool BF = sizeof(T) == 2;
  // LDS pitches: bf16 rows are padded so that the 4 pixel rows of a transpose-read block fall
  // into disjoint 16-bank windows (64 ch -> 192 B, 128 ch -> 320 B); fp32 rows are natural.
  constexpr int PO = BF ? (BCO == 128 ? 320 : 192) : BCO * 4;
  constexpr int PX = BF ? (BCI == 128 ? 320 : 192) : BCI * 4;
  constexpr int O_BYTES = BP * PO, X_BYTES = BP * PX, STAGE = O_BYTES + X_BYTES;
  constexpr int CPRO = BCO / EPC, CPRX = BCI / EPC;
  constexpr int O_IT = BP * CPRO / 256, X_IT = BP * CPRX / 256;
  constexpr int WTO = BCO / 2, WTI = BCI / 2;     // 2 x 2 waves
  constexpr int MI = WTO / 32, NI = WTI / 32;
  static_assert(O_IT >= 1 && X_IT >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int wo_ = wave >> 1, wi_ = wave & 1;
  const int ciTiles = (a.Ci + BCI - 1) / BCI;
  // logical order: (tile, tap) fastest inside a split — those blocks read the same pixel rows
  int bid = xcd_tile((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
  const int bx = bid % gridDim.x; bid /= gridDim.x;
  const int co0 = (bx / ciTiles) * BCO, ci0 = (bx % ciTiles) * BCI;
  const int tap = bid % gridDim.y, kh = tap / a.KW, kw = tap - kh * a.KW;
  const int split = bid / gridDim.y;
  const int p_begin = split * a.chunk;
  const int p_end = min(a.M, p_begin + a.chunk);
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);

  // per-thread staging rows
  int o_row[O_IT], o_c[O_IT];
#pragma unroll
  for (int i = 0; i < O_IT; ++i) {
    const int id = tid + i * 256;
    o_row[i] = id / CPRO;
    o_c[i] = id - o_row[i] * CPRO;
  }
  int x_row[X_IT], x_c[X_IT], x_n[X_IT], x_ho[X_IT], x_wo[X_IT];
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const int id = tid + i * 256;
    x_row[i] = id / CPRX;
    x_c[i] = id - x_row[i] * CPRX;
    const int m = p_begin + x_row[i];
    x_n[i] = m / HoWo;
    const int rem = m - x_n[i] * HoWo;
    x_ho[i] = rem / a.Wo;
    x_wo[i] = rem - x_ho[i] * a.Wo;
  }

  uint4 ro[O_IT], rx[X_IT];
  auto load_tile = [&](int p0) {
#pragma unroll
    for (int i = 0; i < O_IT; ++i) {
      const int m = p0 + o_row[i];
      const int c = co0 + o_c[i] * EPC;
      if (m < p_end && c < a.Co)
        ro[i] = *reinterpret_cast<const uint4*>(dy + (size_t)m * a.ldy + c);
      else
        ro[i] = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int m = p0 + x_row[i];
      const int c = ci0 + x_c[i] * EPC;
      int th = x_ho[i] * a.stride + kh - a.pad, tw = x_wo[i] * a.stride + kw - a.pad;
      bool ok = m < p_end && c < a.Ci && (unsigned)th < (unsigned)a.Hlog && (unsigned)tw < (unsigned)a.Wlog;
      th >>= a.up;
      tw >>= a.up;
      if (ok)
        rx[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(x_n[i] * a.Hi + th) * a.Wi + tw) * a.ldx + c);
      else
        rx[i] = make_uint4(0, 0, 0, 0);
      // advance this row's pixel coordinate by BP for the next step
      x_wo[i] += BP;
      while (x_wo[i] >= a.Wo) {
        x_wo[i] -= a.Wo;
        if (++x_ho[i] == a.Ho) {
          x_ho[i] = 0;
          ++x_n[i];
        }
      }
    }
  };
  auto store_tile = [&](int stage) {
    unsigned char* lo = lds + stage * STAGE;
    unsigned char* lx = lo + O_BYTES;
#pragma unroll
    for (int i = 0; i < O_IT; ++i) *reinterpret_cast<uint4*>(lo + o_row[i] * PO + o_c[i] * 16) = ro[i];
#pragma unroll
    for (int i = 0; i < X_IT; ++i) *reinterpret_cast<uint4*>(lx + x_row[i] * PX + x_c[i] * 16) = rx[i];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int steps = (p_end - p_begin + BP - 1) / BP;
  if (steps > 0) {
    load_tile(p_begin);
    store_tile(0);
  }
  __syncthreads();
  // transpose-read lane geometry: 16-lane group g = lane>>4 covers channels 16*(g&1).. of the
  // wave's 32-channel block and pixels 8*(g>>1)..; lane 4q+p of the group addresses pixel row q,
  // channels 4p..4p+3.
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  for (int st = 0; st < steps; ++st) {
    const int cur = st & 1;
    const bool more = (st + 1) < steps;
    if (more) load_tile(p_begin + (st + 1) * BP);
    const unsigned char* lo = lds + cur * STAGE;
    const unsigned char* lx = lo + O_BYTES;
    if constexpr (BF) {
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        bf16x8 af[MI], bfr[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int col = wo_ * WTO + mi * 32 + 16 * tg + 4 * tp;
          const int prow = ss * 16 + 8 * h + tq;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo + prow * PO + col * 2));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo + (prow + 4) * PO + col * 2));
          af[mi] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int col = wi_ * WTI + ni * 32 + 16 * tg + 4 * tp;
          const int prow = ss * 16 + 8 * h + tq;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lx + prow * PX + col * 2));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lx + (prow + 4) * PX + col * 2));
          bfr[ni] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = mfma_32x32x16<typename TwoByte<T>::type>(af[mi], bfr[ni], acc[mi][ni]);
      }
    } else {
#pragma unroll 4
      for (int s2 = 0; s2 < BP / 2; ++s2) {
        const int prow = 2 * s2 + h;
        float af[MI], bfr[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          af[mi] = *reinterpret_cast<const float*>(lo + prow * PO + (wo_ * WTO + mi * 32 + r32) * 4);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          bfr[ni] = *reinterpret_cast<const float*>(lx + prow * PX + (wi_ * WTI + ni * 32 + r32) * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
      }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  const int taps = a.KH * a.KW;
  float* __restrict__ ws = a.ws + (size_t)split * a.Co * taps * a.Ci;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wo_ * WTO + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ci = ci0 + wi_ * WTI + ni * 32 + r32;
        if (co < a.Co && ci < a.Ci) ws[((size_t)co * taps + tap) * a.Ci + ci] = acc[mi][ni][r];
      }
}

static void wgrad_tiles(int Co, int Ci, int& bco, int& bci) {
  bco = (Co % 128 == 0) ? 128 : 64;
  bci = (Ci % 128 == 0) ? 128 : 64;
}

// nine-tap kernels: 0 = not served; four-wave kernel (wgrad3x3_halo.hpp): 1 = rows of 32-pixel segments, 2 = 16-pixel-wide
// images taken two at a time; eight-wave kernel (wgrad3x3_halo8.hpp): 3 = rows of 64-pixel segments, 4 = 32-pixel-wide images
// taken two at a time.  MI355_WGRAD8=0 keeps every shape on the four-wave kernel (A/B switch).
static int halo_wgrad_mode(int N, int Ho, int Wo, int KH, int KW) {
  static const int use8 = getenv("MI355_WGRAD8") ? atoi(getenv("MI355_WGRAD8")) : 1;
  if (KH != 3 || KW != 3 || Ho % 8 != 0) return 0;
  if (use8 && Wo % 64 == 0) return 3;
  if (use8 && Wo == 32 && N % 2 == 0) return 4;
  if (Wo % 32 == 0) return 1;
  if (Wo == 16 && N % 2 == 0) return 2;
  return 0;
}
// work items of one (x, dy) pair: (image or image pair) x row segment x row band
static int halo_wgrad_items(int mode, int N, int Ho, int Wo, int rb) {
  switch (mode) {
    case 1: return N * (Wo / 32) * (Ho / rb);
    case 3: return N * (Wo / 64) * (Ho / rb);
    default: return (N / 2) * (Ho / rb);             // 2, 4: image pairs
  }
}

extern "C" int mi355_conv2d_wgrad_splits(int N, int Ho, int Wo, int Ci, int Co, int KH, int KW) {
  if (const int mode = halo_wgrad_mode(N, Ho, Wo, KH, KW)) {      // nine-tap kernel: 64x64 tiles, work items = 32-pixel-wide row bands
    const int rb = Ho % 32 == 0 ? 32 : (Ho % 16 == 0 ? 16 : 8);
    const long long items = halo_wgrad_items(mode, N, Ho, Wo, rb);
    const long long tiles = (long long)ceil_div(Co, 64) * ceil_div(Ci, 64);
    // Four-wave kernel: ONE workgroup per CU.  Two are resident (254 VGPRs) and run the kernel 18 % faster on its own (3.4 vs 4.0 ms
    // per Attention U-Net step), but the kernel lives on the side stream next to the data-gradient chain: at one per CU it leaves
    // half of every SIMD's registers to the main stream's kernels, the two interleave instead of queueing, and half as many
    // partial slabs reach the reduce.  Step: 512 / 384 / 320 / 256 / 192 / 128 workgroups = 18.53 / 18.58 / 18.47 / 18.18 /
    // 18.28 / 19.28 ms (round 2).
    // Eight-wave kernel: a workgroup OWNS its CU (512 threads x 256 registers, 115 KB of LDS), so its grid is a share of the chip
    // handed to the side stream for the length of the launch: on 128 CUs the weight gradients run at twice the per-CU rate of the
    // four-wave kernel while the other 128 CUs belong to the main stream's data-gradient / BatchNorm chain alone, and half as many
    // partial slabs are written and reduced.  Step: 64 / 96 / 112 / 128 / 144 / 160 / 192 / 256 workgroups = 17.7 / 16.2 / 16.1 /
    // 15.40 / 15.6 / 15.65 / 15.9 / 16.3 ms against 15.74 for the four-wave kernel (profiles/r04a_wgs_sweep8.txt).
    static const int wgs_env = getenv("MI355_WGRAD_WGS") ? atoi(getenv("MI355_WGRAD_WGS")) : 0;      // (A/B switch)
    int wgs = wgs_env > 0 ? wgs_env : (mode >= 3 ? 128 : 256);
    // (A/B: another grid for the eight-wave kernel on images of at most MI355_WGRAD_DEEP_H rows)
    static const int deep_wgs = getenv("MI355_WGRAD_WGS_DEEP") ? atoi(getenv("MI355_WGRAD_WGS_DEEP")) : 0;
    static const int deep_h = getenv("MI355_WGRAD_DEEP_H") ? atoi(getenv("MI355_WGRAD_DEEP_H")) : 64;
    if (mode >= 3 && deep_wgs > 0 && Ho <= deep_h) wgs = deep_wgs;
    long long s = wgs / tiles;
    if (s > items) s = items;
    const long long slab = (long long)Co * 9 * Ci * 4;
    while (s > 1 && s * slab > (512ll << 20)) --s;
    return (int)(s < 1 ? 1 : s);
  }
  int bco, bci;
  wgrad_tiles(Co, Ci, bco, bci);
  const long long M = (long long)N * Ho * Wo;
  const long long tiles = (long long)ceil_div(Co, bco) * ceil_div(Ci, bci) * KH * KW;
  long long s = 768 / tiles;                           // ~3 workgroups per CU (1536: same kernel time, more partials to reduce)
  const long long max_by_pixels = M / 256 > 0 ? M / 256 : 1;
  if (s > max_by_pixels) s = max_by_pixels;
  const long long slab = (long long)Co * KH * KW * Ci * 4;
  while (s > 1 && s * slab > (512ll << 20)) --s;       // bound the fp32 partial workspace
  if (s < 1) s = 1;
  return (int)s;
}

template <typename T>
static int wgrad_launch(const WgradArgs& a, int splits, hipStream_t s) {
  int bco, bci;
  wgrad_tiles(a.Co, a.Ci, bco, bci);
  dim3 grid(ceil_div(a.Co, bco) * ceil_div(a.Ci, bci), a.KH * a.KW, splits);
  if (bco == 128 && bci == 128)
    hipLaunchKernelGGL((conv_wgrad_kernel<T, 128, 128>), grid, dim3(256), 0, s, a);
  else if (bco == 128)
    hipLaunchKernelGGL((conv_wgrad_kernel<T, 128, 64>), grid, dim3(256), 0, s, a);
  else if (bci == 128)
    hipLaunchKernelGGL((conv_wgrad_kernel<T, 64, 128>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<T, 64, 64>), grid, dim3(256), 0, s, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

template <typename T, bool W16>
static int launch_wgrad3x3(const Wgrad3Args& h, dim3 grid, hipStream_t s) {
  constexpr int lds_bytes = Wgrad3Lds<W16>::BYTES;
  static const hipError_t configured =
      hipFuncSetAttribute((const void*)wgrad3x3_halo_kernel<T, W16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (configured != hipSuccess)
    MI355_FAIL((int)configured, "wgrad3x3_halo: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(configured));
  hipLaunchKernelGGL((wgrad3x3_halo_kernel<T, W16>), grid, dim3(256), lds_bytes, s, h);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
template <typename T, bool W32>
static int launch_wgrad3x3_8(const Wgrad3Args& h, dim3 grid, hipStream_t s) {
  constexpr int lds_bytes = Wgrad8Lds::BYTES;
  static const hipError_t configured =
      hipFuncSetAttribute((const void*)wgrad3x3_halo8_kernel<T, W32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (configured != hipSuccess)
    MI355_FAIL((int)configured, "wgrad3x3_halo8: cannot reserve %d B of LDS: %s", lds_bytes, hipGetErrorString(configured));
  hipLaunchKernelGGL((wgrad3x3_halo8_kernel<T, W32>), grid, dim3(512), lds_bytes, s, h);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
static int launch_wgrad3x3_any(const Wgrad3Args& h, dim3 grid, int hmode, int dtype, hipStream_t s) {
  const bool f16 = dtype == MI355_F16;
  switch (hmode) {
    case 2: return f16 ? launch_wgrad3x3<f16_t, true>(h, grid, s) : launch_wgrad3x3<bf16_t, true>(h, grid, s);
    case 3: return f16 ? launch_wgrad3x3_8<f16_t, false>(h, grid, s) : launch_wgrad3x3_8<bf16_t, false>(h, grid, s);
    case 4: return f16 ? launch_wgrad3x3_8<f16_t, true>(h, grid, s) : launch_wgrad3x3_8<bf16_t, true>(h, grid, s);
    default: return f16 ? launch_wgrad3x3<f16_t, false>(h, grid, s) : launch_wgrad3x3<bf16_t, false>(h, grid, s);
  }
}

extern "C" int mi355_conv2d_wgrad(const void* x, const void* dy, float* ws, int splits, int N, int Hi, int Wi, int Ci,
                                  int ldx, int Ho, int Wo, int Co, int ldy, int KH, int KW, int stride, int pad, int up,
                                  int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(x && dy && ws, "conv2d_wgrad: null pointer");
  MI355_CHECK_ARG(splits >= 1 && splits <= 65535, "conv2d_wgrad: splits=%d out of range", splits);
  const int esz = dtype_is_2byte(dtype) ? 2 : 4;
  const int epc = 16 / esz;
  MI355_CHECK_ARG(Ci % epc == 0 && Co % epc == 0, "conv2d_wgrad: Ci=%d / Co=%d must be multiples of %d", Ci, Co, epc);
  MI355_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && (ldx * esz) % 16 == 0 && (ldy * esz) % 16 == 0,
                  "conv2d_wgrad: pointers / channel strides must be 16-byte aligned");
  MI355_CHECK_ARG((long long)N * Ho * Wo < (1ll << 31) && (long long)N * Hi * Wi < (1ll << 31),
                  "conv2d_wgrad: pixel count overflows int32");
  WgradArgs a;
  a.x = x; a.dy = dy; a.ws = ws;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.ldx = ldx;
  a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.ldy = ldy;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.up = up ? 1 : 0;
  a.M = N * Ho * Wo;
  a.Hlog = up ? 2 * Hi : Hi;
  a.Wlog = up ? 2 * Wi : Wi;
  a.chunk = ceil_div(ceil_div(a.M, splits), 32) * 32;
  static const int use_halo = getenv("MI355_WGRAD_HALO") ? atoi(getenv("MI355_WGRAD_HALO")) : 1;
  const int hmode = halo_wgrad_mode(N, Ho, Wo, KH, KW);
  // (the nine-tap kernel addresses an image — two in the 16-pixel mode — through buffer descriptors with 32-bit lane and row
  //  offsets: 2 GiB or more per image goes to the generic kernel, which carries 64-bit addresses)
  const long long img_lim = (1ll << 31) / (hmode == 2 || hmode == 4 ? 2 : 1);
  const bool fits = (long long)Hi * Wi * ldx * esz < img_lim && (long long)Ho * Wo * ldy * esz < img_lim;
  if (dtype_is_2byte(dtype) && use_halo && stride == 1 && pad == 1 && hmode && Ho == a.Hlog && Wo == a.Wlog && fits) {
    Wgrad3Args h;
    for (int i = 0; i < 6; ++i) { h.xs[i] = x; h.dys[i] = dy; }
    h.ws = ws;
    h.N = N; h.Hi = Hi; h.Wi = Wi; h.Ci = Ci; h.ldx = ldx;
    h.H = Ho; h.W = Wo; h.Co = Co; h.ldy = ldy; h.up = up ? 1 : 0;
    h.RB = Ho % 32 == 0 ? 32 : (Ho % 16 == 0 ? 16 : 8);
    h.items = halo_wgrad_items(hmode, N, Ho, Wo, h.RB);
    h.items_per_app = h.items;
    h.items_per_block = ceil_div(h.items, splits);
    dim3 grid(ceil_div(Co, 64) * ceil_div(Ci, 64), splits);
    return launch_wgrad3x3_any(h, grid, hmode, dtype, (hipStream_t)s);
  }
  return dispatch_dtype(dtype, "conv2d_wgrad", [&](auto tag) { return wgrad_launch<decltype(tag)>(a, splits, (hipStream_t)s); });
}

extern "C" int mi355_conv2d_wgrad_variant(int N, int Ho, int Wo, int KH, int KW, int stride, int pad, int dtype) {
  static const int use_halo = getenv("MI355_WGRAD_HALO") ? atoi(getenv("MI355_WGRAD_HALO")) : 1;
  if (!use_halo || !dtype_is_2byte(dtype) || stride != 1 || pad != 1) return 0;
  return halo_wgrad_mode(N, Ho, Wo, KH, KW);
}

extern "C" int mi355_conv2d_wgrad_multi_ok(int N, int Ho, int Wo, int dtype) {
  static const int use_halo = getenv("MI355_WGRAD_HALO") ? atoi(getenv("MI355_WGRAD_HALO")) : 1;
  return (use_halo && dtype_is_2byte(dtype) && halo_wgrad_mode(N, Ho, Wo, 3, 3)) ? 1 : 0;
}

extern "C" int mi355_conv2d_wgrad_multi(const void* x0, const void* dy0, const void* x1, const void* dy1, const void* x2,
                                        const void* dy2, const void* x3, const void* dy3, const void* x4, const void* dy4,
                                        const void* x5, const void* dy5, int napp, float* ws, int splits, int N, int Hi, int Wi,
                                        int Ci, int ldx, int Ho, int Wo, int Co, int ldy, int up, int dtype, mi355_stream_t s) {
  const void* xs[6] = {x0, x1, x2, x3, x4, x5};
  const void* dys[6] = {dy0, dy1, dy2, dy3, dy4, dy5};
  MI355_CHECK_ARG(napp >= 1 && napp <= 6 && ws && splits >= 1, "conv2d_wgrad_multi: 1..6 operand pairs, a workspace, splits >= 1");
  for (int i = 0; i < napp; ++i)
    MI355_CHECK_ARG(xs[i] && dys[i] && ((uintptr_t)xs[i] % 16) == 0 && ((uintptr_t)dys[i] % 16) == 0,
                    "conv2d_wgrad_multi: pair %d: null or misaligned pointer", i);
  MI355_CHECK_ARG(mi355_conv2d_wgrad_multi_ok(N, Ho, Wo, dtype), "conv2d_wgrad_multi: shape / dtype not served by the nine-tap kernel "
                  "(mi355_conv2d_wgrad_multi_ok == 0): run mi355_conv2d_wgrad per pair");
  MI355_CHECK_ARG(Ci % 8 == 0 && Co % 8 == 0 && (ldx * 2) % 16 == 0 && (ldy * 2) % 16 == 0 && ldx >= Ci && ldy >= Co,
                  "conv2d_wgrad_multi: channel counts / strides must be multiples of 8 elements");
  MI355_CHECK_ARG((up ? (Ho == 2 * Hi && Wo == 2 * Wi) : (Ho == Hi && Wo == Wi)), "conv2d_wgrad_multi: 3x3 / stride 1 / pad 1 geometry only");
  MI355_CHECK_ARG((long long)Hi * Wi * ldx * 4 < (1ll << 31) && (long long)Ho * Wo * ldy * 4 < (1ll << 31),
                  "conv2d_wgrad_multi: an image of 1 GiB or more is beyond the 32-bit offsets of the nine-tap kernel's buffer descriptors");
  const int hmode = halo_wgrad_mode(N, Ho, Wo, 3, 3);
  Wgrad3Args h;
  for (int i = 0; i < 6; ++i) { h.xs[i] = xs[i < napp ? i : 0]; h.dys[i] = dys[i < napp ? i : 0]; }
  h.ws = ws;
  h.N = N; h.Hi = Hi; h.Wi = Wi; h.Ci = Ci; h.ldx = ldx;
  h.H = Ho; h.W = Wo; h.Co = Co; h.ldy = ldy; h.up = up ? 1 : 0;
  h.RB = Ho % 32 == 0 ? 32 : (Ho % 16 == 0 ? 16 : 8);
  h.items_per_app = halo_wgrad_items(hmode, N, Ho, Wo, h.RB);
  h.items = napp * h.items_per_app;
  MI355_CHECK_ARG(splits <= h.items, "conv2d_wgrad_multi: more splits (%d) than work items (%d)", splits, h.items);
  h.items_per_block = ceil_div(h.items, splits);
  dim3 grid(ceil_div(Co, 64) * ceil_div(Ci, 64), splits);
  return launch_wgrad3x3_any(h, grid, hmode, dtype, (hipStream_t)s);
}


// dw[co][ci][kh][kw] (or [ci][co][kh][kw] when transposed) = beta*dw + sum_s ws[s][co][tap][ci]
// One 1024-thread workgroup per (co, 32-channel ci tile).  Thread (kl, cl) sums splits kl, kl+32, ... of channel
// ci0+cl for every tap (slab reads are 128-B ci-contiguous runs, 32 split lanes keep loads in flight),
// the 32 split lanes are folded through LDS, and the result is written transposed so that the
// parameter-gradient stores are contiguous runs of 32*taps floats.
#define WR_CI 32
// TAPS > 0: the tap count at compile time — a thread's partial sums of ALL taps ride in registers and the loads of a split pair
// (2 x TAPS of them) are in flight together; with the run-time count the loop fetched two floats per trip and waited for them.
template <int TAPS>
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, float* __restrict__ dw, int Co,
                                                            int Ci, int Ci_real, int taps_rt, int transposed, float beta, int KL) {
  extern __shared__ float tile[];                 // [KL][taps][WR_CI + 1]
  const int taps = TAPS > 0 ? TAPS : taps_rt;
  const int ciTiles = (Ci + WR_CI - 1) / WR_CI;
  const int co = blockIdx.x / ciTiles, ci0 = (blockIdx.x % ciTiles) * WR_CI;
  const size_t total = (size_t)Co * taps * Ci;
  const int cl = threadIdx.x % WR_CI, kl = threadIdx.x / WR_CI;
  const int ci = ci0 + cl;
  const int tstride = taps * (WR_CI + 1);
  if constexpr (TAPS > 0) {
    float s0[TAPS], s1[TAPS];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) { s0[tap] = 0.f; s1[tap] = 0.f; }
    if (ci < Ci) {
      const float* p = ws + (size_t)co * TAPS * Ci + ci;
      int k = kl;
      for (; k + KL < splits; k += 2 * KL) {
        float a[TAPS], b[TAPS];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          a[tap] = p[(size_t)k * total + (size_t)tap * Ci];
          b[tap] = p[(size_t)(k + KL) * total + (size_t)tap * Ci];
        }
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) { s0[tap] += a[tap]; s1[tap] += b[tap]; }
      }
      if (k < splits) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) s0[tap] += p[(size_t)k * total + (size_t)tap * Ci];
      }
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) tile[kl * tstride + tap * (WR_CI + 1) + cl] = s0[tap] + s1[tap];
  } else {
    for (int tap = 0; tap < taps; ++tap) {
      float s0 = 0.f, s1 = 0.f;
      if (ci < Ci) {
        const float* p = ws + ((size_t)co * taps + tap) * Ci + ci;
        int k = kl;
        for (; k + KL < splits; k += 2 * KL) {
          s0 += p[(size_t)k * total];
          s1 += p[(size_t)(k + KL) * total];
        }
        if (k < splits) s0 += p[(size_t)k * total];
      }
      tile[kl * tstride + tap * (WR_CI + 1) + cl] = s0 + s1;
    }
  }
  __syncthreads();
  const int n = taps * WR_CI;
  for (int t = threadIdx.x; t < n; t += WR_CI * KL) {
    const int c2 = t / taps, tap = t - c2 * taps;
    const int cc = ci0 + c2;
    if (cc >= Ci_real) continue;
    float s = 0.f;
    for (int k = 0; k < KL; ++k) s += tile[k * tstride + tap * (WR_CI + 1) + c2];
    const size_t o = transposed ? ((size_t)cc * Co + co) * taps + tap : ((size_t)co * Ci_real + cc) * taps + tap;
    dw[o] = (beta != 0.f ? beta * dw[o] : 0.f) + s;
  }
}

extern "C" int mi355_conv2d_wgrad_reduce(const float* ws, int splits, float* dw, int Co, int Ci, int Ci_real, int KH,
                                         int KW, int transposed, float beta, mi355_stream_t s) {
  MI355_CHECK_ARG(ws && dw && splits >= 1, "conv2d_wgrad_reduce: bad arguments");
  const int taps = KH * KW;
  const int ciTiles = (Ci + WR_CI - 1) / WR_CI;
  int KL = 15000 / (taps * (WR_CI + 1));          // split lanes per channel: as many as fit 60 KB of LDS, at most 32
  if (KL > 32) KL = 32;
  if (KL > splits) KL = splits;
  if (KL < 1) KL = 1;
  const dim3 grid(Co * ciTiles), block(WR_CI * KL);
  const size_t lds = KL * taps * (WR_CI + 1) * sizeof(float);
  if (taps == 9)          // (same sums in the same order as the run-time loop: two chains per thread, split k before k + KL)
    hipLaunchKernelGGL(wgrad_reduce_kernel<9>, grid, block, lds, (hipStream_t)s, ws, splits, dw, Co, Ci, Ci_real, taps, transposed, beta, KL);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<0>, grid, block, lds, (hipStream_t)s, ws, splits, dw, Co, Ci, Ci_real, taps, transposed, beta, KL);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
