// Implicit-GEMM convolution on MFMA for gfx950 (forward conv, data gradient, ConvTranspose).
//
//   out[m][j] = bias[j] + sum_{tap,c} in[src(m,tap)][c] * wk[j][tap][c]
//
// GEMM view: M = N*Ho*Wo pixel rows, N = Co, K = KH*KW*Ci walked tap-major in BK-channel
// slabs.  One 256-thread workgroup (4 waves, one per SIMD) owns a 128 x BN output tile.
// A rows are gathered straight from the NHWC activation (coalesced 16-B pieces of one
// pixel's channel run, zero for padding / stride holes), B rows from the packed weights
// [Co][tap][Ci]; both are register-staged (issue early, write late) into a double-buffered,
// XOR-swizzled LDS image so every ds_read_b128 fragment read is bank-conflict free.
//   bf16: v_mfma_f32_32x32x16_bf16, BK = 64 (or 32) channels per slab
//   fp32: v_mfma_f32_32x32x2_f32 (exact fmaf chain), BK = 16
#include "common.hpp"
#include <stdlib.h>

struct ConvArgs {
  const void* in;
  const void* wk;
  const float* bias;
  void* out;
  int N, Hi, Wi, Ci, ldi;
  int Ho, Wo, Co, ldo;
  int KH, KW;
  int mul, kmul, off, dshift, up;
  int accumulate;
  int relu;        // epilogue: max(0, conv + bias) before rounding / accumulation (bit 1 of the ABI's `accumulate`)
  int pool2;       // epilogue: sum 2x2 output-pixel groups, out is [N][Ho/2][Wo/2] (bit 2: gradient of a fused nearest x2 up-sampling)
  float* stats;  // optional fused BatchNorm statistics: partial[(mblock*2+q)*Co + c], q = sum / sum of squares
  int M;        // N*Ho*Wo
  int HoWo;
  int Hlog, Wlog;   // logical (post-upsample) input extent
};

template <typename T> struct Frag;   // 16-byte LDS fragment and its MFMA step
template <> struct Frag<bf16_t> {
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                0, 0);
  }
};
template <> struct Frag<f16_t> {
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
    c = mfma_32x32x16<f16_t>(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
  }
};
template <> struct Frag<float> {
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
    const f32x4 av = __builtin_bit_cast(f32x4, a), bv = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], c, 0, 0, 0);
  }
};

template <typename T, int BN, int BK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int BM = 128;
  constexpr int EPC = 16 / (int)sizeof(T);    // elements per 16-B chunk
  constexpr int CPR = BK / EPC;               // chunks per LDS row
  constexpr int P = BK * (int)sizeof(T);      // LDS row pitch in bytes (64 or 128)
  constexpr int RPBS = (P == 128) ? 1 : 2;    // log2(rows per 256-B bank row)
  constexpr int RPP = 256 / CPR;              // rows staged per pass of the 256 threads
  constexpr int A_IT = BM / RPP;
  constexpr int B_IT = (BN >= RPP) ? BN / RPP : 1;
  constexpr int WM = (BN == 32) ? 4 : 2, WN = 4 / WM;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int KQ = CPR / 2;
  constexpr int A_BYTES = BM * P, B_BYTES = BN * P, STAGE = A_BYTES + B_BYTES;
  constexpr bool LDS_EPI = sizeof(T) == 2;
  constexpr int C_PITCH = BN * (int)sizeof(T) + 16;
  constexpr int C_BYTES = LDS_EPI ? BM * C_PITCH : 0;
  constexpr int LDS_BYTES = (2 * STAGE > C_BYTES) ? 2 * STAGE : C_BYTES;
  static_assert(P == 64 || P == 128, "row pitch");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int NT = a.Co / BN;
  const int bid = xcd_tile(blockIdx.x, gridDim.x);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wk = reinterpret_cast<const T*>(a.wk);

  // ---- per-thread staging assignment -------------------------------------------------------
  const int chunk = tid % CPR, srow = tid / CPR;
  int a_nb[A_IT], a_hs[A_IT], a_ws[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + srow + i * RPP;
    if (m < a.M) {
      const int n = m / a.HoWo;
      const int rem = m - n * a.HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      a_nb[i] = n * a.Hi;
      a_hs[i] = ho * a.mul + a.off;
      a_ws[i] = wo * a.mul + a.off;
    } else {
      a_nb[i] = 0;
      a_hs[i] = -(1 << 28);   // never passes the range check
      a_ws[i] = 0;
    }
  }
  const int taps = a.KH * a.KW;
  const size_t wrow = (size_t)taps * a.Ci;
  const int dmask = (1 << a.dshift) - 1;

  uint4 ra[A_IT], rb[B_IT];
  bool a_ok[A_IT];
  auto load_tile = [&](int kh, int kw, int c0) {
    const int dh = kh * a.kmul, dw = kw * a.kmul;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int th = a_hs[i] + dh, tw = a_ws[i] + dw;
      bool ok = ((th | tw) & dmask) == 0;
      th >>= a.dshift;
      tw >>= a.dshift;
      ok = ok && (unsigned)th < (unsigned)a.Hlog && (unsigned)tw < (unsigned)a.Wlog;
      th >>= a.up;
      tw >>= a.up;
      // the load is issued from a CLAMPED pixel whatever `ok` says and masked afterwards: under `if (ok)` hipcc branches around it
      // and waits for it on the spot (s_waitcnt vmcnt(0) per row: A_IT dependent round trips per K tile instead of one)
      const T* p = in + ((size_t)(a_nb[i] + (ok ? th : 0)) * a.Wi + (ok ? tw : 0)) * a.ldi + c0 + chunk * EPC;
      ra[i] = *reinterpret_cast<const uint4*>(p);
      a_ok[i] = ok;
    }
    const int tap = kh * a.KW + kw;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = srow + i * RPP;
      if (BN >= RPP || row < BN) {
        const T* p = wk + (size_t)(n0 + row) * wrow + (size_t)tap * a.Ci + c0 + chunk * EPC;
        rb[i] = *reinterpret_cast<const uint4*>(p);
      }
    }
  };
  auto swz = [](int row, int c) { return (c ^ ((row >> RPBS) & (CPR - 1))) << 4; };
  auto store_tile = [&](int stage) {
    unsigned char* la = lds + stage * STAGE;
    unsigned char* lb = la + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = srow + i * RPP;
      *reinterpret_cast<uint4*>(la + row * P + swz(row, chunk)) = a_ok[i] ? ra[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = srow + i * RPP;
      if (BN >= RPP || row < BN) *reinterpret_cast<uint4*>(lb + row * P + swz(row, chunk)) = rb[i];
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int KT = taps * (a.Ci / BK);
  int kh = 0, kw = 0, c0 = 0;
  load_tile(0, 0, 0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < KT;
    if (more) {
      c0 += BK;
      if (c0 == a.Ci) {
        c0 = 0;
        if (++kw == a.KW) {
          kw = 0;
          ++kh;
        }
      }
      load_tile(kh, kw, c0);
    }
    const unsigned char* la = lds + cur * STAGE;
    const unsigned char* lb = la + A_BYTES;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      const int c = kq * 2 + h;
      uint4 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * WTM + mi * 32 + r32;
        af[mi] = *reinterpret_cast<const uint4*>(la + row * P + swz(row, c));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * WTN + ni * 32 + r32;
        bf[ni] = *reinterpret_cast<const uint4*>(lb + row * P + swz(row, c));
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Frag<T>::mma(af[mi], bf[ni], acc[mi][ni]);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  float bcol[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
    bcol[ni] = a.bias ? a.bias[n0 + wn * WTN + ni * 32 + r32] : 0.f;

  if constexpr (!LDS_EPI) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int m = m0 + row;
          if (m < a.M) {
            T* p = out + (size_t)m * a.ldo + n0 + wn * WTN + ni * 32 + r32;
            float v = acc[mi][ni][r] + bcol[ni];
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.accumulate) v += to_f32<T>(*p);
            *p = from_f32<T>(v);
          }
        }
  } else {
    // stage the tile through LDS so that global stores are 16-B, row-contiguous
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int col = wn * WTN + ni * 32 + r32;
          *reinterpret_cast<T*>(lds + row * C_PITCH + col * (int)sizeof(T)) =
              from_f32<T>(a.relu ? fmaxf(acc[mi][ni][r] + bcol[ni], 0.f) : acc[mi][ni][r] + bcol[ni]);
        }
    __syncthreads();
    constexpr int CPRC = BN / EPC;
    for (int id = tid; id < BM * CPRC; id += 256) {
      const int row = id / CPRC, c = id - row * CPRC;
      const int m = m0 + row;
      if (m < a.M) {
        T* p = out + (size_t)m * a.ldo + n0 + c * EPC;
        Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(lds + row * C_PITCH + c * 16);
        if (a.accumulate) {
          const Vec16<T> o = ld16<T>(p);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) + to_f32<T>(o.v[e]));
        }
        st16<T>(p, v);
      }
    }
  }
}

// ==============================================================================================
// bf16 fast path: the same tiling with the operand tiles streamed by LDS-DMA
// (global_load_lds_dwordx4: 16 B per lane, per-lane gather address, no VGPR staging, no ds_write)
// into an NS-deep LDS ring.  Tile k+NS-1 is issued while tile k is multiplied; a COUNTED
// s_waitcnt vmcnt leaves the newest tiles in flight across the raw s_barrier, so HBM/L2 latency
// is hidden behind NS-1 MFMA phases instead of one.  The XOR swizzle moves to the SOURCE side
// (the DMA writes lane-linear: lane (row, slot) fetches chunk slot ^ swz(row) of its pixel row);
// padding / stride-hole rows read a 256-B zero page.
// ==============================================================================================
#include "dma.hpp"

template <typename T, int BN, int BK, int NS>
__global__ __launch_bounds__(256) void conv_igemm_dma_kernel(const ConvArgs a) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  constexpr int BM = 128;
  constexpr int EPC = 8;
  constexpr int CPR = BK / EPC;               // 16-B chunks per LDS row (4 or 8)
  constexpr int P = BK * 2;                   // row pitch in bytes (64 or 128)
  constexpr int RPBS = (P == 128) ? 1 : 2;
  constexpr int RPW = 64 / CPR;               // rows covered by one wave-instruction (1 KiB)
  constexpr int A_IT = BM / RPW / 4;          // DMA instructions per wave per A tile
  constexpr int B_IT = BN / RPW / 4;
  static_assert(B_IT >= 1, "every wave must issue the same number of DMA instructions");
  constexpr int WM = (BN == 32) ? 4 : 2, WN = 4 / WM;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int KQ = CPR / 2;
  constexpr int A_BYTES = BM * P, B_BYTES = BN * P, STAGE = A_BYTES + B_BYTES;
  constexpr int C_PITCH = BN * 2 + 16;
  constexpr int C_BYTES = BM * C_PITCH;
  constexpr int EPI_BYTES = C_BYTES + WM * 2 * BN * 4;
  constexpr int LDS_BYTES = (NS * STAGE > EPI_BYTES) ? NS * STAGE : EPI_BYTES;
  constexpr int PER_TILE = A_IT + B_IT;       // DMA instructions per wave per K tile
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int NT = a.Co / BN;
  const int bid = xcd_tile(blockIdx.x, gridDim.x);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wk = reinterpret_cast<const T*>(a.wk);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // lane geometry inside one DMA instruction: RPW rows x CPR chunk slots
  const int lrow = lane / CPR, slot = lane % CPR;
  int a_nb[A_IT], a_hs[A_IT], a_ws[A_IT], a_ck[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int row = (wave + 4 * i) * RPW + lrow;
    a_ck[i] = (slot ^ ((row >> RPBS) & (CPR - 1))) * EPC;     // source chunk (swizzle on the source side)
    const int m = m0 + row;
    if (m < a.M) {
      const int n = m / a.HoWo;
      const int rem = m - n * a.HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      a_nb[i] = n * a.Hi;
      a_hs[i] = ho * a.mul + a.off;
      a_ws[i] = wo * a.mul + a.off;
    } else {
      a_nb[i] = 0;
      a_hs[i] = -(1 << 28);
      a_ws[i] = 0;
    }
  }
  const int taps = a.KH * a.KW;
  const size_t wrow = (size_t)taps * a.Ci;
  const T* b_ptr[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int row = (wave + 4 * i) * RPW + lrow;
    b_ptr[i] = wk + (size_t)(n0 + row) * wrow + (slot ^ ((row >> RPBS) & (CPR - 1))) * EPC;
  }
  const int dmask = (1 << a.dshift) - 1;

  auto issue_tile = [&](int stage, int kh, int kw, int c0) {
    unsigned char* la = lds + stage * STAGE;
    unsigned char* lb = la + A_BYTES;
    const int dh = kh * a.kmul, dw = kw * a.kmul;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int th = a_hs[i] + dh, tw = a_ws[i] + dw;
      bool ok = ((th | tw) & dmask) == 0;
      th >>= a.dshift;
      tw >>= a.dshift;
      ok = ok && (unsigned)th < (unsigned)a.Hlog && (unsigned)tw < (unsigned)a.Wlog;
      th >>= a.up;
      tw >>= a.up;
      const char* p = ok ? reinterpret_cast<const char*>(in + ((size_t)(a_nb[i] + th) * a.Wi + tw) * a.ldi + c0 + a_ck[i])
                         : zero + slot * 16;
      dma16(p, lds_addr(la + (wave + 4 * i) * 1024));
    }
    const size_t boff = (size_t)(kh * a.KW + kw) * a.Ci + c0;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) dma16(b_ptr[i] + boff, lds_addr(lb + (wave + 4 * i) * 1024));
  };
  auto swz = [](int row, int c) { return (c ^ ((row >> RPBS) & (CPR - 1))) << 4; };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int KT = taps * (a.Ci / BK);
  // tile iterator of the ISSUE side
  int ikh = 0, ikw = 0, ic0 = 0, issued = 0;
  auto advance = [&]() {
    ic0 += BK;
    if (ic0 == a.Ci) {
      ic0 = 0;
      if (++ikw == a.KW) {
        ikw = 0;
        ++ikh;
      }
    }
  };
  // prologue: NS-1 tiles in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    if (issued < KT) {
      issue_tile(s, ikh, ikw, ic0);
      advance();
      ++issued;
    }
  }
  if (KT >= NS - 1) wait_vmcnt<(NS - 2) * PER_TILE>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  int stage = 0, istage = NS - 1;
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = issued < KT;
    if (more) {
      issue_tile(istage, ikh, ikw, ic0);
      advance();
      ++issued;
      istage = (istage + 1 == NS) ? 0 : istage + 1;
    }
    const unsigned char* la = lds + stage * STAGE;
    const unsigned char* lb = la + A_BYTES;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      const int c = kq * 2 + h;
      uint4 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * WTM + mi * 32 + r32;
        af[mi] = *reinterpret_cast<const uint4*>(la + row * P + swz(row, c));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * WTN + ni * 32 + r32;
        bf[ni] = *reinterpret_cast<const uint4*>(lb + row * P + swz(row, c));
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Frag<T>::mma(af[mi], bf[ni], acc[mi][ni]);
    }
    stage = (stage + 1 == NS) ? 0 : stage + 1;
    // tile kt+1 must have landed: everything but the newest NS-2 tiles (fewer at the tail)
    if (more) wait_vmcnt<(NS - 2) * PER_TILE>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: bias, bf16 rounding, LDS-staged 16-B row-contiguous stores ---------------------
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  float bcol[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bcol[ni] = a.bias ? a.bias[n0 + wn * WTN + ni * 32 + r32] : 0.f;
  float* const red = reinterpret_cast<float*>(lds + C_BYTES);      // [WM][2][BN] behind the C tile
  static_assert(LDS_BYTES >= C_BYTES + WM * 2 * BN * 4, "no room for the statistics scratch");
  if (a.stats) {        // fused BatchNorm statistics of the ROUNDED outputs this tile stores
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = (m0 + row < a.M) ? to_f32<T>(from_f32<T>(a.relu ? fmaxf(acc[mi][ni][r] + bcol[ni], 0.f) : acc[mi][ni][r] + bcol[ni])) : 0.f;
          sm += v;
          sq += v * v;
        }
      sm += __shfl_xor(sm, 32, 64);
      sq += __shfl_xor(sq, 32, 64);
      if (h == 0) {
        red[(wm * 2 + 0) * BN + wn * WTN + ni * 32 + r32] = sm;
        red[(wm * 2 + 1) * BN + wn * WTN + ni * 32 + r32] = sq;
      }
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = wn * WTN + ni * 32 + r32;
        *reinterpret_cast<T*>(lds + row * C_PITCH + col * 2) = from_f32<T>(a.relu ? fmaxf(acc[mi][ni][r] + bcol[ni], 0.f) : acc[mi][ni][r] + bcol[ni]);
      }
  __syncthreads();
  if (a.stats && tid < 2 * BN) {
    const int q = tid / BN, c = tid - q * BN;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) v += red[(w * 2 + q) * BN + c];
    a.stats[((size_t)mt * 2 + q) * a.Co + n0 + c] = v;
  }
  constexpr int CPRC = BN / EPC;
  for (int id = tid; id < BM * CPRC; id += 256) {
    const int row = id / CPRC, c = id - row * CPRC;
    const int m = m0 + row;
    if (m < a.M) {
      T* p = out + (size_t)m * a.ldo + n0 + c * EPC;
      Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(lds + row * C_PITCH + c * 16);
      if (a.accumulate) {
        const Vec16<T> o = ld16<T>(p);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) + to_f32<T>(o.v[e]));
      }
      st16<T>(p, v);
    }
  }
}

template <typename T, int BN, int BK, int NS>
static int launch_dma(const ConvArgs& a, hipStream_t s) {
  const int grid = ceil_div(a.M, 128) * (a.Co / BN);
  hipLaunchKernelGGL((conv_igemm_dma_kernel<T, BN, BK, NS>), dim3(grid), dim3(256), 0, s, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

#include "conv3x3_halo.hpp"
#include "conv3x3_halo_pp.hpp"
#include "conv3x3_halo_pp128.hpp"
#include "conv3x3_ws.hpp"
#include "conv1x1_stream.hpp"
#include "conv_gemm256.hpp"

template <typename T, int BN, int BK>
static int launch(const ConvArgs& a, hipStream_t s) {
  const int grid = ceil_div(a.M, 128) * (a.Co / BN);
  hipLaunchKernelGGL((conv_igemm_kernel<T, BN, BK>), dim3(grid), dim3(256), 0, s, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// (generic kernel: the widest output-channel tile that divides Co, narrower when that grid would be at most half a round of
//  workgroups — ResNet-18 at batch 8, fp32: 512 -> 512 @8² is 4 x 4 tiles of 128 x 128; see dma_tile_n)
static int small_grid_tile_n(long long M, int Co);
template <typename T, int BK>
static int launch_bn(const ConvArgs& a, hipStream_t s) {
  switch (small_grid_tile_n(a.M, a.Co)) {
    case 128: return launch<T, 128, BK>(a, s);
    case 64: return launch<T, 64, BK>(a, s);
    default: return launch<T, 32, BK>(a, s);
  }
}

enum IgemmVariant { IG_GENERIC = 0, IG_DMA, IG_HALO_8x32, IG_HALO_16x16, IG_STREAM1x1, IG_HALO_PP, IG_HALO_PP128, IG_WS64, IG_WS128, IG_GEMM256 };

// ONE place that decides which kernel serves a shape (also used by the statistics-row query).
static IgemmVariant pick_variant(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int mul, int kmul, int off,
                                 int div, int up, int dtype) {
  if (!dtype_is_2byte(dtype)) return IG_GENERIC;
  static const int force_generic = getenv("MI355_IGEMM_VARIANT") ? atoi(getenv("MI355_IGEMM_VARIANT")) == 0 : 0;
  if (force_generic) return IG_GENERIC;
  const int Hlog = up ? 2 * Hi : Hi, Wlog = up ? 2 * Wi : Wi;
  // 3x3 stride-1 pad-1 forward / data gradient on tile-divisible images: halo-patch kernel
  const bool is3x3s1 = KH == 3 && KW == 3 && mul == 1 && div == 1 && Ho == Hlog && Wo == Wlog &&
                       ((kmul == 1 && off == -1) || (kmul == -1 && off == 1 && !up));
  if (is3x3s1 && Co % 64 == 0) {
    // MI355_HALO_PP=1: 16 x 32 tiles, two phase-shifted halves per 512-thread workgroup (conv3x3_halo_pp.hpp).  Measured
    // (profiles/r02a_*): +1-3 % on the 32x32 layers with Ci >= 512, -10-15 % on the 256x256 layers (one workgroup per CU
    // exposes every tile's prologue and epilogue) -> the 4-wave kernel below stays the default; the variant is kept selectable.
    static const int use_pp = getenv("MI355_HALO_PP") ? atoi(getenv("MI355_HALO_PP")) : 0;
    if (use_pp && Wo % 32 == 0 && Ho % 16 == 0) return IG_HALO_PP;
    // 16 x 32 pixels x 128 channels, ping-pong halves (conv3x3_halo_pp128.hpp): half the operand bytes per FLOP; one workgroup
    // per CU, so only where the reduction is deep enough to amortise a tile's prologue and epilogue.  Measured per layer
    // (profiles/r02c_conv_layers.txt against r02a_conv_layers_pp0.txt): Ci >= 512 +14-17 %, Ci = 256 +7-9 %, Ci = 128 -2..+2 % -> threshold 256.
    // MI355_HALO_PP128=0 switches it off (A/B), MI355_HALO_PP128_MINCI moves the threshold.
    static const int pp128 = getenv("MI355_HALO_PP128") ? atoi(getenv("MI355_HALO_PP128")) : 1;
    static const int pp128_min_ci = getenv("MI355_HALO_PP128_MINCI") ? atoi(getenv("MI355_HALO_PP128_MINCI")) : 256;
    if (pp128 && Co % 128 == 0 && Ci % 64 == 0 && Ci >= pp128_min_ci && Wo % 32 == 0 && Ho % 16 == 0) return IG_HALO_PP128;
    // Ci = 64 / 128: weights stationary in registers, persistent workgroups (conv3x3_ws.hpp); MI355_WS64=0 / MI355_WS128=0 switch
    // them off (A/B)
    static const int ws64 = getenv("MI355_WS64") ? atoi(getenv("MI355_WS64")) : 1;
    static const int ws128 = getenv("MI355_WS128") ? atoi(getenv("MI355_WS128")) : 1;
    const int ws_rows = ws_tile_rows(Ci, Ho, Wo, Co);
    if (ws64 && ws_rows == 8) return IG_WS64;
    if (ws128 && ws_rows == 4) return IG_WS128;
    if (Wo % 32 == 0 && Ho % 8 == 0) return IG_HALO_8x32;
    if (Wo % 16 == 0 && Ho % 16 == 0) return IG_HALO_16x16;
  }
  // narrow pointwise convolutions (and their data gradients): register-resident weights, streaming pixels
  if (KH == 1 && KW == 1 && mul == 1 && div == 1 && !up && off == 0 && Ho == Hi && Wo == Wi && stream1x1_shape(Ci, Co))
    return IG_STREAM1x1;
  // padding-free convolutions as plain GEMMs on 256 x 128 tiles (conv_gemm256.hpp): 1x1, 2x2 / stride 2, ConvTranspose2d(2, 2) phases;
  // batch-dependent conditions (row count, grid size) in final_variant.  MI355_GEMM256=0 switches it off (A/B)
  static const int use_gemm256 = getenv("MI355_GEMM256") ? atoi(getenv("MI355_GEMM256")) : 1;
  if (use_gemm256 && gemm256_mode(256, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, mul, kmul, off, div, up)) return IG_GEMM256;
  if (Co % 64 != 0 && Ci % 64 != 0) return IG_GENERIC;      // 32-wide tile with a 32-deep slab: too few DMA pieces per wave
  return IG_DMA;
}

// The 128-channel ping-pong kernel runs ONE 512-thread workgroup per CU: a grid that leaves a quarter of the last round of
// workgroups empty (or does not fill the chip once: 32 images of 32 x 32 x 256 channels = 128 workgroups) is served by the
// 4-wave kernel, whose grid is four times finer.  Batch-dependent, hence not part of the shape-level variant query.
static int device_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return cus;
}
static IgemmVariant resolve_variant(IgemmVariant v, int N, int Ho, int Wo, int Co) {
  // the weight-stationary kernel is persistent (2 workgroups per CU): it wants at least two tiles per workgroup to amortise
  // its weight load, else the 4-wave kernel with its four-times finer grid
  // (MI355_WS64_MIN_TILES lowers the threshold: the parity tests run it on small shapes, one tile per workgroup included)
  if (v == IG_WS64 || v == IG_WS128) {
    static const long long min_tiles = getenv("MI355_WS64_MIN_TILES") ? atoll(getenv("MI355_WS64_MIN_TILES")) : 4ll * device_cus();
    const int th = v == IG_WS64 ? 8 : 4;
    const long long S = (long long)N * (Ho / th) * (Wo / 32);
    // (the 128-channel instantiation loads twice the weights per workgroup for tiles half as tall: measured per layer it wins
    //  from 16 tiles per workgroup up — 128² x 128 -> 128 / 256, 256² x 128 -> 64: +3 ... +10 % — and loses 1-6 % at 8 —
    //  64² x 128 -> 256, 128² x 128 -> 64, batch-16 layers — so its threshold is 12 tiles per workgroup = 24 x CUs)
    static const int mult128 = getenv("MI355_WS128_TILE_MULT") ? atoi(getenv("MI355_WS128_TILE_MULT")) : 6;      // (A/B switch)
    if (S >= 8 && S * (Co / 64) >= min_tiles * (v == IG_WS128 ? mult128 : 1)) return v;
    return Ho % 8 == 0 ? IG_HALO_8x32 : (Ho % 16 == 0 && Wo % 16 == 0 ? IG_HALO_16x16 : IG_DMA);
  }
  if (v != IG_HALO_PP128) return v;
  const int cus = device_cus();
  const long long grid = (long long)N * (Ho / 16) * (Wo / 32) * (Co / 128);
  const long long rounds = (grid + cus - 1) / cus;
  static const int fill = getenv("MI355_PP128_FILL") ? atoi(getenv("MI355_PP128_FILL")) : 80;      // (A/B switch, percent)
  return grid * 100 >= rounds * cus * fill ? IG_HALO_PP128 : IG_HALO_8x32;     // >= 80 % of the last round filled
}

// Output-channel tile of the LDS-DMA ring kernel: the widest that divides Co — unless its grid leaves most of the chip idle (the
// frozen ResNet-50 encoder's 8 x 8 and 16 x 16 layers at batch 32: M = 2048 rows x 512 channels = 64 workgroups of 128 x 128), then
// the next narrower one (twice / four times the workgroups).  MI355_DMA_SMALLGRID=0: always the widest (A/B).
static int small_grid_tile_n(long long M, int Co) {
  static const int small = getenv("MI355_DMA_SMALLGRID") ? atoi(getenv("MI355_DMA_SMALLGRID")) : 1;
  const long long rows = (M + 127) / 128;
  int bn = Co % 128 == 0 ? 128 : (Co % 64 == 0 ? 64 : 32);
  if (small) {
    const int cus = device_cus();
    if (bn == 128 && rows * (Co / 128) * 2 <= cus) bn = 64;
    if (bn == 64 && rows * (Co / 64) * 2 <= cus) bn = 32;
  }
  return bn;
}

static int dma_tile_n(long long M, int Ci, int Co) {
  static const int small = getenv("MI355_DMA_SMALLGRID") ? atoi(getenv("MI355_DMA_SMALLGRID")) : 1;
  const long long rows = (M + 127) / 128;
  int bn = Co % 128 == 0 ? 128 : (Co % 64 == 0 ? 64 : 32);
  if (small) {
    const int cus = device_cus();
    if (bn == 128 && rows * (Co / 128) * 2 <= cus) bn = 64;                       // at most half a round of workgroups
    if (bn == 64 && Ci % 64 == 0 && rows * (Co / 64) * 2 <= cus) bn = 32;         // (the 32-wide instance runs 64-channel K tiles)
  }
  return bn;
}

static bool halo_family(IgemmVariant v) {
  return v == IG_HALO_8x32 || v == IG_HALO_16x16 || v == IG_HALO_PP || v == IG_HALO_PP128 || v == IG_WS64 || v == IG_WS128;
}
static bool image_fits_descriptor(int Hi, int Wi, int ldi, int esz) { return (long long)Hi * Wi * ldi * esz < (1ll << 31); }
static IgemmVariant final_variant(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int mul, int kmul, int off,
                                  int div, int up, int dtype) {
  IgemmVariant v = resolve_variant(pick_variant(Hi, Wi, Ci, Ho, Wo, Co, KH, KW, mul, kmul, off, div, up, dtype), N, Ho, Wo, Co);
  if (v == IG_GEMM256) {      // whole 256-row tiles only, and enough of them to fill the chip once (one workgroup per CU)
    static const long long min_tiles = getenv("MI355_GEMM256_MIN_TILES") ? atoll(getenv("MI355_GEMM256_MIN_TILES")) : 128;
    const int gm = gemm256_mode(N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, mul, kmul, off, div, up);
    if (!gm || gemm256_tiles(gm, N, Hi, Wi, Ho, Wo, Co) < min_tiles) v = IG_DMA;
  }
  return halo_family(v) && !image_fits_descriptor(Hi, Wi, Ci, 2) ? IG_DMA : v;
}

extern "C" int mi355_conv2d_igemm_variant(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int mul, int kmul, int off,
                                          int div, int up, int dtype) {
  return (int)pick_variant(Hi, Wi, Ci, Ho, Wo, Co, KH, KW, mul, kmul, off, div, up, dtype);
}

extern "C" int mi355_conv2d_igemm_variant_n(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int mul, int kmul,
                                            int off, int div, int up, int dtype) {
  return (int)final_variant(N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, mul, kmul, off, div, up, dtype);
}

/* output-channel tile (128 / 64 / 32) the LDS-DMA ring kernel (variant 1) runs for N x Ho x Wo rows, Ci -> Co channels */
extern "C" int mi355_conv2d_igemm_dma_tile(int N, int Ho, int Wo, int Ci, int Co) { return dma_tile_n((long long)N * Ho * Wo, Ci, Co); }
extern "C" int mi355_conv2d_igemm_generic_tile(int N, int Ho, int Wo, int Co) { return small_grid_tile_n((long long)N * Ho * Wo, Co); }

extern "C" int mi355_conv2d_igemm_stat_rows(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int mul,
                                            int kmul, int off, int div, int up, int dtype) {
  switch (final_variant(N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, mul, kmul, off, div, up, dtype)) {
    case IG_HALO_PP:
    case IG_HALO_PP128: return N * (Ho / 16) * (Wo / 32);
    case IG_WS128: return ws_groups(N, Ho, Wo, Co, 4, device_cus());      // persistent kernels: one row per workgroup range
    case IG_WS64: return ws_groups(N, Ho, Wo, Co, 8, device_cus());
    case IG_HALO_8x32: return N * (Ho / 8) * (Wo / 32);
    case IG_HALO_16x16: return N * (Ho / 16) * (Wo / 16);
    case IG_DMA: return ceil_div((long long)N * Ho * Wo, 128);
    case IG_STREAM1x1: return stream1x1_grid((long long)N * Ho * Wo);
    case IG_GEMM256:         // one row per 256-pixel tile; the ConvTranspose2d phases have no statistics epilogue
      return gemm256_mode(N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, mul, kmul, off, div, up) == 1 ? (int)((long long)N * Ho * Wo / 256) : 0;
    default: return 0;       // generic kernel: no fused statistics, run mi355_bn_stats
  }
}

extern "C" int mi355_conv2d_igemm(const void* in, const void* wk, const float* bias, void* out, int N, int Hi, int Wi,
                                  int Ci, int ldi, int Ho, int Wo, int Co, int ldo, int KH, int KW, int mul, int kmul,
                                  int off, int div, int up, int accumulate, float* stats, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(in && wk && out, "conv2d_igemm: null pointer");
  MI355_CHECK_ARG(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && KH > 0 && KW > 0, "conv2d_igemm: bad extent");
  MI355_CHECK_ARG(div == 1 || div == 2 || div == 4, "conv2d_igemm: div must be 1, 2 or 4 (got %d)", div);
  MI355_CHECK_ARG(Co % 32 == 0, "conv2d_igemm: Co=%d must be a multiple of 32", Co);
  MI355_CHECK_ARG(ldi >= Ci && ldo >= Co, "conv2d_igemm: channel stride smaller than channel count");
  MI355_CHECK_ARG((long long)N * Ho * Wo < (1ll << 31) && (long long)N * Hi * Wi < (1ll << 31),
                  "conv2d_igemm: pixel count overflows int32");
  MI355_CHECK_ARG(dtype == MI355_BF16 || dtype == MI355_F16 || dtype == MI355_F32, "conv2d_igemm: unknown dtype %d", dtype);
  const int esz = dtype_is_2byte(dtype) ? 2 : 4;
  MI355_CHECK_ARG(((uintptr_t)in % 16) == 0 && ((uintptr_t)wk % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                      (ldi * esz) % 16 == 0 && (ldo * esz) % 16 == 0,
                  "conv2d_igemm: pointers / channel strides must be 16-byte aligned");
  MI355_CHECK_ARG(Ci % (esz == 2 ? 32 : 16) == 0, "conv2d_igemm: Ci=%d must be a multiple of %d", Ci, esz == 2 ? 32 : 16);
  // the halo / ping-pong / weight-stationary kernels address an image through a buffer descriptor with 32-bit lane offsets and a
  // 2^31-byte range check (dma.hpp): an image of 2 GiB or more would be zero-filled silently — final_variant sends such shapes to
  // the LDS-DMA ring kernel with its 64-bit addresses (judged on Ci, which the statistics-row query knows too; a channel STRIDE
  // that alone pushes an image over the limit is refused below)
  const IgemmVariant v = final_variant(N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, mul, kmul, off, div, up, dtype);
  MI355_CHECK_ARG(!halo_family(v) || image_fits_descriptor(Hi, Wi, ldi, esz),
                  "conv2d_igemm: an image of %d x %d pixels at a channel stride of %d is 2 GiB or more: beyond the 32-bit lane offsets of "
                  "the halo kernels' buffer descriptors", Hi, Wi, ldi);
  MI355_CHECK_ARG(!stats || (v != IG_GENERIC && !(accumulate & 1)),
                  "conv2d_igemm: fused statistics are not available for this shape/dtype (mi355_conv2d_igemm_stat_rows == 0)");
  ConvArgs a;
  a.in = in; a.wk = wk; a.bias = bias; a.out = out;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.ldi = ldi;
  a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.ldo = ldo;
  a.KH = KH; a.KW = KW; a.mul = mul; a.kmul = kmul; a.off = off;
  a.dshift = div == 1 ? 0 : (div == 2 ? 1 : 2);
  a.up = up ? 1 : 0;
  a.accumulate = accumulate & 1;
  a.relu = (accumulate >> 1) & 1;
  a.pool2 = (accumulate >> 2) & 1;
  MI355_CHECK_ARG(!a.pool2 || ((v == IG_HALO_8x32 || v == IG_HALO_16x16 || v == IG_HALO_PP || v == IG_HALO_PP128 || v == IG_WS64 || v == IG_WS128) && !stats),
                  "conv2d_igemm: the 2x2-sum epilogue exists for the halo kernel only (mi355_conv2d_igemm_variant >= 2)");
  a.stats = stats;
  a.M = N * Ho * Wo;
  a.HoWo = Ho * Wo;
  a.Hlog = up ? 2 * Hi : Hi;
  a.Wlog = up ? 2 * Wi : Wi;
  hipStream_t st = (hipStream_t)s;
  const bool k64 = Ci % 64 == 0;
  if (esz == 4) return launch_bn<float, 16>(a, st);      // (32-channel K tiles measured: +2 % on 128-wide tiles, -13 % on 64-wide ones)
  return dispatch_dtype(dtype, "conv2d_igemm", [&](auto tag) -> int {
    using T = decltype(tag);
    if constexpr (sizeof(T) == 2) {
      switch (v) {
        case IG_HALO_PP: return launch_halo_pp<T>(a, st);
        case IG_HALO_PP128: return launch_halo_pp128<T>(a, st);
        case IG_WS64: return launch_ws<T, 64, 8>(a, st, device_cus());
        case IG_WS128: return launch_ws<T, 128, 4>(a, st, device_cus());
        case IG_HALO_8x32: return launch_halo_rw<T, 8, 32>(a, st);
        case IG_HALO_16x16: return launch_halo_rw<T, 16, 16>(a, st);
        case IG_STREAM1x1: return launch_stream1x1<T>(a, st);
        case IG_GEMM256: return launch_gemm256<T>(a, gemm256_mode(N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, mul, kmul, off, div, up), st);
        case IG_DMA:
          // measured on MI355X (AttentionUNet shapes): the 2-deep BK=64 ring (2 workgroups/CU) wins for 128-wide tiles,
          // the 3-deep BK=32 ring (3-4 workgroups/CU) for 64-wide tiles and for Ci % 64 != 0
          switch (dma_tile_n(a.M, Ci, Co)) {
            case 128: return k64 ? launch_dma<T, 128, 64, 2>(a, st) : launch_dma<T, 128, 32, 3>(a, st);
            case 64: return launch_dma<T, 64, 32, 3>(a, st);
            default: return launch_dma<T, 32, 64, 3>(a, st);
          }
        default: return k64 ? launch_bn<T, 64>(a, st) : launch_bn<T, 32>(a, st);
      }
    }
    return MI355_ERR_UNSUPPORTED;
  });
}
