// Pointwise (1x1, stride 1) convolution for NARROW channel counts, as a streaming kernel.
//
// The attention gates' W_g / W_x projections and their data gradients (AttentionUNet.py:33-45) are 1x1 convolutions with
// 32..128 channels on either side over 10^5..10^6 pixels: 0.5-1 flop per byte, i.e. HBM-bound by a wide margin, and with
// ONE K tile the staged implicit-GEMM kernel is all prologue and epilogue (measured 1.5-3.2 TB/s).  Here the whole weight
// matrix lives in registers as MFMA fragments for the life of the workgroup, pixels stream from global memory straight into
// fragment registers (a lane's 16 bytes are 8 consecutive channels of one pixel, which is exactly the 16x16x32 operand
// layout: no LDS on the way in), and the waves of a workgroup never synchronise: each sweeps its own pixel tiles.
//
// The MFMAs run with the weight fragment as the A operand (as in conv3x3_halo.hpp), so a lane ends up with FOUR CONSECUTIVE
// OUTPUT CHANNELS of ONE pixel; the tile is staged through a wave-private LDS strip with 8-byte writes and leaves as 16-byte
// row-contiguous stores (whole 64..256-B pixel rows).
//
// Epilogue features match the other forward kernels: bias, ReLU, fp32 accumulate onto the destination, and the fused
// BatchNorm statistics of the ROUNDED outputs (one partial row per WORKGROUP: mi355_conv2d_igemm_stat_rows).
#pragma once
#include "common.hpp"

static inline int stream1x1_grid(long long M) {
  long long g = (M + 255) / 256;
  if (g > 2048) g = 2048;      // 8 workgroups per CU, grid-stride beyond
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T, int CI, int CO>
struct Stream1x1Cfg {
  static constexpr int KK = CI / 32, NB = CO / 16;
  // pixel blocks (of 16) per wave pass, sized so that weights + pixels + accumulators stay near 128 VGPRs
  static constexpr int MB = (KK * NB <= 4) ? 4 : 2;
  static constexpr int PITCH = CO * 2 + 16;
  static constexpr int STRIP = MB * 16 * PITCH + CO * 4;        // wave-private staging strip + fp32 bias copy
  static constexpr int LDS_BYTES = 4 * STRIP;
};

template <typename T, int CI, int CO>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(const ConvArgs a) {
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  using Cfg = Stream1x1Cfg<T, CI, CO>;
  constexpr int KK = Cfg::KK, NB = Cfg::NB, MB = Cfg::MB, PITCH = Cfg::PITCH;
  constexpr int EPC = 8, CPRC = CO / EPC, ROWS = MB * 16;
  static_assert(Cfg::LDS_BYTES >= 4 * 2 * CO * 4, "statistics scratch must fit");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, c4 = lane >> 4;
  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wk = reinterpret_cast<const T*>(a.wk);
  T* __restrict__ out = reinterpret_cast<T*>(a.out);
  unsigned char* const strip = lds + wave * Cfg::STRIP;

  // the whole [CO][CI] weight matrix as fragments: block nb holds rows nb*16 + l16, channels kk*32 + c4*8 .. +7
  bf16x8 wfr[NB][KK];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      wfr[nb][kk] = *reinterpret_cast<const bf16x8*>(wk + (size_t)(nb * 16 + l16) * CI + kk * 32 + c4 * 8);
  // bias: a wave-private fp32 copy behind the strip, read back as the STARTING accumulator of every block (no registers held)
  float* const biasl = reinterpret_cast<float*>(strip + ROWS * PITCH);
  for (int c = lane; c < CO; c += 64) biasl[c] = a.bias ? a.bias[c] : 0.f;
  const float lo = a.relu ? 0.f : -INFINITY;
  // statistics are taken in the store loop, where a lane always meets the same 16-byte channel chunk (id % CPRC == lane % CPRC)
  static_assert(64 % CPRC == 0, "a lane must keep its channel chunk across the store loop");
  float sm[EPC], sq[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
  struct alignas(8) Pack4 { T v[4]; };
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const long long M = a.M;
  const long long stride = (long long)gridDim.x * 4 * ROWS;
  // pixel fragments of one wave pass; rows past the end re-read the last pixel (never stored).  The NEXT pass's fragments are
  // requested as soon as this pass's MFMAs have consumed the registers, so the loads fly during the store phase.
  bf16x8 xfr[MB][KK];
  auto fetch = [&](long long m0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const long long m = m0 + mb * 16 + l16;
      const T* p = in + (size_t)(m < M ? m : M - 1) * a.ldi + c4 * 8;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) xfr[mb][kk] = *reinterpret_cast<const bf16x8*>(p + kk * 32);
    }
  };
  const long long mfirst = ((long long)blockIdx.x * 4 + wave) * ROWS;
  if (mfirst < M) fetch(mfirst);
  for (long long m0 = mfirst; m0 < M; m0 += stride) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(biasl + nb * 16 + 4 * c4);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) acc = mfma_16x16x32<T>(wfr[nb][kk], xfr[mb][kk], acc);      // D[channel][pixel]
        Pack4 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pk.v[r] = from_f32<T>(fmaxf(acc[r], lo));
        *reinterpret_cast<Pack4*>(strip + (mb * 16 + l16) * PITCH + (nb * 16 + 4 * c4) * 2) = pk;
      }
    if (m0 + stride < M) fetch(m0 + stride);
    // the strip is wave-private: LDS executes a wave's accesses in order, the fences only pin the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int id = lane; id < ROWS * CPRC; id += 64) {
      const int row = id / CPRC, c = id - row * CPRC;
      const long long m = m0 + row;
      if (m < M) {
        T* p = out + (size_t)m * a.ldo + c * EPC;
        Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(strip + row * PITCH + c * 16);
        if (a.stats) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float f = to_f32<T>(v.v[e]);
            sm[e] += f;
            sq[e] += f * f;
          }
        }
        if (a.accumulate) {
          const Vec16<T> o = ld16<T>(p);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) + to_f32<T>(o.v[e]));
        }
        st16<T>(p, v);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  if (a.stats) {        // one partial row per workgroup: fold the lanes that share a channel chunk, then the four waves through LDS
    __syncthreads();
    float* const red = reinterpret_cast<float*>(lds);      // [4 waves][2][CO]
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float s1 = sm[e], s2 = sq[e];
#pragma unroll
      for (int m = CPRC; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
      if (lane < CPRC) {
        red[(wave * 2 + 0) * CO + lane * EPC + e] = s1;
        red[(wave * 2 + 1) * CO + lane * EPC + e] = s2;
      }
    }
    __syncthreads();
    for (int id = tid; id < 2 * CO; id += 256) {
      const int q = id / CO, c = id - q * CO;
      a.stats[((size_t)blockIdx.x * 2 + q) * CO + c] = (red[(0 * 2 + q) * CO + c] + red[(1 * 2 + q) * CO + c]) +
                                                       (red[(2 * 2 + q) * CO + c] + red[(3 * 2 + q) * CO + c]);
    }
  }
}

// shapes served: (Ci, Co) with the weight fragments in at most 64 VGPRs
static inline bool stream1x1_shape(int Ci, int Co) {
  return (Ci == 64 && Co == 32) || (Ci == 32 && Co == 64) || (Ci == 128 && Co == 64) || (Ci == 64 && Co == 128) ||
         (Ci == 64 && Co == 64) || (Ci == 32 && Co == 32);
}

template <typename T, int CI, int CO>
static int launch_stream1x1_shape(const ConvArgs& a, hipStream_t s) {
  using Cfg = Stream1x1Cfg<T, CI, CO>;
  hipLaunchKernelGGL((conv1x1_stream_kernel<T, CI, CO>), dim3(stream1x1_grid(a.M)), dim3(256), Cfg::LDS_BYTES, s, a);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

template <typename T>
static int launch_stream1x1(const ConvArgs& a, hipStream_t s) {
  if (a.Ci == 64 && a.Co == 32) return launch_stream1x1_shape<T, 64, 32>(a, s);
  if (a.Ci == 32 && a.Co == 64) return launch_stream1x1_shape<T, 32, 64>(a, s);
  if (a.Ci == 128 && a.Co == 64) return launch_stream1x1_shape<T, 128, 64>(a, s);
  if (a.Ci == 64 && a.Co == 128) return launch_stream1x1_shape<T, 64, 128>(a, s);
  if (a.Ci == 64 && a.Co == 64) return launch_stream1x1_shape<T, 64, 64>(a, s);
  if (a.Ci == 32 && a.Co == 32) return launch_stream1x1_shape<T, 32, 32>(a, s);
  mi355_set_error("conv1x1 stream kernel: no instance for Ci=%d Co=%d", a.Ci, a.Co);
  return MI355_ERR_ARG;
}
