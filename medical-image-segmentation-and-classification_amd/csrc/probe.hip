// Hardware ceiling probes (not on the product path): what does the chip sustain on the MFMA shape the conv kernels
// use, with RANDOM operands, when nothing else is in the way?  The roofline denominators in DESIGN.md come from
// MI355X_MICROARCH.md; these loops measure, on the box at hand, the clock-limited ceiling that section describes.
//   mode 0: 16x16x32 bf16 MFMAs back to back, operands in registers, 8 independent accumulators per wave
//   mode 1: the same MFMA stream with the halo kernel's LDS diet: 18 ds_read_b128 fragment reads per 48 MFMAs
//   mode 2: 32x32x16 bf16, registers only
#include "common.hpp"

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe_mfma_kernel(const uint4* __restrict__ rnd, int iters, float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += 256) reinterpret_cast<uint4*>(lds)[i] = rnd[(blockIdx.x * 2048 + i) & 65535];
  __syncthreads();
  bf16x8 a[6], b[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, rnd[(tid * 13 + i * 7 + blockIdx.x) & 65535]);
    b[i] = __builtin_bit_cast(bf16x8, rnd[(tid * 29 + i * 3 + blockIdx.x + 4096) & 65535]);
  }
  if (MODE == 2) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + i) % 6], b[k], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) sink[0] = s;
    return;
  }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* base = lds + ((lane & 15) * 64 + ((lane >> 4) ^ (((lane & 15) >> 2) & 1) << 1) * 16);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {                      // 18 fragment reads feeding the 48 MFMAs below (conflict-free, as in the halo kernel)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(base + ((it + i) & 15) * 1024);
        b[i] = *reinterpret_cast<const bf16x8*>(base + 16384 + ((it + 2 * i) & 15) * 1024);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(base + ((it + i + 7) & 15) * 1024);
        a[i] = i & 1 ? x : a[i];
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + k) % 6], b[(i / 3 + k) % 6], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
}

/* rnd: 65536 x 16 B of random bf16 bit patterns (finite values); grid = blocks x 256 threads; returns the FLOPs issued */
extern "C" int mi355_probe_mfma(int mode, const void* rnd, int blocks, int iters, float* sink, mi355_stream_t s) {
  MI355_CHECK_ARG(rnd && sink && blocks > 0 && iters > 0 && mode >= 0 && mode <= 2, "probe_mfma: bad arguments");
  if (mode == 0) hipLaunchKernelGGL(probe_mfma_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const uint4*)rnd, iters, sink);
  else if (mode == 1) hipLaunchKernelGGL(probe_mfma_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const uint4*)rnd, iters, sink);
  else hipLaunchKernelGGL(probe_mfma_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const uint4*)rnd, iters, sink);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}


// ---- LDS-DMA through a buffer descriptor: what the range check writes ------------------------------------------------------
// 64 lanes fetch 16 B each from `src` (+ soff bytes); lanes with bit k of `pad_mask` set use the always-out-of-range offset;
// valid == 0 gives the descriptor num_records = 0.  out[256] = the 1 KiB that arrived in LDS (pre-filled with 0x7f bytes).
#include "dma.hpp"
__global__ void probe_bufdma_kernel(const void* src, unsigned soff, unsigned long long pad_mask, int valid, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
  const int lane = threadIdx.x;
  reinterpret_cast<uint4*>(lds)[lane] = make_uint4(0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu);
  __syncthreads();
  const bufdesc_t d = make_buf(src, valid != 0);
  dma16_buf(d, ((pad_mask >> lane) & 1) ? DMA_PAD : lane * 16u, __builtin_amdgcn_readfirstlane(soff), lds_addr(lds));
  wait_vmcnt<0>();
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = reinterpret_cast<unsigned*>(lds)[i];
}
extern "C" int mi355_probe_bufdma(const void* src, unsigned soff, unsigned long long pad_mask, int valid, unsigned* out, mi355_stream_t s) {
  hipLaunchKernelGGL(probe_bufdma_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, src, soff, pad_mask, valid, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
