// GPU side of the input pipeline: what the reference does per sample on CPU DataLoader workers with
// Albumentations / OpenCV (utils/trainer.py:52-115 transforms, utils/dataset.py:100-134 mask handling), as two batched
// kernels over uint8 images that are already resident in HBM:
//   warp_u8      : dst <- bilinear / nearest sample of src under a per-sample inverse affine map (A.Resize =
//                  half-pixel scale map with replicated border; A.ShiftScaleRotate + A.HorizontalFlip = rotation /
//                  scale / shift matrix with BORDER_REFLECT_101), uint8 in, uint8 out (rounded like cv2)
//   normalize_u8 : A.RandomBrightnessContrast (alpha * v + beta * 255, clipped and rounded in the uint8 domain) +
//                  A.Normalize ((v / 255 - mean) / std) + ToTensorV2 (HWC -> CHW fp32); masks: v / 255
// HBM-bound and tiny next to a train step (a 32-image batch is 6 MB); one thread per output pixel.
#include "common.hpp"

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

// m: [N][6] row-major 2x3, maps dst pixel (x, y) to src coordinates: sx = m0*x + m1*y + m2, sy = m3*x + m4*y + m5
__global__ void warp_u8_kernel(const uint8_t* __restrict__ src, int Hs, int Ws, const float* __restrict__ m, uint8_t* __restrict__ dst,
                               int H, int W, int C, int nearest, int reflect, long long total) {
#pragma clang fp contract(off)      // no FMA contraction: the interpolation is then bit-reproducible against a plain IEEE evaluation
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long long r = i / W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    const float* mm = m + (size_t)n * 6;
    const float sx = mm[0] * x + mm[1] * y + mm[2], sy = mm[3] * x + mm[4] * y + mm[5];
    const uint8_t* s = src + (size_t)n * Hs * Ws * C;
    uint8_t* d = dst + ((size_t)(n * H + y) * W + x) * C;
    auto at = [&](int yy, int xx, int c) -> float {
      if (reflect) { yy = reflect101(yy, Hs); xx = reflect101(xx, Ws); }
      else { yy = min(max(yy, 0), Hs - 1); xx = min(max(xx, 0), Ws - 1); }
      return (float)s[((size_t)yy * Ws + xx) * C + c];
    };
    if (nearest) {
      const int xi = (int)floorf(sx + 0.5f), yi = (int)floorf(sy + 0.5f);
      for (int c = 0; c < C; ++c) d[c] = (uint8_t)at(yi, xi, c);
    } else {
      const float fx = floorf(sx), fy = floorf(sy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float ax = sx - fx, ay = sy - fy;
      for (int c = 0; c < C; ++c) {
        const float top = at(y0, x0, c) * (1.f - ax) + at(y0, x0 + 1, c) * ax;
        const float bot = at(y0 + 1, x0, c) * (1.f - ax) + at(y0 + 1, x0 + 1, c) * ax;
        d[c] = (uint8_t)fminf(fmaxf(rintf(top * (1.f - ay) + bot * ay), 0.f), 255.f);
      }
    }
  }
}

extern "C" int mi355_warp_u8(const uint8_t* src, int N, int Hs, int Ws, int C, const float* m, uint8_t* dst, int H, int W, int nearest,
                             int reflect, mi355_stream_t s) {
  MI355_CHECK_ARG(src && m && dst && N > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0 && C > 0 && C <= 4, "warp_u8: bad arguments");
  const long long total = (long long)N * H * W;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(warp_u8_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, src, Hs, Ws, m, dst, H, W, C, nearest, reflect, total);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// out[n][c][y][x] = ((clip(round(alpha_n * v + beta_n * 255)) / 255) - mean_c) / std_c; bc == NULL: alpha 1, beta 0;
// mean == NULL: out = v / 255 (masks)
__global__ void normalize_u8_kernel(const uint8_t* __restrict__ src, const float* __restrict__ bc, const float* __restrict__ mean,
                                    const float* __restrict__ stdv, float* __restrict__ out, int HW, int C, long long total) {
#pragma clang fp contract(off)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int n = (int)(i / HW);
    const float alpha = bc ? bc[2 * n] : 1.f, beta = bc ? bc[2 * n + 1] : 0.f;
    for (int c = 0; c < C; ++c) {
      float v = (float)src[((size_t)n * HW + p) * C + c];
      if (bc) v = fminf(fmaxf(rintf(alpha * v + beta * 255.f), 0.f), 255.f);
      v *= (1.f / 255.f);
      if (mean) v = (v - mean[c]) / stdv[c];
      out[((size_t)n * C + c) * HW + p] = v;
    }
  }
}

extern "C" int mi355_normalize_u8(const uint8_t* src, int N, int H, int W, int C, const float* bc, const float* mean, const float* stdv,
                                  float* out, mi355_stream_t s) {
  MI355_CHECK_ARG(src && out && N > 0 && H > 0 && W > 0 && C > 0 && (!mean || stdv), "normalize_u8: bad arguments");
  const long long total = (long long)N * H * W;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(normalize_u8_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, src, bc, mean, stdv, out, H * W, C, total);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
