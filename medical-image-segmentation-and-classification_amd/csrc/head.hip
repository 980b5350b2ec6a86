// Classifier heads: global max/avg pooling, tiny fp32 Linear layers, dropout (ResNet.py:112-115,
// VGG.py:103-122, helpers.py:124-143).  M = batch is tiny here, so these are latency-bound VALU
// kernels; they exist so that the whole step runs on the library without torch math.
#include "common.hpp"

// one workgroup per (n, 64-channel group): 4 waves split the HW positions, lanes own channels
template <typename T>
__global__ __launch_bounds__(256) void global_pool_fwd_kernel(const T* __restrict__ x, int ldx, float* __restrict__ y,
                                                              int32_t* __restrict__ argmax, int HW, int C, int is_max) {
  __shared__ float sv[4][64];
  __shared__ int si[4][64];
  const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  float best = is_max ? -INFINITY : 0.f;
  int bi = 0;
  if (c < C) {
    for (int p = wave; p < HW; p += 4) {
      const float v = to_f32<T>(x[((size_t)n * HW + p) * ldx + c]);
      if (is_max) {
        if (v > best) { best = v; bi = p; }
      } else {
        best += v;
      }
    }
  }
  sv[wave][threadIdx.x & 63] = best;
  si[wave][threadIdx.x & 63] = bi;
  __syncthreads();
  if (wave == 0 && c < C) {
    float r = sv[0][threadIdx.x];
    int ri = si[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) {
      const float v = sv[w][threadIdx.x];
      const int vi = si[w][threadIdx.x];
      if (is_max) {
        if (v > r || (v == r && vi < ri)) { r = v; ri = vi; }   // first maximum in scan order
      } else {
        r += v;
      }
    }
    y[(size_t)n * C + c] = is_max ? r : r / (float)HW;
    if (argmax) argmax[(size_t)n * C + c] = ri;
  }
}

extern "C" int mi355_global_pool_fwd(const void* x, int ldx, float* y, int32_t* argmax, int N, int HW, int C, int is_max,
                                     int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(x && y && (!is_max || argmax), "global_pool_fwd: null pointer");
  dim3 grid(ceil_div(C, 64), N);
  return dispatch_dtype(dtype, "global_pool_fwd_kernel", [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((global_pool_fwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, y, argmax, HW, C, is_max);
    MI355_LAUNCH_CHECK();
    return (int)MI355_OK;
  });
}

template <typename T>
__global__ void global_pool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ argmax, T* __restrict__ dx,
                                       int lddx, int HW, int C, int is_max, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long np = i / C;
    const int p = (int)(np % HW);
    const long long n = np / HW;
    const float g = dy[n * C + c];
    const float v = is_max ? (argmax[n * C + c] == p ? g : 0.f) : g / (float)HW;
    dx[(size_t)np * lddx + c] = from_f32<T>(v);
  }
}

extern "C" int mi355_global_pool_bwd(const float* dy, const int32_t* argmax, void* dx, int lddx, int N, int HW, int C,
                                     int is_max, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(dy && dx && (!is_max || argmax), "global_pool_bwd: null pointer");
  const long long total = (long long)N * HW * C;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return dispatch_dtype(dtype, "global_pool_bwd_kernel", [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((global_pool_bwd_kernel<T>), dim3((int)blocks), dim3(256), 0, (hipStream_t)s, dy, argmax, (T*)dx, lddx, HW, C, is_max, total);
    MI355_LAUNCH_CHECK();
    return (int)MI355_OK;
  });
}

// ---- Linear (fp32): one wave per output element -------------------------------------------------------
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B, int I,
                                                         int O, int relu) {
  const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (gw >= B * O) return;
  const int b = gw / O, o = gw - b * O;
  float acc = 0.f;
  for (int i = lane; i < I; i += 64) acc += x[(size_t)b * I + i] * w[(size_t)o * I + i];
  acc = wave_sum(acc);
  if (lane == 0) {
    acc += bias ? bias[o] : 0.f;
    y[gw] = relu ? fmaxf(acc, 0.f) : acc;
  }
}

// ---- wide Linear layers (torchvision VGG head: 25088 -> 4096 -> 4096) ------------------------------------------------
// M = batch is tiny, so these are streaming passes over the weight matrix: every kernel below touches W (or dW)
// exactly once per 16 batch rows with 16-byte accesses, keeps the batch-side operand in registers and never uses
// atomics.  Algorithmic bytes: forward O*I*4 read; backward O*I*4 read (dx) + O*I*4 written (dW).
constexpr int LIN_BT = 16;      // batch rows per pass

// y[b][o] = act(sum_i x[b][i] w[o][i] + bias[o]): a wave owns LIN_R consecutive output rows and strides them in float4s, so a
// 16-byte piece of every batch row (L1 / L2 traffic: 16 loads) meets LIN_R weight loads instead of one — with one row per wave the
// kernel moved 17 bytes through the L1 per weight byte and ran at 0.66 TB/s (scripts/linear_bench.py).
#ifndef LIN_R_ROWS
#define LIN_R_ROWS 4
#endif
constexpr int LIN_R = LIN_R_ROWS;
typedef float lin_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lin_ldnt(const float4* p) {      // streaming 16-byte load / store of the weight matrix
  const lin_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const lin_f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lin_stnt(float4* p, const float4& a) {
  const lin_f32x4 v = {a.x, a.y, a.z, a.w};
  __builtin_nontemporal_store(v, reinterpret_cast<lin_f32x4*>(p));
}
__global__ __launch_bounds__(256) void linear_fwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y, int B, int I,
                                                              int O, int relu) {
  // a workgroup owns LIN_R output rows; its four waves take a quarter of the columns each (1024 workgroups for O = 4096: with one
  // wave per SIMD the 98 dependent load-then-multiply trips of a row were latency-bound) and meet in LDS
  __shared__ float red[4][LIN_R * LIN_BT];
  const int o0 = blockIdx.x * LIN_R, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b0 = blockIdx.y * LIN_BT, nb = min(LIN_BT, B - b0);
  float acc[LIN_R][LIN_BT];
#pragma unroll
  for (int r = 0; r < LIN_R; ++r)
#pragma unroll
    for (int b = 0; b < LIN_BT; ++b) acc[r][b] = 0.f;
  const int I4 = I >> 2;
  const int chunk = ((I4 + 3) / 4 + 63) / 64 * 64;              // columns (in float4s) per wave, a multiple of the wave's stride
  const int i_end = min(I4, (wave + 1) * chunk);
  const float4* wr[LIN_R];
#pragma unroll
  for (int r = 0; r < LIN_R; ++r) wr[r] = reinterpret_cast<const float4*>(w + (size_t)min(o0 + r, O - 1) * I);      // (rows past the end: clamped, not stored)
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)b0 * I);
  // the next trip's weight pieces are requested before this trip's multiply-adds (columns past the end: clamped, weights zeroed)
  int i = wave * chunk + lane;
  float4 wv[LIN_R];
#pragma unroll
  for (int r = 0; r < LIN_R; ++r) wv[r] = lin_ldnt(wr[r] + min(i, I4 - 1));
  for (; i < i_end; i += 64) {
    float4 wn[LIN_R];
    const int inx = min(i + 64, I4 - 1);
#pragma unroll
    for (int r = 0; r < LIN_R; ++r) wn[r] = lin_ldnt(wr[r] + inx);
#pragma unroll
    for (int b = 0; b < LIN_BT; ++b) {
      const float4 xv = xr[(size_t)min(b, nb - 1) * I4 + i];      // (unconditional, clamped: a guarded load is waited for before the next)
#pragma unroll
      for (int r = 0; r < LIN_R; ++r) acc[r][b] += wv[r].x * xv.x + wv[r].y * xv.y + wv[r].z * xv.z + wv[r].w * xv.w;
    }
#pragma unroll
    for (int r = 0; r < LIN_R; ++r) wv[r] = wn[r];
  }
#pragma unroll
  for (int r = 0; r < LIN_R; ++r)
#pragma unroll
    for (int b = 0; b < LIN_BT; ++b) {
      const float v = wave_sum(acc[r][b]);
      if (lane == 0) red[wave][r * LIN_BT + b] = v;
    }
  __syncthreads();
  if (threadIdx.x < LIN_R * LIN_BT) {
    const int r = threadIdx.x / LIN_BT, b = threadIdx.x - r * LIN_BT;
    if (b < nb && o0 + r < O) {
      const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]) + (bias ? bias[o0 + r] : 0.f);
      y[(size_t)(b0 + b) * O + o0 + r] = relu ? fmaxf(t, 0.f) : t;
    }
  }
}

// The backward kernels keep the masked output gradient of their rows, g[o][b] = dy[b][o] * (relu ? y[b][o] > 0 : 1), in LDS
// (16 floats per output row, read back as four broadcast ds_read_b128): fetched per output row through wave-uniform global
// loads, 32 dependent scalar loads sat in front of every weight row (dx 0.85 TB/s, dW 1.4 TB/s).
constexpr int LIN_GROWS = 128;      // output rows per workgroup of the dx kernel (an 8 KB tile of g)
__device__ __forceinline__ void linear_stage_g(float (*lg)[LIN_BT], const float* __restrict__ y, const float* __restrict__ dy, int b0,
                                               int nb, int o0, int rows, int O, int relu) {
  for (int idx = threadIdx.x; idx < rows * LIN_BT; idx += blockDim.x) {
    const int r = idx / LIN_BT, b = idx - r * LIN_BT;
    float g = 0.f;
    if (b < nb) {
      g = dy[(size_t)(b0 + b) * O + o0 + r];
      if (relu && !(y[(size_t)(b0 + b) * O + o0 + r] > 0.f)) g = 0.f;
    }
    lg[r][b] = g;
  }
  __syncthreads();
}

// partial dx: part[split][b][i] = sum_{o in split} g[b][o] w[o][i]; a thread owns four consecutive i
__global__ __launch_bounds__(256) void linear_dx_wide_kernel(const float* __restrict__ w, const float* __restrict__ y,
                                                             const float* __restrict__ dy, float* __restrict__ part, int B, int I,
                                                             int O, int relu, int o_per_split) {
  __shared__ __attribute__((aligned(16))) float lg[LIN_GROWS][LIN_BT];
  const int i4 = blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.z * LIN_BT, nb = min(LIN_BT, B - b0);
  const int o0 = blockIdx.y * o_per_split, rows = min(o_per_split, O - o0);
  linear_stage_g(lg, y, dy, b0, nb, o0, rows, O, relu);
  if (i4 * 4 >= I) return;
  float4 acc[LIN_BT];
#pragma unroll
  for (int b = 0; b < LIN_BT; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* wp = reinterpret_cast<const float4*>(w + (size_t)o0 * I) + i4;
  const size_t stride = (size_t)(I >> 2);
  int r = 0;
  for (; r + 4 <= rows; r += 4) {                    // four weight rows in flight
    float4 wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wv[u] = lin_ldnt(wp + (size_t)(r + u) * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < LIN_BT / 4; ++q) {
        const float4 g = *reinterpret_cast<const float4*>(&lg[r + u][4 * q]);
        const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float4& a = acc[4 * q + k];
          a.x += gs[k] * wv[u].x; a.y += gs[k] * wv[u].y; a.z += gs[k] * wv[u].z; a.w += gs[k] * wv[u].w;
        }
      }
  }
  for (; r < rows; ++r) {
    const float4 wv = wp[(size_t)r * stride];
#pragma unroll
    for (int b = 0; b < LIN_BT; ++b) {
      const float g = lg[r][b];
      acc[b].x += g * wv.x; acc[b].y += g * wv.y; acc[b].z += g * wv.z; acc[b].w += g * wv.w;
    }
  }
#pragma unroll
  for (int b = 0; b < LIN_BT; ++b)
    if (b < nb) reinterpret_cast<float4*>(part + ((size_t)blockIdx.y * B + b0 + b) * I)[i4] = acc[b];
}

__global__ void linear_dx_fold_kernel(const float* __restrict__ part, float* __restrict__ dx, int splits, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
    dx[i] = s;
  }
}

// dW[o][i] = beta*dW + sum_b g[b][o] x[b][i] (+ db): a thread keeps x[:, i..i+3] in registers and walks 64 output rows
__global__ __launch_bounds__(256) void linear_dw_wide_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy, float* __restrict__ dw,
                                                             float* __restrict__ db, int B, int I, int O, int relu, float beta) {
  __shared__ __attribute__((aligned(16))) float lg[64][LIN_BT];
  const int i4 = blockIdx.x * 256 + threadIdx.x;
  const int o0 = blockIdx.y * 64, rows = min(64, O - o0);
  const bool live = i4 * 4 < I;
  const int i4c = live ? i4 : 0;
  for (int b0 = 0; b0 < B; b0 += LIN_BT) {                     // (B <= 16 in every configuration: one trip)
    const int nb = min(LIN_BT, B - b0);
    if (b0) __syncthreads();
    linear_stage_g(lg, y, dy, b0, nb, o0, rows, O, relu);
    float4 xv[LIN_BT];
#pragma unroll
    for (int b = 0; b < LIN_BT; ++b) xv[b] = reinterpret_cast<const float4*>(x + (size_t)(b0 + min(b, nb - 1)) * I)[i4c];      // (rows past nb meet g == 0)
    const float keep = (b0 == 0) ? beta : 1.f;                  // later batch chunks accumulate onto the first
    for (int r = 0; r < rows; ++r) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float sb = 0.f;
#pragma unroll
      for (int q = 0; q < LIN_BT / 4; ++q) {
        const float4 g = *reinterpret_cast<const float4*>(&lg[r][4 * q]);
        const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4& v = xv[4 * q + k];
          acc.x += gs[k] * v.x; acc.y += gs[k] * v.y; acc.z += gs[k] * v.z; acc.w += gs[k] * v.w;
          sb += gs[k];
        }
      }
      if (live) {
        float4* p = reinterpret_cast<float4*>(dw + (size_t)(o0 + r) * I) + i4;
        if (keep != 0.f) {
          const float4 old = *p;
          acc.x += keep * old.x; acc.y += keep * old.y; acc.z += keep * old.z; acc.w += keep * old.w;
        }
        lin_stnt(p, acc);
      }
      if (db && blockIdx.x == 0 && threadIdx.x == 0) db[o0 + r] = (keep != 0.f ? keep * db[o0 + r] : 0.f) + sb;
    }
  }
}

// output-row ranges of the dx kernel: enough workgroups to fill the chip (the 4096 x 4096 layer has four column blocks), at most
// LIN_GROWS rows each
static inline int linear_dx_splits(int I, int O) {
  const int iblocks = ceil_div(I / 4, 256);
  int splits = ceil_div(768, iblocks);
  if (splits > 64) splits = 64;                       // (every range costs a B x I partial: written, then folded)
  if (splits < ceil_div(O, LIN_GROWS)) splits = ceil_div(O, LIN_GROWS);
  if (splits > O) splits = O;
  return splits;
}

static inline bool linear_wide(int B, int I, int O) { return I % 4 == 0 && (long long)I * O >= (1 << 20) && B >= 1; }

extern "C" int mi355_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int I, int O, int relu,
                                mi355_stream_t s) {
  MI355_CHECK_ARG(x && w && y, "linear_fwd: null pointer");
  if (linear_wide(B, I, O))
    hipLaunchKernelGGL(linear_fwd_wide_kernel, dim3(ceil_div(O, LIN_R), ceil_div(B, LIN_BT)), dim3(256), 0, (hipStream_t)s, x, w, bias,
                       y, B, I, O, relu);
  else
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(ceil_div((long long)B * O, 4)), dim3(256), 0, (hipStream_t)s, x, w, bias, y, B, I, O,
                       relu);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

/* scratch floats mi355_linear_bwd needs for the input gradient of a wide layer (0: none) */
extern "C" int mi355_linear_bwd_scratch(int B, int I, int O) {
  if (!linear_wide(B, I, O)) return 0;
  const long long n = (long long)linear_dx_splits(I, O) * B * I;
  return n > 0x7fffffff ? -1 : (int)n;
}

// g = dy * (relu ? y > 0 : 1);  dx[b][i] = sum_o g[b][o] w[o][i];  dw[o][i] = beta*dw + sum_b g[b][o] x[b][i]
__global__ void linear_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ y, const float* __restrict__ dy,
                                     float* __restrict__ dx, int B, int I, int O, int relu) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * I) return;
  const int b = idx / I, i = idx - b * I;
  float acc = 0.f;
  for (int o = 0; o < O; ++o) {
    float g = dy[(size_t)b * O + o];
    if (relu && !(y[(size_t)b * O + o] > 0.f)) g = 0.f;
    acc += g * w[(size_t)o * I + i];
  }
  dx[idx] = acc;
}

__global__ void linear_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                     float* __restrict__ dw, float* __restrict__ db, int B, int I, int O, int relu,
                                     float beta) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= O * I) return;
  const int o = idx / I, i = idx - o * I;
  float acc = 0.f, accb = 0.f;
  for (int b = 0; b < B; ++b) {
    float g = dy[(size_t)b * O + o];
    if (relu && !(y[(size_t)b * O + o] > 0.f)) g = 0.f;
    acc += g * x[(size_t)b * I + i];
    accb += g;
  }
  dw[idx] = (beta != 0.f ? beta * dw[idx] : 0.f) + acc;
  if (db && i == 0) db[o] = (beta != 0.f ? beta * db[o] : 0.f) + accb;
}

extern "C" int mi355_linear_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                                float* db, int B, int I, int O, int relu, float beta, float* scratch, mi355_stream_t s) {
  MI355_CHECK_ARG(x && w && dy && (!relu || y), "linear_bwd: null pointer");
  if (linear_wide(B, I, O)) {
    if (dx) {
      MI355_CHECK_ARG(scratch, "linear_bwd: a wide layer needs mi355_linear_bwd_scratch(B, I, O) floats of scratch for dx");
      const int splits0 = linear_dx_splits(I, O), ops = ceil_div(O, splits0), splits = ceil_div(O, ops);
      hipLaunchKernelGGL(linear_dx_wide_kernel, dim3(ceil_div(I / 4, 256), splits, ceil_div(B, LIN_BT)), dim3(256), 0, (hipStream_t)s, w,
                         y, dy, scratch, B, I, O, relu, ops);
      hipLaunchKernelGGL(linear_dx_fold_kernel, dim3(ceil_div((long long)B * I, 256) > 2048 ? 2048 : ceil_div((long long)B * I, 256)),
                         dim3(256), 0, (hipStream_t)s, scratch, dx, splits, (long long)B * I);
    }
    if (dw)
      hipLaunchKernelGGL(linear_dw_wide_kernel, dim3(ceil_div(I / 4, 256), ceil_div(O, 64)), dim3(256), 0, (hipStream_t)s, x, y, dy, dw,
                         db, B, I, O, relu, beta);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  }
  if (dx) hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(ceil_div((long long)B * I, 256)), dim3(256), 0, (hipStream_t)s, w, y, dy, dx, B, I, O, relu);
  if (dw) hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(ceil_div((long long)O * I, 256)), dim3(256), 0, (hipStream_t)s, x, y, dy, dw, db, B, I, O, relu, beta);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// ---- dropout ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint64_t k) {   // splitmix64 finaliser
  k += 0x9E3779B97F4A7C15ull;
  k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull;
  k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((k ^ (k >> 31)) >> 32);
}

__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask, long long n,
                                   float p, uint64_t seed, const int32_t* __restrict__ counter) {
  const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  if (counter) seed ^= (uint64_t)(uint32_t)counter[0] * 0xD1342543DE82EF95ull;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float u = (float)(mix32(seed * 0x100000001B3ull + (uint64_t)i) >> 8) * (1.0f / 16777216.0f);
    const uint8_t keep = u >= p;
    mask[i] = keep;
    y[i] = keep ? x[i] * scale : 0.f;
  }
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask, float* __restrict__ dx,
                                   long long n, float p) {
  const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = mask[i] ? dy[i] * scale : 0.f;
}

extern "C" int mi355_dropout_fwd(const float* x, float* y, uint8_t* mask, long long n, float p, uint64_t seed,
                                 const int32_t* counter, mi355_stream_t s) {
  MI355_CHECK_ARG(x && y && mask && p >= 0.f && p <= 1.f, "dropout_fwd: bad arguments");
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)s, x, y, mask, n, p, seed, counter);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

extern "C" int mi355_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, long long n, float p, mi355_stream_t s) {
  MI355_CHECK_ARG(dy && mask && dx, "dropout_bwd: null pointer");
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)s, dy, mask, dx, n, p);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// ---- joint pipeline glue (utils/pipeline.py:324-357, 359-418) --------------------------------------------------------
// classify -> keep the samples of one class -> segment those -> binary masks.  Per sample: pred = argmax softmax,
// conf = 100 * max softmax; the kept samples' batch indices are compacted in order (the count goes back to the
// host once, to size the segmentation launch).
__global__ void cls_decide_kernel(const float* __restrict__ logits, int B, int C, int keep_class, int32_t* __restrict__ pred,
                                  float* __restrict__ conf, int32_t* __restrict__ kept, int32_t* __restrict__ n_kept) {
  __shared__ int flag[1024];
  const int b = threadIdx.x;
  int mine = 0;
  if (b < B) {
    const float* z = logits + (size_t)b * C;
    float m = z[0];
    int am = 0;
    for (int c = 1; c < C; ++c)
      if (z[c] > m) { m = z[c]; am = c; }          // first maximum (torch.max tie rule)
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(z[c] - m);
    pred[b] = am;
    conf[b] = 100.f / den;
    mine = am == keep_class;
  }
  flag[b] = mine;
  __syncthreads();
  if (b == 0) {
    int n = 0;
    for (int i = 0; i < B; ++i)
      if (flag[i]) kept[n++] = i;
    n_kept[0] = n;
  }
}

extern "C" int mi355_cls_decide(const float* logits, int B, int C, int keep_class, int32_t* pred, float* conf, int32_t* kept,
                                int32_t* n_kept, mi355_stream_t s) {
  MI355_CHECK_ARG(logits && pred && conf && kept && n_kept && B > 0 && B <= 1024 && C > 0, "cls_decide: bad arguments (B=%d)", B);
  hipLaunchKernelGGL(cls_decide_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, logits, B, C, keep_class, pred, conf, kept, n_kept);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// y[i] = x[idx[i]] for rows of `row` fp32 elements (row % 4 == 0): the kept images, compacted
__global__ void gather_rows_kernel(const float4* __restrict__ x, const int32_t* __restrict__ idx, long long row4,
                                   float4* __restrict__ y) {
  const float4* src = x + (size_t)idx[blockIdx.y] * row4;
  float4* dst = y + (size_t)blockIdx.y * row4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < row4; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int mi355_gather_rows(const float* x, const int32_t* idx, int n, long long row, float* y, mi355_stream_t s) {
  MI355_CHECK_ARG(x && idx && y && n > 0 && row > 0 && row % 4 == 0, "gather_rows: bad arguments");
  long long bx = (row / 4 + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((int)bx, n), dim3(256), 0, (hipStream_t)s, (const float4*)x, idx, row / 4, (float4*)y);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// out[idx[i]][p] = sigmoid(logit[i][p]) > thr ? 255 : 0 for the n compacted samples (out: [B][per] uint8, zeroed by the caller)
__global__ void mask_scatter_kernel(const float* __restrict__ logit, const int32_t* __restrict__ idx, long long per, float thr,
                                    uint8_t* __restrict__ out) {
  const float* src = logit + (size_t)blockIdx.y * per;
  uint8_t* dst = out + (size_t)idx[blockIdx.y] * per;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x)
    dst[i] = 1.f / (1.f + expf(-src[i])) > thr ? 255 : 0;
}

extern "C" int mi355_mask_scatter(const float* logit, const int32_t* idx, int n, long long per, float thr, uint8_t* out,
                                  mi355_stream_t s) {
  MI355_CHECK_ARG(logit && idx && out && n > 0 && per > 0, "mask_scatter: bad arguments");
  long long bx = (per + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(mask_scatter_kernel, dim3((int)bx, n), dim3(256), 0, (hipStream_t)s, logit, idx, per, thr, out);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// ---- AdaptiveAvgPool2d((OH,OW)) + Flatten (torchvision VGG: 512 x 7 x 7 -> 25088) --------------------------------
// NHWC activations -> fp32 [N][C*OH*OW] in torch's NCHW flatten order; window of cell o: [floor(o*H/OH), ceil((o+1)*H/OH)).
template <typename T>
__global__ void adaptive_avgpool_fwd_kernel(const T* __restrict__ x, int ldx, float* __restrict__ y, int H, int W, int C, int OH,
                                            int OW, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);                       // channel fastest: coalesced reads
    long long r = i / C;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    const int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH;
    const int w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
    float acc = 0.f;
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) acc += to_f32<T>(x[((size_t)(n * H + h) * W + w) * ldx + c]);
    y[((size_t)n * C + c) * OH * OW + oh * OW + ow] = acc / (float)((h1 - h0) * (w1 - w0));
  }
}

template <typename T>
__global__ void adaptive_avgpool_bwd_kernel(const float* __restrict__ dy, T* __restrict__ dx, int lddx, int H, int W, int C, int OH,
                                            int OW, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float acc = 0.f;
    for (int oh = 0; oh < OH; ++oh) {
      const int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH;
      if (h < h0 || h >= h1) continue;
      for (int ow = 0; ow < OW; ++ow) {
        const int w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
        if (w < w0 || w >= w1) continue;
        acc += dy[((size_t)n * C + c) * OH * OW + oh * OW + ow] / (float)((h1 - h0) * (w1 - w0));
      }
    }
    dx[((size_t)(n * H + h) * W + w) * lddx + c] = from_f32<T>(acc);
  }
}

extern "C" int mi355_adaptive_avgpool_fwd(const void* x, int ldx, float* y, int N, int H, int W, int C, int OH, int OW, int dtype,
                                          mi355_stream_t s) {
  MI355_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "adaptive_avgpool_fwd: bad arguments");
  const long long total = (long long)N * OH * OW * C;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return dispatch_dtype(dtype, "adaptive_avgpool_fwd", [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((adaptive_avgpool_fwd_kernel<T>), dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, y, H, W, C,
                       OH, OW, total);
    MI355_LAUNCH_CHECK();
    return (int)MI355_OK;
  });
}

extern "C" int mi355_adaptive_avgpool_bwd(const float* dy, void* dx, int lddx, int N, int H, int W, int C, int OH, int OW, int dtype,
                                          mi355_stream_t s) {
  MI355_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "adaptive_avgpool_bwd: bad arguments");
  const long long total = (long long)N * H * W * C;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return dispatch_dtype(dtype, "adaptive_avgpool_bwd", [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((adaptive_avgpool_bwd_kernel<T>), dim3((int)blocks), dim3(256), 0, (hipStream_t)s, dy, (T*)dx, lddx, H, W, C, OH,
                       OW, total);
    MI355_LAUNCH_CHECK();
    return (int)MI355_OK;
  });
}
