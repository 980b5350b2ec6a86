// BatchNorm2d pieces for NHWC tensors (train-mode statistics, apply(+ReLU,+add), backward).
// All kernels are HBM-bound streaming passes built on rowred.hpp (16-B per lane, deterministic
// per-workgroup partials, fp32/fp64 statistics regardless of the storage dtype).
#include "rowred.hpp"

// ---- forward statistics --------------------------------------------------------------------------
template <typename T> struct BnStatsOp {
  static constexpr int NQ = 2;
  static constexpr bool WRITES = false;
  typedef double Acc;
  static constexpr int EPC = 16 / (int)sizeof(T);
  const T* x;
  int ld;
  __device__ void load_cols(int) {}
  __device__ void apply(size_t row, int c0, Acc (&acc)[NQ][EPC]) const {
    const Vec16<T> v = ld16<T>(x + row * ld + c0);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const double f = (double)to_f32<T>(v.v[e]);
      acc[0][e] += f;
      acc[1][e] += f * f;
    }
  }
};


// ---- folding per-workgroup partials -----------------------------------------------------------------
// Finalize kernels run 1024-thread workgroups of CH channels x BL row lanes: thread (bl, cl) sums partial
// rows b = bl, bl+BL, ... for channel c0+cl (fp64, four independent loads in flight), for NQ quantities at once
// (column offsets off[q]); the row lanes of a wave are then folded with shuffles and the 16 waves through ONE
// LDS exchange (one barrier instead of the 2 x log2(BL) of a shared-memory tree: these kernels are pure
// latency on the critical path, ~70 launches per step).  <32,32> for the row-reduce partials of narrow grids;
// <4,256> when there are more than 128 partial rows.  The sums are valid in threads tid < CH.
template <int CH, int BL, int NQ>
__device__ __forceinline__ void fold_partials(const float* __restrict__ partial, int nblocks, int rowlen, const int (&off)[NQ], int C,
                                              double* red /* [NQ][16][CH] */, double (&out)[NQ]) {
  static_assert(CH * BL == 1024 && CH <= 64 && (CH & (CH - 1)) == 0, "1024-thread workgroup, power-of-two channel count");
  const int cl = threadIdx.x % CH, bl = threadIdx.x / CH;
  const int c = blockIdx.x * CH + cl;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (c < C) {
      const float* p = partial + off[q] + c;
      int b = bl;
      for (; b + 3 * BL < nblocks; b += 4 * BL) {
        s0 += (double)p[(size_t)b * rowlen];
        s1 += (double)p[(size_t)(b + BL) * rowlen];
        s2 += (double)p[(size_t)(b + 2 * BL) * rowlen];
        s3 += (double)p[(size_t)(b + 3 * BL) * rowlen];
      }
      for (; b < nblocks; b += BL) s0 += (double)p[(size_t)b * rowlen];
    }
    s[q] = (s0 + s1) + (s2 + s3);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int m = CH; m < 64; m <<= 1) s[q] += __shfl_xor(s[q], m, 64);      // the 64 / CH row lanes of this wave
    if (lane < CH) red[(q * 16 + wave) * CH + lane] = s[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double t = 0;
    if (threadIdx.x < CH) {
#pragma unroll
      for (int w = 0; w < 16; ++w) t += red[(q * 16 + w) * CH + threadIdx.x];
    }
    out[q] = t;
  }
}

extern "C" int mi355_rowreduce_blocks(long long M) { return rowreduce_blocks(M); }

extern "C" int mi355_bn_stats(const void* x, float* partial, long long M, int C, int ld, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(x && partial && M > 0, "bn_stats: bad arguments");
  return dispatch_dtype(dtype, "bn_stats", [&](auto tag) {
    using T = decltype(tag);
    BnStatsOp<T> op{(const T*)x, ld};
    return rowred_launch<T>(op, M, C, partial, (hipStream_t)s);
  });
}

template <int CH, int BL>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int nblocks, double M, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, int64_t* nbt, float momentum,
                                   float eps, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  __shared__ double red[2 * 16 * CH];
  double sums2[2];
  fold_partials<CH, BL, 2>(partial, nblocks, 2 * C, {0, C}, C, red, sums2);
  const double s = sums2[0], q = sums2[1];
  const int c = blockIdx.x * CH + threadIdx.x;
  if (threadIdx.x >= CH) return;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  const double mean = s / M;
  double var = q / M - mean * mean;
  if (var < 0) var = 0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) {
    const double unb = M > 1 ? var * M / (M - 1) : var;
    rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
    rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
  }
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
}

extern "C" int mi355_bn_finalize(const float* partial, int nblocks, long long M, int C, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                                 float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                                 mi355_stream_t s) {
  MI355_CHECK_ARG(partial && gamma && beta && scale && shift && mean && invstd, "bn_finalize: null pointer");
  // many partial rows: 4 channels x 256 row lanes per workgroup (C/4 workgroups, at most 4 dependent loads per thread for the
  // 1024 rows of a row reduction); few rows: 32 x 32
  if (nblocks > 128)
    hipLaunchKernelGGL((bn_finalize_kernel<4, 256>), dim3(ceil_div(C, 4)), dim3(1024), 0, (hipStream_t)s, partial, nblocks, (double)M, C,
                       gamma, beta, running_mean, running_var, nbt, momentum, eps, scale, shift, mean, invstd);
  else
    hipLaunchKernelGGL((bn_finalize_kernel<32, 32>), dim3(ceil_div(C, 32)), dim3(1024), 0, (hipStream_t)s, partial, nblocks, (double)M, C,
                       gamma, beta, running_mean, running_var, nbt, momentum, eps, scale, shift, mean, invstd);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// One wave row = 64 consecutive floats of a partial row (256 B); four row lanes per workgroup, four loads in flight per lane.
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* __restrict__ in, int rows, int rowlen, float* __restrict__ out,
                                                        int nsplit) {
  __shared__ double red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int per = (rows + nsplit - 1) / nsplit;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (c < rowlen) {
    int r = r0 + rl;
    for (; r + 12 < r1; r += 16) {
      s0 += (double)in[(size_t)r * rowlen + c];
      s1 += (double)in[(size_t)(r + 4) * rowlen + c];
      s2 += (double)in[(size_t)(r + 8) * rowlen + c];
      s3 += (double)in[(size_t)(r + 12) * rowlen + c];
    }
    for (; r < r1; r += 4) s0 += (double)in[(size_t)r * rowlen + c];
  }
  red[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && c < rowlen) out[(size_t)blockIdx.y * rowlen + c] = (float)((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}

extern "C" int mi355_fold_rows(const float* partial, int rows, int rowlen, float* out, int nsplit, mi355_stream_t s) {
  MI355_CHECK_ARG(partial && out && rows > 0 && rowlen > 0 && nsplit > 0 && nsplit <= rows, "fold_rows: bad arguments");
  hipLaunchKernelGGL(fold_rows_kernel, dim3(ceil_div(rowlen, 64), nsplit), dim3(256), 0, (hipStream_t)s, partial, rows, rowlen, out,
                     nsplit);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// eval-mode folding: conv(x; w*scale) + (scale*bias + shift) == bn(conv(x; w) + bias)
__global__ void bn_fold_bias_kernel(const float* bias, const float* scale, const float* shift, float* out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) out[c] = scale[c] * (bias ? bias[c] : 0.f) + shift[c];
}

extern "C" int mi355_bn_fold_bias(const float* bias, const float* scale, const float* shift, float* out, int C, mi355_stream_t s) {
  MI355_CHECK_ARG(scale && shift && out && C > 0, "bn_fold_bias: bad arguments");
  hipLaunchKernelGGL(bn_fold_bias_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, (hipStream_t)s, bias, scale, shift, out, C);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

extern "C" int mi355_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, int C, float* scale, float* shift,
                                    mi355_stream_t s) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, (hipStream_t)s, gamma, beta, running_mean,
                     running_var, eps, C, scale, shift);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// ---- apply (+ second normalised operand, + residual, + ReLU) ------------------------------------------
// H2 / HR (a second normalised operand, a residual operand) are compile-time: a load under a run-time condition is waited for on the
// spot, which takes the rest of a fetch batch out of flight (common.hpp, ld16_pol).  The raw convolution output is read streaming
// (not read again before the backward pass; the default policy measured +0.09 ms per step).
template <typename T, bool H2 = false, bool HR = false> struct BnActOp {
  static constexpr int EPC = 16 / (int)sizeof(T);
  const T* x; int ldx; const float* scale; const float* shift;
  const T* x2; int ldx2; const float* scale2; const float* shift2;
  const T* res; int ldr;
  T* y; int ldy;
  int act;
  float sc[EPC], sh[EPC], sc2[EPC];
  __device__ void load_cols(int c0) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      sc[e] = scale ? scale[c0 + e] : 1.f;
      sh[e] = scale ? shift[c0 + e] : 0.f;
      sc2[e] = H2 ? scale2[c0 + e] : 0.f;
      if constexpr (H2) sh[e] += shift2[c0 + e];
    }
  }
#ifndef BN_ACT_FETCH
#define BN_ACT_FETCH 8
#endif
  static constexpr int FETCH_ROWS = BN_ACT_FETCH;
  struct In { Vec16<T> v, v2, vr; };
  __device__ In fetch(size_t row, int c0) const {
    In in;
    in.v = ld16_nt<T>(x + row * ldx + c0);
    if constexpr (H2) in.v2 = ld16<T>(x2 + row * ldx2 + c0);
    if constexpr (HR) in.vr = ld16<T>(res + row * ldr + c0);
    return in;
  }
  __device__ void pin(In& in) const {
    pin16(in.v);
    if constexpr (H2) pin16(in.v2);
    if constexpr (HR) pin16(in.vr);
  }
  __device__ void finish(const In& in, size_t row, int c0) const {
    float f[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) f[e] = __builtin_fmaf(to_f32<T>(in.v.v[e]), sc[e], sh[e]);      // (fused: the pool-aware backward recomputes it bit for bit)
    if constexpr (H2) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) f[e] = __builtin_fmaf(to_f32<T>(in.v2.v[e]), sc2[e], f[e]);      // (mi355_gate_bn_bwd_* recompute it)
    }
    // act bit0: ReLU; bit1: the residual is added AFTER the activation (recurrent block x + relu(bn(.)),
    // R2AttU_Net.py:44) instead of before it (ResNet.py:43)
    const bool relu = act & 1, post = act & 2;
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float v = f[e];
      if (HR && !post) v += to_f32<T>(in.vr.v[e]);
      if (relu) v = fmaxf(v, 0.f);
      if (HR && post) v += to_f32<T>(in.vr.v[e]);
      o.v[e] = from_f32<T>(v);
    }
    st16<T>(y + row * ldy + c0, o);
  }
};

extern "C" int mi355_bn_act(const void* x, int ldx, const float* scale, const float* shift, const void* x2, int ldx2,
                            const float* scale2, const float* shift2, const void* res, int ldr, void* y, int ldy,
                            long long M, int C, int act, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(x && y && M > 0, "bn_act: bad arguments");
  MI355_CHECK_ARG(!x2 || (scale2 && shift2), "bn_act: second operand needs scale2/shift2");
  return dispatch_dtype(dtype, "bn_act", [&](auto tag) {
    using T = decltype(tag);
    auto run = [&](auto op) {
      op.x = (const T*)x; op.ldx = ldx; op.scale = scale; op.shift = shift; op.x2 = (const T*)x2; op.ldx2 = ldx2;
      op.scale2 = scale2; op.shift2 = shift2; op.res = (const T*)res; op.ldr = ldr; op.y = (T*)y; op.ldy = ldy; op.act = act;
      return rowmap_launch<T>(op, M, C, (hipStream_t)s);
    };
    if (x2) return res ? run(BnActOp<T, true, true>{}) : run(BnActOp<T, true, false>{});
    return res ? run(BnActOp<T, false, true>{}) : run(BnActOp<T, false, false>{});
  });
}

#ifndef BN_ACT_POOL_WGS
#define BN_ACT_POOL_WGS 16     // workgroups per CU of the window-ordered forward apply passes (4 / 8 / 16: 5.72 / 5.81 / 5.91 TB/s, profiles/r04n_actwg.txt)
#endif
// ---- apply + ReLU with the following MaxPool2d(2, 2) in the same pass (AttentionUNet.py:61,89-95: every encoder level) --------
// A thread owns one 16-byte channel chunk of one 2 x 2 pixel group: four reads of the raw convolution output, four writes of the
// activation (the skip connection / gate / next convolution read it), one write of the pooled tensor — the separate pooling pass
// re-read the whole activation.  Values are pooled AFTER rounding to the storage type, i.e. exactly what mi355_maxpool_fwd reads.
template <typename T, bool HR>
__global__ __launch_bounds__(256) void bn_act_pool2_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, T* __restrict__ y, int ldy,
                                                           T* __restrict__ p, int ldp, int N, int H, int W, int C, int act,
                                                           const T* __restrict__ res = nullptr, int ldr = 0) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int cp = C / EPC, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)N * Ho * Wo * cp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cp) * EPC;
    long long q = i / cp;
    const int wo = (int)(q % Wo); q /= Wo;
    const int ho = (int)(q % Ho);
    const int n = (int)(q / Ho);
    float sc[EPC], sh[EPC], m[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; }
    const size_t r0 = ((size_t)(n * H + 2 * ho) * W + 2 * wo);
    Vec16<T> in[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      in[k] = ld16_nt<T>(x + (r0 + (size_t)(k >> 1) * W + (k & 1)) * ldx + c0);      // (streaming, as mi355_bn_act)
    Vec16<T> rs[4];
    if constexpr (HR) {       // (mi355_bn_act's residual operand: added before the activation, or after it when act bit 1 is set)
#pragma unroll
      for (int k = 0; k < 4; ++k) rs[k] = ld16<T>(res + (r0 + (size_t)(k >> 1) * W + (k & 1)) * ldr + c0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float v = __builtin_fmaf(to_f32<T>(in[k].v[e]), sc[e], sh[e]);
        if (HR && !(act & 2)) v += to_f32<T>(rs[k].v[e]);
        if (act & 1) v = fmaxf(v, 0.f);
        if (HR && (act & 2)) v += to_f32<T>(rs[k].v[e]);
        o.v[e] = from_f32<T>(v);
        const float r = to_f32<T>(o.v[e]);
        m[e] = k == 0 ? r : fmaxf(m[e], r);
      }
      st16<T>(y + (r0 + (size_t)(k >> 1) * W + (k & 1)) * ldy + c0, o);
    }
    if (p) {
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < EPC; ++e) o.v[e] = from_f32<T>(m[e]);
      st16<T>(p + ((size_t)(n * Ho + ho) * Wo + wo) * ldp + c0, o);
    }
  }
}

extern "C" int mi355_bn_act_pool2(const void* x, int ldx, const float* scale, const float* shift, void* y, int ldy, void* p, int ldp,
                                  int N, int H, int W, int C, int act, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(x && scale && shift && y && N > 0, "bn_act_pool2: bad arguments");
  MI355_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "bn_act_pool2: %d x %d is not divisible into 2 x 2 groups", H, W);
  const int epc = dtype_is_2byte(dtype) ? 8 : 4;
  MI355_CHECK_ARG(C % epc == 0, "bn_act_pool2: C=%d must be a multiple of %d", C, epc);
  long long blocks = ((long long)N * (H / 2) * (W / 2) * (C / epc) + 255) / 256;
  if (blocks > 256 * BN_ACT_POOL_WGS) blocks = 256 * BN_ACT_POOL_WGS;
  return dispatch_dtype(dtype, "bn_act_pool2", [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((bn_act_pool2_kernel<T, false>), dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, scale, shift, (T*)y, ldy,
                       (T*)p, ldp, N, H, W, C, act);
    MI355_LAUNCH_CHECK();
    return (int)MI355_OK;
  });
}

// mi355_bn_act's plain and residual forms (no second normalised operand) in the window order of the kernel above, on even images
extern "C" int mi355_bn_act_windows(const void* x, int ldx, const float* scale, const float* shift, const void* res, int ldr, void* y,
                                    int ldy, int N, int H, int W, int C, int act, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(x && scale && shift && y && N > 0, "bn_act_windows: bad arguments");
  MI355_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "bn_act_windows: %d x %d is not divisible into 2 x 2 groups", H, W);
  const int epc = dtype_is_2byte(dtype) ? 8 : 4;
  MI355_CHECK_ARG(C % epc == 0, "bn_act_windows: C=%d must be a multiple of %d", C, epc);
  long long blocks = ((long long)N * (H / 2) * (W / 2) * (C / epc) + 255) / 256;
  if (blocks > 256 * BN_ACT_POOL_WGS) blocks = 256 * BN_ACT_POOL_WGS;
  return dispatch_dtype(dtype, "bn_act_windows", [&](auto tag) {
    using T = decltype(tag);
    if (res)
      hipLaunchKernelGGL((bn_act_pool2_kernel<T, true>), dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, scale, shift,
                         (T*)y, ldy, (T*)nullptr, 0, N, H, W, C, act, (const T*)res, ldr);
    else
      hipLaunchKernelGGL((bn_act_pool2_kernel<T, false>), dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, scale, shift,
                         (T*)y, ldy, (T*)nullptr, 0, N, H, W, C, act, (const T*)nullptr, 0);
    MI355_LAUNCH_CHECK();
    return (int)MI355_OK;
  });
}

// ---- backward ------------------------------------------------------------------------------------
// Cache policy of the REDUCE passes' operand reads: the apply pass re-reads the same tensors right behind them, and with the
// default policy finds part of them in the memory-side cache (end of round 3: -0.13 ms per step against streaming reads — round 2
// had measured the opposite on a plan with more passes between the two; the apply passes' own reads stay streaming).
// MI355_BN_REDUCE_NT=1 restores the streaming reads (A/B).
// MI355_BN_KEEP_MB: operand pairs larger than this (MB) cannot be found again in the 256 MB memory-side cache by the pass behind
// them — they are read with streaming loads whatever the switches say (0 = no limit).
static inline long long bn_keep_limit() {
  static const long long mb = getenv("MI355_BN_KEEP_MB") ? atoll(getenv("MI355_BN_KEEP_MB")) : 0;
  return mb > 0 ? mb * 1000000ll : 0;
}
static inline int bn_reduce_keeps(long long bytes = 0) {
  static const int nt = getenv("MI355_BN_REDUCE_NT") ? atoi(getenv("MI355_BN_REDUCE_NT")) : 0;
  if (bn_keep_limit() && bytes > bn_keep_limit()) return 0;
  return nt ? 0 : 1;
}

// ... and of the APPLY passes' reads: the default policy as well since the end of round 3 (-0.13 ms per step; round 2 had measured
// streaming reads ahead by 0.02).  MI355_BN_APPLY_NT=1 restores them (A/B).  The forward apply pass keeps its streaming read of the
// raw convolution output (default policy there: +0.09 ms).
static inline int bn_apply_keeps(long long bytes = 0) {
  static const int nt = getenv("MI355_BN_APPLY_NT") ? atoi(getenv("MI355_BN_APPLY_NT")) : 0;
  static const int lim = getenv("MI355_BN_KEEP_APPLY") ? atoi(getenv("MI355_BN_KEEP_APPLY")) : 0;      // (the size rule for the apply pass too)
  if (lim && bn_keep_limit() && bytes > bn_keep_limit()) return 0;
  return nt ? 0 : 1;
}

// The apply pass re-reads what the reduce pass just read.  Walking the rows from the END meets the reduce pass's most recent lines
// first: when the pair of tensors is larger than the 256 MB memory-side cache a second forward sweep finds nothing (each line was
// evicted before its turn comes again), a backward sweep finds whatever the cache still holds.  MI355_BN_APPLY_REV (A/B).
static inline int bn_apply_reversed() {
  static const int rev = getenv("MI355_BN_APPLY_REV") ? atoi(getenv("MI355_BN_APPLY_REV")) : 0;
  return rev;
}

// HASY (the activated tensor is read for the mask) and KEEP (cache policy of the operand reads) are compile-time: see ld16_pol.
template <typename T, bool HASY = false, bool KEEP = true> struct BnBwdReduceOp {
  static constexpr int NQ = 2;
  static constexpr bool WRITES = false;
  typedef float Acc;
  static constexpr int EPC = 16 / (int)sizeof(T);
  const T* dy; int lddy; const T* y; int ldy; const T* x; int ldx;
  const float* mean; const float* invstd; const float* mscale; const float* mshift; int act;
  float mu[EPC], is[EPC], ms[EPC], mt[EPC];
  __device__ void load_cols(int c0) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      mu[e] = mean[c0 + e];
      is[e] = invstd[c0 + e];
      ms[e] = (act && !HASY) ? mscale[c0 + e] : 0.f;
      mt[e] = (act && !HASY) ? mshift[c0 + e] : 0.f;
    }
  }
#ifndef BN_RED_FETCH
#define BN_RED_FETCH 4
#endif
  static constexpr int FETCH_ROWS = BN_RED_FETCH;
#ifdef BN_RED_WGS
  static constexpr int MAX_WGS = BN_RED_WGS;      // (A/B: workgroups of the read-only pass)
#endif
  struct In { Vec16<T> g, xv, yv; };
  __device__ In fetch(size_t row, int c0) const {
    In in;
    // KEEP: the lines stay in the memory-side cache for the apply pass that re-reads them next (bn_reduce_keeps); else streaming
    // loads (round 2: with the three BatchNorm passes reading their operands that way the step was 0.27 ms shorter; round 3, on a
    // plan with fewer passes between the two, the default policy won by 0.13 ms) — A/B of all combinations in DESIGN.md section 4
    in.g = ld16_pol<KEEP, T>(dy + row * lddy + c0);
    in.xv = ld16_pol<KEEP, T>(x + row * ldx + c0);
    if constexpr (HASY) in.yv = ld16<T>(y + row * ldy + c0);
    return in;
  }
  __device__ void pin(In& in) const {
    pin16(in.g);
    pin16(in.xv);
    if constexpr (HASY) pin16(in.yv);
  }
  __device__ void finish(const In& in, size_t, int, Acc (&acc)[NQ][EPC]) const {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float gg = to_f32<T>(in.g.v[e]);
      const float xf = to_f32<T>(in.xv.v[e]);
      if (act) {
        const bool on = HASY ? (to_f32<T>(in.yv.v[e]) > 0.f) : (__builtin_fmaf(xf, ms[e], mt[e]) > 0.f);
        if (!on) gg = 0.f;
      }
      acc[0][e] += gg;
      acc[1][e] += gg * (xf - mu[e]) * is[e];
    }
  }
};

extern "C" int mi355_bn_bwd_reduce_rows(long long M) { return rowred_grid<BnBwdReduceOp<bf16_t>>(M); }

extern "C" int mi355_bn_bwd_reduce(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                                   const float* mean, const float* invstd, const float* mscale, const float* mshift,
                                   float* partial, long long M, int C, int act, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(dy && x && mean && invstd && partial && (!act || y || (mscale && mshift)), "bn_bwd_reduce: null pointer");
  return dispatch_dtype(dtype, "bn_bwd_reduce", [&](auto tag) {
    using T = decltype(tag);
    const bool keep = bn_reduce_keeps(2ll * M * C * (long long)sizeof(T)), hasy = act && y;
    auto run = [&](auto op) {
      op.dy = (const T*)dy; op.lddy = lddy; op.y = (const T*)y; op.ldy = ldy; op.x = (const T*)x; op.ldx = ldx;
      op.mean = mean; op.invstd = invstd; op.mscale = mscale; op.mshift = mshift; op.act = act;
      return rowred_launch<T>(op, M, C, partial, (hipStream_t)s);
    };
    if (hasy) return keep ? run(BnBwdReduceOp<T, true, true>{}) : run(BnBwdReduceOp<T, true, false>{});
    return keep ? run(BnBwdReduceOp<T, false, true>{}) : run(BnBwdReduceOp<T, false, false>{});
  });
}

template <int CH, int BL>
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                               float* __restrict__ sums, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float accf, int nq, int q0, int q1) {
  __shared__ double red[2 * 16 * CH];
  double sums2[2];
  fold_partials<CH, BL, 2>(partial, nblocks, nq * C, {q0 * C, q1 * C}, C, red, sums2);
  const double s0 = sums2[0], s1 = sums2[1];
  const int c = blockIdx.x * CH + threadIdx.x;
  if (threadIdx.x >= CH || c >= C) return;
  sums[c] = (float)s0;
  sums[C + c] = (float)s1;
  if (dbeta) dbeta[c] = (accf != 0.f ? accf * dbeta[c] : 0.f) + (float)s0;
  if (dgamma) dgamma[c] = (accf != 0.f ? accf * dgamma[c] : 0.f) + (float)s1;
}

extern "C" int mi355_bn_bwd_finalize_at(const float* partial, int nblocks, int nq, int q0, int q1, int C, float* sums, float* dgamma,
                                        float* dbeta, float acc, mi355_stream_t s) {
  MI355_CHECK_ARG(partial && sums, "bn_bwd_finalize: null pointer");
  MI355_CHECK_ARG(nq >= 2 && q0 >= 0 && q0 < nq && q1 >= 0 && q1 < nq, "bn_bwd_finalize: quantities %d, %d of %d per partial row", q0, q1, nq);
  if (nblocks > 128)      // (see mi355_bn_finalize)
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<4, 256>), dim3(ceil_div(C, 4)), dim3(1024), 0, (hipStream_t)s, partial, nblocks, C,
                       sums, dgamma, dbeta, acc, nq, q0, q1);
  else
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<32, 32>), dim3(ceil_div(C, 32)), dim3(1024), 0, (hipStream_t)s, partial, nblocks, C,
                       sums, dgamma, dbeta, acc, nq, q0, q1);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

extern "C" int mi355_bn_bwd_finalize(const float* partial, int nblocks, int C, float* sums, float* dgamma, float* dbeta,
                                     float acc, mi355_stream_t s) {
  return mi355_bn_bwd_finalize_at(partial, nblocks, 2, 0, 1, C, sums, dgamma, dbeta, acc, s);
}

// NEX = 4: the post-activation operand's gradient also takes up to four EARLIER incoming gradients (the recurrent block's
// x + relu(bn(.)) applied t times to the same x, R2AttU_Net.py:41-44): d x = sum over the applications of their incoming gradients,
// added here in ONE pass by the last of them — one fp32 sum, one rounding — instead of a read-modify-write of d x in every one.
// HASY / PACC (the activated tensor is read for the mask; dpost accumulates) are compile-time: a load under a run-time condition is
// branched around AND waited for on the spot (vmcnt(0)), which would take the other rows' loads of the batch out of flight.
template <typename T, int NEX = 0, bool HASY = false, bool PACC = false, bool KEEP = true> struct BnBwdApplyOp {
  static constexpr int NQ = 1;
  static constexpr bool WRITES = true;
  typedef float Acc;
  static constexpr int EPC = 16 / (int)sizeof(T);
  static constexpr int NEXA = NEX > 0 ? NEX : 1;
  const T* dy; int lddy; const T* y; int ldy; const T* x; int ldx;
  const float* gamma; const float* mean; const float* invstd; const float* mscale; const float* mshift; const float* sums;
  T* dx; int lddx; T* dres; int lddres; T* dpost; int lddpost; int post_acc;
  float invM; int C; int act;
  long long last;      // >= 0: the pass walks the rows from the END (row r of the sweep is tensor row last - r), see bn_apply_reversed
  const T* ex[NEXA]; int ldex, nex;      // (NEX > 0) earlier incoming gradients: the first nex are added; all of row pitch ldex
  float mu[EPC], is[EPC], k0[EPC], k1[EPC], gi[EPC], ms[EPC], mt[EPC];
  __device__ void load_cols(int c0) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      ms[e] = (act && !y) ? mscale[c0 + e] : 0.f;
      mt[e] = (act && !y) ? mshift[c0 + e] : 0.f;
      mu[e] = mean[c0 + e];
      is[e] = invstd[c0 + e];
      gi[e] = gamma[c0 + e] * is[e];
      k0[e] = sums[c0 + e] * invM;
      k1[e] = sums[C + c0 + e] * invM;
    }
  }
#ifndef BN_APPLY_FETCH
#define BN_APPLY_FETCH 4
#endif
  static constexpr int FETCH_ROWS = NEX > 0 ? 2 : BN_APPLY_FETCH;
  struct In { Vec16<T> g, xv, yv, pv, ev[NEXA]; };
  __device__ In fetch(size_t row, int c0) const {
    In in;
    if (last >= 0) row = (size_t)last - row;
    in.g = ld16_pol<KEEP, T>(dy + row * lddy + c0);
    in.xv = ld16_pol<KEEP, T>(x + row * ldx + c0);
    if constexpr (HASY) in.yv = ld16<T>(y + row * ldy + c0);
    if constexpr (PACC) in.pv = ld16<T>(dpost + row * lddpost + c0);
    if constexpr (NEX > 0) {      // (absent ones point at ex[0]: loaded, not added)
#pragma unroll
      for (int j = 0; j < NEX; ++j) in.ev[j] = ld16_nt<T>(ex[j] + row * ldex + c0);      // (their last read)
    }
    return in;
  }
  __device__ void pin(In& in) const {
    pin16(in.g);
    pin16(in.xv);
    if constexpr (HASY) pin16(in.yv);
    if constexpr (PACC) pin16(in.pv);
    if constexpr (NEX > 0) {
#pragma unroll
      for (int j = 0; j < NEX; ++j) pin16(in.ev[j]);
    }
  }
  __device__ void finish(const In& in, size_t row, int c0, Acc (&acc)[NQ][EPC]) const {
    if (last >= 0) row = (size_t)last - row;
    if (dpost) {      // gradient of an operand added after the activation: the incoming gradient itself
      Vec16<T> pg = in.g;
      if constexpr (PACC || NEX > 0) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float v = to_f32<T>(in.g.v[e]);
          if constexpr (PACC) v += to_f32<T>(in.pv.v[e]);
          if constexpr (NEX > 0) {
#pragma unroll
            for (int j = 0; j < NEX; ++j)
              if (j < nex) v += to_f32<T>(in.ev[j].v[e]);
          }
          pg.v[e] = from_f32<T>(v);
        }
      }
      st16<T>(dpost + row * lddpost + c0, pg);
    }
    Vec16<T> o, r;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float gg = to_f32<T>(in.g.v[e]);
      const float xf = to_f32<T>(in.xv.v[e]);
      if (act) {
        const bool on = HASY ? (to_f32<T>(in.yv.v[e]) > 0.f) : (__builtin_fmaf(xf, ms[e], mt[e]) > 0.f);
        if (!on) gg = 0.f;
      }
      const float xh = (xf - mu[e]) * is[e];
      const float d = bn_dx(gi[e], gg, k0[e], xh, k1[e]);
      o.v[e] = from_f32<T>(d);
      r.v[e] = from_f32<T>(gg);
      acc[0][e] += d;
    }
    st16<T>(dx + row * lddx + c0, o);
    if (dres) st16<T>(dres + row * lddres + c0, r);
  }
};

template <typename T, int NEX, bool HASY, bool PACC, bool KEEP>
static int bn_bwd_apply_launch(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                               const float* gamma, const float* mean, const float* invstd, const float* mscale,
                               const float* mshift, const float* sums, void* dx, int lddx, void* dres, int lddres,
                               void* dpost, int lddpost, const void* const* ex, int ldex,
                               float* dbias_partial, long long M, int C, int act, mi355_stream_t s) {
  const float invM = (float)(1.0 / (double)M);
  const long long last = bn_apply_reversed() ? M - 1 : -1ll;
  BnBwdApplyOp<T, NEX, HASY, PACC, KEEP> op{(const T*)dy, lddy, (const T*)y, ldy, (const T*)x, ldx, gamma, mean, invstd,
                                            mscale, mshift, sums, (T*)dx, lddx, (T*)dres, lddres, (T*)dpost, lddpost, PACC ? 1 : 0, invM, C,
                                            act, last, {nullptr}, ldex, 0};
  if constexpr (NEX > 0) {
    for (int j = 0; j < NEX; ++j) {
      op.ex[j] = (const T*)(ex[j] ? ex[j] : ex[0]);
      if (ex[j]) op.nex = j + 1;
    }
  }
  return rowred_launch<T>(op, M, C, dbias_partial, (hipStream_t)s);
}

static int bn_bwd_apply_impl(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                             const float* gamma, const float* mean, const float* invstd, const float* mscale,
                             const float* mshift, const float* sums, void* dx, int lddx, void* dres, int lddres,
                             void* dpost, int lddpost, int post_acc, const void* const* ex, int ldex,
                             float* dbias_partial, long long M, int C, int act, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(dy && x && gamma && mean && invstd && sums && dx && (!act || y || (mscale && mshift)), "bn_bwd_apply: null pointer");
  if (ex) {
    for (int j = 1; j < 4; ++j) MI355_CHECK_ARG(!ex[j] || ex[j - 1], "bn_bwd_apply_post4: the earlier gradients must be given without gaps");
  }
  return dispatch_dtype(dtype, "bn_bwd_apply", [&](auto tag) {
    using T = decltype(tag);
    const bool hasy = act && y, pacc = dpost && post_acc;
    const bool keep = bn_apply_keeps(2ll * M * C * (long long)sizeof(T));
#define MI355_APPLY(NEX, HASY, PACC)                                                                                                       \
  (keep ? bn_bwd_apply_launch<T, NEX, HASY, PACC, true>(dy, lddy, y, ldy, x, ldx, gamma, mean, invstd, mscale, mshift, sums, dx, lddx,     \
                                                        dres, lddres, dpost, lddpost, ex, ldex, dbias_partial, M, C, act, s)               \
        : bn_bwd_apply_launch<T, NEX, HASY, PACC, false>(dy, lddy, y, ldy, x, ldx, gamma, mean, invstd, mscale, mshift, sums, dx, lddx,    \
                                                         dres, lddres, dpost, lddpost, ex, ldex, dbias_partial, M, C, act, s))
    if (ex) {
      if (hasy) return pacc ? MI355_APPLY(4, true, true) : MI355_APPLY(4, true, false);
      return pacc ? MI355_APPLY(4, false, true) : MI355_APPLY(4, false, false);
    }
    if (hasy) return pacc ? MI355_APPLY(0, true, true) : MI355_APPLY(0, true, false);
    return pacc ? MI355_APPLY(0, false, true) : MI355_APPLY(0, false, false);
#undef MI355_APPLY
  });
}

extern "C" int mi355_bn_bwd_apply(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                                  const float* gamma, const float* mean, const float* invstd, const float* mscale,
                                  const float* mshift, const float* sums, void* dx, int lddx, void* dres, int lddres,
                                  void* dpost, int lddpost, int post_acc,
                                  float* dbias_partial, long long M, int C, int act, int dtype, mi355_stream_t s) {
  return bn_bwd_apply_impl(dy, lddy, y, ldy, x, ldx, gamma, mean, invstd, mscale, mshift, sums, dx, lddx, dres, lddres, dpost, lddpost,
                           post_acc, nullptr, 0, dbias_partial, M, C, act, dtype, s);
}

extern "C" int mi355_bn_bwd_apply_post4(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                                        const float* gamma, const float* mean, const float* invstd, const float* mscale,
                                        const float* mshift, const float* sums, void* dx, int lddx, void* dpost, int lddpost,
                                        int post_acc, const void* ex0, const void* ex1, const void* ex2, const void* ex3, int ldex,
                                        long long M, int C, int act, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(dpost && ex0, "bn_bwd_apply_post4: the post-activation operand's gradient and at least one earlier gradient expected");
  const void* ex[4] = {ex0, ex1, ex2, ex3};
  return bn_bwd_apply_impl(dy, lddy, y, ldy, x, ldx, gamma, mean, invstd, mscale, mshift, sums, dx, lddx, nullptr, 0, dpost, lddpost,
                           post_acc, ex, ldex, nullptr, M, C, act, dtype, s);
}

// ---- plain column sums (bias gradients) -------------------------------------------------------------
// ---- BatchNorm(+ReLU) backward of a layer whose activation also feeds a fused MaxPool2d(2, 2) (mi355_bn_act_pool2) -------------
// The pooled tensor's gradient dp is NOT scattered into the activation's gradient by a pass of its own (mi355_maxpool_bwd: read
// da, a and dp, write da): both BatchNorm passes add it on the fly, g = da + [pixel is the FIRST maximum of its window] * dp, with
// the activation recomputed from the raw convolution output exactly as the forward rounded it.  A thread owns one 2 x 2 window per
// batch (the mapping of bn_act_pool2_kernel: a wave instruction covers 64 / tpr pixels two apart, the four together every byte of
// two image-row segments), a workgroup pass rp windows = two image rows x 2 * rp pixels.  Geometry the host checks: tpr = C / EPC
// a power of two, W a power-of-two multiple of 2 * rp.
template <typename T, bool HASDY, bool KEEP> struct BnBwdPool2 {
  static constexpr int EPC = 16 / (int)sizeof(T);
  const T* dy; int lddy; const T* dp; int lddp; const T* x; int ldx;
  const float* mean; const float* invstd; const float* mscale; const float* mshift;
  int W, lrp, lwb;               // rp = 1 << lrp windows of a workgroup pass, W = (2 * rp) << lwb
  float mu[EPC], is[EPC], ms[EPC], mt[EPC];      // (KEEP: cache policy of the operand reads, compile-time: see ld16_pol)
  __device__ void load_common(int c0) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) { mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e]; ms[e] = mscale[c0 + e]; mt[e] = mshift[c0 + e]; }
  }
  struct Px { size_t pix[4]; Vec16<T> gv[4], xv[4], pv; };      // rows in window scan order: (h, w), (h, w + 1), (h + 1, w), (h + 1, w + 1)
  __device__ void fetch(long long r, int ty, int c0, Px& p) const {
    const long long q = r >> (lrp + 2);                    // batch index
    const long long R = q >> lwb;                          // pair of image rows (pairs never straddle images: H is even)
    const int wcol = ((int)(q & ((1 << lwb) - 1)) << lrp) + ty;
#pragma unroll
    for (int b = 0; b < 4; ++b) p.pix[b] = (size_t)(2 * R + (b >> 1)) * W + 2 * wcol + (b & 1);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if constexpr (HASDY) p.gv[b] = ld16_pol<KEEP, T>(dy + p.pix[b] * lddy + c0);
      p.xv[b] = ld16_pol<KEEP, T>(x + p.pix[b] * ldx + c0);
    }
    p.pv = ld16_pol<KEEP, T>(dp + ((size_t)R * (W >> 1) + wcol) * lddp + c0);
  }
  __device__ void pin(Px& p) const {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if constexpr (HASDY) pin16(p.gv[b]);
      pin16(p.xv[b]);
    }
    pin16(p.pv);
  }
  // g[b]: the gradient reaching the activation of row b, ReLU-masked
  __device__ void grads(const Px& p, int e, float (&g)[4]) const {
    float a[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) a[b] = to_f32<T>(from_f32<T>(fmaxf(__builtin_fmaf(to_f32<T>(p.xv[b].v[e]), ms[e], mt[e]), 0.f)));
    // first maximum in scan order (torch's tie rule, mi355_maxpool_bwd): the earliest element equal to the window's maximum
    const float mx = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
    const bool w0 = a[0] == mx, w1 = !w0 && a[1] == mx, w2 = !(w0 || w1) && a[2] == mx, w3 = !(w0 || w1 || w2);
    const float pg = to_f32<T>(p.pv.v[e]);
    const bool win[4] = {w0, w1, w2, w3};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      // (HASDY == false: the pooling is the activation's ONLY consumer — VGG.py's feature stack —, its gradient is the routed dp alone)
      const float t = (HASDY ? to_f32<T>(p.gv[b].v[e]) : 0.f) + (win[b] ? pg : 0.f);
      g[b] = a[b] > 0.f ? t : 0.f;
    }
  }
};

template <typename T, bool HASDY, bool KEEP = true> struct BnBwdReducePool2Op : BnBwdPool2<T, HASDY, KEEP> {
  static constexpr int NQ = 2;
  static constexpr bool WRITES = false;
  static constexpr int BATCH_ROWS = 4;
  typedef float Acc;
  static constexpr int EPC = 16 / (int)sizeof(T);
  __device__ void load_cols(int c0) { this->load_common(c0); }
  __device__ void finish(const typename BnBwdPool2<T, HASDY, KEEP>::Px& p, int, int, Acc (&acc)[NQ][EPC]) const {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float g[4];
      this->grads(p, e, g);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        acc[0][e] += g[b];
        acc[1][e] += g[b] * (to_f32<T>(p.xv[b].v[e]) - this->mu[e]) * this->is[e];
      }
    }
  }
};

template <typename T, bool HASDY, bool KEEP = true> struct BnBwdApplyPool2Op : BnBwdPool2<T, HASDY, KEEP> {
  static constexpr int NQ = 1;
  static constexpr bool WRITES = true;
  static constexpr int BATCH_ROWS = 4;
  typedef float Acc;
  static constexpr int EPC = 16 / (int)sizeof(T);
  const float* gamma; const float* sums; T* dx; int lddx; float invM; int C;
  float k0[EPC], k1[EPC], gi[EPC];
  __device__ void load_cols(int c0) {
    this->load_common(c0);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      gi[e] = gamma[c0 + e] * this->is[e];
      k0[e] = sums[c0 + e] * invM;
      k1[e] = sums[C + c0 + e] * invM;
    }
  }
  __device__ void finish(const typename BnBwdPool2<T, HASDY, KEEP>::Px& p, int, int c0, Acc (&acc)[NQ][EPC]) const {
    Vec16<T> o[4];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float g[4];
      this->grads(p, e, g);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float xh = (to_f32<T>(p.xv[b].v[e]) - this->mu[e]) * this->is[e];
        const float d = bn_dx(gi[e], g[b], k0[e], xh, k1[e]);
        o[b].v[e] = from_f32<T>(d);
        acc[0][e] += d;
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) st16<T>(dx + p.pix[b] * lddx + c0, o[b]);
  }
};

// log2 of a power of two, -1 otherwise
static inline int exact_log2(long long v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1LL << l) < v) ++l;
  return l;
}

extern "C" int mi355_bn_bwd_reduce_pool2_rows(long long M) { return rowred_grid<BnBwdReducePool2Op<bf16_t, true>>(M); }

extern "C" int mi355_bn_bwd_pool2_ok(int H, int W, int C, int dtype) {
  const int epc = dtype_is_2byte(dtype) ? 8 : 4;
  if (C % epc || H % 2 || W % 2) return 0;
  const int tpr = C / epc;
  if (tpr > 256 || exact_log2(tpr) < 0) return 0;
  const int rp = 256 / tpr;
  return W % (2 * rp) == 0 && exact_log2(W / (2 * rp)) >= 0;
}

template <typename T, typename Op> static void fill_pool2(Op& op, const void* dy, int lddy, const void* dp, int lddp, const void* x, int ldx,
                                                          const float* mean, const float* invstd, const float* mscale,
                                                          const float* mshift, int W, int C) {
  const int tpr = C / (16 / (int)sizeof(T)), rp = 256 / tpr;
  op.dy = (const T*)dy; op.lddy = lddy; op.dp = (const T*)dp; op.lddp = lddp; op.x = (const T*)x; op.ldx = ldx;
  op.mean = mean; op.invstd = invstd; op.mscale = mscale; op.mshift = mshift;
  op.W = W; op.lrp = exact_log2(rp); op.lwb = exact_log2(W / (2 * rp));
}

extern "C" int mi355_bn_bwd_reduce_pool2(const void* dy, int lddy, const void* dp, int lddp, const void* x, int ldx, const float* mean,
                                         const float* invstd, const float* mscale, const float* mshift, float* partial,
                                         int N, int H, int W, int C, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(dp && x && mean && invstd && mscale && mshift && partial && N > 0, "bn_bwd_reduce_pool2: null pointer");
  MI355_CHECK_ARG(mi355_bn_bwd_pool2_ok(H, W, C, dtype), "bn_bwd_reduce_pool2: %d x %d x %d is not a geometry of the window-ordered pass", H, W, C);
  return dispatch_dtype(dtype, "bn_bwd_reduce_pool2", [&](auto tag) {
    using T = decltype(tag);
    auto run = [&](auto op) {
      fill_pool2<T>(op, dy, lddy, dp, lddp, x, ldx, mean, invstd, mscale, mshift, W, C);
      return rowred_launch<T>(op, (long long)N * H * W, C, partial, (hipStream_t)s);
    };
    const bool keep = bn_reduce_keeps(2ll * N * H * W * C * (long long)sizeof(T));
    if (dy) return keep ? run(BnBwdReducePool2Op<T, true, true>{}) : run(BnBwdReducePool2Op<T, true, false>{});
    return keep ? run(BnBwdReducePool2Op<T, false, true>{}) : run(BnBwdReducePool2Op<T, false, false>{});
  });
}

extern "C" int mi355_bn_bwd_apply_pool2(const void* dy, int lddy, const void* dp, int lddp, const void* x, int ldx, const float* gamma,
                                        const float* mean, const float* invstd, const float* mscale, const float* mshift,
                                        const float* sums, void* dx, int lddx, int N, int H, int W, int C, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(dp && x && gamma && mean && invstd && mscale && mshift && sums && dx && N > 0, "bn_bwd_apply_pool2: null pointer");
  MI355_CHECK_ARG(mi355_bn_bwd_pool2_ok(H, W, C, dtype), "bn_bwd_apply_pool2: %d x %d x %d is not a geometry of the window-ordered pass", H, W, C);
  const long long M = (long long)N * H * W;
  return dispatch_dtype(dtype, "bn_bwd_apply_pool2", [&](auto tag) {
    using T = decltype(tag);
    auto run = [&](auto op) {
      fill_pool2<T>(op, dy, lddy, dp, lddp, x, ldx, mean, invstd, mscale, mshift, W, C);
      op.gamma = gamma; op.sums = sums; op.dx = (T*)dx; op.lddx = lddx; op.invM = (float)(1.0 / (double)M); op.C = C;
      return rowred_launch<T>(op, M, C, nullptr, (hipStream_t)s);
    };
    const bool keep = bn_apply_keeps(2ll * M * C * (long long)sizeof(T));
    if (dy) return keep ? run(BnBwdApplyPool2Op<T, true, true>{}) : run(BnBwdApplyPool2Op<T, true, false>{});
    return keep ? run(BnBwdApplyPool2Op<T, false, true>{}) : run(BnBwdApplyPool2Op<T, false, false>{});
  });
}

template <typename T> struct ColSumOp {
  static constexpr int NQ = 1;
  static constexpr bool WRITES = false;
  typedef float Acc;
  static constexpr int EPC = 16 / (int)sizeof(T);
  const T* x; int ld;
  __device__ void load_cols(int) {}
  __device__ void apply(size_t row, int c0, Acc (&acc)[NQ][EPC]) const {
    const Vec16<T> v = ld16<T>(x + row * ld + c0);
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[0][e] += to_f32<T>(v.v[e]);
  }
};

extern "C" int mi355_colsum(const void* x, int ld, float* partial, long long M, int C, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(x && partial, "colsum: null pointer");
  return dispatch_dtype(dtype, "colsum", [&](auto tag) {
    using T = decltype(tag);
    ColSumOp<T> op{(const T*)x, ld};
    return rowred_launch<T>(op, M, C, partial, (hipStream_t)s);
  });
}

__global__ void colsum_finalize_kernel(const float* __restrict__ partial, int nblocks, int stride, int C,
                                       float* __restrict__ out, float accf) {
  __shared__ double red[16 * 32];
  double sums1[1];
  fold_partials<32, 32, 1>(partial, nblocks, stride * C, {0}, C, red, sums1);
  const double s0 = sums1[0];
  const int c = blockIdx.x * 32 + threadIdx.x;
  if (threadIdx.x >= 32 || c >= C) return;
  out[c] = (accf != 0.f ? accf * out[c] : 0.f) + (float)s0;
}

extern "C" int mi355_colsum_finalize(const float* partial, int nblocks, int stride, int C, float* out, float acc,
                                     mi355_stream_t s) {
  MI355_CHECK_ARG(partial && out, "colsum_finalize: null pointer");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(ceil_div(C, 32)), dim3(1024), 0, (hipStream_t)s, partial, nblocks, stride, C,
                     out, acc);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
