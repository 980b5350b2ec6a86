// Losses, gradient-norm clipping, fused AdamW on flat fp32 buffers, segmentation metric counters.
// (utils/helpers.py:244-246, 251, 332-336; utils/tester.py:92-193.)  Everything the optimiser needs
// (lr, step, clip coefficient, inf flag) lives in device memory so a captured hipGraph replays it.
#include "rowred.hpp"

__device__ __forceinline__ float block_sum_f(float v) {
  __shared__ float red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}

// ---- BCEWithLogits (mean) ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ z, const float* __restrict__ t,
                                                         float* __restrict__ loss, float* __restrict__ dz,
                                                         const float* __restrict__ gscale, long long n) {
  const float gs = (gscale ? gscale[0] : 1.f) / (float)n;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float x = z[i], y = t[i];
    acc += fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)));
    if (dz) dz[i] = (1.f / (1.f + __expf(-x)) - y) * gs;
  }
  acc = block_sum_f(acc);
  if (threadIdx.x == 0) atomicAdd(loss, acc / (float)n);
}

extern "C" int mi355_bce_logits(const float* z, const float* t, float* loss, float* dz, const float* gscale, long long n,
                                mi355_stream_t s) {
  MI355_CHECK_ARG(z && t && loss && n > 0, "bce_logits: bad arguments");
  hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), (hipStream_t)s);
  if (e != hipSuccess) MI355_FAIL((int)e, "bce_logits: memset failed: %s", hipGetErrorString(e));
  long long blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(bce_logits_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, z, t, loss, dz, gscale, n);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// ---- CrossEntropy with label smoothing (one workgroup; B*C is tiny) ----------------------------------------
__global__ __launch_bounds__(256) void ce_smooth_kernel(const float* __restrict__ z, const int64_t* __restrict__ y,
                                                        float* __restrict__ loss, float* __restrict__ dz,
                                                        const float* __restrict__ gscale, int B, int C, float sm) {
  const float gs = (gscale ? gscale[0] : 1.f) / (float)B;
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* row = z + (size_t)b * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(row[c] - mx);
    const float lse = mx + __logf(se);
    const int lab = (int)y[b];
    float sum_logp = 0.f;
    for (int c = 0; c < C; ++c) sum_logp += row[c] - lse;
    acc += -(1.f - sm) * (row[lab] - lse) - sm * sum_logp / (float)C;
    if (dz)
      for (int c = 0; c < C; ++c) {
        const float p = __expf(row[c] - lse);
        const float tgt = (1.f - sm) * (c == lab ? 1.f : 0.f) + sm / (float)C;
        dz[(size_t)b * C + c] = (p - tgt) * gs;
      }
  }
  acc = block_sum_f(acc);
  if (threadIdx.x == 0) loss[0] = acc / (float)B;
}

extern "C" int mi355_ce_smooth(const float* z, const int64_t* y, float* loss, float* dz, const float* gscale, int B, int C,
                               float smoothing, mi355_stream_t s) {
  MI355_CHECK_ARG(z && y && loss && B > 0 && C > 0, "ce_smooth: bad arguments");
  hipLaunchKernelGGL(ce_smooth_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, z, y, loss, dz, gscale, B, C, smoothing);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// ---- gradient norm + clip coefficient ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, float* __restrict__ partial, long long n) {
  double acc = 0;
  const long long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  // four 16-byte loads in flight per thread, four independent double chains (was one load and one dependent chain per trip; the
  // pass is a small part of the step either way — scripts/opt_time.py measures the CALL, which is bound by its host side)
  const long long st = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double a1 = 0, a2 = 0, a3 = 0;
  for (; i + 3 * st < n4; i += 4 * st) {
    const float4 v0 = g4[i], v1 = g4[i + st], v2 = g4[i + 2 * st], v3 = g4[i + 3 * st];
    acc += (double)v0.x * v0.x + (double)v0.y * v0.y + (double)v0.z * v0.z + (double)v0.w * v0.w;
    a1 += (double)v1.x * v1.x + (double)v1.y * v1.y + (double)v1.z * v1.z + (double)v1.w * v1.w;
    a2 += (double)v2.x * v2.x + (double)v2.y * v2.y + (double)v2.z * v2.z + (double)v2.w * v2.w;
    a3 += (double)v3.x * v3.x + (double)v3.y * v3.y + (double)v3.z * v3.z + (double)v3.w * v3.w;
  }
  for (; i < n4; i += st) {
    const float4 v = g4[i];
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  acc = (acc + a1) + (a2 + a3);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    acc += (double)v * v;
  }
  __shared__ double red[4];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

extern "C" int mi355_sumsq_partial(const float* g, float* partial, long long n, mi355_stream_t s) {
  MI355_CHECK_ARG(g && partial && ((uintptr_t)g % 16) == 0, "sumsq_partial: bad arguments");
  hipLaunchKernelGGL(sumsq_kernel, dim3(rowreduce_blocks(n)), dim3(256), 0, (hipStream_t)s, g, partial, n);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

__global__ void clip_coef_kernel(const float* __restrict__ partial, int nblocks, float max_norm, float inv_scale,
                                 const float* dev_scale, float* norm, float* coef, float* found_inf, int32_t* step) {
  double acc = 0;
  for (int b = threadIdx.x; b < nblocks; b += 64) acc += (double)partial[b];
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) {
    const float nrm = (float)sqrt(acc) * inv_scale * (dev_scale ? dev_scale[0] : 1.f);
    const bool bad = !isfinite(nrm);
    norm[0] = nrm;
    coef[0] = max_norm > 0.f ? fminf(1.f, max_norm / (nrm + 1e-6f)) : 1.f;
    if (found_inf) found_inf[0] = bad ? 1.f : 0.f;
    if (step && !bad) step[0] += 1;
  }
}

extern "C" int mi355_clip_coef(const float* partial, int nblocks, float max_norm, float inv_scale, const float* dev_scale,
                               float* norm, float* coef, float* found_inf, int32_t* step, mi355_stream_t s) {
  MI355_CHECK_ARG(partial && norm && coef, "clip_coef: null pointer");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, partial, nblocks, max_norm, inv_scale, dev_scale, norm,
                     coef, found_inf, step);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// ---- AdamW (torch.optim.AdamW semantics: decoupled decay, bias-corrected, eps outside sqrt(bc2)) ------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, const float* __restrict__ lr_p,
                                                    float beta1, float beta2, float eps, float wd,
                                                    const float* __restrict__ coef, float inv_scale,
                                                    const float* __restrict__ dev_scale,
                                                    const float* __restrict__ found_inf, const int32_t* __restrict__ step) {
  if (found_inf && found_inf[0] != 0.f) return;
  const float lr = lr_p[0];
  const float gs = (coef ? coef[0] : 1.f) * inv_scale * (dev_scale ? dev_scale[0] : 1.f);
  const float t = (float)step[0];
  const float bc1 = 1.f - powf(beta1, t);
  const float bc2s = sqrtf(1.f - powf(beta2, t));
  const float step_size = lr / bc1;
  const float decay = 1.f - lr * wd;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gg = g[i] * gs;
    float pp = p[i] * decay;
    const float mm = beta1 * m[i] + (1.f - beta1) * gg;
    const float vv = beta2 * v[i] + (1.f - beta2) * gg * gg;
    pp -= step_size * mm / (sqrtf(vv) / bc2s + eps);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
  }
}

extern "C" int mi355_adamw(float* p, const float* g, float* m, float* v, long long n, const float* lr, float beta1,
                           float beta2, float eps, float wd, const float* coef, float inv_scale, const float* dev_scale,
                           const float* found_inf, const int32_t* step, mi355_stream_t s) {
  MI355_CHECK_ARG(p && g && m && v && lr && step && n > 0, "adamw: bad arguments");
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, lr, beta1, beta2, eps, wd, coef,
                     inv_scale, dev_scale, found_inf, step);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// ---- loss scaling (torch.amp.GradScaler semantics, all state on the device) -------------------------------------
__global__ void step_tick_kernel(int32_t* step, const float* found_inf) {
  if (!found_inf || found_inf[0] == 0.f) step[0] += 1;
}

extern "C" int mi355_step_tick(int32_t* step, const float* found_inf, mi355_stream_t s) {
  MI355_CHECK_ARG(step, "step_tick: null pointer");
  hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, step, found_inf);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

__global__ void amp_update_kernel(float* scale, float* inv_scale, int32_t* growth_tracker, const float* found_inf, float growth,
                                  float backoff, int interval) {
  if (found_inf[0] != 0.f) {
    scale[0] *= backoff;
    growth_tracker[0] = 0;
  } else if (++growth_tracker[0] == interval) {
    const float grown = scale[0] * growth;
    if (isfinite(grown)) scale[0] = grown;     // (torch: never grow into infinity)
    growth_tracker[0] = 0;
  }
  inv_scale[0] = 1.f / scale[0];
}

extern "C" int mi355_amp_update(float* scale, float* inv_scale, int32_t* growth_tracker, const float* found_inf, float growth,
                                float backoff, int interval, mi355_stream_t s) {
  MI355_CHECK_ARG(scale && inv_scale && growth_tracker && found_inf && interval > 0, "amp_update: bad arguments");
  hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, scale, inv_scale, growth_tracker, found_inf, growth,
                     backoff, interval);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

__global__ void fill_kernel(float* p, float v, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

extern "C" int mi355_fill_f32(float* p, float v, long long n, mi355_stream_t s) {
  MI355_CHECK_ARG(p && n >= 0, "fill_f32: bad arguments");
  if (n == 0) return MI355_OK;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, p, v, n);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// ---- gradient buckets on the wire ("fp32 (parity) or bf16 (perf) buckets") ----------------------------------------
// A bucket of the flat fp32 gradient buffer is rounded into a 2-byte staging buffer, summed over the ranks there, and widened back
// in place of the local gradients.  Four gradients per thread and trip: one 16-byte access on the fp32 side, 8 bytes of wire (a
// bucket starts on a parameter boundary = a multiple of four floats, mi355/engine.py ALIGN).
template <typename T> struct alignas(8) Wire4 { T v[4]; };

template <typename T>
__global__ __launch_bounds__(256) void grads_to_wire_kernel(const float* __restrict__ g, T* __restrict__ w, long long n) {
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(g + 4 * i);
    Wire4<T> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = from_f32<T>(a[e]);
    *reinterpret_cast<Wire4<T>*>(w + 4 * i) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) w[4 * n4 + threadIdx.x] = from_f32<T>(g[4 * n4 + threadIdx.x]);
}

template <typename T>
__global__ __launch_bounds__(256) void grads_from_wire_kernel(const T* __restrict__ w, float* __restrict__ g, long long n) {
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const Wire4<T> v = *reinterpret_cast<const Wire4<T>*>(w + 4 * i);
    f32x4 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = to_f32<T>(v.v[e]);
    *reinterpret_cast<f32x4*>(g + 4 * i) = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) g[4 * n4 + threadIdx.x] = to_f32<T>(w[4 * n4 + threadIdx.x]);
}

static int wire_blocks(long long n) {
  long long blocks = (n / 4 + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
}

extern "C" int mi355_grads_to_wire(const float* g, void* wire, long long n, int dtype, mi355_stream_t s) {
  if (n == 0) return MI355_OK;      // (an empty bucket has no storage to point at)
  MI355_CHECK_ARG(g && wire && n >= 0 && ((uintptr_t)g % 16 == 0) && ((uintptr_t)wire % 8 == 0) && dtype_is_2byte(dtype),
                  "grads_to_wire: gradients 16-byte aligned, wire 8-byte aligned, 2-byte wire dtype expected (dtype=%d)", dtype);
  return dispatch_dtype(dtype, "grads_to_wire", [&](auto tag) {
    using T = decltype(tag);
    if constexpr (sizeof(T) == 2) {
      hipLaunchKernelGGL((grads_to_wire_kernel<T>), dim3(wire_blocks(n)), dim3(256), 0, (hipStream_t)s, g, (T*)wire, n);
      MI355_LAUNCH_CHECK();
    }
    return (int)MI355_OK;
  });
}

extern "C" int mi355_grads_from_wire(const void* wire, float* g, long long n, int dtype, mi355_stream_t s) {
  if (n == 0) return MI355_OK;      // (an empty bucket has no storage to point at)
  MI355_CHECK_ARG(g && wire && n >= 0 && ((uintptr_t)g % 16 == 0) && ((uintptr_t)wire % 8 == 0) && dtype_is_2byte(dtype),
                  "grads_from_wire: gradients 16-byte aligned, wire 8-byte aligned, 2-byte wire dtype expected (dtype=%d)", dtype);
  return dispatch_dtype(dtype, "grads_from_wire", [&](auto tag) {
    using T = decltype(tag);
    if constexpr (sizeof(T) == 2) {
      hipLaunchKernelGGL((grads_from_wire_kernel<T>), dim3(wire_blocks(n)), dim3(256), 0, (hipStream_t)s, (const T*)wire, g, n);
      MI355_LAUNCH_CHECK();
    }
    return (int)MI355_OK;
  });
}

// ---- segmentation counters ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_counts_kernel(const float* __restrict__ pr, const float* __restrict__ tg,
                                                         float* __restrict__ counts, long long per, int is_logit, float thr) {
  const int b = blockIdx.y;
  const float* p = pr + (size_t)b * per;
  const float* t = tg + (size_t)b * per;
  float tp = 0, pp = 0, tt = 0, eq = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)gridDim.x * 256) {
    float v = p[i];
    if (is_logit) v = 1.f / (1.f + __expf(-v));
    const bool pb = v > thr, tb = t[i] > thr;
    tp += (pb && tb);
    pp += pb;
    tt += tb;
    eq += (pb == tb);
  }
  tp = block_sum_f(tp);
  pp = block_sum_f(pp);
  tt = block_sum_f(tt);
  eq = block_sum_f(eq);
  if (threadIdx.x == 0) {
    atomicAdd(counts + b * 4 + 0, tp);
    atomicAdd(counts + b * 4 + 1, pp);
    atomicAdd(counts + b * 4 + 2, tt);
    atomicAdd(counts + b * 4 + 3, eq);
  }
}

extern "C" int mi355_seg_counts(const float* prob_or_logit, const float* target, float* counts, int B, long long per,
                                int is_logit, float thr, mi355_stream_t s) {
  MI355_CHECK_ARG(prob_or_logit && target && counts && B > 0 && per > 0, "seg_counts: bad arguments");
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(float) * 4 * B, (hipStream_t)s);
  if (e != hipSuccess) MI355_FAIL((int)e, "seg_counts: memset failed: %s", hipGetErrorString(e));
  long long bx = (per + 2047) / 2048;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(seg_counts_kernel, dim3((int)bx, B), dim3(256), 0, (hipStream_t)s, prob_or_logit, target, counts, per, is_logit,
                     thr);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
