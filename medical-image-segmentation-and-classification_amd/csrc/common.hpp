// Shared device/host helpers for libmi355conv (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mi355conv.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MI355_WAVE 64

// ---- error plumbing (no exceptions cross the C ABI) -------------------------------------
void mi355_set_error(const char* fmt, ...);
#define MI355_FAIL(code, ...)          \
  do {                                 \
    mi355_set_error(__VA_ARGS__);      \
    return (code);                     \
  } while (0)
#define MI355_CHECK_ARG(cond, ...)                         \
  do {                                                     \
    if (!(cond)) MI355_FAIL(MI355_ERR_ARG, __VA_ARGS__);   \
  } while (0)
#define MI355_LAUNCH_CHECK()                                                        \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) MI355_FAIL((int)e__, "launch failed: %s", hipGetErrorString(e__)); \
  } while (0)

// ---- XCD-aware tile order ------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin (workgroup b -> XCD b % 8, each with a private L2).  Tiles that
// are neighbours in the logical order share operand panels (the same input patch under different channel tiles,
// the same pixel rows under different weight tiles), so every XCD gets ONE CONTIGUOUS RANGE of the logical order
// and the shared panels meet in one L2 instead of being fetched from HBM by up to eight.  Bijective for any n.
__device__ __forceinline__ int xcd_tile(int b, int n) {
  const int q = n >> 3, r = n & 7;           // XCD k owns q + (k < r) tiles, starting at k*q + min(k, r)
  const int k = b & 7;
  return k * q + (k < r ? k : r) + (b >> 3);
}

// ---- scalar conversions ------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }

// 16-byte vector of T (4 x f32 or 8 x bf16)
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float v[4];
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  bf16_t v[8];
};
template <> struct Vec16<f16_t> {
  static constexpr int N = 8;
  f16_t v[8];
};

// ---- 2-byte MFMA by element type (fragments travel as raw 16-byte vectors typed bf16x8) ---------------------
template <typename T> __device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma_32x32x16<bf16_t>(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma_32x32x16<f16_t>(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma_16x16x32<bf16_t>(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma_16x16x32<f16_t>(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// In-place forms (inline asm, accumulator tied to the destination).  The builtin lets the register allocator give D a tuple
// different from C (the early-clobber 3-address form): accumulators then wander through the file and every MFMA needs a spare
// tuple — fine with registers to spare, 200+ spills in a kernel that keeps 128 accumulators in a 256-register budget.
// The compiler inserts the operand waits (lgkmcnt of the fragment reads) but NOT the MFMA -> VALU read wait states of its hazard
// recogniser: a kernel using these calls mfma_results_ready() before any non-MFMA instruction reads an accumulator.
template <typename T> __device__ __forceinline__ void mfma_16x16x32_acc(bf16x8 a, bf16x8 b, f32x4& c);
template <> __device__ __forceinline__ void mfma_16x16x32_acc<bf16_t>(bf16x8 a, bf16x8 b, f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_16x16x32_acc<f16_t>(bf16x8 a, bf16x8 b, f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <typename T> __device__ __forceinline__ void mfma_16x16x32_first(bf16x8 a, bf16x8 b, f32x4& c);      // c = a * b
template <> __device__ __forceinline__ void mfma_16x16x32_first<bf16_t>(bf16x8 a, bf16x8 b, f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_16x16x32_first<f16_t>(bf16x8 a, bf16x8 b, f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
// (the sched_barrier keeps every later instruction — a pure VALU read of an accumulator has no data dependence on the nops —
// behind them)
__device__ __forceinline__ void mfma_results_ready() {
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: quad_perm [1,0,3,2], [2,3,0,1], then
// row_half_mirror and row_mirror (after the quad steps every lane of a quad holds the quad sum, so mirroring pairs distinct
// quads / halves).  Four VALU adds with a DPP source, no LDS traffic (__shfl_xor compiles to ds_bpermute_b32).
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  return dpp_add<0x140>(v);
}

// streaming read: the line is not kept in the caches behind it (read-once operands of the HBM-bound passes)
template <typename T> __device__ __forceinline__ Vec16<T> ld16_nt(const T* p) {
  Vec16<T> r;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<u32x4*>(&r) = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return r;
}
template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
  Vec16<T> r;
  *reinterpret_cast<uint4*>(&r) = *reinterpret_cast<const uint4*>(p);
  return r;
}
// (the same load under a name the per-file `#define ld16 ld16_nt` of the streaming kernels does not capture)
template <typename T> __device__ __forceinline__ Vec16<T> ld16_plain(const T* p) {
  Vec16<T> r;
  *reinterpret_cast<uint4*>(&r) = *reinterpret_cast<const uint4*>(p);
  return r;
}
// ... with the cache policy as a COMPILE-TIME choice: a run-time switch between the two loads (`keep ? ld16(p) : ld16_nt(p)`, or an
// if / else around them) makes hipcc wait for the loads inside the branch (s_waitcnt vmcnt(0)) and takes the other rows of a
// fetch batch out of flight — measured on the BatchNorm backward passes: reduce 3.9 -> 4.1-4.5 TB/s, apply 5.0 -> 5.2-5.7
template <bool KEEP, typename T> __device__ __forceinline__ Vec16<T> ld16_pol(const T* p) {
  if constexpr (KEEP) return ld16_plain<T>(p);
  else return ld16_nt<T>(p);
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& r) {
  *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(&r);
}

// ---- reductions --------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- dtype dispatch of the launchers: f(T{}) with T = float / bf16_t / f16_t ------------------------------
template <typename F> static inline int dispatch_dtype(int dtype, const char* who, F&& f) {
  switch (dtype) {
    case MI355_F32: return f(float{});
    case MI355_BF16: return f(bf16_t{});
    case MI355_F16: return f(f16_t{});
    default: MI355_FAIL(MI355_ERR_UNSUPPORTED, "%s: unknown dtype %d", who, dtype);
  }
}
static inline bool dtype_is_2byte(int dtype) { return dtype == MI355_BF16 || dtype == MI355_F16; }

// dx of a train-mode BatchNorm, gi * (g - k0 - xhat * k1), with the fused multiply-add written out: every kernel that evaluates it
// (plain / pool-aware / gate / post4 apply passes, each in several instantiations) then rounds the same way — left to the compiler,
// the contraction differed between two instantiations of the same op and the results in the last bit (seen in fp16 outputs).
__device__ __forceinline__ float bn_dx(float gi, float g, float k0, float xhat, float k1) { return gi * __builtin_fmaf(-xhat, k1, g - k0); }
