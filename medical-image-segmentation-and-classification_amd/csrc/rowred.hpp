// Column (per-channel) reduction skeleton over an NHWC tensor viewed as [M rows][C channels].
//
// HBM-bound by construction: every lane moves 16 contiguous bytes per access, consecutive
// lanes cover consecutive channel chunks of one pixel row (full-line coalescing), each
// workgroup walks a contiguous band of rows, keeps its per-channel partial sums in registers,
// folds the row-parallel thread groups through LDS once, and writes ONE partial row per
// workgroup:  partial[(block*NQ + q)*C + c].  A tiny finalize kernel sums the partials in a
// fixed order (deterministic, no float atomics).
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"

#define MI355_RR_MAX_BLOCKS 1024

static inline int rowreduce_blocks(long long M) {
  long long b = (M + 63) / 64;
  if (b < 1) b = 1;
  if (b > MI355_RR_MAX_BLOCKS) b = MI355_RR_MAX_BLOCKS;
  return (int)b;
}

struct RowRedGeom {
  long long M;
  int C;
  int rows_per_block;
  int tpr;   // threads per row (<= 256)
  int rp;    // rows walked in parallel by one block
  int nb_rows;   // rows of `partial` the caller folds (== rowreduce_blocks(M)); rows past the grid are written as zeros
};

template <typename T>
static inline RowRedGeom rowred_geom(long long M, int C, int nblocks) {
  RowRedGeom g;
  const int epc = 16 / (int)sizeof(T);
  const int cp = C / epc;
  g.M = M;
  g.C = C;
  g.rows_per_block = (int)((M + nblocks - 1) / nblocks);
  g.tpr = cp < 256 ? cp : 256;
  g.rp = 256 / g.tpr;
  return g;
}

// An op may split apply() into `In fetch(row, c0) const` (loads only) and `finish(in, row, c0[, acc])` (arithmetic and stores),
// with `static constexpr int FETCH_ROWS`: the kernels then fetch that many rows before finishing any of them.
template <typename Op, typename = void> struct has_max_wgs : std::false_type {};
template <typename Op> struct has_max_wgs<Op, std::void_t<decltype(Op::MAX_WGS)>> : std::true_type {};
template <typename Op, typename = void> struct has_fetch : std::false_type {};
template <typename Op> struct has_fetch<Op, std::void_t<typename Op::In>> : std::true_type {};
// An op with `static constexpr int BATCH_ROWS`, `struct Px`, `fetch(r, ty, c0, Px&)` and `finish(Px&, ty, c0, acc)` maps each batch of BATCH_ROWS * rp consecutive "virtual"
// rows (r = its first one) to pixels ITSELF (2 x 2-window aware orders); M is a multiple of the batch and every thread is
// active (the host checks both).
template <typename Op, typename = void> struct has_batch : std::false_type {};
// An op with `pin(In&)` (or `pin(Px&)`) passes every register its fetch loaded through pin16 below (an empty volatile asm with a
// memory clobber): the kernels call it for row b right in front of finish(row b), behind the fetch loop, which makes "every load
// of the batch is ISSUED before any row is finished" a dependence the compiler has to keep (loads do not move across the first
// pin, a row's arithmetic does not move in front of its own), while rows b + 1 .. stay in flight during row b's arithmetic.
// Left alone, hipcc hoists the first rows' arithmetic in between the loads and sinks the last row's loads behind the first waits —
// bn_bwd_reduce issued its eighth load behind an s_waitcnt vmcnt(0): one dependent round trip per batch, 3.3 TB/s.
template <typename Op, typename X, typename = void> struct has_pin : std::false_type {};
template <typename Op, typename X> struct has_pin<Op, X, std::void_t<decltype(std::declval<const Op&>().pin(std::declval<X&>()))>> : std::true_type {};
template <typename V> __device__ __forceinline__ void pin16(V& v) {
  static_assert(sizeof(V) == 16, "a 16-byte load result");
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t w = __builtin_bit_cast(u32x4_t, v);
  asm volatile("" : "+v"(w) : : "memory");
  v = __builtin_bit_cast(V, w);
}
template <typename Op> struct has_batch<Op, std::void_t<decltype(Op::BATCH_ROWS)>> : std::true_type {};

// Op contract:
//   static constexpr int NQ;                       number of reduced quantities
//   static constexpr bool WRITES;                  apply() also stores an output tensor
//   typedef ... Acc;                               float or double
//   __device__ void load_cols(int c0);             (optional per-chunk constants)
//   __device__ void apply(size_t row, int c0, Acc (&acc)[NQ][EPC]);   may also write outputs
template <typename T, typename Op>
__global__ __launch_bounds__(256) void rowred_kernel(Op op, RowRedGeom g, float* __restrict__ partial) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int NQ = Op::NQ;
  typedef typename Op::Acc Acc;
  __shared__ Acc red[256 * NQ * EPC / 2 + 1];   // only ty >= 1 rows are staged; sized for the worst case
  const int tid = threadIdx.x;
  const int tx = tid % g.tpr, ty = tid / g.tpr;
  const int chunk = blockIdx.y * 256 + tx;
  const int c0 = chunk * EPC;
  const bool active = ty < g.rp && c0 < g.C;
  Acc acc[NQ][EPC];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[q][e] = 0;
  if (active) {
    op.load_cols(c0);
    const long long r0 = (long long)blockIdx.x * g.rows_per_block;
    long long r1 = r0 + g.rows_per_block;
    if (r1 > g.M) r1 = g.M;
    if constexpr (has_batch<Op>::value) {
      // (issuing the next batch's loads before the current one's arithmetic — ping-pong register sets — and two or three workgroups
      // per CU were measured within noise of this loop: profiles/r03c_ab_poolbwd.txt)
      constexpr int B = Op::BATCH_ROWS;
      const long long st = (long long)gridDim.x * g.rp * B;
      long long r = (long long)blockIdx.x * g.rp * B;
#ifndef RR_NO_RING
      // the NEXT batch is requested as soon as this one has landed, in front of its arithmetic (these ops recompute activations
      // and window maxima: a microsecond of VALU work per batch during which nothing was in flight); the last trip requests its
      // own batch again — a clamped address, not a load under a condition — and ignores it
      if (r < g.M) {
        typename Op::Px p;
        op.fetch(r, ty, c0, p);
        for (;;) {
          const long long rn = r + st;
          const bool more = rn < g.M;
          if constexpr (has_pin<Op, typename Op::Px>::value) op.pin(p);
          typename Op::Px cur = p;
          op.fetch(more ? rn : r, ty, c0, p);
          op.finish(cur, ty, c0, acc);
          r = rn;
          if (!more) break;
        }
      }
#else
      for (; r < g.M; r += st) {
        typename Op::Px p;
        op.fetch(r, ty, c0, p);
        if constexpr (has_pin<Op, typename Op::Px>::value) op.pin(p);      // (all loads of the batch issued before the arithmetic: see has_pin)
        op.finish(p, ty, c0, acc);
      }
#endif
    } else if constexpr (has_fetch<Op>::value) {
      // ops that also store: the compiler may not move a load above the previous row's store (the tensors can alias), so
      // the rows of a trip are fetched explicitly before any of them is finished — FETCH_ROWS x the bytes in flight
      // A workgroup pass covers B * rp consecutive rows and the grid sweeps the tensor as one moving window (workgroups that
      // each own a far-apart contiguous band were measured slower: see DESIGN.md, streaming kernels).
      constexpr int B = Op::FETCH_ROWS;
      const long long st = (long long)gridDim.x * g.rp * B;
      long long r = (long long)blockIdx.x * g.rp * B + ty;
#ifndef RR_NO_RING
      // The B rows in flight are a RING: as soon as row b of this batch has been waited for, row b of the workgroup's NEXT batch
      // is requested into the same slot, in front of row b's arithmetic — the loads never drain between batches.  The last
      // batch requests its own rows again (a clamped, unconditional address: a load under `if (more)` would be waited for on the
      // spot) and ignores them.
      if (r + (long long)(B - 1) * g.rp < g.M) {
        typename Op::In in[B];
#pragma unroll
        for (int b = 0; b < B; ++b) in[b] = op.fetch((size_t)(r + (long long)b * g.rp), c0);
        for (;;) {
          const long long rn = r + st;
          const bool more = rn + (long long)(B - 1) * g.rp < g.M;
          const long long rf = more ? rn : r;
#pragma unroll
          for (int b = 0; b < B; ++b) {
            if constexpr (has_pin<Op, typename Op::In>::value) op.pin(in[b]);      // (waits for row b only; the rest of the ring stays in flight)
            const typename Op::In cur = in[b];
            in[b] = op.fetch((size_t)(rf + (long long)b * g.rp), c0);
            op.finish(cur, (size_t)(r + (long long)b * g.rp), c0, acc);
          }
          r = rn;
          if (!more) break;
        }
      }
      for (long long q = r; q < g.M; q += g.rp) op.finish(op.fetch((size_t)q, c0), (size_t)q, c0, acc);      // the rows behind the last whole batch
#else
      for (; r < g.M; r += st) {
        if (r + (long long)(B - 1) * g.rp < g.M) {
          typename Op::In in[B];
#pragma unroll
          for (int b = 0; b < B; ++b) in[b] = op.fetch((size_t)(r + (long long)b * g.rp), c0);
          // every load of the batch is ISSUED before the first row is finished: left alone, the scheduler sinks the later rows'
          // loads towards their uses (a read-only reduction has no store to hold them back) and two or three rows are in flight
          // instead of B: Op::pin, see has_pin above
#pragma unroll
          for (int b = 0; b < B; ++b) {
            if constexpr (has_pin<Op, typename Op::In>::value) op.pin(in[b]);      // (waits for row b only; rows b + 1 .. stay in flight)
            op.finish(in[b], (size_t)(r + (long long)b * g.rp), c0, acc);
          }
        } else {
          for (long long q = r; q < g.M; q += g.rp) op.finish(op.fetch((size_t)q, c0), (size_t)q, c0, acc);
        }
      }
#endif
    } else if constexpr (!Op::WRITES && sizeof(typename Op::Acc) == 4) {
      // read-only reductions: two rows per trip keep twice the bytes in flight (+5 % on bn_bwd_reduce; ops that also
      // store — bn_bwd_apply — were measured slower with it)
#pragma unroll 2
      for (long long r = r0 + ty; r < r1; r += g.rp) op.apply((size_t)r, c0, acc);
    } else {
      for (long long r = r0 + ty; r < r1; r += g.rp) op.apply((size_t)r, c0, acc);
    }
  }
  // fold the rp row-groups: log-step tree over ty through LDS
  for (int stride = 1; stride < g.rp; stride <<= 1) {
    // pairs (ty, ty+stride) with ty % (2*stride) == 0
    const bool sender = active && (ty % (2 * stride)) == stride;
    const bool recver = active && (ty % (2 * stride)) == 0 && (ty + stride) < g.rp;
    __syncthreads();
    if (sender) {
      Acc* dst = red + (size_t)(((ty - stride) / (2 * stride)) * g.tpr + tx) * NQ * EPC;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < EPC; ++e) dst[q * EPC + e] = acc[q][e];
    }
    __syncthreads();
    if (recver) {
      const Acc* src = red + (size_t)((ty / (2 * stride)) * g.tpr + tx) * NQ * EPC;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[q][e] += src[q * EPC + e];
    }
  }
  if (active && ty == 0 && partial) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < EPC; ++e) partial[((size_t)blockIdx.x * NQ + q) * g.C + c0 + e] = (float)acc[q][e];
    for (int b = blockIdx.x + gridDim.x; b < g.nb_rows; b += gridDim.x)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < EPC; ++e) partial[((size_t)b * NQ + q) * g.C + c0 + e] = 0.f;
  }
}

// Workgroups a row reduction of op `Op` runs on = the partial rows it can leave non-zero (the rest, up to rowreduce_blocks(M), are
// written as zeros): what a fold has to read.
template <typename Op> static inline int rowred_grid(long long M) {
  int nb = rowreduce_blocks(M);
  static const int rr_wgs = getenv("MI355_RR_WGS") ? atoi(getenv("MI355_RR_WGS")) : 0;      // (A/B switch: one cap for every op)
  if constexpr (has_fetch<Op>::value || has_batch<Op>::value) {
    int cap = 256;
    if constexpr (has_max_wgs<Op>::value) cap = Op::MAX_WGS;      // (narrow rows: rowdot_bwd 256 / 512 / 1024 workgroups = 0.307 / 0.218 / 0.212 ms per step)
    if (rr_wgs > 0) cap = rr_wgs;
    nb = nb < cap ? nb : cap;
  }
  return nb;
}

template <typename T, typename Op>
static inline int rowred_launch(const Op& op, long long M, int C, float* partial, hipStream_t s) {
  const int epc = 16 / (int)sizeof(T);
  if (C % epc != 0) {
    mi355_set_error("row reduction: C=%d must be a multiple of %d", C, epc);
    return MI355_ERR_ARG;
  }
  // The caller sized `partial` (and tells the finalize kernels) rowreduce_blocks(M) rows.  Ops that keep several rows in
  // flight per thread stream faster from ONE workgroup per CU (bn_bwd_apply 256^2 x 64: 1024 / 512 / 256 workgroups =
  // 4.97 / 5.03 / 5.35 TB/s); the rows of `partial` they do not produce are zero-filled.
  const int nb_rows = rowreduce_blocks(M);
  const int nb = rowred_grid<Op>(M);
  RowRedGeom g = rowred_geom<T>(M, C, nb);
  g.nb_rows = nb_rows;
  const int cp = C / epc;
  dim3 grid(nb, (cp + 255) / 256);
  hipLaunchKernelGGL((rowred_kernel<T, Op>), grid, dim3(256), 0, s, op, g, partial);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mi355_set_error("row reduction launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return MI355_OK;
}

// Elementwise skeleton over [M][C]: Op::load_cols(c0) once per thread, Op::apply(row, c0) per 16-B chunk.
template <typename T, typename Op>
__global__ __launch_bounds__(256) void rowmap_kernel(Op op, long long M, int cp) {
  constexpr int EPC = 16 / (int)sizeof(T);
  // tpr lanes cover one pixel row's chunks (strided when cp > 256); rp rows per workgroup pass.
  // No per-element division: the (row, chunk) split is fixed per thread.
  const int tpr = cp < 256 ? cp : 256;
  const int rp = 256 / tpr;
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
  if (ty >= rp) return;
  for (int c = tx; c < cp; c += tpr) {          // one trip unless a row has more than 256 chunks
    op.load_cols(c * EPC);                       // per-channel constants live in registers for the whole band
    if constexpr (has_fetch<Op>::value) {        // (see rowred_kernel: explicit fetch-then-finish batches)
      // a workgroup pass covers B * rp CONSECUTIVE rows (thread rows r, r + rp, ...): the chip still sweeps the tensor as one
      // moving window (batches a grid stride apart were measured 18 % slower than no batching at all)
      constexpr int B = Op::FETCH_ROWS;
      const long long st = (long long)gridDim.x * rp * B;
      for (long long r = (long long)blockIdx.x * rp * B + ty; r < M; r += st) {
        if (r + (long long)(B - 1) * rp < M) {
          typename Op::In in[B];
#pragma unroll
          for (int b = 0; b < B; ++b) in[b] = op.fetch((size_t)(r + b * rp), c * EPC);
#pragma unroll
          for (int b = 0; b < B; ++b) {
            if constexpr (has_pin<Op, typename Op::In>::value) op.pin(in[b]);      // (all loads of the batch issued before the first finish: see has_pin)
            op.finish(in[b], (size_t)(r + b * rp), c * EPC);
          }
        } else {
          for (long long q = r; q < M; q += rp) op.finish(op.fetch((size_t)q, c * EPC), (size_t)q, c * EPC);
        }
      }
    } else {
      for (long long r = (long long)blockIdx.x * rp + ty; r < M; r += (long long)gridDim.x * rp) op.apply((size_t)r, c * EPC);
    }
  }
}

template <typename T, typename Op>
static inline int rowmap_launch(const Op& op, long long M, int C, hipStream_t s) {
  const int epc = 16 / (int)sizeof(T);
  if (C % epc != 0) {
    mi355_set_error("elementwise: C=%d must be a multiple of %d", C, epc);
    return MI355_ERR_ARG;
  }
  const int cp = C / epc;
  const int rp = 256 / (cp < 256 ? cp : 256);
  long long rpb = rp;
  if constexpr (has_fetch<Op>::value) rpb *= Op::FETCH_ROWS;
  long long blocks = (M + rpb - 1) / rpb;
  // grid-stride beyond 16 workgroups per CU; ops that keep several rows in flight per thread stream faster from 4 per CU
  // (bn_act 256^2 x 64: 4096 / 1024 / 512 workgroups = 4.18 / 4.45 / 4.45 TB/s)
  static const int rm_wgs = getenv("MI355_RM_WGS") ? atoi(getenv("MI355_RM_WGS")) : 256 * 4;      // (A/B switch)
  const long long cap = has_fetch<Op>::value ? rm_wgs : 256 * 16;
  if (blocks > cap) blocks = cap;     // grid-stride beyond 16 workgroups per CU
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((rowmap_kernel<T, Op>), dim3((int)blocks), dim3(256), 0, s, op, M, cp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mi355_set_error("elementwise launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return MI355_OK;
}
