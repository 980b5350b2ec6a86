// Attention-gate tail and single-output 1x1 convolutions (Co == 1): per-pixel dot products.
// These are HBM-bound row reductions, not GEMMs: TPR lanes (a power of two <= 64) share one pixel
// row, each lane moving 16-B channel chunks, folded with xor-shuffles; per-workgroup (sum, sum^2)
// partials feed the one-channel BatchNorm that follows psi (AttentionUNet.py:40-44).
#include "rowred.hpp"
// Every 16-byte load of this file is a streaming read of an operand the kernel touches once: nontemporal (A/B over a train step:
// -0.07 ms for the gate kernels, -0.12 ms for the pooling / add / up-sampling ones; -DKEEP_CACHED restores the default policy)
#ifndef KEEP_CACHED
#define ld16 ld16_nt
#endif

#define ROWDOT_MAXCH 4

// Workgroups of the moving-window row-dot kernels (the callers size `partial` for rowreduce_blocks(M) rows: the rest is zero-filled)
static inline int rowdot_grid(int nb) {
  static const int cap = getenv("MI355_ROWDOT_WGS") ? atoi(getenv("MI355_ROWDOT_WGS")) : 1024;      // (A/B switch)
  return nb < cap ? nb : cap;
}

static inline int pow2_tpr(int cp) {
  int t = 1;
  while (t < cp && t < 64) t <<= 1;
  return t;
}

__device__ __forceinline__ float seg_sum(float v, int tpr) {
  for (int o = tpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-level sum of two doubles -> partial[blk*2 + {0,1}]
__device__ __forceinline__ void block_pair_sum(double s0, double s1, float* partial, int nb_rows = 0) {
  __shared__ double red[8];
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave * 2] = s0;
    red[wave * 2 + 1] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0 && partial) {
    partial[blockIdx.x * 2 + 0] = (float)(red[0] + red[2] + red[4] + red[6]);
    partial[blockIdx.x * 2 + 1] = (float)(red[1] + red[3] + red[5] + red[7]);
    for (int b = blockIdx.x + gridDim.x; b < nb_rows; b += gridDim.x) partial[b * 2] = partial[b * 2 + 1] = 0.f;      // rows past the grid
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ z,
                                                         float* __restrict__ partial, long long M, int C,
                                                         int tpr, int hw, int K, int nb_rows) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int cp = C / EPC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % tpr, rsub = lane / tpr, rpw = 64 / tpr;
  float wr[ROWDOT_MAXCH][EPC];
#pragma unroll
  for (int k = 0; k < ROWDOT_MAXCH; ++k) {
    const int ck = sub + k * tpr;
    const int cb = (ck < cp ? ck : 0) * EPC;        // (unconditional loads from a clamped chunk: see gate_psi_fwd_kernel)
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float wv = w[cb + e];
      wr[k][e] = ck < cp ? wv : 0.f;
    }
  }
  const float bias = b ? b[0] : 0.f;
  // one moving window over the tensor (see gate_psi_fwd_kernel), U row groups in flight per lane when a row is one chunk per lane
  constexpr int U = 4;
  const long long r1 = M;
  const long long chunk = (long long)U * 4 * rpw;
  double s0 = 0, s1 = 0;
  for (long long base = blockIdx.x * chunk + wave * rpw; base < r1; base += (long long)gridDim.x * chunk) {
    Vec16<T> v0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {        // (unconditional loads from a clamped (row, chunk): see gate_psi_fwd_kernel)
      const long long r = base + (long long)u * 4 * rpw + rsub;
      v0[u] = ld16<T>(x + (size_t)(r < r1 ? r : r1 - 1) * ldx + (sub < cp ? sub : 0) * EPC);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = base + (long long)u * 4 * rpw + rsub;
      float acc = 0.f;
      if (r < r1) {
#pragma unroll
        for (int k = 0; k < ROWDOT_MAXCH; ++k) {
          const int ck = sub + k * tpr;
          if (ck < cp) {
            const Vec16<T> v = k == 0 ? v0[u] : ld16<T>(x + (size_t)r * ldx + ck * EPC);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc += to_f32<T>(v.v[e]) * wr[k][e];
          }
        }
      }
      acc = seg_sum(acc, tpr);
      if (r < r1 && sub == 0) {
        const float zz = acc + bias;
        z[K == 1 ? r : r + (r / hw) * (long long)(K - 1) * hw] = zz;      // K > 1: channel plane of an NCHW [N][K][hw] map
        s0 += zz;
        s1 += (double)zz * zz;
      }
    }
  }
  block_pair_sum(s0, s1, partial, nb_rows);
}

extern "C" int mi355_rowdot_fwd(const void* x, int ldx, const float* w, const float* b, float* z, float* partial,
                                long long M, int C, int HW, int K, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(x && w && z, "rowdot_fwd: null pointer");
  MI355_CHECK_ARG(K >= 1 && (K == 1 || (HW > 0 && M % HW == 0)), "rowdot_fwd: K=%d planes need HW | M", K);
  const int epc = dtype_is_2byte(dtype) ? 8 : 4;
  MI355_CHECK_ARG(C % epc == 0 && C / epc <= 64 * ROWDOT_MAXCH, "rowdot_fwd: unsupported C=%d", C);
  const int nb = rowreduce_blocks(M);
  const int tpr = pow2_tpr(C / epc);
  return dispatch_dtype(dtype, "rowdot_fwd", [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((rowdot_fwd_kernel<T>), dim3(rowdot_grid(nb)), dim3(256), 0, (hipStream_t)s, (const T*)x, ldx, w, b, z, partial, M, C,
                       tpr, HW, K, nb);
    MI355_LAUNCH_CHECK();
    return (int)MI355_OK;
  });
}

// dx[m][c] = dz[m]*w[c]*(mask ? x[m][c] > 0 : 1);  partial: q0 = sum_m dz[m]*x[m][c], q1 = sum_m dz[m]
// ACC (dx accumulates: a further output channel of a multi-channel head) is compile-time and its operand part of the fetch: a load
// under a run-time condition inside finish() would be waited for on the spot (common.hpp, ld16_pol)
template <typename T, bool ACC = false> struct RowdotBwdOp {
  static constexpr int NQ = 2;
  static constexpr bool WRITES = true;
  typedef float Acc;
  static constexpr int EPC = 16 / (int)sizeof(T);
  const float* dz; const T* x; int ldx; const float* w; T* dx; int lddx; int mask; int hw; int K;
  float wr[EPC];
  __device__ void load_cols(int c0) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) wr[e] = w[c0 + e];
  }
  static constexpr int FETCH_ROWS = 4;        // (rowred.hpp: rows fetched before any is finished)
  static constexpr int MAX_WGS = 1024;        // one fp32 + one 2-byte stream per row: more workgroups than the BN reductions' one per CU
  struct In { Vec16<T> v, old; float d; };
  __device__ In fetch(size_t row, int c0) const {
    In in;
    in.v = ld16<T>(x + row * ldx + c0);
    if constexpr (ACC) in.old = ld16<T>(dx + row * lddx + c0);
    in.d = dz[K == 1 ? row : row + (row / hw) * (size_t)(K - 1) * hw];
    return in;
  }
  __device__ void pin(In& in) const {
    pin16(in.v);
    if constexpr (ACC) pin16(in.old);
    asm volatile("" : "+v"(in.d) : : "memory");
  }
  __device__ void finish(const In& in, size_t row, int c0, Acc (&acc)[NQ][EPC]) const {
    const float d = in.d;
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float xv = to_f32<T>(in.v.v[e]);
      acc[0][e] += d * xv;
      acc[1][e] += d;
      o.v[e] = from_f32<T>((mask && !(xv > 0.f)) ? 0.f : d * wr[e]);
    }
    if constexpr (ACC) {                      // a further output channel of a multi-channel head adds its share
#pragma unroll
      for (int e = 0; e < EPC; ++e) o.v[e] = from_f32<T>(to_f32<T>(in.old.v[e]) + to_f32<T>(o.v[e]));
    }
    if (dx) st16<T>(dx + row * lddx + c0, o);
  }
};

extern "C" int mi355_rowdot_bwd(const float* dz, const void* x, int ldx, const float* w, void* dx, int lddx,
                                float* partial, long long M, int C, int relu_mask, int HW, int K, int accumulate, int dtype,
                                mi355_stream_t s) {
  MI355_CHECK_ARG(dz && x && w && partial, "rowdot_bwd: null pointer");
  MI355_CHECK_ARG(K >= 1 && (K == 1 || (HW > 0 && M % HW == 0)), "rowdot_bwd: K=%d planes need HW | M", K);
  return dispatch_dtype(dtype, "rowdot_bwd", [&](auto tag) {
    using T = decltype(tag);
    if (dx && accumulate) {
      RowdotBwdOp<T, true> op{dz, (const T*)x, ldx, w, (T*)dx, lddx, relu_mask, HW, K};
      return rowred_launch<T>(op, M, C, partial, (hipStream_t)s);
    }
    RowdotBwdOp<T, false> op{dz, (const T*)x, ldx, w, (T*)dx, lddx, relu_mask, HW, K};
    return rowred_launch<T>(op, M, C, partial, (hipStream_t)s);
  });
}

// ---- psi_in = relu(BN_g(g1) + BN_x(x1)) and the one-channel psi convolution in one pass (AttentionUNet.py:48-52) --------------
// z[m] = b + sum_c w[c] * psi_in[m][c] with psi_in computed from the raw branch outputs as mi355_bn_act would have stored it
// (same fmaf chain, rounded to the storage type) — and NOT stored: the backward (mi355_gate_bn_bwd_*) recomputes it too.  The
// dot product sums in mi355_rowdot_fwd's order.  Four rows per lane group are in flight (two 16-byte loads each).
template <typename T, bool TWO>
__global__ __launch_bounds__(256) void gate_psi_fwd_kernel(const T* __restrict__ g1, int ldg, const T* __restrict__ x1, int ldx,
                                                           const float* __restrict__ scale_g, const float* __restrict__ shift_g,
                                                           const float* __restrict__ scale_x, const float* __restrict__ shift_x,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           float* __restrict__ z, float* __restrict__ partial, long long M, int C,
                                                           int tpr, int nb_rows) {
  constexpr int EPC = 16 / (int)sizeof(T);
#ifndef GATE_PSI_U
#define GATE_PSI_U 4
#endif
  constexpr int U = GATE_PSI_U;
  const int cp = C / EPC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % tpr, rsub = lane / tpr, rpw = 64 / tpr;
  const bool on = sub < cp;
  // per-channel constants: unconditional loads from a clamped chunk (a select per element made the compiler issue 40 dependent
  // single-dword loads, each waited for: ≈15 us of prologue per launch), lanes past the last chunk zero their weights
  float sg[EPC], sx[EPC], sh[EPC], wr[EPC];
  const int cb = (on ? sub : 0) * EPC;
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    sg[e] = scale_g[cb + e];
    sx[e] = TWO ? scale_x[cb + e] : 0.f;
    sh[e] = TWO ? shift_g[cb + e] + shift_x[cb + e] : shift_g[cb + e];
    const float wv = w[cb + e];
    wr[e] = on ? wv : 0.f;
  }
  const float bias = b ? b[0] : 0.f;
  // the grid sweeps the tensor as ONE moving window of gridDim.x chunks of U * 4 * rpw consecutive rows (workgroups that each
  // stream a far-apart band of their own were 1.3-2x slower: 1024 concurrent DRAM streams)
  const long long r1 = M;
  const long long chunk = (long long)U * 4 * rpw;
  double s0 = 0, s1 = 0;
  for (long long base = blockIdx.x * chunk + wave * rpw; base < r1; base += (long long)gridDim.x * chunk) {
    Vec16<T> gv[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {        // unconditional loads from a clamped (row, chunk): a guarded load is waited for before the next is issued
      const long long r = base + (long long)u * 4 * rpw + rsub;
      const size_t rc = (size_t)(r < r1 ? r : r1 - 1);
      gv[u] = ld16<T>(g1 + rc * ldg + cb);
      if constexpr (TWO) xv[u] = ld16<T>(x1 + rc * ldx + cb);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = base + (long long)u * 4 * rpw + rsub;
      float acc = 0.f;      // (no guard: lanes past the last chunk carry zero weights, rows past the end are not stored)
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float f = __builtin_fmaf(to_f32<T>(gv[u].v[e]), sg[e], sh[e]);
        if constexpr (TWO) f = __builtin_fmaf(to_f32<T>(xv[u].v[e]), sx[e], f);
        acc += to_f32<T>(from_f32<T>(fmaxf(f, 0.f))) * wr[e];
      }
      acc = seg_sum(acc, tpr);
      if (r < r1 && sub == 0) {
        const float zz = acc + bias;
        z[r] = zz;
        s0 += zz;
        s1 += (double)zz * zz;
      }
    }
  }
  block_pair_sum(s0, s1, partial, nb_rows);
}

extern "C" int mi355_gate_psi_fwd_ok(int C, int dtype) {
  const int epc = dtype_is_2byte(dtype) ? 8 : 4;
  return C % epc == 0 && C / epc <= 64;
}

extern "C" int mi355_gate_psi_fwd(const void* g1, int ldg, const void* x1, int ldx, const float* scale_g, const float* shift_g,
                                  const float* scale_x, const float* shift_x, const float* w, const float* b, float* z,
                                  float* partial, long long M, int C, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(g1 && scale_g && shift_g && w && z && (!x1 || (scale_x && shift_x)), "gate_psi_fwd: null pointer");
  MI355_CHECK_ARG(mi355_gate_psi_fwd_ok(C, dtype), "gate_psi_fwd: unsupported C=%d", C);
  const int epc = dtype_is_2byte(dtype) ? 8 : 4;
  const int nb = rowreduce_blocks(M);      // (the partial rows the one-channel BatchNorm folds: every workgroup writes its own)
  const int tpr = pow2_tpr(C / epc);
  return dispatch_dtype(dtype, "gate_psi_fwd", [&](auto tag) {
    using T = decltype(tag);
    if (x1)
      hipLaunchKernelGGL((gate_psi_fwd_kernel<T, true>), dim3(rowdot_grid(nb)), dim3(256), 0, (hipStream_t)s, (const T*)g1, ldg, (const T*)x1, ldx,
                         scale_g, shift_g, scale_x, shift_x, w, b, z, partial, M, C, tpr, nb);
    else      // one normalised operand: relu(bn(y)) in front of a one-channel convolution (the logit head, AttentionUNet.py:84)
      hipLaunchKernelGGL((gate_psi_fwd_kernel<T, false>), dim3(rowdot_grid(nb)), dim3(256), 0, (hipStream_t)s, (const T*)g1, ldg, (const T*)nullptr, 0,
                         scale_g, shift_g, nullptr, nullptr, w, b, z, partial, M, C, tpr, nb);
    MI355_LAUNCH_CHECK();
    return (int)MI355_OK;
  });
}

// ---- backward of the gate's two normalised branches, W_g / W_x (AttentionUNet.py:32-38,48-52) ---------------------------------
// p = relu(bn_g(g1) + bn_x(x1)) feeds the one-channel psi convolution; given dz (the gradient of that convolution's output) the
// gradient of p is dz[m] * w[c] where p > 0 — a tensor mi355_rowdot_bwd would write and FOUR BatchNorm passes (two branches x
// reduce / apply) would read.  These two passes recompute p from the raw branch outputs exactly as mi355_bn_act rounded it and
// serve both branches at once: reduce reads g1, x1 and dz and leaves five quantities per channel,
//   q0 = sum dp, q1 = sum dp * xhat_g, q2 = sum dp * xhat_x, q3 = sum dz * p (the psi weight's gradient), q4 = sum dz (its bias's);
// apply reads the same and writes both branches' input gradients.
template <typename T, bool TWO, bool KEEP> struct GateBnBwd {
  static constexpr int EPC = 16 / (int)sizeof(T);
  const float* dz; const T* g1; int ldg; const T* x1; int ldx;
  const float* scale_g; const float* shift_g; const float* scale_x; const float* shift_x;
  const float* mean_g; const float* invstd_g; const float* mean_x; const float* invstd_x; const float* w;
  float sg[EPC], sx[EPC], sh[EPC], mg[EPC], ig[EPC], mx[EPC], ix[EPC], wr[EPC];
  __device__ void load_common(int c0) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      sg[e] = scale_g[c0 + e]; mg[e] = mean_g[c0 + e]; ig[e] = invstd_g[c0 + e];
      if constexpr (TWO) {
        sx[e] = scale_x[c0 + e]; sh[e] = shift_g[c0 + e] + shift_x[c0 + e];      // (BnActOp::load_cols)
        mx[e] = mean_x[c0 + e]; ix[e] = invstd_x[c0 + e];
      } else {
        sx[e] = 0.f; sh[e] = shift_g[c0 + e]; mx[e] = 0.f; ix[e] = 0.f;
      }
      wr[e] = w[c0 + e];
    }
  }
  static constexpr int FETCH_ROWS = 4;
  struct In { Vec16<T> gv, xv; float d; };
  __device__ In fetch(size_t row, int c0) const {
    In in;
    in.gv = ld16_pol<KEEP, T>(g1 + row * ldg + c0);      // (KEEP: cache policy of the branch reads, compile-time: see ld16_pol)
    if constexpr (TWO) in.xv = ld16_pol<KEEP, T>(x1 + row * ldx + c0);
    in.d = dz[row];
    return in;
  }
  __device__ void pin(In& in) const {
    pin16(in.gv);
    if constexpr (TWO) pin16(in.xv);
    asm volatile("" : "+v"(in.d) : : "memory");
  }
  // the activation as the forward stored it (BnActOp::finish with a second operand and ReLU)
  __device__ float act(const In& in, int e) const {
    float f = __builtin_fmaf(to_f32<T>(in.gv.v[e]), sg[e], sh[e]);
    if constexpr (TWO) f = __builtin_fmaf(to_f32<T>(in.xv.v[e]), sx[e], f);
    return to_f32<T>(from_f32<T>(fmaxf(f, 0.f)));
  }
  // ... and its gradient as mi355_rowdot_bwd would have STORED it (rounded to the storage type): the two-pass backward is then
  // bit-identical to the separate passes, and a 2-byte training trajectory does not depend on which of the two ran
  __device__ float dpsi(const In& in, int e, float p) const { return p > 0.f ? to_f32<T>(from_f32<T>(in.d * wr[e])) : 0.f; }
};

template <typename T, bool TWO, bool KEEP = true> struct GateBnBwdReduceOp : GateBnBwd<T, TWO, KEEP> {
  static constexpr int MAX_WGS = 512;         // ≈20 VALU instructions per element: two waves per SIMD (scripts/gate_bench.py: 256 / 512 / 768 = 3.9 / 4.8 / 4.6 TB/s)
  static constexpr int NQ = 5;
  static constexpr bool WRITES = false;
  typedef float Acc;
  static constexpr int EPC = 16 / (int)sizeof(T);
  using typename GateBnBwd<T, TWO, KEEP>::In;
  __device__ void load_cols(int c0) { this->load_common(c0); }
  __device__ void finish(const In& in, size_t, int, Acc (&acc)[NQ][EPC]) const {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float p = this->act(in, e);
      const float dp = this->dpsi(in, e, p);
      acc[0][e] += dp;
      acc[1][e] += dp * (to_f32<T>(in.gv.v[e]) - this->mg[e]) * this->ig[e];
      if constexpr (TWO) acc[2][e] += dp * (to_f32<T>(in.xv.v[e]) - this->mx[e]) * this->ix[e];
      acc[3][e] += in.d * p;
      acc[4][e] += in.d;
    }
  }
};

template <typename T, bool TWO, bool KEEP = true> struct GateBnBwdApplyOp : GateBnBwd<T, TWO, KEEP> {
  static constexpr int MAX_WGS = 512;         // (256 / 512 / 768 workgroups = 5.3 / 5.9 / 5.9 TB/s)
  static constexpr int NQ = 1;
  static constexpr bool WRITES = true;
  typedef float Acc;
  static constexpr int EPC = 16 / (int)sizeof(T);
  using typename GateBnBwd<T, TWO, KEEP>::In;
  const float* gamma_g; const float* gamma_x; const float* sums_g; const float* sums_x;
  T* dg; int lddg; T* dx; int lddx; float invM; int C;
  float kg0[EPC], kg1[EPC], gg[EPC], kx0[EPC], kx1[EPC], gx[EPC];
  __device__ void load_cols(int c0) {
    this->load_common(c0);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      gg[e] = gamma_g[c0 + e] * this->ig[e]; kg0[e] = sums_g[c0 + e] * invM; kg1[e] = sums_g[C + c0 + e] * invM;
      if constexpr (TWO) { gx[e] = gamma_x[c0 + e] * this->ix[e]; kx0[e] = sums_x[c0 + e] * invM; kx1[e] = sums_x[C + c0 + e] * invM; }
    }
  }
  __device__ void finish(const In& in, size_t row, int c0, Acc (&)[NQ][EPC]) const {
    Vec16<T> og, ox;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float dp = this->dpsi(in, e, this->act(in, e));
      const float hg = (to_f32<T>(in.gv.v[e]) - this->mg[e]) * this->ig[e];
      og.v[e] = from_f32<T>(bn_dx(gg[e], dp, kg0[e], hg, kg1[e]));
      if constexpr (TWO) {
        const float hx = (to_f32<T>(in.xv.v[e]) - this->mx[e]) * this->ix[e];
        ox.v[e] = from_f32<T>(bn_dx(gx[e], dp, kx0[e], hx, kx1[e]));
      }
    }
    st16<T>(dg + row * lddg + c0, og);
    if constexpr (TWO) st16<T>(dx + row * lddx + c0, ox);
  }
};

extern "C" int mi355_gate_bn_bwd_reduce_rows(long long M) { return rowred_grid<GateBnBwdReduceOp<bf16_t, true>>(M); }

template <typename T, typename Op> static void fill_gate_bn(Op& op, const float* dz, const void* g1, int ldg, const void* x1, int ldx,
                                                            const float* const* co, const float* w) {
  op.dz = dz; op.g1 = (const T*)g1; op.ldg = ldg; op.x1 = (const T*)x1; op.ldx = ldx;
  op.scale_g = co[0]; op.shift_g = co[1]; op.mean_g = co[2]; op.invstd_g = co[3];
  op.scale_x = co[4]; op.shift_x = co[5]; op.mean_x = co[6]; op.invstd_x = co[7];
  op.w = w;
}

extern "C" int mi355_gate_bn_bwd_reduce(const float* dz, const void* g1, int ldg, const void* x1, int ldx, const float* scale_g,
                                        const float* shift_g, const float* mean_g, const float* invstd_g, const float* scale_x,
                                        const float* shift_x, const float* mean_x, const float* invstd_x, const float* w,
                                        float* partial, long long M, int C, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(dz && g1 && scale_g && shift_g && mean_g && invstd_g && w && partial && (!x1 || (scale_x && shift_x && mean_x && invstd_x)),
                  "gate_bn_bwd_reduce: null pointer");
  const float* co[8] = {scale_g, shift_g, mean_g, invstd_g, scale_x, shift_x, mean_x, invstd_x};
  return dispatch_dtype(dtype, "gate_bn_bwd_reduce", [&](auto tag) {
    using T = decltype(tag);
    static const int nt = getenv("MI355_BN_REDUCE_NT") ? atoi(getenv("MI355_BN_REDUCE_NT")) : 0;      // (see bn.hip: bn_reduce_keeps)
    auto run = [&](auto op) {
      fill_gate_bn<T>(op, dz, g1, ldg, x1, x1 ? ldx : 0, co, w);
      return rowred_launch<T>(op, M, C, partial, (hipStream_t)s);
    };
    if (x1) return nt ? run(GateBnBwdReduceOp<T, true, false>{}) : run(GateBnBwdReduceOp<T, true, true>{});
    // one normalised operand (quantity 2 of the partial rows stays zero)
    return nt ? run(GateBnBwdReduceOp<T, false, false>{}) : run(GateBnBwdReduceOp<T, false, true>{});
  });
}

extern "C" int mi355_gate_bn_bwd_apply(const float* dz, const void* g1, int ldg, const void* x1, int ldx, const float* scale_g,
                                       const float* shift_g, const float* mean_g, const float* invstd_g, const float* scale_x,
                                       const float* shift_x, const float* mean_x, const float* invstd_x, const float* w,
                                       const float* gamma_g, const float* gamma_x, const float* sums_g, const float* sums_x,
                                       void* dg1, int lddg, void* dx1, int lddx, long long M, int C, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(dz && g1 && scale_g && shift_g && mean_g && invstd_g && w && gamma_g && sums_g && dg1 &&
                  (!x1 || (scale_x && shift_x && mean_x && invstd_x && gamma_x && sums_x && dx1)), "gate_bn_bwd_apply: null pointer");
  const float* co[8] = {scale_g, shift_g, mean_g, invstd_g, scale_x, shift_x, mean_x, invstd_x};
  return dispatch_dtype(dtype, "gate_bn_bwd_apply", [&](auto tag) {
    using T = decltype(tag);
    static const int apply_nt = getenv("MI355_BN_APPLY_NT") ? atoi(getenv("MI355_BN_APPLY_NT")) : 0;      // (see bn.hip: bn_apply_keeps)
    auto run = [&](auto op) {
      fill_gate_bn<T>(op, dz, g1, ldg, x1, ldx, co, w);
      op.gamma_g = gamma_g; op.gamma_x = gamma_x; op.sums_g = sums_g; op.sums_x = sums_x;
      op.dg = (T*)dg1; op.lddg = lddg; op.dx = (T*)dx1; op.lddx = lddx; op.invM = (float)(1.0 / (double)M); op.C = C;
      return rowred_launch<T>(op, M, C, nullptr, (hipStream_t)s);
    };
    if (x1) return apply_nt ? run(GateBnBwdApplyOp<T, true, false>{}) : run(GateBnBwdApplyOp<T, true, true>{});
    return apply_nt ? run(GateBnBwdApplyOp<T, false, false>{}) : run(GateBnBwdApplyOp<T, false, true>{});
  });
}

// ---- x * sigmoid(bn1(z)) ---------------------------------------------------------------------------------
template <typename T> struct GateMulOp {
  static constexpr int EPC = 16 / (int)sizeof(T);
  const T* x; int ldx; const float* z; const float* scale; const float* shift; T* y; int ldy;
  __device__ void load_cols(int) {}
  __device__ void apply(size_t row, int c0) const {
    const float psi = 1.f / (1.f + __expf(-(z[row] * scale[0] + shift[0])));
    Vec16<T> v = ld16<T>(x + row * ldx + c0);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v.v[e] = from_f32<T>(to_f32<T>(v.v[e]) * psi);
    st16<T>(y + row * ldy + c0, v);
  }
};

extern "C" int mi355_gate_mul_fwd(const void* x, int ldx, const float* z, const float* scale, const float* shift, void* y,
                                  int ldy, long long M, int C, int dtype, mi355_stream_t s) {
  MI355_CHECK_ARG(x && z && scale && shift && y, "gate_mul_fwd: null pointer");
  return dispatch_dtype(dtype, "gate_mul_fwd", [&](auto tag) {
    using T = decltype(tag);
    GateMulOp<T> op{(const T*)x, ldx, z, scale, shift, (T*)y, ldy};
    return rowmap_launch<T>(op, M, C, (hipStream_t)s);
  });
}

template <typename T>
__global__ __launch_bounds__(256) void gate_mul_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                           const float* __restrict__ z, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, T* __restrict__ dx, int lddx,
                                                           int accumulate, float* __restrict__ dzn,
                                                           float* __restrict__ partial, long long M, int C, int rows_per_block,
                                                           int tpr) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int cp = C / EPC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % tpr, rsub = lane / tpr, rpw = 64 / tpr;
  const float sc = scale[0], sh = shift[0], mu = mean[0], is = invstd[0];
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(M, r0 + rows_per_block);
  double s0 = 0, s1 = 0;
  if (cp <= tpr) {
    // one chunk per lane (C <= 512 in the 2-byte types): U row groups in flight, unconditional loads from a clamped (row, chunk) —
    // a guarded load is waited for before the next is issued (gate_psi_fwd_kernel) —, lanes past the last chunk write nothing
    constexpr int U = 4;
    const bool on = sub < cp;
    const int cb = (on ? sub : 0) * EPC;
    for (long long base = r0 + wave * rpw; base < r1; base += (long long)U * 4 * rpw) {
      Vec16<T> gv[U], xv[U], ov[U];
      float zz[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long r = base + (long long)u * 4 * rpw + rsub;
        const size_t rc = (size_t)(r < r1 ? r : r1 - 1);
        gv[u] = ld16<T>(dy + rc * lddy + cb);
        xv[u] = ld16<T>(x + rc * ldx + cb);
        if (accumulate) ov[u] = ld16_plain<T>(dx + rc * lddx + cb);
        zz[u] = z[rc];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long r = base + (long long)u * 4 * rpw + rsub;
        const float psi = 1.f / (1.f + __expf(-(zz[u] * sc + sh)));
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float gg = to_f32<T>(gv[u].v[e]);
          acc += gg * to_f32<T>(xv[u].v[e]);
          const float d = gg * psi;
          ov[u].v[e] = from_f32<T>(accumulate ? to_f32<T>(ov[u].v[e]) + d : d);
        }
        if (!on) acc = 0.f;
        if (r < r1 && on) st16<T>(dx + (size_t)r * lddx + cb, ov[u]);
        acc = seg_sum(acc, tpr);
        if (r < r1 && sub == 0) {
          const float d = acc * psi * (1.f - psi);
          dzn[r] = d;
          s0 += d;
          s1 += (double)d * ((zz[u] - mu) * is);
        }
      }
    }
    block_pair_sum(s0, s1, partial);
    return;
  }
  for (long long base = r0 + wave * rpw; base < r1; base += 4 * rpw) {
    const long long r = base + rsub;
    float acc = 0.f, psi = 0.f, zz = 0.f;
    if (r < r1) {
      zz = z[r];
      psi = 1.f / (1.f + __expf(-(zz * sc + sh)));
      for (int ck = sub; ck < cp; ck += tpr) {
        const Vec16<T> g = ld16<T>(dy + (size_t)r * lddy + ck * EPC);
        const Vec16<T> xv = ld16<T>(x + (size_t)r * ldx + ck * EPC);
        T* o = dx + (size_t)r * lddx + ck * EPC;
        Vec16<T> ov;
        if (accumulate) ov = ld16<T>(o);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float gg = to_f32<T>(g.v[e]);
          acc += gg * to_f32<T>(xv.v[e]);
          const float d = gg * psi;
          ov.v[e] = from_f32<T>(accumulate ? to_f32<T>(ov.v[e]) + d : d);
        }
        st16<T>(o, ov);
      }
    }
    acc = seg_sum(acc, tpr);
    if (r < r1 && sub == 0) {
      const float d = acc * psi * (1.f - psi);
      dzn[r] = d;
      s0 += d;
      s1 += (double)d * ((zz - mu) * is);
    }
  }
  block_pair_sum(s0, s1, partial);
}

extern "C" int mi355_gate_mul_bwd(const void* dy, int lddy, const void* x, int ldx, const float* z, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, void* dx, int lddx,
                                  int accumulate, float* dzn, float* partial, long long M, int C, int dtype,
                                  mi355_stream_t s) {
  MI355_CHECK_ARG(dy && x && z && scale && shift && mean && invstd && dx && dzn && partial, "gate_mul_bwd: null pointer");
  const int epc = dtype_is_2byte(dtype) ? 8 : 4;
  MI355_CHECK_ARG(C % epc == 0, "gate_mul_bwd: C=%d must be a multiple of %d", C, epc);
  const int nb = rowreduce_blocks(M);
  const int rpb = (int)((M + nb - 1) / nb);
  const int tpr = pow2_tpr(C / epc);
  return dispatch_dtype(dtype, "gate_mul_bwd", [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((gate_mul_bwd_kernel<T>), dim3(nb), dim3(256), 0, (hipStream_t)s, (const T*)dy, lddy, (const T*)x, ldx, z, scale,
                       shift, mean, invstd, (T*)dx, lddx, accumulate, dzn, partial, M, C, rpb, tpr);
    MI355_LAUNCH_CHECK();
    return (int)MI355_OK;
  });
}

__global__ void bn1_bwd_apply_kernel(const float* __restrict__ dzn, const float* __restrict__ z, const float* gamma,
                                     const float* mean, const float* invstd, const float* sums, float* __restrict__ dz,
                                     long long M) {
  const float is = invstd[0], mu = mean[0], gi = gamma[0] * is;
  const float k0 = sums[0] / (float)M, k1 = sums[1] / (float)M;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x)
    dz[i] = gi * (dzn[i] - k0 - (z[i] - mu) * is * k1);
}

extern "C" int mi355_bn1_bwd_apply(const float* dzn, const float* z, const float* gamma, const float* mean,
                                   const float* invstd, const float* sums, float* dz, long long M, mi355_stream_t s) {
  MI355_CHECK_ARG(dzn && z && gamma && mean && invstd && sums && dz, "bn1_bwd_apply: null pointer");
  long long blocks = (M + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn1_bwd_apply_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, dzn, z, gamma, mean, invstd, sums, dz, M);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}
