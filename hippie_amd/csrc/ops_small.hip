// HBM/L2-bound and latency-bound ops of the cVAE step: BatchNorm apply / backward,
// stem and tail convolutions (C_in = 1 / C_out = 1), pooling, the small Linear heads,
// reparameterisation + KL, MSE, AdamW.  Semantics: include/hippie_hip.h.
//
// Shared thread mapping ("column-fixed"): a 256-thread block covers CW = min(C,256)
// channels x RL = 256/CW row lanes; a thread keeps ONE channel for its whole life, so
// per-channel coefficients live in registers and per-channel sums need one LDS fold and
// one fp64 atomic per block.  Consecutive lanes touch consecutive channels (coalesced).
#include "hp_common.h"
#include "linear_mfma.h"
#include "heads_fused.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int kRowsPerLane = 16;

struct ColMap {
  int c, row, rstep, rend;
  int cw, rl, rlane;
  bool active;
};

__device__ __forceinline__ ColMap colmap(int M, int C, int rpl = kRowsPerLane) {
  ColMap m;
  m.cw = C < 256 ? C : 256;
  m.rl = 256 / m.cw;
  const int tid = threadIdx.x;
  m.rlane = tid / m.cw;
  m.c = blockIdx.y * m.cw + (tid - m.rlane * m.cw);
  const int rpb = m.rl * rpl;
  m.row = blockIdx.x * rpb + m.rlane;
  m.rstep = m.rl;
  m.rend = min(M, (int)(blockIdx.x + 1) * rpb);
  m.active = (m.rlane < m.rl) && (m.c < C);
  return m;
}

inline dim3 colgrid(int M, int C, int rpl = kRowsPerLane) {
  const int cw = C < 256 ? C : 256;
  const int rpb = (256 / cw) * rpl;
  return dim3(hp::cdiv(M, rpb), hp::cdiv(C, cw));
}

// fold per-thread partials over the row lanes of a block; returns the total in row lane 0
template <int NV>
__device__ __forceinline__ void fold_rowlanes(const ColMap& m, double (&v)[NV], double* lds /* [NV][256] */) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) lds[k * 256 + threadIdx.x] = m.active ? v[k] : 0.0;
  __syncthreads();
  if (m.active && m.rlane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double s = 0.0;
      for (int r = 0; r < m.rl; ++r) s += lds[k * 256 + r * m.cw + (threadIdx.x)];
      v[k] = s;
    }
  }
}

// ---- BN apply --------------------------------------------------------------------
// V = 4: each thread owns 4 consecutive channels (one 16-byte access per row and tensor);
// V = 1: scalar fallback for the heads' odd widths (C = 5, 10, ...).
template <int V> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };
template <int V> __device__ __forceinline__ float& at(typename Vec<V>::T& v, int j);
template <> __device__ __forceinline__ float& at<4>(float4& v, int j) { return (&v.x)[j]; }
template <> __device__ __forceinline__ float& at<1>(float& v, int) { return v; }
// V consecutive elements of an activation tensor stored fp32 (H = false) or bf16 (H = true: HP_FLAG_ACT_BF16; only V = 4)
template <int V, bool H> __device__ __forceinline__ typename Vec<V>::T ldv(const float* base, const size_t idx);
template <> __device__ __forceinline__ float4 ldv<4, false>(const float* base, const size_t idx) { return *reinterpret_cast<const float4*>(base + idx); }
template <> __device__ __forceinline__ float4 ldv<4, true>(const float* base, const size_t idx) { return aload4<true>(base, idx); }
template <> __device__ __forceinline__ float ldv<1, false>(const float* base, const size_t idx) { return base[idx]; }
template <> __device__ __forceinline__ float ldv<1, true>(const float* base, const size_t idx) { return aload1<true>(base, idx); }
template <int V, bool H> __device__ __forceinline__ void stv(float* base, const size_t idx, const typename Vec<V>::T v);
template <> __device__ __forceinline__ void stv<4, false>(float* base, const size_t idx, const float4 v) { *reinterpret_cast<float4*>(base + idx) = v; }
template <> __device__ __forceinline__ void stv<4, true>(float* base, const size_t idx, const float4 v) { astore4<true>(base, idx, v); }
template <> __device__ __forceinline__ void stv<1, false>(float* base, const size_t idx, const float v) { base[idx] = v; }
template <> __device__ __forceinline__ void stv<1, true>(float* base, const size_t idx, const float v) { astore1<true>(base, idx, v); }
// run-time form for the scalar kernels at the backbones' edges (stem, pool, repeat, tail): `h` is uniform
__device__ __forceinline__ float ald(const float* base, const size_t idx, const int h) { return h ? aload1<true>(base, idx) : base[idx]; }
__device__ __forceinline__ void ast(float* base, const size_t idx, const float v, const int h) { if (h) astore1<true>(base, idx, v); else base[idx] = v; }
__device__ __forceinline__ float4 ald4(const float* base, const size_t idx, const int h) { return h ? aload4<true>(base, idx) : *reinterpret_cast<const float4*>(base + idx); }

template <int V>
__device__ __forceinline__ ColMap colmap_v(int M, int C, int rpl, int bx, int by) {
  ColMap m;
  const int cg = C / V;
  m.cw = cg < 256 ? cg : 256;
  m.rl = 256 / m.cw;
  const int tid = threadIdx.x;
  m.rlane = tid / m.cw;
  m.c = (by * m.cw + (tid - m.rlane * m.cw)) * V;
  const int rpb = m.rl * rpl;
  m.row = bx * rpb + m.rlane;
  m.rstep = m.rl;
  m.rend = min(M, (bx + 1) * rpb);
  m.active = (m.rlane < m.rl) && (m.c < C);
  return m;
}
template <int V>
inline dim3 colgrid_v(int M, int C, int rpl) {
  const int cg = C / V;
  const int cw = cg < 256 ? cg : 256;
  return dim3(hp::cdiv(M, (256 / cw) * rpl), hp::cdiv(cg, cw));
}
#ifndef HP_BN_ROWS
#define HP_BN_ROWS 4       // rows per thread of the float4 BatchNorm passes (tools/micro/bn_sweep.py builds 1 / 2 / 4 / 8)
#endif
constexpr int kBnRows = HP_BN_ROWS;
// rows per thread of a BatchNorm pass over an [M][C] tensor: the float4 form takes exactly kBnRows; the scalar form (C not a
// multiple of 4: the heads' z_dim-wide BatchNorms) up to kRowsPerLane, fewer on short tensors so that ~64 workgroups exist
// (a [512][10] tensor used to be 2 workgroups walking 16 dependent rows per thread)
template <int V> __host__ __device__ inline int bn_rows(int M, int C) {
  if (V == 4) return kBnRows;
  const int cw = C < 256 ? C : 256, rl = 256 / cw;
  const int want = (M + rl * 64 - 1) / (rl * 64);
  return want < 1 ? 1 : (want > kRowsPerLane ? kRowsPerLane : want);
}

// fold NV partials per thread over the row lanes (thread index layout of colmap_v): every (statistic, column) pair is
// summed by a thread of its own, in row-lane order (a column group of 5 threads used to walk 12 x 51 LDS values each)
template <int NV>
__device__ __forceinline__ void fold_rows(const ColMap& m, double (&v)[NV], double* lds /* [NV][256] */) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) lds[k * 256 + threadIdx.x] = m.active ? v[k] : 0.0;
  __syncthreads();
  for (int t = threadIdx.x; t < NV * m.cw; t += 256) {
    const int k = t / m.cw, cl = t - k * m.cw;
    double s = 0.0;
    for (int r = 0; r < m.rl; ++r) s += lds[k * 256 + r * m.cw + cl];
    lds[k * 256 + cl] = s;          // row lane 0's slot: nobody else reads it
  }
  __syncthreads();
  if (m.active && m.rlane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = lds[k * 256 + threadIdx.x];
  }
}

struct BnApplyArgs {
  const float* raw; float* out; const double* stats; const float* gamma; const float* beta;
  float* rmean; float* rvar; float* save;
  const float* res; const double* stats2; const float* gamma2; const float* beta2;
  float* rmean2; float* rvar2; float* save2;
  int M, C, res_mode, training, act;
  int Mstat;          // rows behind the statistics: M, or M * world under sync-BatchNorm (HP_OP_STATS_SYNC)
  float slope, eps, momentum;
  int abf;            // HP_FLAG_ACT_BF16: RAW, OUT, RES hold bf16
};

#ifdef HP_BN_TS       // timing experiment (tools/micro/bn_phases.py): 100 MHz timestamps of block 0 / the last block into SAVE2
#define BN_TS(k) if (threadIdx.x == 0 && p.save2 != nullptr && by == 0) { unsigned long long* q_ = (unsigned long long*)p.save2; \
    if (bx == 0) q_[k] = wall_clock64(); else if (bx == (int)gridDim.x - 1) q_[8 + (k)] = wall_clock64(); }
#else
#define BN_TS(k)
#endif
// RES (= BnApplyArgs::res_mode) is a template parameter: with the residual forms as run-time branches the compiler
// re-read the coefficients from LDS for every row and waited for each row's STORE before touching the next row
// (vmcnt(0) per row: 1.4 us of serialised store latency per launch, tools/micro/bn_phases.py).
template <int V, int RES, bool H>
__device__ __forceinline__ void bn_apply_impl(const BnApplyArgs& p, const int bx, const int by, float (*s_coef)[256 * V]) {
  using T = typename Vec<V>::T;
  BN_TS(0)
  const ColMap m = colmap_v<V>(p.M, p.C, bn_rows<V>(p.M, p.C), bx, by);
  // The kernel is latency-bound (one round trip for the statistics, one for the rows): issue the row loads first so
  // both round trips overlap.  UNCONDITIONAL loads (row / column clamped into the tensor; only the stores are
  // guarded): every row of the thread is in flight at once.  (V = 4: exactly kBnRows rows per thread.)
  constexpr int NR = V == 4 ? kBnRows : 1;
  T xs[NR], rs[NR];
  if (V == 4) {
    const int cc = m.active ? m.c : 0;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = min(m.row + k * m.rstep, p.M - 1);
      const size_t idx = (size_t)r * p.C + cc;
      xs[k] = ldv<V, H>(p.raw, idx);
      if (RES != 0) rs[k] = ldv<V, H>(p.res, idx);
    }
  }
  // each channel's (replicated) statistics are summed ONCE per block, not once per thread
  for (int ci = threadIdx.x; ci < m.cw * V; ci += 256) {
    const int c = by * m.cw * V + ci;
    if (c >= p.C) continue;
    const BnCoef k = bn_coef(p.training, p.Mstat, p.stats, p.C, c, p.gamma, p.beta, p.rmean, p.rvar, p.eps);
    s_coef[0][ci] = k.scale; s_coef[1][ci] = k.shift;
    BnCoef k2 = k;
    if (RES == 2) {
      k2 = bn_coef(p.training, p.Mstat, p.stats2, p.C, c, p.gamma2, p.beta2, p.rmean2, p.rvar2, p.eps);
      s_coef[2][ci] = k2.scale; s_coef[3][ci] = k2.shift;
    }
    if (p.training && bx == 0) {
      bn_side_effects(k, p.Mstat, p.C, c, p.save, p.rmean, p.rvar, p.momentum);
      if (RES == 2) bn_side_effects(k2, p.Mstat, p.C, c, p.save2, p.rmean2, p.rvar2, p.momentum);
    }
  }
  BN_TS(1)
  __syncthreads();
  BN_TS(2)
  if (!m.active) return;
  const int ci0 = (threadIdx.x - m.rlane * m.cw) * V;
  float sc[V], sh[V], sc2[V], sh2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    sc[j] = s_coef[0][ci0 + j]; sh[j] = s_coef[1][ci0 + j];
    sc2[j] = RES == 2 ? s_coef[2][ci0 + j] : 0.f; sh2[j] = RES == 2 ? s_coef[3][ci0 + j] : 0.f;
  }
  const bool act = p.act;
  const float slope = p.slope;
  auto value = [&](T x, T rsd) -> T {
    T o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float v = fmaf(at<V>(x, j), sc[j], sh[j]);
      if (RES == 1) v += at<V>(rsd, j);
      else if (RES == 2) v += fmaf(at<V>(rsd, j), sc2[j], sh2[j]);
      at<V>(o, j) = act ? lrelu(v, slope) : v;
    }
    return o;
  };
  if (V == 4) {
    T os[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) os[k] = value(xs[k], rs[k]);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = m.row + k * m.rstep;
      if (r < m.rend) stv<V, H>(p.out, (size_t)r * p.C + m.c, os[k]);
    }
    BN_TS(3)
    return;
  }
  for (int r = m.row; r < m.rend; r += m.rstep) {
    const size_t idx = (size_t)r * p.C + m.c;
    T x = ldv<V, H>(p.raw, idx);
    T rsd = x;
    if (RES != 0) rsd = ldv<V, H>(p.res, idx);
    stv<V, H>(p.out, idx, value(x, rsd));
  }
}
template <int V>
__device__ __forceinline__ void bn_apply_body(const BnApplyArgs& p, const int bx, const int by, float (*s_coef)[256 * V]) {
  if (V == 4 && p.abf) {          // (uniform) activations stored as bf16: HP_FLAG_ACT_BF16
    if (p.res_mode == 0) bn_apply_impl<V, 0, V == 4>(p, bx, by, s_coef);
    else if (p.res_mode == 1) bn_apply_impl<V, 1, V == 4>(p, bx, by, s_coef);
    else bn_apply_impl<V, 2, V == 4>(p, bx, by, s_coef);
    return;
  }
  if (p.res_mode == 0) bn_apply_impl<V, 0, false>(p, bx, by, s_coef);
  else if (p.res_mode == 1) bn_apply_impl<V, 1, false>(p, bx, by, s_coef);
  else bn_apply_impl<V, 2, false>(p, bx, by, s_coef);
}
// single and paired launch forms (HP_OP_PAIR: two independent ops, one launch, flattened 2-D grids)
#define HP_BN_KERNELS(NAME, ARGS, SHARED_DECL, SHARED_ARG)                                                        \
  template <int V> __global__ __launch_bounds__(256) void NAME##_kernel(ARGS p) {                                 \
    SHARED_DECL;                                                                                                  \
    NAME##_body<V>(p, blockIdx.x, blockIdx.y, SHARED_ARG);                                                        \
  }                                                                                                               \
  template <int V> __global__ __launch_bounds__(256) void NAME##_pair_kernel(ARGS a, ARGS b, int gxa, int na, int gxb) { \
    SHARED_DECL;                                                                                                  \
    int id = blockIdx.x;                                                                                          \
    if (id < na) NAME##_body<V>(a, id % gxa, id / gxa, SHARED_ARG);                                               \
    else { id -= na; NAME##_body<V>(b, id % gxb, id / gxb, SHARED_ARG); }                                         \
  }
HP_BN_KERNELS(bn_apply, BnApplyArgs, __shared__ float s_coef[4][256 * V], s_coef)

// ---- BN backward -----------------------------------------------------------------
struct BnBwdReduceArgs {
  const float* g1; const float* g2; const float* act; float* gout;
  const float* raw; const float* save; double* bs;
  const float* raw2; const float* save2; double* bs2;
  const float* coef;     // act == nullptr: the activation was never stored; its sign is that of fma(raw, scale, shift)
  int M, C, has_second;
  int rpl;               // rows per thread (bn_red_rpl): sets the grid and the number of atomic adds per statistics replica
  float slope;
  int abf;               // HP_FLAG_ACT_BF16: G1, G2, ACT, GOUT, RAW, RAW2 hold bf16
};
// Rows per thread of HP_OP_BN_BWD_REDUCE.  Every workgroup ends in one fp64 atomic per column and statistic; with
// 8 rows per workgroup a [2048][512] tensor sent 128 adds to each replica address (2 replicas at C = 512) and the
// kernel took 16.8 us against 7.6 us at C = 64.  Taller workgroups (up to 8 batches of rows per thread) until at most
// ~32 workgroups share a replica, as long as at least 64 workgroups remain.
inline int bn_red_rpl(int M, int C) {
  if (C % 4 != 0) return bn_rows<1>(M, C);
  const int cg = C / 4, cw = cg < 256 ? cg : 256, rl = 256 / cw, groups = hp::cdiv(cg, cw);
  const int R = hp_stat_repl(C);
  int nb = 1;
  while (nb < 8) {
    if (hp::cdiv(M, rl * kBnRows * nb) <= 32 * R) break;
    if (hp::cdiv(M, rl * kBnRows * nb * 2) * groups < 64) break;
    nb *= 2;
  }
  return kBnRows * nb;
}

template <int V, bool H>
__device__ __forceinline__ void bn_bwd_reduce_impl(const BnBwdReduceArgs& p, const int bx, const int by, double* lds) {
  using T = typename Vec<V>::T;
  const ColMap m = colmap_v<V>(p.M, p.C, p.rpl, bx, by);
  double v[3 * V];
#pragma unroll
  for (int j = 0; j < 3 * V; ++j) v[j] = 0.0;
  if (m.active) {
    float mean[V], invstd[V], mean2[V], invstd2[V], csc[V], csh[V];
    const bool from_raw = p.act == nullptr;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      mean[j] = p.save[m.c + j]; invstd[j] = p.save[p.C + m.c + j];
      mean2[j] = 0.f; invstd2[j] = 0.f; csc[j] = 0.f; csh[j] = 0.f;
      if (p.has_second) { mean2[j] = p.save2[m.c + j]; invstd2[j] = p.save2[p.C + m.c + j]; }
      if (from_raw) { csc[j] = p.coef[m.c + j]; csh[j] = p.coef[p.C + m.c + j]; }
    }
    auto accumulate = [&](T g, T a, T x, T gg, T x2, size_t idx) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float gv = at<V>(g, j);
        if (p.g2 != nullptr) gv += at<V>(gg, j);
        gv *= lrelu_grad(from_raw ? fmaf(at<V>(x, j), csc[j], csh[j]) : at<V>(a, j), p.slope);
        at<V>(g, j) = gv;
        v[3 * j + 0] += (double)gv;
        v[3 * j + 1] += (double)gv * (double)((at<V>(x, j) - mean[j]) * invstd[j]);
        if (p.has_second) v[3 * j + 2] += (double)gv * (double)((at<V>(x2, j) - mean2[j]) * invstd2[j]);
      }
      stv<V, H>(p.gout, idx, g);
    };
    if (V == 4) {
      // batches of kBnRows rows per thread (p.rpl / kBnRows of them): all loads of a batch in flight before the first use
      constexpr int NR = kBnRows;
      T g[NR], a[NR], x[NR], gg[NR], x2[NR];
      // unconditional loads from a clamped row (so that they form one straight-line batch), then a scheduling
      // barrier: otherwise the compiler folds each row's mask computation into its load block and the four
      // rows become four dependent round trips
      const int rlast = m.rend - 1;
      for (int r0 = m.row; r0 <= rlast; r0 += NR * m.rstep) {
        size_t idx[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          idx[k] = (size_t)min(r0 + k * m.rstep, rlast) * p.C + m.c;
          g[k] = ldv<V, H>(p.g1, idx[k]);
          x[k] = ldv<V, H>(p.raw, idx[k]);
        }
        if (!from_raw) {
#pragma unroll
          for (int k = 0; k < NR; ++k) a[k] = ldv<V, H>(p.act, idx[k]);
        }
        if (p.g2 != nullptr) {
#pragma unroll
          for (int k = 0; k < NR; ++k) gg[k] = ldv<V, H>(p.g2, idx[k]);
        }
        if (p.has_second) {
#pragma unroll
          for (int k = 0; k < NR; ++k) x2[k] = ldv<V, H>(p.raw2, idx[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          const int r = r0 + k * m.rstep;
          if (r <= rlast) accumulate(g[k], a[k], x[k], gg[k], x2[k], (size_t)r * p.C + m.c);
        }
      }
    } else {
      for (int r = m.row; r < m.rend; r += m.rstep) {
        const size_t idx = (size_t)r * p.C + m.c;
        T g = ldv<V, H>(p.g1, idx);
        T x = ldv<V, H>(p.raw, idx);
        T a = x;
        if (!from_raw) a = ldv<V, H>(p.act, idx);
        T gg = g, x2 = x;
        if (p.g2 != nullptr) gg = ldv<V, H>(p.g2, idx);
        if (p.has_second) x2 = ldv<V, H>(p.raw2, idx);
        accumulate(g, a, x, gg, x2, idx);
      }
    }
  }
  fold_rows<3 * V>(m, v, lds);
  // lds[(3 * j + s) * 256 + q] = statistic s of channel 4q + j of this workgroup's column group.  One atomic per channel and
  // statistic from CONSECUTIVE threads on consecutive addresses (an atomic instruction costs per cache line it touches;
  // thread q adding its own four channels spread every instruction over cw / 4 lines)
  const int ncol = m.cw * V, c_base = by * ncol;
  for (int o = threadIdx.x; o < ncol; o += 256) {
    const int q = o / V, j = o - q * V, c = c_base + o;
    if (c >= p.C) continue;
    const double s0 = lds[(3 * j + 0) * 256 + q], s1 = lds[(3 * j + 1) * 256 + q];
    double* b1 = stat_replica(p.bs, p.C, bx);
    atomic_add_f64(b1 + c, s0);
    atomic_add_f64(b1 + p.C + c, s1);
    if (p.has_second) {
      double* b2 = stat_replica(p.bs2, p.C, bx);
      atomic_add_f64(b2 + c, s0);
      atomic_add_f64(b2 + p.C + c, lds[(3 * j + 2) * 256 + q]);
    }
  }
}

template <int V>
__device__ __forceinline__ void bn_bwd_reduce_body(const BnBwdReduceArgs& p, const int bx, const int by, double* lds) {
  if (V == 4 && p.abf) bn_bwd_reduce_impl<V, V == 4>(p, bx, by, lds);      // (uniform) HP_FLAG_ACT_BF16
  else bn_bwd_reduce_impl<V, false>(p, bx, by, lds);
}
HP_BN_KERNELS(bn_bwd_reduce, BnBwdReduceArgs, __shared__ double lds[3 * V * 256], lds)

struct BnBwdApplyArgs {
  const float* g; const float* raw; const float* save; const double* bs; const float* gamma;
  float* dr; float* dgamma; float* dbeta;
  int M, C;
  int Mstat;          // M * world under sync-BatchNorm: BS then holds the sums over all ranks
  float gscale;       // 1 / world: dgamma / dbeta are written so that the data-parallel MEAN of the ranks gives the sum
  int abf;            // HP_FLAG_ACT_BF16: G, RAW, DR hold bf16
};

template <int V, bool H>
__device__ __forceinline__ void bn_bwd_apply_impl(const BnBwdApplyArgs& p, const int bx, const int by, float (*s_coef)[256 * V]) {
  using T = typename Vec<V>::T;
  const ColMap m = colmap_v<V>(p.M, p.C, bn_rows<V>(p.M, p.C), bx, by);
  constexpr int NR = V == 4 ? kBnRows : 1;          // row loads first: overlaps the statistics round trip (see bn_apply_body)
  T xs[NR], gs[NR];
  if (V == 4) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = m.row + k * m.rstep;
      if (m.active && r < m.rend) {
        const size_t idx = (size_t)r * p.C + m.c;
        xs[k] = ldv<V, H>(p.raw, idx);
        gs[k] = ldv<V, H>(p.g, idx);
      }
    }
  }
  for (int ci = threadIdx.x; ci < m.cw * V; ci += 256) {
    const int c = by * m.cw * V + ci;
    if (c >= p.C) continue;
    const float mean = p.save[c], invstd = p.save[p.C + c], gam = p.gamma[c];
    double sg, sgx;
    stat_sum2(p.bs, p.C, c, sg, sgx);
    const BnDrCoef k = bn_dr_coef(mean, invstd, gam, sg, sgx, p.Mstat);
    s_coef[0][ci] = k.A; s_coef[1][ci] = k.B; s_coef[2][ci] = k.C;
    if (bx == 0) { p.dgamma[c] = (float)(sgx * (double)p.gscale); p.dbeta[c] = (float)(sg * (double)p.gscale); }
  }
  __syncthreads();
  if (!m.active) return;
  const int ci0 = (threadIdx.x - m.rlane * m.cw) * V;
  float cA[V], cB[V], cC[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { cA[j] = s_coef[0][ci0 + j]; cB[j] = s_coef[1][ci0 + j]; cC[j] = s_coef[2][ci0 + j]; }
  auto apply = [&](T x, T g, size_t idx) {
    T o;
#pragma unroll
    for (int j = 0; j < V; ++j) at<V>(o, j) = bn_dr(at<V>(g, j), at<V>(x, j), cA[j], cB[j], cC[j]);
    stv<V, H>(p.dr, idx, o);
  };
  if (V == 4) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = m.row + k * m.rstep;
      if (r < m.rend) apply(xs[k], gs[k], (size_t)r * p.C + m.c);
    }
    return;
  }
  for (int r = m.row; r < m.rend; r += m.rstep) {
    const size_t idx = (size_t)r * p.C + m.c;
    apply(ldv<V, H>(p.raw, idx), ldv<V, H>(p.g, idx), idx);
  }
}

template <int V>
__device__ __forceinline__ void bn_bwd_apply_body(const BnBwdApplyArgs& p, const int bx, const int by, float (*s_coef)[256 * V]) {
  if (V == 4 && p.abf) bn_bwd_apply_impl<V, V == 4>(p, bx, by, s_coef);      // (uniform) HP_FLAG_ACT_BF16
  else bn_bwd_apply_impl<V, false>(p, bx, by, s_coef);
}
HP_BN_KERNELS(bn_bwd_apply, BnBwdApplyArgs, __shared__ float s_coef[5][256 * V], s_coef)   // mean, invstd, c1, c2, gamma*invstd

// ---- stem conv (C_in = 1) --------------------------------------------------------
struct StemArgs { const float* x; const float* w; float* out; double* stats; const float* dr; float* dw; int B, Lin, Lout, C; int abf; };

__global__ __launch_bounds__(256) void stem_fwd_kernel(StemArgs p) {
  __shared__ double lds[2 * 256];
  const int M = p.B * p.Lout;
  const ColMap m = colmap(M, p.C);
  double v[2] = {0.0, 0.0};
  if (m.active) {
    const float w0 = p.w[m.c * 3 + 0], w1 = p.w[m.c * 3 + 1], w2 = p.w[m.c * 3 + 2];
    for (int r = m.row; r < m.rend; r += m.rstep) {
      const int b = r / p.Lout, l = r - b * p.Lout;
      const float* xb = p.x + (size_t)b * p.Lin;
      const int j = 2 * l - 1;
      const float x0 = j >= 0 ? xb[j] : 0.f;
      const float x1 = xb[j + 1];
      const float x2 = (j + 2 < p.Lin) ? xb[j + 2] : 0.f;
      const float o = fmaf(x2, w2, fmaf(x1, w1, x0 * w0));
      ast(p.out, (size_t)r * p.C + m.c, o, p.abf);
      v[0] += (double)o;
      v[1] += (double)o * (double)o;
    }
  }
  fold_rowlanes<2>(m, v, lds);
  if (p.stats != nullptr && m.active && m.rlane == 0) {
    double* st = stat_replica(p.stats, p.C, blockIdx.x);
    atomic_add_f64(st + m.c, v[0]);
    atomic_add_f64(st + p.C + m.c, v[1]);
  }
}

// Rows per thread of the stem / tail weight-gradient reductions (batches of 8 loads); HIPPIE_WG_ROWS overrides it for
// tools/micro/small_wgrad_sweep.py.  Measured at batch 512 (time model's stem, 25 600 rows x 64 channels), per launch in a
// graph: 8 rows -> 60.8 us, 16 -> 33.2, 32 -> 21.0, 64 -> 18.4, 128 -> 23.9: every workgroup ends in one fp32 atomic per
// output and same-address atomics retire at ~70 ns each, so FEWER, taller workgroups win until the serial row batches
// take over.
// measurement knobs are honoured only under HIPPIE_DEBUG_KNOBS=1 (hippie_amd/program.py: debug_knob)
inline int debug_knob_int(const char* name) {
  const char* on = getenv("HIPPIE_DEBUG_KNOBS");
  if (on == nullptr || on[0] != '1') return 0;
  const char* e = getenv(name);
  return e ? atoi(e) : 0;
}
inline int small_wgrad_rows(int dflt) {
  static const int forced = debug_knob_int("HIPPIE_WG_ROWS");
  return forced > 0 ? forced : dflt;
}
constexpr int kStemRows = 64;
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemArgs p, int rows) {
  __shared__ double lds[3 * 256];
  const int M = p.B * p.Lout;
  const ColMap m = colmap(M, p.C, rows);      // many rows per block: the 3*C outputs are one atomic target per block
  double v[3] = {0.0, 0.0, 0.0};
  if (m.active) {
    // kRowsPerLane rows per thread in batches of 8: every load of a batch is issued before the first use
    for (int k0 = 0; k0 < rows; k0 += 8) {
      float d[8], x0[8], x1[8], x2[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = m.row + (k0 + k) * m.rstep;
        const bool ok = r < m.rend;
        const int rr = ok ? r : m.rend - 1;
        const int b = rr / p.Lout, l = rr - b * p.Lout;
        const float* xb = p.x + (size_t)b * p.Lin;
        const int j = 2 * l - 1;
        d[k] = ok ? ald(p.dr, (size_t)rr * p.C + m.c, p.abf) : 0.f;
        x0[k] = j >= 0 ? xb[j] : 0.f;
        x1[k] = xb[j + 1];
        x2[k] = j + 2 < p.Lin ? xb[j + 2] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[0] += (double)(d[k] * x0[k]);
        v[1] += (double)(d[k] * x1[k]);
        v[2] += (double)(d[k] * x2[k]);
      }
    }
  }
  fold_rowlanes<3>(m, v, lds);
  if (m.active && m.rlane == 0) {
    atomic_add_f32(p.dw + m.c * 3 + 0, (float)v[0]);
    atomic_add_f32(p.dw + m.c * 3 + 1, (float)v[1]);
    atomic_add_f32(p.dw + m.c * 3 + 2, (float)v[2]);
  }
}

// float4 form (C a multiple of 4): 16 channel quads x 16 row lanes per workgroup, 16 rows per thread in two batches of 8
// loads — the same 256 rows and the same number of atomics per workgroup as the scalar form's 4 row lanes x 64 rows, with
// two dependent memory round trips instead of eight (time model's stem at batch 512: 18.4 -> see tools/micro/small_wgrad_sweep.py)
constexpr int kStemRows4 = 16;
__global__ __launch_bounds__(256) void stem_wgrad4_kernel(StemArgs p, int rows) {
  __shared__ double lds[12 * 256];
  const int M = p.B * p.Lout;
  const ColMap m = colmap_v<4>(M, p.C, rows, blockIdx.x, blockIdx.y);
  double v[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) v[j] = 0.0;
  if (m.active) {
    for (int k0 = 0; k0 < rows; k0 += 8) {
      float4 d[8];
      float x0[8], x1[8], x2[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = m.row + (k0 + k) * m.rstep;
        const bool ok = r < m.rend;
        const int rr = ok ? r : m.rend - 1;
        const int b = rr / p.Lout, l = rr - b * p.Lout;
        const float* xb = p.x + (size_t)b * p.Lin;
        const int j = 2 * l - 1;
        d[k] = ald4(p.dr, (size_t)rr * p.C + m.c, p.abf);
        if (!ok) d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        x0[k] = j >= 0 ? xb[j] : 0.f;
        x1[k] = xb[j + 1];
        x2[k] = j + 2 < p.Lin ? xb[j + 2] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dj = (&d[k].x)[j];
          v[3 * j + 0] += (double)(dj * x0[k]);
          v[3 * j + 1] += (double)(dj * x1[k]);
          v[3 * j + 2] += (double)(dj * x2[k]);
        }
      }
    }
  }
  fold_rows<12>(m, v, lds);
  // lds[k * 256 + quad] (k = 3 * (channel & 3) + tap) holds the workgroup's sums.  One atomic per OUTPUT, issued by
  // consecutive threads on consecutive addresses: an atomic instruction costs per cache line it touches, and twelve
  // strided instructions from 16 threads touched 72 lines where 192 consecutive outputs are 6.
  const int c_base = blockIdx.y * m.cw * 4;
  for (int o = threadIdx.x; o < m.cw * 12; o += 256) {
    const int cl = o / 3, t = o - cl * 3;
    if (c_base + cl < p.C) atomic_add_f32(p.dw + (size_t)(c_base + cl) * 3 + t, (float)lds[(3 * (cl & 3) + t) * 256 + (cl >> 2)]);
  }
}

// ---- pool / repeat ---------------------------------------------------------------
__global__ void pool_fwd_kernel(const float* in, float* out, int B, int L, int C, int abf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * C) return;
  const int b = idx / C, c = idx - b * C;
  float s = 0.f;
  for (int l = 0; l < L; ++l) s += ald(in, ((size_t)b * L + l) * C + c, abf);
  out[idx] = s / (float)L;
}
__global__ void pool_bwd_kernel(const float* d, float* g, int B, int L, int C, int abf) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * L * C) return;
  const int c = idx % C;
  const int b = (idx / C) / L;
  ast(g, idx, d[(size_t)b * C + c] / (float)L, abf);
}
__global__ void repeat_fwd_kernel(const float* in, float* out, int B, int R, int C, int abf) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * R * C) return;
  const int c = idx % C;
  const int b = (idx / C) / R;
  ast(out, idx, in[(size_t)b * C + c], abf);
}
__global__ void repeat_bwd_kernel(const float* g1, const float* g2, float* d, int B, int R, int C, int abf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * C) return;
  const int b = idx / C, c = idx - b * C;
  float s = 0.f;
  for (int l = 0; l < R; ++l) {
    const size_t j = ((size_t)b * R + l) * C + c;
    s += ald(g1, j, abf);
    if (g2 != nullptr) s += ald(g2, j, abf);
  }
  d[idx] = s;
}

// ---- concat / embedding ----------------------------------------------------------
struct ConcatArgs { float* out; const float* src[4]; const int64_t* idx[4]; int kind[4], w[4], ld[4], rows[4]; int B, nseg, ldo; };
__device__ __forceinline__ void concat_body(const ConcatArgs& p, int bx) {
  const int id = bx * 256 + threadIdx.x;
  if (id >= p.B * p.ldo) return;
  const int b = id / p.ldo;
  int col = id - b * p.ldo;
  float v = 0.f;
  for (int j = 0; j < p.nseg; ++j) {
    if (col < p.w[j]) {
      if (p.kind[j] == 0) v = p.src[j][(size_t)b * p.ld[j] + col];
      else if (p.kind[j] == 1) {
        const int64_t row = p.idx[j][b];            // out-of-range label: zero row, no memory access
        if (row >= 0 && row < (int64_t)p.rows[j]) v = p.src[j][(size_t)row * p.ld[j] + col];
      }
      break;
    }
    col -= p.w[j];
  }
  p.out[id] = v;
}
__global__ __launch_bounds__(256) void concat_kernel(ConcatArgs p) { concat_body(p, blockIdx.x); }
struct EmbArgs { const float* d; const int64_t* idx; float* dt; int B, w, ld, col0, rows; int det; };
// A small table (rows * w <= 1024 floats: the 5 x 5 source / class tables): one WAVE per table element scans the batch
// and adds its fixed-order sum to the element it owns with ONE atomic (at most two ops contribute to a table per backward
// pass: 0 + a + b does not depend on the order) — bit-reproducible, and 2 560 global atomics on one cache line (4.9 us)
// became 25 (2.2 us).  Larger tables: one fp32 atomic per (sample, column) — unless flags & 1 (deterministic_wgrad): then
// the wave-per-element form at any size (ordered sums; slower for big tables, which the reference's scripts never build).
__device__ __forceinline__ void emb_bwd_body(const EmbArgs& p, int bx) {
  const int n = p.rows * p.w;
  if (n <= 1024 || p.det) {
    const int e = bx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= n) return;
    const int row = e / p.w, k = e - row * p.w;
    float s = 0.f;
    for (int b0 = lane; b0 < p.B; b0 += 8 * 64) {       // 8 samples per lane and round trip: labels first, then the matching rows
      int64_t id[8];
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) id[j] = b0 + j * 64 < p.B ? p.idx[b0 + j * 64] : -1;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = id[j] == (int64_t)row ? p.d[(size_t)(b0 + j * 64) * p.ld + p.col0 + k] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    s = wave_sum(s);
    if (lane == 0) atomic_add_f32(p.dt + e, s);      // (two ops on one table may share a parallel group; 0 + a + b is order-independent)
    return;
  }
  const int id = bx * 256 + threadIdx.x;
  if (id >= p.B * p.w) return;
  const int b = id / p.w, k = id - b * p.w;
  const int64_t row = p.idx[b];
  if (row < 0 || row >= (int64_t)p.rows) return;      // never write outside the table's gradient
  atomic_add_f32(p.dt + (size_t)row * p.w + k, p.d[(size_t)b * p.ld + p.col0 + k]);
}
__global__ __launch_bounds__(256) void emb_bwd_kernel(EmbArgs p) { emb_bwd_body(p, blockIdx.x); }

// ---- Linear ----------------------------------------------------------------------
struct LinArgs {
  const float* X; const float* W; const float* Bv; float* Y; double* stats;
  const float* DY; float* DX; const float* ACT; float* DW; float* DB;
  int M, N, K, ldx, ldy, act, has_mask, lda, accumulate;
  float slope;
};
// one thread per output (small K).  Statistics (the following BatchNorm's sum / sum of squares per column): the workgroup's
// 256 outputs are folded per column in LDS and leave as ONE atomic per column and statistic from consecutive threads —
// an atomic per output element (M*N*2 of them, ~7 cache lines per wave instruction) made the three BatchNorm-fed heads
// 4.4-4.9 us launches against 2.6 us for the same shape without statistics.
constexpr int kLinLdsW = 12288, kLinLdsX = 1024;      // floats of LDS for the staged weight / input rows
__device__ __forceinline__ void linear_fwd_thread_body(const LinArgs& p, int bx) {
  __shared__ double s_st[2][256];
  __shared__ float s_w[kLinLdsW], s_x[kLinLdsX];
  const int id = bx * 256 + threadIdx.x;
  const bool valid = id < p.M * p.N;
  const int m = valid ? id / p.N : 0, n = valid ? id - m * p.N : 0;
  float s = 0.f;
  // The weight rows and the input rows this workgroup's 256 outputs need are staged in LDS with coalesced loads (odd row
  // stride: conflict-free column reads); the dot product then runs out of LDS in the same k order.  Read straight from
  // global memory the loop cost ~0.12 us per k (x[k] and w[n*K+k]: two scalar round trips per step): decoder.linear_out
  // (K = 64) was a 9.3 us launch.
  const int KP = p.K | 1;
  const int WN = p.N < 256 ? p.N : 256;
  const int first = bx * 256;
  const int m_first = first / p.N, n_first = p.N <= 256 ? 0 : first - m_first * p.N;
  const int last = min(first + 255, p.M * p.N - 1);
  const int nrows = last / p.N - m_first + 1;
  // (measured per launch in a graph, batch 512: linear_out N=100 K=64 9.3 -> 5.7 us, encoder_fc.0 N=20 K=30 4.8 -> 4.4;
  //  N=512 K=20 got SLOWER, 7.4 -> 9.9 us — 20 staging iterations per thread for 20 k-steps — hence N <= 256, K >= 24)
  if (p.N <= 256 && p.K >= 24 && WN * KP <= kLinLdsW && nrows * KP <= kLinLdsX) {      // (uniform)
    __syncthreads();                 // (the body may be called repeatedly by one workgroup)
    for (int i = threadIdx.x; i < WN * p.K; i += 256) {
      const int r = i / p.K, k = i - r * p.K;
      int nn = n_first + r;
      if (nn >= p.N) nn -= p.N;
      s_w[r * KP + k] = p.W[(size_t)nn * p.K + k];
    }
    for (int i = threadIdx.x; i < nrows * p.K; i += 256) {
      const int r = i / p.K, k = i - r * p.K;
      s_x[r * KP + k] = p.X[(size_t)(m_first + r) * p.ldx + k];
    }
    __syncthreads();
    if (valid) {
      int wr = n - n_first;
      if (wr < 0) wr += p.N;
      const float* x = s_x + (m - m_first) * KP;
      const float* w = s_w + wr * KP;
      for (int k = 0; k < p.K; ++k) s = fmaf(x[k], w[k], s);
      if (p.Bv != nullptr) s += p.Bv[n];
    }
  } else if (valid) {
    const float* x = p.X + (size_t)m * p.ldx;
    const float* w = p.W + (size_t)n * p.K;
    for (int k = 0; k < p.K; ++k) s = fmaf(x[k], w[k], s);
    if (p.Bv != nullptr) s += p.Bv[n];
  }
  if (p.stats != nullptr) {          // (uniform)
    if (p.N > 256) {
      if (valid) {
        double* st = stat_replica(p.stats, p.N, m);
        atomic_add_f64(st + n, (double)s);
        atomic_add_f64(st + p.N + n, (double)s * (double)s);
      }
    } else {
      __syncthreads();               // (the body may be called repeatedly by one workgroup: the previous call's readers are done)
      s_st[0][threadIdx.x] = valid ? (double)s : 0.0;
      s_st[1][threadIdx.x] = valid ? (double)s * (double)s : 0.0;
      __syncthreads();
      if ((int)threadIdx.x < p.N) {
        // outputs bx*256 + i of this workgroup with (bx*256 + i) % N == threadIdx.x
        const int first = (int)(((int64_t)threadIdx.x - ((int64_t)bx * 256) % p.N + p.N) % p.N);
        double a = 0.0, b = 0.0;
        for (int i = first; i < 256; i += p.N) { a += s_st[0][i]; b += s_st[1][i]; }
        double* st = stat_replica(p.stats, p.N, bx);
        atomic_add_f64(st + threadIdx.x, a);
        atomic_add_f64(st + p.N + threadIdx.x, b);
      }
    }
  }
  if (!valid) return;
  if (p.act) s = lrelu(s, p.slope);
  p.Y[(size_t)m * p.ldy + n] = s;
}
__global__ __launch_bounds__(256) void linear_fwd_thread_kernel(LinArgs p) { linear_fwd_thread_body(p, blockIdx.x); }
// one wave per output (long K): coalesced reads of both operands
__global__ __launch_bounds__(256) void linear_fwd_wave_kernel(LinArgs p) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= p.M * p.N) return;
  const int m = wid / p.N, n = wid - m * p.N;
  const float* x = p.X + (size_t)m * p.ldx;
  const float* w = p.W + (size_t)n * p.K;
  float s = 0.f;
  for (int k = lane; k < p.K; k += 64) s = fmaf(x[k], w[k], s);
  s = wave_sum(s);
  if (lane == 0) {
    if (p.Bv != nullptr) s += p.Bv[n];
    if (p.stats != nullptr) {
      double* st = stat_replica(p.stats, p.N, m);
      atomic_add_f64(st + n, (double)s);
      atomic_add_f64(st + p.N + n, (double)s * (double)s);
    }
    if (p.act) s = lrelu(s, p.slope);
    p.Y[(size_t)m * p.ldy + n] = s;
  }
}
__device__ __forceinline__ void linear_bwd_x_body(const LinArgs& p, int bx) {
  const int id = bx * 256 + threadIdx.x;
  if (id >= p.M * p.K) return;
  const int m = id / p.K, k = id - m * p.K;
  const float* dy = p.DY + (size_t)m * p.ldy;
  float s = 0.f;
  for (int n = 0; n < p.N; ++n) s = fmaf(dy[n], p.W[(size_t)n * p.K + k], s);
  if (p.has_mask) s *= lrelu_grad(p.ACT[(size_t)m * p.lda + k], p.slope);
  float* dst = p.DX + (size_t)m * p.ldx + k;
  *dst = p.accumulate ? *dst + s : s;
}
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(LinArgs p) { linear_bwd_x_body(p, blockIdx.x); }
// one wave per output when the contraction (N) is long: coalesced DY reads, strided W reads hit L2
__global__ __launch_bounds__(256) void linear_bwd_x_wave_kernel(LinArgs p) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= p.M * p.K) return;
  const int m = wid / p.K, k = wid - m * p.K;
  const float* dy = p.DY + (size_t)m * p.ldy;
  float s = 0.f;
  for (int n = lane; n < p.N; n += 64) s = fmaf(dy[n], p.W[(size_t)n * p.K + k], s);
  s = wave_sum(s);
  if (lane == 0) {
    if (p.has_mask) s *= lrelu_grad(p.ACT[(size_t)m * p.lda + k], p.slope);
    float* dst = p.DX + (size_t)m * p.ldx + k;
    *dst = p.accumulate ? *dst + s : s;
  }
}
// block = (output row n, chunk of <=256 input columns, slice of M); fp32 atomics into zeroed DW/DB
__device__ __forceinline__ void linear_bwd_w_body(const LinArgs& p, int rows_per_z, int bx, int by, int bz) {
  __shared__ double lds[2 * 256];
  __syncthreads();                      // (the body may be called repeatedly by one workgroup: the previous call's readers are done)
  const int n = bx;
  const int kw = p.K < 256 ? p.K : 256;
  const int ml = 256 / kw;
  const int tid = threadIdx.x, mlane = tid / kw;
  const int k = by * kw + (tid - mlane * kw);
  const bool active = mlane < ml && k < p.K;
  const int mbeg = bz * rows_per_z, mend = min(p.M, mbeg + rows_per_z);
  double v[2] = {0.0, 0.0};
  if (active && mbeg < mend) {
    for (int m0 = mbeg + mlane; m0 < mend; m0 += 8 * ml) {      // batches of 8 rows: loads first, then use
      float d[8], x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int m = m0 + j * ml;
        const int mm = m < mend ? m : mend - 1;
        d[j] = m < mend ? p.DY[(size_t)mm * p.ldy + n] : 0.f;
        x[j] = p.X[(size_t)mm * p.ldx + k];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[0] += (double)(d[j] * x[j]);
        v[1] += (double)d[j];
      }
    }
  }
  lds[tid] = active ? v[0] : 0.0;
  lds[256 + tid] = active ? v[1] : 0.0;
  __syncthreads();
  if (active && mlane == 0) {
    double s = 0.0, sb = 0.0;
    for (int r = 0; r < ml; ++r) { s += lds[r * kw + tid]; sb += lds[256 + r * kw + tid]; }
    atomic_add_f32(p.DW + (size_t)n * p.K + k, (float)s);
    if (p.DB != nullptr && k == 0) atomic_add_f32(p.DB + n, (float)sb);
  }
}
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(LinArgs p, int rows_per_z) {
  linear_bwd_w_body(p, rows_per_z, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- Linear on the matrix cores (linear_mfma.h): the same three ops once the batch makes them GEMMs -----------------------------
template <bool B_KN>
__global__ __launch_bounds__(kLinMMThreads) void lin_mm_kernel(LinMM p) {
  __shared__ __attribute__((aligned(16))) float smem[kLinMMLds];
  lin_mm_body<B_KN>(p, blockIdx.x, smem);
}
__global__ __launch_bounds__(256) void lin_wgrad_kernel(LinWg p, int ntiles) {
  __shared__ __attribute__((aligned(16))) float smem[kLinWgLds];
  lin_wgrad_body(p, blockIdx.x % ntiles, blockIdx.x / ntiles, smem);
}

// ---- reparameterisation / losses ---------------------------------------------------
__device__ __forceinline__ void block_atomic_f64(double v, double* dst) {
  __shared__ double red[4];
  __syncthreads();                      // re-entrant: a previous call's reader (thread 0) is done with red[]
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    atomic_add_f64(dst, s);
  }
}
struct ReparamArgs { const float* mulv; const float* eps; float* z; double* loss; const float* dz; float* dmulv; int B, zd, lddz; float beta; };
__device__ __forceinline__ void reparam_kl_fwd_body(const ReparamArgs& p, int bx) {
  const float* mulv = p.mulv; const float* eps = p.eps; float* z = p.z; double* loss = p.loss;
  const int B = p.B, zd = p.zd;
  const int id = bx * 256 + threadIdx.x;
  double kl = 0.0;
  if (id < B * zd) {
    const int b = id / zd, j = id - b * zd;
    const float mu = mulv[(size_t)b * 2 * zd + j], lv = mulv[(size_t)b * 2 * zd + zd + j];
    z[id] = fmaf(eps[id], expf(0.5f * lv), mu);
    kl = -0.5 * (double)(1.f + lv - mu * mu - expf(lv));
  }
  block_atomic_f64(kl, loss + 0);
}
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(ReparamArgs p) { reparam_kl_fwd_body(p, blockIdx.x); }
__device__ __forceinline__ void reparam_kl_bwd_body(const ReparamArgs& p, int bx) {
  const float* mulv = p.mulv; const float* eps = p.eps; const float* dz = p.dz; float* dmulv = p.dmulv;
  const int B = p.B, zd = p.zd, lddz = p.lddz;
  const float beta = p.beta;
  const int id = bx * 256 + threadIdx.x;
  if (id >= B * zd) return;
  const int b = id / zd, j = id - b * zd;
  const float mu = mulv[(size_t)b * 2 * zd + j], lv = mulv[(size_t)b * 2 * zd + zd + j];
  const float g = dz[(size_t)b * lddz + j];
  const float invB = 1.f / (float)B;
  dmulv[(size_t)b * 2 * zd + j] = g + beta * mu * invB;
  dmulv[(size_t)b * 2 * zd + zd + j] = g * eps[id] * 0.5f * expf(0.5f * lv) + beta * 0.5f * (expf(lv) - 1.f) * invB;
}
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(ReparamArgs p) { reparam_kl_bwd_body(p, blockIdx.x); }
struct MseArgs { const float* x; const float* rec; float* drec; double* loss; int n, slot; float w; };
__device__ __forceinline__ void mse_body(const MseArgs& p, int bx) {
  const int id = bx * 256 + threadIdx.x;
  double sq = 0.0;
  if (id < p.n) {
    const float d = p.rec[id] - p.x[id];
    sq = (double)d * (double)d;
    p.drec[id] = p.w * 2.f * d / (float)p.n;
  }
  block_atomic_f64(sq, p.loss + p.slot);
}
__global__ __launch_bounds__(256) void mse_kernel(MseArgs p) { mse_body(p, blockIdx.x); }
struct LossArgs { const double* loss; float* out; int B, n1, n2; float beta, w1, w2; };
__device__ __forceinline__ void loss_finalize_body(const LossArgs& p) {
  const double* loss = p.loss; float* out = p.out;
  const int B = p.B, n1 = p.n1, n2 = p.n2;
  const float beta = p.beta, w1 = p.w1, w2 = p.w2;
  if (threadIdx.x != 0) return;
  const double kl = loss[0] / (double)B;
  const double m1 = loss[1] / (double)n1;
  const double m2 = n2 > 0 ? loss[2] / (double)n2 : 0.0;
  out[0] = (float)((double)w1 * m1 + (double)w2 * m2 + (double)beta * kl);
  out[1] = (float)m1;
  out[2] = (float)m2;
  out[3] = (float)kl;
}
__global__ void loss_finalize_kernel(LossArgs p) { if (blockIdx.x == 0) loss_finalize_body(p); }

// ---- decoder tail (C_out = 1) -----------------------------------------------------
__global__ void tail_fwd_kernel(const float* act, const float* w, const float* bias, float* out, int B, int Lh, int C, int abf) {
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int Lo = 2 * Lh;
  if (wid >= B * Lo) return;
  const int b = wid / Lo, pz = wid - b * Lo;
  float s = 0.f;
  for (int t = 0; t < 3; ++t) {
    const int u = pz + t - 1;
    if (u < 0 || u >= Lo) continue;
    const size_t row = ((size_t)b * Lh + (u >> 1)) * C;
    for (int c = lane; c < C; c += 64) s = fmaf(ald(act, row + c, abf), w[c * 3 + t], s);
  }
  s = wave_sum(s);
  if (lane == 0) out[wid] = s + bias[0];
}
__global__ void tail_bwd_x_kernel(const float* dt, const float* w, float* dact, int B, int Lh, int C, int abf) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (size_t)B * Lh * C) return;
  const int c = id % C;
  const int j = (id / C) % Lh;
  const int b = (id / C) / Lh;
  const int Lo = 2 * Lh;
  const float* d = dt + (size_t)b * Lo;
  float s = 0.f;
  for (int e = 0; e < 2; ++e)
    for (int t = 0; t < 3; ++t) {
      const int q = 2 * j + e - t + 1;
      if (q >= 0 && q < Lo) s = fmaf(d[q], w[c * 3 + t], s);
    }
  ast(dact, id, s, abf);
}
// dw[c][t] = sum_{b,h} act[b,h,c] * D_t[b,h] with D_t[b,h] = dt[b][2h+1-t] + dt[b][2h+2-t] (in range), db = sum dt.
// thread = (channel, row lane); kTailRows rows (b,h) per thread, loaded in batches of 8 before use (a per-row
// load -> use loop is one dependent memory round trip per row); fp32 atomics into zeroed DW/DB.
constexpr int kTailRows = 32;
__global__ __launch_bounds__(256) void tail_bwd_w_kernel(const float* dt, const float* act, float* dw, float* db, int B, int Lh, int C, int rows, int abf) {
  __shared__ double lds[4 * 256];
  const int cw = C < 256 ? C : 256, rl = 256 / cw;
  const int tid = threadIdx.x, rlane = tid / cw;
  const int c = blockIdx.y * cw + (tid - rlane * cw);
  const bool active = rlane < rl && c < C;
  const int M = B * Lh, Lo = 2 * Lh;
  const int row0 = blockIdx.x * rl * rows + rlane;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  if (active) {
    for (int k0 = 0; k0 < rows; k0 += 8) {
      float a[8], d0[8], d1[8], d2[8], d3[8];
      bool ok[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = row0 + (k0 + k) * rl;
        ok[k] = r < M;
        const int rr = ok[k] ? r : M - 1;
        const int b = rr / Lh, h = rr - b * Lh;
        const float* dtb = dt + (size_t)b * Lo + 2 * h;
        a[k] = ald(act, (size_t)rr * C + c, abf);
        d0[k] = h > 0 ? dtb[-1] : 0.f;
        d1[k] = dtb[0];
        d2[k] = dtb[1];
        d3[k] = h + 1 < Lh ? dtb[2] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!ok[k]) continue;
        v[0] += (double)(a[k] * (d2[k] + d3[k]));      // tap 0: pz = 2h+1, 2h+2
        v[1] += (double)(a[k] * (d1[k] + d2[k]));      // tap 1: pz = 2h,   2h+1
        v[2] += (double)(a[k] * (d0[k] + d1[k]));      // tap 2: pz = 2h-1, 2h
        v[3] += (double)(d1[k] + d2[k]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) lds[j * 256 + tid] = active ? v[j] : 0.0;
  __syncthreads();
  if (active && rlane == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double sum = 0.0;
      for (int r = 0; r < rl; ++r) sum += lds[j * 256 + r * cw + tid];
      v[j] = sum;
    }
    atomic_add_f32(dw + c * 3 + 0, (float)v[0]);
    atomic_add_f32(dw + c * 3 + 1, (float)v[1]);
    atomic_add_f32(dw + c * 3 + 2, (float)v[2]);
    if (c == 0) atomic_add_f32(db, (float)v[3]);
  }
}
// float4 form of the same reduction (C a multiple of 4): 16 row lanes x 16 rows per thread (see stem_wgrad4_kernel)
constexpr int kTailRows4 = 16;
__global__ __launch_bounds__(256) void tail_bwd_w4_kernel(const float* dt, const float* act, float* dw, float* db, int B, int Lh, int C, int rows, int abf) {
  __shared__ double lds[13 * 256];
  const int M = B * Lh, Lo = 2 * Lh;
  const ColMap m = colmap_v<4>(M, C, rows, blockIdx.x, blockIdx.y);
  double v[13];
#pragma unroll
  for (int j = 0; j < 13; ++j) v[j] = 0.0;
  if (m.active) {
    for (int k0 = 0; k0 < rows; k0 += 8) {
      float4 a[8];
      float d0[8], d1[8], d2[8], d3[8];
      bool ok[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = m.row + (k0 + k) * m.rstep;
        ok[k] = r < m.rend;
        const int rr = ok[k] ? r : m.rend - 1;
        const int b = rr / Lh, h = rr - b * Lh;
        const float* dtb = dt + (size_t)b * Lo + 2 * h;
        a[k] = ald4(act, (size_t)rr * C + m.c, abf);
        d0[k] = h > 0 ? dtb[-1] : 0.f;
        d1[k] = dtb[0];
        d2[k] = dtb[1];
        d3[k] = h + 1 < Lh ? dtb[2] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!ok[k]) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float aj = (&a[k].x)[j];
          v[3 * j + 0] += (double)(aj * (d2[k] + d3[k]));      // tap 0: pz = 2h+1, 2h+2
          v[3 * j + 1] += (double)(aj * (d1[k] + d2[k]));      // tap 1: pz = 2h,   2h+1
          v[3 * j + 2] += (double)(aj * (d0[k] + d1[k]));      // tap 2: pz = 2h-1, 2h
        }
        v[12] += (double)(d1[k] + d2[k]);
      }
    }
  }
  fold_rows<13>(m, v, lds);
  // one atomic per output from consecutive threads (see stem_wgrad4_kernel)
  const int c_base = blockIdx.y * m.cw * 4;
  for (int o = threadIdx.x; o < m.cw * 12; o += 256) {
    const int cl = o / 3, t = o - cl * 3;
    if (c_base + cl < C) atomic_add_f32(dw + (size_t)(c_base + cl) * 3 + t, (float)lds[(3 * (cl & 3) + t) * 256 + (cl >> 2)]);
  }
  if (threadIdx.x == 0 && blockIdx.y == 0) atomic_add_f32(db, (float)lds[12 * 256 + 0]);
}

// ---- slab reduce / optimiser ------------------------------------------------------
__global__ void slab_reduce_kernel(const float* slab, float* out, int n, int nsplit, int stride) {
  const int nv = n >> 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    float4 s = *reinterpret_cast<const float4*>(slab + (size_t)i * 4);
    for (int k = 1; k < nsplit; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)k * stride + (size_t)i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)i * 4) = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int i = (nv << 2) + threadIdx.x;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slab[(size_t)k * stride + i];
    out[i] = s;
  }
}
__global__ __launch_bounds__(256) void gradnorm_kernel(const float* g, double* norm2, int n) {
  double s = 0.0;
  const int nv = n >> 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    const float4 v = *reinterpret_cast<const float4*>(g + (size_t)i * 4);
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(nv << 2) + threadIdx.x]; s += (double)v * v; }
  block_atomic_f64(s, norm2);
}

struct AdamArgs { float* p; const float* g; float* m; float* v; const int64_t* step; const double* norm2; int n; float lr, b1, b2, eps, wd, clip, omb1, omb2; };
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a, float coef, float step_size, float bc2s) {
  g *= coef;
  p *= (1.f - a.lr * a.wd);
  m = m + a.omb1 * (g - m);          // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.b2 + a.omb2 * g * g;
  const float denom = sqrtf(v) / bc2s + a.eps;
  p = p - step_size * (m / denom);
}
__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  const double t = (double)a.step[0];
  // beta reconstructed from the fp32 complements: 1 - beta^t keeps ~1e-7 relative accuracy at small t
  const double bc1 = 1.0 - pow(1.0 - (double)a.omb1, t), bc2 = 1.0 - pow(1.0 - (double)a.omb2, t);
  const float step_size = (float)((double)a.lr / bc1), bc2s = (float)sqrt(bc2);
  float coef = 1.f;
  if (a.clip > 0.f) {
    const float c = a.clip / ((float)sqrt(a.norm2[0]) + 1e-6f);
    coef = c < 1.f ? c : 1.f;
  }
  const int nv = a.n >> 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    float4 p = reinterpret_cast<float4*>(a.p)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 m = reinterpret_cast<float4*>(a.m)[i];
    float4 v = reinterpret_cast<float4*>(a.v)[i];
    adam1(p.x, g.x, m.x, v.x, a, coef, step_size, bc2s);
    adam1(p.y, g.y, m.y, v.y, a, coef, step_size, bc2s);
    adam1(p.z, g.z, m.z, v.z, a, coef, step_size, bc2s);
    adam1(p.w, g.w, m.w, v.w, a, coef, step_size, bc2s);
    reinterpret_cast<float4*>(a.p)[i] = p;
    reinterpret_cast<float4*>(a.m)[i] = m;
    reinterpret_cast<float4*>(a.v)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const int i = (nv << 2) + threadIdx.x;
    adam1(a.p[i], a.g[i], a.m[i], a.v[i], a, coef, step_size, bc2s);
  }
}
// ---- Schedule-Free AdamW (hippie/optimizers.py:105-209) -------------------------------------------
// per-step scalars, fp64 like the reference's Python floats (:118-138)
__global__ void sf_schedule_kernel(const int64_t* step, double* st, int warmup, float lr, float omb2, float r, float power) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double k = (double)step[0];
  const double sched = k < (double)warmup ? (k + 1.0) / (double)warmup : 1.0;
  const double bc2 = 1.0 - pow(1.0 - (double)omb2, k + 1.0);
  const double lr_t = (double)lr * sched * sqrt(bc2);
  const double lr_max = lr_t > st[0] ? lr_t : st[0];
  const double weight = pow(k + 1.0, (double)r) * pow(lr_max, (double)power);
  const double wsum = st[1] + weight;
  st[0] = lr_max;
  st[1] = wsum;
  st[2] = lr_t;
  st[3] = wsum != 0.0 ? weight / wsum : 0.0;
}
// torch.lerp: start + w*(end-start) for |w| < 0.5, else end - (end-start)*(1-w)
__device__ __forceinline__ float torch_lerp(float a, float b, float w) {
  const float d = b - a;
  return fabsf(w) < 0.5f ? a + w * d : b - d * (1.f - w);
}
struct SfArgs {
  float *y, *z, *v;
  const float* g;
  const int64_t* step;
  const double *st, *norm2;
  int n;
  float b1, b2, eps, wd, clip, omb2;
};
__device__ __forceinline__ void sf1(float& y, float g, float& z, float& v, const SfArgs& a, float coef, float ckp1,
                                    float ay, float lr_t, bool first) {
  g *= coef;
  if (first) z = y;
  v = v * a.b2 + a.omb2 * g * g;
  float gn = g / (sqrtf(v) + a.eps);
  if (a.wd != 0.f) gn += a.wd * y;
  y = torch_lerp(y, z, ckp1);
  y += ay * gn;
  z -= lr_t * gn;
}
__global__ __launch_bounds__(256) void adamw_sf_kernel(SfArgs a) {
  const bool first = a.step[0] == 0;
  const double lr_d = a.st[2], ck_d = a.st[3];
  const float lr_t = (float)lr_d, ckp1 = (float)ck_d;
  const float ay = (float)(lr_d * ((double)a.b1 * (1.0 - ck_d) - 1.0));
  float coef = 1.f;
  if (a.clip > 0.f) {
    const float c = a.clip / ((float)sqrt(a.norm2[0]) + 1e-6f);
    coef = c < 1.f ? c : 1.f;
  }
  const int nv = a.n >> 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    float4 y = reinterpret_cast<float4*>(a.y)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    float4 z = reinterpret_cast<float4*>(a.z)[i];
    float4 v = reinterpret_cast<float4*>(a.v)[i];
    sf1(y.x, g.x, z.x, v.x, a, coef, ckp1, ay, lr_t, first);
    sf1(y.y, g.y, z.y, v.y, a, coef, ckp1, ay, lr_t, first);
    sf1(y.z, g.z, z.z, v.z, a, coef, ckp1, ay, lr_t, first);
    sf1(y.w, g.w, z.w, v.w, a, coef, ckp1, ay, lr_t, first);
    reinterpret_cast<float4*>(a.y)[i] = y;
    reinterpret_cast<float4*>(a.z)[i] = z;
    reinterpret_cast<float4*>(a.v)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const int i = (nv << 2) + threadIdx.x;
    sf1(a.y[i], a.g[i], a.z[i], a.v[i], a, coef, ckp1, ay, lr_t, first);
  }
}
__global__ __launch_bounds__(256) void lerp_kernel(float* y, const float* z, int n, float w) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = torch_lerp(y[i], z[i], w);
}
// F.interpolate(mode="linear", align_corners=False) per row (+ optional log(x+1)): dataloading.py:78,93,96
__global__ void resample_linear_kernel(const float* in, float* out, int N, int W, int L, int log1) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= N * L) return;
  const int n = id / L, i = id - n * L;
  const float scale = (float)W / (float)L;
  float src = scale * ((float)i + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  int x0 = (int)src;
  x0 = x0 < W - 1 ? x0 : W - 1;
  const int x1 = x0 + 1 < W ? x0 + 1 : W - 1;
  const float w1 = src - (float)x0, w0 = 1.f - w1;
  float a = in[(size_t)n * W + x0], b = in[(size_t)n * W + x1];
  if (log1) { a = logf(a + 1.f); b = logf(b + 1.f); }
  out[id] = w0 * a + w1 * b;
}
// HP_OP_ZERO as a plain kernel node (not hipMemsetAsync): inside a captured hipGraph a memset node is executed by
// a different engine than the kernel nodes around it, and replays of graphs that START with memset nodes were
// observed to race with the previous replay's tail on this runtime (wrong gradients on the second replay of the
// zipped two-model program, tools/debug/pair_probe.py).  A kernel is ordered like every other node.
__global__ __launch_bounds__(256) void zero_kernel(uint4* p16, size_t n16, uint8_t* tail, int ntail) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p16[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
// ---- HP_OP_STAGE_BATCH: index gather from HBM-resident tables + Philox reparameterisation noise --------------------------
struct StageArgs {
  const float* table; const float* table2; const int64_t* labels; const int64_t* perm; const int64_t* cursor; const int64_t* seed;
  float* x; float* x2; int64_t* src; float* eps;
  int B, L, L2, z, spe, world, rank, N;
};
// Philox4x32-10 (Salmon et al., SC'11): counter-based, so a step's noise is a pure function of (seed, cursor, element)
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(uint32_t v) { return ((float)(v >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__global__ __launch_bounds__(256) void stage_batch_kernel(StageArgs p) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  const uint64_t cur = (uint64_t)p.cursor[0];
  const long j = (long)(cur % (uint64_t)p.spe) * p.world + p.rank;
  const long nx = (long)p.B * p.L, nx2 = (long)p.B * p.L2;
  auto row_of = [&](int b) -> long {
    const int64_t r = p.perm[j * p.B + b];
    return (r >= 0 && r < (int64_t)p.N) ? (long)r : 0;
  };
  if (id < nx) {
    const int b = (int)(id / p.L), c = (int)(id - (long)b * p.L);
    p.x[id] = p.table[row_of(b) * p.L + c];
    return;
  }
  long q = id - nx;
  if (q < nx2) {
    const int b = (int)(q / p.L2), c = (int)(q - (long)b * p.L2);
    p.x2[q] = p.table2[row_of(b) * p.L2 + c];
    return;
  }
  q -= nx2;
  if (q < p.B) {
    p.src[q] = p.labels[row_of((int)q)];
    return;
  }
  q -= p.B;
  const long ne = (long)p.B * p.z;
  if (q * 4 < ne) {
    const uint64_t sd = (uint64_t)p.seed[0];
    const uint4 v = philox4x32_10(make_uint4((uint32_t)cur, (uint32_t)(cur >> 32), (uint32_t)q, (uint32_t)p.rank), make_uint2((uint32_t)sd, (uint32_t)(sd >> 32)));
    const float r0 = sqrtf(-2.f * logf(u01(v.x))), r1 = sqrtf(-2.f * logf(u01(v.z)));
    float s0, c0, s1, c1;
    sincosf(6.28318530717958647692f * u01(v.y), &s0, &c0);
    sincosf(6.28318530717958647692f * u01(v.w), &s1, &c1);
    const float n4[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (q * 4 + t < ne) p.eps[q * 4 + t] = n4[t];
  }
}

__global__ void step_inc_kernel(int64_t* step) { if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1; }

inline int blocks_for(int64_t n, int per = 256) { return (int)((n + per - 1) / per); }

// The Linear ops take the matrix-core kernels from this many rows on (below it a launch is floor-bound either way and the scalar
// forms, sized for the 512-row heads, stay).  (the knob: tools/micro/linear_sweep.py and the ragged unit-test shapes)
inline int lin_mfma_min_rows() {
  static const int forced = debug_knob_int("HIPPIE_LIN_MFMA_MIN_M");
  return forced > 0 ? forced : 1024;
}
// alignment class of the 4-float pieces `base + row*ld + 4*j`: 4 = 16 bytes, 2 = 8 bytes, 1 = 4 bytes (linear_mfma.h)
inline int lin_align(const void* base, int ld) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(base);
  if ((a & 15) == 0 && (ld & 3) == 0) return 4;
  if ((a & 7) == 0 && (ld & 1) == 0) return 2;
  return 1;
}
// rows per M-split of the matrix-core weight gradient: 16 splits of >= 256 rows (a 64x64 tile's 4096 atomics per split stay small
// next to its MFMA work); flags & 1 (deterministic_wgrad): one split, no cross-workgroup atomics
inline int lin_wgrad_rows(int M, bool one_split) {
  if (one_split) return hp::cdiv(M, 32) * 32;
  return min(1024, max(256, hp::cdiv(hp::cdiv(M, 16), 32) * 32));
}
LinWg lin_wgrad_args(const HpOp& op, void* const* bases) {
  using hp::ptr;
  LinWg a;
  a.DY = ptr<const float>(op, 0, bases); a.X = ptr<const float>(op, 1, bases); a.DW = ptr<float>(op, 2, bases); a.DB = ptr<float>(op, 3, bases);
  a.M = op.i[0]; a.N = op.i[1]; a.K = op.i[2]; a.ldy = op.i[3]; a.ldx = op.i[4];
  a.alY = lin_align(a.DY, a.ldy); a.alX = lin_align(a.X, a.ldx);
  a.rows_per_split = lin_wgrad_rows(a.M, op.flags & 1);
  return a;
}

// M-slices of HP_OP_LINEAR_BWD_W: every slice ends in one fp32 atomic per (n, k).  Rows per slice = 16 per row lane of the
// workgroup (256 / min(K, 256) lanes), within [32, 128]: two batches of 8 row loads per thread.  Measured per launch in a
// graph at M = 512 (tools/micro/linear_sweep.py): the former "1024 workgroups, 16 rows each" took 8.7-8.9 us on the
// 20 x 20 / 20 x 30 heads (32 atomics per address, 640 workgroups of two rows per thread) against 3.2-3.5 us now.
inline int linear_bwd_w_slices(int M, int N, int K, bool one_slice = false) {
  if (one_slice) return 1;           // flags & 1: no cross-workgroup atomics, bit-reproducible
  static const int forced = debug_knob_int("HIPPIE_LBW_ROWS");   // (the sweep)
  const int kw = K < 256 ? K : 256, ml = 256 / kw;
  const int rows = forced > 0 ? forced : min(128, max(32, 16 * ml));
  return max(1, hp::cdiv(M, rows));
}
BnApplyArgs bn_apply_args(const HpOp& op, void* const* bases) {
  using hp::ptr;
  const int32_t* I = op.i;
  BnApplyArgs a;
  a.raw = ptr<const float>(op, 0, bases); a.out = ptr<float>(op, 1, bases); a.stats = ptr<const double>(op, 2, bases);
  a.gamma = ptr<const float>(op, 3, bases); a.beta = ptr<const float>(op, 4, bases);
  a.rmean = ptr<float>(op, 5, bases); a.rvar = ptr<float>(op, 6, bases); a.save = ptr<float>(op, 7, bases);
  a.res = ptr<const float>(op, 8, bases); a.stats2 = ptr<const double>(op, 9, bases);
  a.gamma2 = ptr<const float>(op, 10, bases); a.beta2 = ptr<const float>(op, 11, bases);
  a.rmean2 = ptr<float>(op, 12, bases); a.rvar2 = ptr<float>(op, 13, bases); a.save2 = ptr<float>(op, 14, bases);
  a.M = I[0]; a.C = I[1]; a.res_mode = I[2]; a.training = I[3]; a.act = I[4];
  a.Mstat = I[0] * (I[5] > 1 ? I[5] : 1);
  a.slope = op.f[0]; a.eps = op.f[1]; a.momentum = op.f[2];
  a.abf = (op.flags & HP_FLAG_ACT_BF16) ? 1 : 0;
  return a;
}
BnBwdReduceArgs bn_bwd_reduce_args(const HpOp& op, void* const* bases) {
  using hp::ptr;
  const int32_t* I = op.i;
  BnBwdReduceArgs a;
  a.g1 = ptr<const float>(op, 0, bases); a.g2 = I[2] ? ptr<const float>(op, 1, bases) : nullptr;
  a.act = ptr<const float>(op, 2, bases); a.gout = ptr<float>(op, 3, bases);
  a.raw = ptr<const float>(op, 4, bases); a.save = ptr<const float>(op, 5, bases); a.bs = ptr<double>(op, 6, bases);
  a.raw2 = ptr<const float>(op, 7, bases); a.save2 = ptr<const float>(op, 8, bases); a.bs2 = ptr<double>(op, 9, bases);
  a.coef = ptr<const float>(op, 10, bases);
  a.M = I[0]; a.C = I[1]; a.has_second = I[3]; a.slope = op.f[0];
  a.rpl = bn_red_rpl(a.M, a.C);
  a.abf = (op.flags & HP_FLAG_ACT_BF16) ? 1 : 0;
  return a;
}
BnBwdApplyArgs bn_bwd_apply_args(const HpOp& op, void* const* bases) {
  using hp::ptr;
  BnBwdApplyArgs a;
  a.g = ptr<const float>(op, 0, bases); a.raw = ptr<const float>(op, 1, bases); a.save = ptr<const float>(op, 2, bases);
  a.bs = ptr<const double>(op, 3, bases); a.gamma = ptr<const float>(op, 4, bases); a.dr = ptr<float>(op, 5, bases);
  a.dgamma = ptr<float>(op, 6, bases); a.dbeta = ptr<float>(op, 7, bases);
  a.M = op.i[0]; a.C = op.i[1];
  const int w = op.i[2] > 1 ? op.i[2] : 1;
  a.Mstat = a.M * w; a.gscale = 1.f / (float)w;
  a.abf = (op.flags & HP_FLAG_ACT_BF16) ? 1 : 0;
  return a;
}


// ---- HP_OP_WFRAG: three-term MFMA fragments of a conv weight tensor (include/hippie_hip.h) -----------------------------------------
struct WfragArgs {
  const float* W; unsigned short* F; unsigned short* G;
  int T, N, K, which;
};
// One workgroup: the 32 x 32 block (n32, k32) of slab t, read once (rows of 128 contiguous bytes), split, and written as its two F chunks
// (waves 0, 1: the block's two 16-wide k slabs) and its two G chunks (waves 2, 3: its two 16-row n slabs).
__device__ __forceinline__ void wfrag_body(const WfragArgs& p, const int blk, float* tile /* [32][33] */) {
  const int kb = p.K >> 5, nb = (p.N + 31) >> 5;
  const int k32 = blk % kb, n32 = (blk / kb) % nb, t = blk / (kb * nb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const int r = tid >> 3, c = (tid & 7) << 2;          // 32 rows x 8 four-float pieces
    const int n = n32 * 32 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < p.N) v = *reinterpret_cast<const float4*>(p.W + ((size_t)t * p.N + n) * p.K + k32 * 32 + c);
    tile[r * 33 + c] = v.x; tile[r * 33 + c + 1] = v.y; tile[r * 33 + c + 2] = v.z; tile[r * 33 + c + 3] = v.w;
  }
  __syncthreads();
  const int j = lane & 31, h = lane >> 5, q = wave & 1;
  const bool g_form = wave >= 2;
  if (!(p.which & (g_form ? 2 : 1))) return;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = g_form ? tile[(q * 16 + 8 * h + e) * 33 + j] : tile[j * 33 + q * 16 + 8 * h + e];
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    unsigned short hb[2], mb[2], lb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float v = x[2 * d + u];
      const __bf16 h1 = (__bf16)v;
      const float r1 = v - (float)h1;
      const __bf16 m1 = (__bf16)r1;
      const __bf16 l1 = (__bf16)(r1 - (float)m1);
      hb[u] = *reinterpret_cast<const unsigned short*>(&h1); mb[u] = *reinterpret_cast<const unsigned short*>(&m1); lb[u] = *reinterpret_cast<const unsigned short*>(&l1);
    }
    hh[d] = (unsigned)hb[0] | ((unsigned)hb[1] << 16); mm[d] = (unsigned)mb[0] | ((unsigned)mb[1] << 16); ll[d] = (unsigned)lb[0] | ((unsigned)lb[1] << 16);
  }
  // chunk (t, ks, jn) of the image [T][KK/16][JN][3][64][8]: F: KK = K, JN = ceil(N/32), ks = 2 k32 + q, jn = n32;  G: KK = N, JN = K/32, ks = 2 n32 + q, jn = k32
  size_t chunk;
  unsigned short* img;
  if (!g_form) { img = p.F; chunk = ((size_t)t * (p.K >> 4) + 2 * k32 + q) * nb + n32; }
  else         { img = p.G; chunk = ((size_t)t * (((p.N + 31) >> 5) * 2) + 2 * n32 + q) * kb + k32; }
  uint4* dst = reinterpret_cast<uint4*>(img + chunk * (3 * 64 * 8)) + lane;
  dst[0] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
  dst[64] = make_uint4(mm[0], mm[1], mm[2], mm[3]);
  dst[128] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}
__global__ __launch_bounds__(256) void wfrag_kernel(WfragArgs p) {
  __shared__ float tile[32 * 33];
  wfrag_body(p, blockIdx.x, tile);
}

// ---- argument records of the small ops that are launched from a table (the small-leaf group) or share their argument
// decoding with it --------------------------------------------------------------------------------------------------
struct SmallEntry {
  int op, variant;          // HP_OP_* ; variant: LINEAR_BWD_W: 1 = matrix-core form (a.wg; gx = tiles, gy = splits)
  int gx, gy, gz;           // virtual grid of 256-thread blocks
  int rows_per_z;           // LINEAR_BWD_W
  union { LinArgs lin; LinWg wg; EmbArgs emb; ReparamArgs rp; MseArgs mse; LossArgs loss; WfragArgs wf; } a;
};

// args + grid of one such op; false for any other opcode
bool small_entry(const HpOp& op, void* const* bases, SmallEntry& e) {
  using hp::ptr;
  const int32_t* I = op.i;
  e.op = op.op; e.variant = 0; e.gx = e.gy = e.gz = 1; e.rows_per_z = 0;
  switch (op.op) {
    case HP_OP_WFRAG: {
      WfragArgs a{ptr<const float>(op, 0, bases), ptr<unsigned short>(op, 1, bases), ptr<unsigned short>(op, 2, bases), I[0], I[1], I[2], I[3]};
      e.a.wf = a;
      e.gx = I[0] * hp::cdiv(I[1], 32) * (I[2] / 32);
      return true;
    }
    case HP_OP_EMB_BWD: {
      EmbArgs a{ptr<const float>(op, 0, bases), ptr<const int64_t>(op, 1, bases), ptr<float>(op, 2, bases), I[0], I[1], I[2], I[3], I[4], op.flags & 1};
      e.a.emb = a;
      e.gx = (a.rows * a.w <= 1024 || a.det) ? hp::cdiv(a.rows * a.w, 4) : blocks_for((int64_t)I[0] * I[1]);      // (emb_bwd_body: a wave per element)
      return true;
    }
    case HP_OP_LINEAR_BWD_W: {
      if (I[0] >= lin_mfma_min_rows()) {
        const LinWg w = lin_wgrad_args(op, bases);
        e.variant = 1; e.a.wg = w;
        e.gx = hp::cdiv(w.N, 64) * hp::cdiv(w.K, 64); e.gy = hp::cdiv(w.M, w.rows_per_split);
        return true;
      }
      LinArgs a{};
      a.DY = ptr<const float>(op, 0, bases); a.X = ptr<const float>(op, 1, bases); a.DW = ptr<float>(op, 2, bases);
      a.DB = ptr<float>(op, 3, bases);
      a.M = I[0]; a.N = I[1]; a.K = I[2]; a.ldy = I[3]; a.ldx = I[4];
      const int kw = a.K < 256 ? a.K : 256;
      const int ky = hp::cdiv(a.K, kw);
      const int nz = linear_bwd_w_slices(a.M, a.N, a.K, op.flags & 1);
      e.rows_per_z = hp::cdiv(a.M, nz);
      e.a.lin = a; e.gx = a.N; e.gy = ky; e.gz = hp::cdiv(a.M, e.rows_per_z);
      return true;
    }
    case HP_OP_REPARAM_KL_FWD: {
      ReparamArgs a{};
      a.mulv = ptr<const float>(op, 0, bases); a.eps = ptr<const float>(op, 1, bases); a.z = ptr<float>(op, 2, bases);
      a.loss = ptr<double>(op, 3, bases); a.B = I[0]; a.zd = I[1];
      e.a.rp = a; e.gx = blocks_for((int64_t)I[0] * I[1]);
      return true;
    }
    case HP_OP_REPARAM_KL_BWD: {
      ReparamArgs a{};
      a.mulv = ptr<const float>(op, 0, bases); a.eps = ptr<const float>(op, 1, bases); a.dz = ptr<const float>(op, 2, bases);
      a.dmulv = ptr<float>(op, 3, bases); a.B = I[0]; a.zd = I[1]; a.lddz = I[2]; a.beta = op.f[0];
      e.a.rp = a; e.gx = blocks_for((int64_t)I[0] * I[1]);
      return true;
    }
    case HP_OP_MSE_FWD_BWD: {
      MseArgs a{ptr<const float>(op, 0, bases), ptr<const float>(op, 1, bases), ptr<float>(op, 2, bases), ptr<double>(op, 3, bases), I[0], I[1], op.f[0]};
      e.a.mse = a; e.gx = blocks_for(I[0]);
      return true;
    }
    case HP_OP_LOSS_FINALIZE: {
      LossArgs a{ptr<const double>(op, 0, bases), ptr<float>(op, 1, bases), I[0], I[1], I[2], op.f[0], op.f[1], op.f[2]};
      e.a.loss = a;
      return true;
    }
    default:
      return false;
  }
}

// Small-leaf group (HP_FLAG_GROUP_SHIFT): the members are independent — every workgroup runs ONE virtual block of one
// member.  Only the leaf reductions the planner groups (hp::groupable) are dispatched here: a kernel carries the LDS of
// EVERY body it can reach (a first version went through a switch over all small bodies and ran two workgroups per CU: 56-61 us
// for what ten stand-alone launches do in 31 us).
template <bool MFMA>
__global__ __launch_bounds__(256) void small_group_kernel(const SmallEntry* __restrict__ entries, int n) {
  // (MFMA: some member is a matrix-core weight gradient; the scalar-only instantiation — every group at batch 512 — does not carry its
  // 16 KB of LDS and its registers)
  __shared__ __attribute__((aligned(16))) float smem[MFMA ? kLinWgLds : 4];
  int b = blockIdx.x;
  for (int k = 0; k < n; ++k) {
    const SmallEntry& e = entries[k];
    const int nb = e.gx * e.gy * e.gz;
    if (b < nb) {
      const int bx = b % e.gx, by = (b / e.gx) % e.gy, bz = b / (e.gx * e.gy);
      if (e.op == HP_OP_LINEAR_BWD_W) {
        if (MFMA && e.variant == 1) {
          const LinWg w = e.a.wg;      // by value: one scalar load of the record up front
          lin_wgrad_body(w, bx, by, smem);
        } else {
          linear_bwd_w_body(e.a.lin, e.rows_per_z, bx, by, bz);
        }
      } else if (e.op == HP_OP_EMB_BWD) {
        emb_bwd_body(e.a.emb, bx);
      }
      return;
    }
    b -= nb;
  }
}

// a small-leaf group of HP_OP_WFRAG records (a group holds only such records): all conv weights of a model in one launch
__global__ __launch_bounds__(256) void wfrag_group_kernel(const SmallEntry* __restrict__ entries, int n) {
  __shared__ float tile[32 * 33];
  int b = blockIdx.x;
  for (int k = 0; k < n; ++k) {
    const int nb = entries[k].gx;
    if (b < nb) {
      const WfragArgs w = entries[k].a.wf;      // by value: one scalar load of the record up front
      wfrag_body(w, b, tile);
      return;
    }
    b -= nb;
  }
}

}  // namespace

template <int V> inline int rows_per_thread(const BnApplyArgs& a) { return bn_rows<V>(a.M, a.C); }
template <int V> inline int rows_per_thread(const BnBwdApplyArgs& a) { return bn_rows<V>(a.M, a.C); }
template <int V> inline int rows_per_thread(const BnBwdReduceArgs& a) { return a.rpl; }
// one launch for two independent BatchNorm-family ops of the same opcode and vector width
#define HP_PAIR_CASE(OPCODE, NAME, ARGFN)                                                                       \
  case OPCODE: {                                                                                                \
    const auto a = ARGFN(opa, bases);                                                                           \
    const auto b = ARGFN(opb, bases);                                                                           \
    if ((a.C % 4 == 0) != (b.C % 4 == 0)) return hipErrorInvalidValue;                                          \
    if (a.C % 4 == 0) {                                                                                         \
      const dim3 ga = colgrid_v<4>(a.M, a.C, rows_per_thread<4>(a)), gb = colgrid_v<4>(b.M, b.C, rows_per_thread<4>(b)); \
      hipLaunchKernelGGL(NAME##_pair_kernel<4>, dim3(ga.x * ga.y + gb.x * gb.y), dim3(256), 0, s, a, b,         \
                         (int)ga.x, (int)(ga.x * ga.y), (int)gb.x);                                             \
    } else {                                                                                                    \
      const dim3 ga = colgrid_v<1>(a.M, a.C, rows_per_thread<1>(a)), gb = colgrid_v<1>(b.M, b.C, rows_per_thread<1>(b)); \
      hipLaunchKernelGGL(NAME##_pair_kernel<1>, dim3(ga.x * ga.y + gb.x * gb.y), dim3(256), 0, s, a, b,         \
                         (int)ga.x, (int)(ga.x * ga.y), (int)gb.x);                                             \
    }                                                                                                           \
    break;                                                                                                      \
  }
hipError_t hp::launch_small_pair(const HpOp& opa, const HpOp& opb, void* const* bases, hipStream_t s) {
  if (opa.op != opb.op) return hipErrorInvalidValue;
  switch (opa.op) {
    HP_PAIR_CASE(HP_OP_BN_APPLY, bn_apply, bn_apply_args)
    HP_PAIR_CASE(HP_OP_BN_BWD_REDUCE, bn_bwd_reduce, bn_bwd_reduce_args)
    HP_PAIR_CASE(HP_OP_BN_BWD_APPLY, bn_bwd_apply, bn_bwd_apply_args)
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
#undef HP_PAIR_CASE

hipError_t hp::launch_small(const HpOp& op, void* const* bases, hipStream_t s) {
  using hp::ptr;
  const int32_t* I = op.i;
  const int abf = (op.flags & HP_FLAG_ACT_BF16) ? 1 : 0;      // activation-typed buffers hold bf16 (include/hippie_hip.h)
  switch (op.op) {
    case HP_OP_SLAB_REDUCE: {
      const int n = I[0];
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(min(2048, max(1, blocks_for(n >> 2)))), dim3(256), 0, s,
                         ptr<const float>(op, 0, bases), ptr<float>(op, 1, bases), n, I[1], I[2]);
      break;
    }
    case HP_OP_BN_APPLY: {
      const BnApplyArgs a = bn_apply_args(op, bases);
      if (a.C % 4 == 0) hipLaunchKernelGGL(bn_apply_kernel<4>, colgrid_v<4>(a.M, a.C, bn_rows<4>(a.M, a.C)), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(bn_apply_kernel<1>, colgrid_v<1>(a.M, a.C, bn_rows<1>(a.M, a.C)), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_BN_BWD_REDUCE: {
      const BnBwdReduceArgs a = bn_bwd_reduce_args(op, bases);
      if (a.C % 4 == 0) hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, colgrid_v<4>(a.M, a.C, a.rpl), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, colgrid_v<1>(a.M, a.C, a.rpl), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_BN_BWD_APPLY: {
      const BnBwdApplyArgs a = bn_bwd_apply_args(op, bases);
      if (a.C % 4 == 0) hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, colgrid_v<4>(a.M, a.C, bn_rows<4>(a.M, a.C)), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, colgrid_v<1>(a.M, a.C, bn_rows<1>(a.M, a.C)), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_STEM_FWD: {
      StemArgs a{};
      a.x = ptr<const float>(op, 0, bases); a.w = ptr<const float>(op, 1, bases); a.out = ptr<float>(op, 2, bases);
      a.stats = ptr<double>(op, 3, bases); a.B = I[0]; a.Lin = I[1]; a.Lout = I[2]; a.C = I[3]; a.abf = (op.flags & HP_FLAG_ACT_BF16) ? 1 : 0;
      hipLaunchKernelGGL(stem_fwd_kernel, colgrid(a.B * a.Lout, a.C), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_STEM_WGRAD: {
      StemArgs a{};
      a.dr = ptr<const float>(op, 0, bases); a.x = ptr<const float>(op, 1, bases); a.dw = ptr<float>(op, 2, bases);
      a.B = I[0]; a.Lin = I[1]; a.Lout = I[2]; a.C = I[3]; a.abf = (op.flags & HP_FLAG_ACT_BF16) ? 1 : 0;
      if (a.C % 4 == 0) {
        // flags & 1: ONE workgroup per column group walks all rows — no cross-workgroup atomics, bit-reproducible
        const int rows = (op.flags & 1) ? hp::cdiv(hp::cdiv(a.B * a.Lout, 16), 8) * 8 : small_wgrad_rows(kStemRows4);
        hipLaunchKernelGGL(stem_wgrad4_kernel, colgrid_v<4>(a.B * a.Lout, a.C, rows), dim3(256), 0, s, a, rows);
      } else {
        const int rows = small_wgrad_rows(kStemRows);
        hipLaunchKernelGGL(stem_wgrad_kernel, colgrid(a.B * a.Lout, a.C, rows), dim3(256), 0, s, a, rows);
      }
      break;
    }
    case HP_OP_POOL_FWD:
      hipLaunchKernelGGL(pool_fwd_kernel, dim3(blocks_for((int64_t)I[0] * I[2])), dim3(256), 0, s,
                         ptr<const float>(op, 0, bases), ptr<float>(op, 1, bases), I[0], I[1], I[2], abf);
      break;
    case HP_OP_POOL_BWD:
      hipLaunchKernelGGL(pool_bwd_kernel, dim3(blocks_for((int64_t)I[0] * I[1] * I[2])), dim3(256), 0, s,
                         ptr<const float>(op, 0, bases), ptr<float>(op, 1, bases), I[0], I[1], I[2], abf);
      break;
    case HP_OP_REPEAT_FWD:
      hipLaunchKernelGGL(repeat_fwd_kernel, dim3(blocks_for((int64_t)I[0] * I[1] * I[2])), dim3(256), 0, s,
                         ptr<const float>(op, 0, bases), ptr<float>(op, 1, bases), I[0], I[1], I[2], abf);
      break;
    case HP_OP_REPEAT_BWD:
      hipLaunchKernelGGL(repeat_bwd_kernel, dim3(blocks_for((int64_t)I[0] * I[2])), dim3(256), 0, s,
                         ptr<const float>(op, 0, bases), I[3] ? ptr<const float>(op, 1, bases) : nullptr,
                         ptr<float>(op, 2, bases), I[0], I[1], I[2], abf);
      break;
    case HP_OP_CONCAT: {
      ConcatArgs a{};
      a.out = ptr<float>(op, 0, bases); a.B = I[0]; a.nseg = I[1]; a.ldo = I[2];
      for (int j = 0; j < 4; ++j) {
        a.kind[j] = I[4 + 3 * j]; a.w[j] = I[5 + 3 * j]; a.ld[j] = I[6 + 3 * j]; a.rows[j] = I[16 + j];
        a.src[j] = ptr<const float>(op, 1 + 2 * j, bases); a.idx[j] = ptr<const int64_t>(op, 2 + 2 * j, bases);
      }
      hipLaunchKernelGGL(concat_kernel, dim3(blocks_for((int64_t)a.B * a.ldo)), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_EMB_BWD: {
      SmallEntry e;
      small_entry(op, bases, e);
      hipLaunchKernelGGL(emb_bwd_kernel, dim3(e.gx), dim3(256), 0, s, e.a.emb);
      break;
    }
    case HP_OP_WFRAG: {
      SmallEntry e;
      small_entry(op, bases, e);
      hipLaunchKernelGGL(wfrag_kernel, dim3(e.gx), dim3(256), 0, s, e.a.wf);
      break;
    }
    case HP_OP_LINEAR_FWD: {
      LinArgs a{};
      a.X = ptr<const float>(op, 0, bases); a.W = ptr<const float>(op, 1, bases); a.Bv = ptr<const float>(op, 2, bases);
      a.Y = ptr<float>(op, 3, bases); a.stats = I[6] ? ptr<double>(op, 4, bases) : nullptr;
      a.M = I[0]; a.N = I[1]; a.K = I[2]; a.ldx = I[3]; a.ldy = I[4]; a.act = I[5]; a.slope = op.f[0];
      if (a.M >= lin_mfma_min_rows()) {
        LinMM g{};
        g.A = a.X; g.lda = a.ldx; g.B = a.W; g.ldb = a.K; g.C = a.Y; g.ldc = a.ldy;
        g.M = a.M; g.N = a.N; g.Kc = a.K; g.alA = lin_align(g.A, g.lda); g.alB = lin_align(g.B, g.ldb);
        g.bias = a.Bv; g.stats = a.stats; g.act = a.act; g.slope = a.slope;
        hipLaunchKernelGGL(lin_mm_kernel<false>, dim3(hp::cdiv(g.M, 64) * hp::cdiv(g.N, 64)), dim3(kLinMMThreads), 0, s, g);
        break;
      }
      if (a.K >= 128) hipLaunchKernelGGL(linear_fwd_wave_kernel, dim3(blocks_for((int64_t)a.M * a.N, 4)), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(linear_fwd_thread_kernel, dim3(blocks_for((int64_t)a.M * a.N)), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_LINEAR_BWD_X: {
      LinArgs a{};
      a.DY = ptr<const float>(op, 0, bases); a.W = ptr<const float>(op, 1, bases); a.DX = ptr<float>(op, 2, bases);
      a.ACT = ptr<const float>(op, 3, bases);
      a.M = I[0]; a.N = I[1]; a.K = I[2]; a.ldy = I[3]; a.ldx = I[4]; a.has_mask = I[5]; a.lda = I[6]; a.accumulate = I[7];
      a.slope = op.f[0];
      if (a.M >= lin_mfma_min_rows()) {
        // DX[m][k] = sum_n DY[m][n] * W[n][k]: output columns k, contraction over the layer's N, W read as its transpose
        LinMM g{};
        g.A = a.DY; g.lda = a.ldy; g.B = a.W; g.ldb = a.K; g.C = a.DX; g.ldc = a.ldx;
        g.M = a.M; g.N = a.K; g.Kc = a.N; g.alA = lin_align(g.A, g.lda); g.alB = lin_align(g.B, g.ldb);
        g.mask = a.has_mask ? a.ACT : nullptr; g.ldm = a.lda; g.accumulate = a.accumulate; g.slope = a.slope;
        hipLaunchKernelGGL(lin_mm_kernel<true>, dim3(hp::cdiv(g.M, 64) * hp::cdiv(g.N, 64)), dim3(kLinMMThreads), 0, s, g);
        break;
      }
      if (a.N >= 128 && (int64_t)a.M * a.K <= (1 << 20))
        hipLaunchKernelGGL(linear_bwd_x_wave_kernel, dim3(blocks_for((int64_t)a.M * a.K, 4)), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(blocks_for((int64_t)a.M * a.K)), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_LINEAR_BWD_W: {
      if (I[0] >= lin_mfma_min_rows()) {
        const LinWg w = lin_wgrad_args(op, bases);
        const int ntiles = hp::cdiv(w.N, 64) * hp::cdiv(w.K, 64);
        hipLaunchKernelGGL(lin_wgrad_kernel, dim3(ntiles * hp::cdiv(w.M, w.rows_per_split)), dim3(256), 0, s, w, ntiles);
        break;
      }
      LinArgs a{};
      a.DY = ptr<const float>(op, 0, bases); a.X = ptr<const float>(op, 1, bases); a.DW = ptr<float>(op, 2, bases);
      a.DB = ptr<float>(op, 3, bases);
      a.M = I[0]; a.N = I[1]; a.K = I[2]; a.ldy = I[3]; a.ldx = I[4];
      const int kw = a.K < 256 ? a.K : 256;
      const int ky = hp::cdiv(a.K, kw);
      const int nz = linear_bwd_w_slices(a.M, a.N, a.K, op.flags & 1);
      const int rows_per_z = hp::cdiv(a.M, nz);
      dim3 grid(a.N, ky, hp::cdiv(a.M, rows_per_z));
      hipLaunchKernelGGL(linear_bwd_w_kernel, grid, dim3(256), 0, s, a, rows_per_z);
      break;
    }
    case HP_OP_REPARAM_KL_FWD: {
      SmallEntry e;
      small_entry(op, bases, e);
      hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(e.gx), dim3(256), 0, s, e.a.rp);
      break;
    }
    case HP_OP_REPARAM_KL_BWD: {
      SmallEntry e;
      small_entry(op, bases, e);
      hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(e.gx), dim3(256), 0, s, e.a.rp);
      break;
    }
    case HP_OP_MSE_FWD_BWD: {
      SmallEntry e;
      small_entry(op, bases, e);
      hipLaunchKernelGGL(mse_kernel, dim3(e.gx), dim3(256), 0, s, e.a.mse);
      break;
    }
    case HP_OP_TAIL_FWD:
      hipLaunchKernelGGL(tail_fwd_kernel, dim3(blocks_for((int64_t)I[0] * 2 * I[1], 4)), dim3(256), 0, s,
                         ptr<const float>(op, 0, bases), ptr<const float>(op, 1, bases), ptr<const float>(op, 2, bases),
                         ptr<float>(op, 3, bases), I[0], I[1], I[2], abf);
      break;
    case HP_OP_TAIL_BWD_X:
      hipLaunchKernelGGL(tail_bwd_x_kernel, dim3(blocks_for((int64_t)I[0] * I[1] * I[2])), dim3(256), 0, s,
                         ptr<const float>(op, 0, bases), ptr<const float>(op, 1, bases), ptr<float>(op, 2, bases), I[0], I[1], I[2], abf);
      break;
    case HP_OP_TAIL_BWD_W: {
      const int cw = I[2] < 256 ? I[2] : 256;
      if (I[2] % 4 == 0) {
        const int rows = (op.flags & 1) ? hp::cdiv(hp::cdiv(I[0] * I[1], 16), 8) * 8 : small_wgrad_rows(kTailRows4);      // flags & 1: see STEM_WGRAD
        hipLaunchKernelGGL(tail_bwd_w4_kernel, colgrid_v<4>(I[0] * I[1], I[2], rows), dim3(256), 0, s, ptr<const float>(op, 0, bases),
                           ptr<const float>(op, 1, bases), ptr<float>(op, 2, bases), ptr<float>(op, 3, bases), I[0], I[1], I[2], rows, abf);
        break;
      }
      const int rows = small_wgrad_rows(kTailRows);
      const dim3 grid(hp::cdiv(I[0] * I[1], (256 / cw) * rows), hp::cdiv(I[2], cw));
      hipLaunchKernelGGL(tail_bwd_w_kernel, grid, dim3(256), 0, s, ptr<const float>(op, 0, bases),
                         ptr<const float>(op, 1, bases), ptr<float>(op, 2, bases), ptr<float>(op, 3, bases), I[0], I[1], I[2], rows, abf);
      break;
    }
    case HP_OP_LOSS_FINALIZE: {
      SmallEntry e;
      small_entry(op, bases, e);
      hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, e.a.loss);
      break;
    }
    case HP_OP_GRADNORM:
      hipLaunchKernelGGL(gradnorm_kernel, dim3(min(256, max(1, blocks_for(I[0] >> 2)))), dim3(256), 0, s,
                         ptr<const float>(op, 0, bases), ptr<double>(op, 1, bases), I[0]);
      break;
    case HP_OP_ADAMW: {
      AdamArgs a;
      a.p = ptr<float>(op, 0, bases); a.g = ptr<const float>(op, 1, bases); a.m = ptr<float>(op, 2, bases);
      a.v = ptr<float>(op, 3, bases); a.step = ptr<const int64_t>(op, 4, bases); a.norm2 = ptr<const double>(op, 5, bases);
      a.n = I[0]; a.lr = op.f[0]; a.b1 = op.f[1]; a.b2 = op.f[2]; a.eps = op.f[3]; a.wd = op.f[4]; a.clip = op.f[5]; a.omb1 = op.f[6]; a.omb2 = op.f[7];
      hipLaunchKernelGGL(adamw_kernel, dim3(min(2048, max(1, blocks_for(a.n >> 2)))), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_STATS_SYNC:
      return hipSuccess;          // marker: the host sums buf[0] over the data-parallel ranks at this point
    case HP_OP_HEADS:
      return hipSuccess;          // closing record of a fused chain: executed one by one (hp_run_op), its members did the work
    case HP_OP_SF_SCHEDULE:
      hipLaunchKernelGGL(sf_schedule_kernel, dim3(1), dim3(64), 0, s, ptr<const int64_t>(op, 0, bases), ptr<double>(op, 1, bases),
                         I[0], op.f[0], op.f[1], op.f[2], op.f[3]);
      break;
    case HP_OP_ADAMW_SF: {
      SfArgs a;
      a.y = ptr<float>(op, 0, bases); a.g = ptr<const float>(op, 1, bases); a.z = ptr<float>(op, 2, bases);
      a.v = ptr<float>(op, 3, bases); a.step = ptr<const int64_t>(op, 4, bases); a.st = ptr<const double>(op, 5, bases);
      a.norm2 = ptr<const double>(op, 6, bases);
      a.n = I[0]; a.b1 = op.f[0]; a.b2 = op.f[1]; a.eps = op.f[2]; a.wd = op.f[3]; a.clip = op.f[4]; a.omb2 = op.f[5];
      hipLaunchKernelGGL(adamw_sf_kernel, dim3(min(2048, max(1, blocks_for(a.n >> 2)))), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_LERP:
      hipLaunchKernelGGL(lerp_kernel, dim3(min(2048, max(1, blocks_for(I[0])))), dim3(256), 0, s, ptr<float>(op, 0, bases),
                         ptr<const float>(op, 1, bases), I[0], op.f[0]);
      break;
    case HP_OP_RESAMPLE_LINEAR:
      hipLaunchKernelGGL(resample_linear_kernel, dim3(blocks_for((int64_t)I[0] * I[2])), dim3(256), 0, s,
                         ptr<const float>(op, 0, bases), ptr<float>(op, 1, bases), I[0], I[1], I[2], op.flags & 1);
      break;
    case HP_OP_STEP_INC:
      hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, s, ptr<int64_t>(op, 0, bases));
      break;
    case HP_OP_STAGE_BATCH: {
      StageArgs a;
      a.table = ptr<const float>(op, 0, bases); a.table2 = ptr<const float>(op, 1, bases); a.labels = ptr<const int64_t>(op, 2, bases);
      a.perm = ptr<const int64_t>(op, 3, bases); a.cursor = ptr<const int64_t>(op, 4, bases); a.x = ptr<float>(op, 5, bases);
      a.x2 = ptr<float>(op, 6, bases); a.src = ptr<int64_t>(op, 7, bases); a.eps = ptr<float>(op, 8, bases); a.seed = ptr<const int64_t>(op, 9, bases);
      a.B = I[0]; a.L = I[1]; a.L2 = a.table2 != nullptr ? I[2] : 0; a.z = I[3]; a.spe = I[4]; a.world = I[5]; a.rank = I[6]; a.N = I[7];
      const int64_t items = (int64_t)a.B * a.L + (int64_t)a.B * a.L2 + a.B + ((int64_t)a.B * a.z + 3) / 4;
      hipLaunchKernelGGL(stage_batch_kernel, dim3(blocks_for(items)), dim3(256), 0, s, a);
      break;
    }
    case HP_OP_ZERO: {
      uint8_t* dst = ptr<uint8_t>(op, 0, bases);
      const size_t nbytes = (size_t)(uint32_t)I[0] + ((size_t)(uint32_t)I[1] << 32);
      if (nbytes == 0) return hipSuccess;
      if (((uintptr_t)dst & 15) != 0) return hipMemsetAsync(dst, 0, nbytes, s);      // never the case for planner output
      const size_t n16 = nbytes >> 4;
      const int grid = (int)std::min<size_t>(2048, std::max<size_t>(1, (n16 + 255) / 256));
      hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<uint4*>(dst), n16, dst + (n16 << 4), (int)(nbytes & 15));
      break;
    }
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}


// ---- small-leaf group: host side ------------------------------------------------------------------------------------
bool hp::groupable(const HpOp& op) { return op.op == HP_OP_LINEAR_BWD_W || op.op == HP_OP_EMB_BWD || op.op == HP_OP_WFRAG; }

hipError_t hp::build_small_group(const HpOp* members, int count, void* const* bases, void** d_entries) {
  std::vector<SmallEntry> entries(count);
  for (int j = 0; j < count; ++j)
    if (!hp::groupable(members[j]) || !small_entry(members[j], bases, entries[j]) ||
        (members[j].op == HP_OP_WFRAG) != (members[0].op == HP_OP_WFRAG)) return hipErrorInvalidValue;      // (WFRAG records group among themselves)
  *d_entries = nullptr;
  hipError_t e = hipMalloc(d_entries, entries.size() * sizeof(SmallEntry));
  if (e != hipSuccess) return e;
  e = hipMemcpy(*d_entries, entries.data(), entries.size() * sizeof(SmallEntry), hipMemcpyHostToDevice);
  if (e != hipSuccess) { hipFree(*d_entries); *d_entries = nullptr; }
  return e;
}

hipError_t hp::launch_small_group(const HpOp* members, const void* d_entries, int count, hipStream_t s) {
  int blocks = 0;
  bool mfma = false;
  void* const zero_bases[HP_NUM_SPACES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int j = 0; j < count; ++j) {
    SmallEntry e;
    if (!small_entry(members[j], zero_bases, e)) return hipErrorInvalidValue;      // (grid only: the device table holds the pointers)
    blocks += e.gx * e.gy * e.gz;
    mfma = mfma || e.variant == 1;
  }
  if (members[0].op == HP_OP_WFRAG) {
    hipLaunchKernelGGL(wfrag_group_kernel, dim3(blocks), dim3(256), 0, s, (const SmallEntry*)d_entries, count);
    return hipGetLastError();
  }
  if (mfma) hipLaunchKernelGGL(small_group_kernel<true>, dim3(blocks), dim3(256), 0, s, (const SmallEntry*)d_entries, count);
  else hipLaunchKernelGGL(small_group_kernel<false>, dim3(blocks), dim3(256), 0, s, (const SmallEntry*)d_entries, count);
  return hipGetLastError();
}


// ---- fused heads: host side ---------------------------------------------------------------------------------------------------------
namespace {
struct HeadsDims { int B, Z; };
// The chain the kernels implement, record by record; `why` names the first mismatch.
bool heads_match(const HpOp* m, int count, int kind, HeadsDims& d, const char** why) {
  auto bad = [&](const char* w) { if (why) *why = w; return false; };
  static const int fwd[] = {HP_OP_CONCAT, HP_OP_LINEAR_FWD, HP_OP_BN_APPLY, HP_OP_LINEAR_FWD, HP_OP_BN_APPLY, HP_OP_LINEAR_FWD, HP_OP_REPARAM_KL_FWD,
                            HP_OP_CONCAT, HP_OP_LINEAR_FWD, HP_OP_LINEAR_FWD, HP_OP_BN_APPLY};
  const int* pat = fwd;
  const int n = 11;
  if (kind != 0) return bad("heads: kind must be 0 (the training forward chain)");
  if (count != n) return bad("heads: wrong number of member records");
  for (int k = 0; k < n; ++k) {
    if (m[k].op != pat[k]) return bad("heads: member opcodes are not the heads chain");
    if (!(m[k].flags & HP_FLAG_MEMBER)) return bad("heads: a member record lacks HP_FLAG_MEMBER");
  }
  constexpr int H = 5;
  auto lin = [&](const HpOp& o, int M, int N, int K, int ldx, int ldy) { return o.i[0] == M && o.i[1] == N && o.i[2] == K && o.i[3] == ldx && o.i[4] == ldy; };
  {
    const int B = m[0].i[0], Z = m[6].i[1], Z2 = 2 * Z, NC0 = Z2 + 2 * H, NC1 = Z + 2 * H;
    d.B = B; d.Z = Z;
    if (B < 1 || B > kHeadsThreads || (Z != 5 && Z != 10)) return bad("heads: needs 1 <= batch <= 512 and z_dim 5 or 10");
    const HpOp& c0 = m[0];
    const HpOp& c1 = m[7];
    auto cat_ok = [&](const HpOp& c, int w0, int ldo) {
      return c.i[0] == B && c.i[1] == 3 && c.i[2] == ldo && c.i[4] == 0 && c.i[5] == w0 && c.i[6] == w0 && c.i[7] == 1 && c.i[8] == H && c.i[9] == H &&
             (c.i[10] == 1 || c.i[10] == 2) && c.i[11] == H && c.buf[3] != HP_NULL && c.buf[4] != HP_NULL && (c.i[10] == 2 || (c.i[12] == H && c.buf[5] != HP_NULL && c.buf[6] != HP_NULL));
    };
    if (!cat_ok(c0, Z2, NC0) || !cat_ok(c1, Z, NC1)) return bad("heads: a CONCAT member is not (dense, source embedding, class embedding | zeros)");
    if (c0.buf[3] != c1.buf[3] || c0.buf[4] != c1.buf[4] || c0.i[10] != c1.i[10] || c0.buf[5] != c1.buf[5] || c0.buf[6] != c1.buf[6] || c0.i[16 + 1] != c1.i[16 + 1])
      return bad("heads: the two CONCAT members use different embedding tables / labels");
    if (!lin(m[1], B, Z2, NC0, NC0, Z2) || !m[1].i[6] || m[1].i[5] || !lin(m[3], B, Z, Z2, Z2, Z) || !m[3].i[6] || m[3].i[5] || !lin(m[5], B, Z2, Z, Z, Z2) || m[5].i[6] ||
        m[5].i[5] || !lin(m[8], B, Z2, NC1, NC1, Z2) || m[8].i[6] || !m[8].i[5] || !lin(m[9], B, Z2, Z2, Z2, Z2) || !m[9].i[6] || m[9].i[5])
      return bad("heads: a LINEAR_FWD member has an unexpected shape / flags");
    auto bn_ok = [&](const HpOp& o, int C) { return o.i[0] == B && o.i[1] == C && o.i[2] == 0 && o.i[3] == 1 && o.i[4] == 1 && o.i[5] <= 1; };
    if (!bn_ok(m[2], Z2) || !bn_ok(m[4], Z) || !bn_ok(m[10], Z2)) return bad("heads: a BN_APPLY member is not (training, LeakyReLU, no residual, local statistics)");
    if (m[6].i[0] != B) return bad("heads: REPARAM_KL_FWD batch");
    // data flow: every member consumes its predecessor's output, statistics slots match
    const bool flow = m[1].buf[0] == c0.buf[0] && m[2].buf[0] == m[1].buf[3] && m[2].buf[2] == m[1].buf[4] && m[3].buf[0] == m[2].buf[1] &&
                      m[4].buf[0] == m[3].buf[3] && m[4].buf[2] == m[3].buf[4] && m[5].buf[0] == m[4].buf[1] && m[6].buf[0] == m[5].buf[3] &&
                      c1.buf[1] == m[6].buf[2] && m[8].buf[0] == c1.buf[0] && m[9].buf[0] == m[8].buf[3] && m[10].buf[0] == m[9].buf[3] && m[10].buf[2] == m[9].buf[4];
    if (!flow) return bad("heads: the forward members do not feed each other");
    if (m[2].f[0] != m[4].f[0] || m[2].f[0] != m[10].f[0] || m[2].f[0] != m[8].f[0] || m[2].f[1] != m[4].f[1] || m[2].f[1] != m[10].f[1] || m[2].f[2] != m[4].f[2] || m[2].f[2] != m[10].f[2])
      return bad("heads: the members disagree on slope / eps / momentum");
    return true;
  }
}
HeadsBn heads_bn_fwd(const HpOp& o, void* const* bases) {
  using hp::ptr;
  return HeadsBn{ptr<const float>(o, 3, bases), ptr<const float>(o, 4, bases), ptr<float>(o, 5, bases), ptr<float>(o, 6, bases), ptr<float>(o, 7, bases)};
}
}  // namespace

bool hp::check_heads(const HpOp* members, int count, int kind, const char** why) {
  HeadsDims d;
  return heads_match(members, count, kind, d, why);
}

hipError_t hp::launch_heads(const HpOp* m, int count, int kind, void* const* bases, hipStream_t s) {
  using hp::ptr;
  HeadsDims d;
  if (!heads_match(m, count, kind, d, nullptr)) return hipErrorInvalidValue;
  {
    HeadsFwd a{};
    a.B = d.B; a.n_src = m[0].i[16 + 1]; a.n_cls = m[0].i[10] == 1 ? m[0].i[16 + 2] : 0;
    a.slope = m[2].f[0]; a.eps = m[2].f[1]; a.momentum = m[2].f[2];
    a.h = ptr<const float>(m[0], 1, bases);
    a.semb = ptr<const float>(m[0], 3, bases); a.src = ptr<const int64_t>(m[0], 4, bases);
    a.cemb = m[0].i[10] == 1 ? ptr<const float>(m[0], 5, bases) : nullptr; a.cls = m[0].i[10] == 1 ? ptr<const int64_t>(m[0], 6, bases) : nullptr;
    a.c0 = ptr<float>(m[0], 0, bases);
    a.w0 = ptr<const float>(m[1], 1, bases); a.b0 = ptr<const float>(m[1], 2, bases); a.u1 = ptr<float>(m[1], 3, bases); a.st1 = ptr<double>(m[1], 4, bases);
    a.bn1 = heads_bn_fwd(m[2], bases); a.a1 = ptr<float>(m[2], 1, bases);
    a.w3 = ptr<const float>(m[3], 1, bases); a.b3 = ptr<const float>(m[3], 2, bases); a.u2 = ptr<float>(m[3], 3, bases); a.st2 = ptr<double>(m[3], 4, bases);
    a.bn4 = heads_bn_fwd(m[4], bases); a.encv = ptr<float>(m[4], 1, bases);
    a.wz = ptr<const float>(m[5], 1, bases); a.bz = ptr<const float>(m[5], 2, bases); a.mulv = ptr<float>(m[5], 3, bases);
    a.epsn = ptr<const float>(m[6], 1, bases); a.zz = ptr<float>(m[6], 2, bases); a.loss = ptr<double>(m[6], 3, bases);
    a.c1 = ptr<float>(m[7], 0, bases);
    a.wf0 = ptr<const float>(m[8], 1, bases); a.bf0 = ptr<const float>(m[8], 2, bases); a.u3 = ptr<float>(m[8], 3, bases);
    a.wf2 = ptr<const float>(m[9], 1, bases); a.bf2 = ptr<const float>(m[9], 2, bases); a.u4 = ptr<float>(m[9], 3, bases); a.st4 = ptr<double>(m[9], 4, bases);
    a.bn3 = heads_bn_fwd(m[10], bases); a.dv = ptr<float>(m[10], 1, bases);
    if (d.Z == 10) hipLaunchKernelGGL((heads_fwd_kernel<10, 5>), dim3(1), dim3(kHeadsThreads), 0, s, a);
    else hipLaunchKernelGGL((heads_fwd_kernel<5, 5>), dim3(1), dim3(kHeadsThreads), 0, s, a);
    return hipGetLastError();
  }
}
