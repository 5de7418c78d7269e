// The cVAE's heads between the two backbones as ONE launch of ONE workgroup (training forward, batches of up to 512 rows):
//   hippie/model.py:51-72:  cat(enc, source_emb, class_emb) -> encoder_fc (Linear + BatchNorm + LeakyReLU(0.2), twice)
//   -> z_mean | z_log_var -> reparameterize + KL (:46-49, :104) -> cat(z, source_emb, class_emb)
//   -> decoder_fc (Linear + LeakyReLU, Linear + BatchNorm + LeakyReLU)
// At batch 512 these are 11 launches of 2-5 us each — every one a dependent-launch floor plus two L2 round trips — on the critical
// path of a model-step; their arithmetic is 1 800 multiply-adds per row.  Here a thread owns a ROW and keeps it in registers from one
// layer to the next (all widths are compile-time: 2z + 2H, 2z, z, z + 2H floats); the layers' weights are uniform across the workgroup
// (scalar loads through the constant address space); the only cross-row steps are the BatchNorm statistics — fp64 column sums over a
// padded LDS image of the rows, fixed order — and the KL sum.  Every tensor the un-fused ops write (the backward pass and the tests
// read them) is still written, straight from the registers.
//
// The arithmetic per element is that of the generic kernels, expression for expression (linear_fwd_thread_body: fmaf chain in k order,
// then + bias; bn_apply: lrelu(fmaf(x, scale, shift)); bn_coef_sums shared with them; reparam_kl_fwd_body); only the ORDER of the fp64
// statistic sums differs (a fixed tree here, atomics there).
//
// Measured (profiles/r04_heads_fused.txt): 30.2 us against 38.8 us for the eleven launches (one CU's VALU is the bound: ~12 k dynamic
// instructions per wave, two waves per SIMD), -9 us of a 2.74 ms model-step; the pair-step is unchanged within its spread.  A backward
// twin (twelve launches, 42 us) was built and tested to the same 1e-6 but reached only 94 us: the transposed weight access leaves no
// contiguous scalar loads, and every LDS / register formulation tried made the compiler keep whole layers of weights live (256 VGPRs
// + 380 spilled); removed.
//
// The planner emits the un-fused records as MEMBERS (HP_FLAG_MEMBER: hp_run_op and the interpreter execute them one by one) closed
// by an HP_OP_HEADS record; the executor launches this kernel instead.  ops_small.hip decodes the members into HeadsFwd and refuses
// any chain that is not exactly this one.
#pragma once
#include "hp_mfma.h"
#include <type_traits>

struct HeadsBn { const float* gamma; const float* beta; float* rmean; float* rvar; float* save; };

struct HeadsFwd {
  int B, n_src, n_cls;
  float slope, eps, momentum;
  const float* h;                               // [B][2z] encoder.linear output
  const float* semb; const int64_t* src; const float* cemb; const int64_t* cls;     // cemb == nullptr: zeros (no class labels)
  float* c0;
  const float* w0; const float* b0; float* u1; double* st1; HeadsBn bn1; float* a1;
  const float* w3; const float* b3; float* u2; double* st2; HeadsBn bn4; float* encv;
  const float* wz; const float* bz; float* mulv;
  const float* epsn; float* zz; double* loss;
  float* c1;
  const float* wf0; const float* bf0; float* u3;
  const float* wf2; const float* bf2; float* u4; double* st4; HeadsBn bn3; float* dv;
};

constexpr int kHeadsThreads = 512;

// ---- shared pieces ------------------------------------------------------------------------------------------------------------
// A thread's row of W contiguous floats straight between global memory and its registers, in the widest pieces the row's alignment
// allows (rows are dense, 256-byte-aligned tensors: W % 4 == 0 -> 16-byte pieces, W % 2 == 0 -> 8-byte, else 4).  One wave instruction
// then touches 64 rows W*4 bytes apart — ~40 cache lines — but there are only W/4 of them per tensor; the first version staged every
// tensor through an LDS image for coalesced access and spent 5x the arithmetic on index math, LDS traffic and barriers (39 us forward
// against 39 us for the eleven launches it replaced).
__device__ __forceinline__ void gstore2(float* p, const float a, const float b) {
  hp_v2f t; t.x = a; t.y = b;
  *(hp_v2f __attribute__((address_space(1)))*)(p) = t;
}
template <int W>
__device__ __forceinline__ void heads_row_load(const float* g, const int row, float (&v)[W]) {
  const float* p = g + (size_t)row * W;
  if (W % 4 == 0) {
#pragma unroll
    for (int j = 0; j < W; j += 4) { const float4 q = gload4(p + j); v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w; }
  } else if (W % 2 == 0) {
#pragma unroll
    for (int j = 0; j < W; j += 2) { const float2 q = gload2(p + j); v[j] = q.x; v[j + 1] = q.y; }
  } else {
#pragma unroll
    for (int j = 0; j < W; ++j) v[j] = gload1(p + j);
  }
}
template <int W>
__device__ __forceinline__ void heads_row_store(float* g, const int row, const float (&v)[W]) {
  float* p = g + (size_t)row * W;
  if (W % 4 == 0) {
#pragma unroll
    for (int j = 0; j < W; j += 4) gstore4(p + j, make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]));
  } else if (W % 2 == 0) {
#pragma unroll
    for (int j = 0; j < W; j += 2) gstore2(p + j, v[j], v[j + 1]);
  } else {
#pragma unroll
    for (int j = 0; j < W; ++j) gstore1(p + j, v[j]);
  }
}
template <int W, int WP>
__device__ __forceinline__ void heads_put_row(float* tile, const int row, const float (&v)[W]) {
#pragma unroll
  for (int j = 0; j < W; ++j) tile[row * WP + j] = v[j];
}
template <int W, int WP>
__device__ __forceinline__ void heads_get_row(const float* tile, const int row, float (&v)[W]) {
#pragma unroll
  for (int j = 0; j < W; ++j) v[j] = tile[row * WP + j];
}
// Weights and biases are uniform across the workgroup and never written by these kernels: read through the CONSTANT address space,
// i.e. with scalar loads into SGPRs (a pointer that arrives inside the by-value argument struct is generic to the compiler, which
// then loads every weight into a VGPR of every lane: 450 vector loads and, in the backward kernel, 400 spilled registers).
typedef const float __attribute__((address_space(4))) * heads_cptr;
__device__ __forceinline__ heads_cptr heads_const(const float* p) { return (heads_cptr)(p); }

// y[n] = sum_k x[k] * w[n*K + k] (+ b[n]) in k order, as linear_fwd_thread_body
template <int N, int K>
__device__ __forceinline__ void heads_linear(const float (&x)[K], const float* wg, const float* bg, float (&y)[N]) {
  const heads_cptr w = heads_const(wg), b = heads_const(bg);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s = fmaf(x[k], w[n * K + k], s);
    if (bg != nullptr) s += b[n];
    y[n] = s;
    if (n % 2 == 1) __builtin_amdgcn_sched_barrier(0);
  }
}
// Column sums over the B rows of NS interleaved float columns per channel of the LDS image: out[s*C + c] = sum_r f_s(row r, channel c), in
// fp64, fixed order (parts of rows in row order, then the parts in order).  F(row pointer, c, s) -> double term.
template <int C, int NS, typename F>
__device__ __forceinline__ void heads_colsum(const int B, double* part, double* out, F term) {
  constexpr int NP = kHeadsThreads / (C * NS);          // row parts per (statistic, channel)
  const int t = threadIdx.x;
  const int sc = t % (C * NS), p = t / (C * NS);
  if (p < NP) {
    double a = 0.0;
    for (int r = p; r < B; r += NP) a += term(r, sc % C, sc / C);
    part[sc * NP + p] = a;
  }
  __syncthreads();
  if (t < C * NS) {
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < NP; ++q) a += part[t * NP + q];
    out[t] = a;
  }
  __syncthreads();
}

constexpr int heads_wp(int w) { return w | 1; }

// ---- forward --------------------------------------------------------------------------------------------------------------------
template <int Z, int H>
__global__ __launch_bounds__(kHeadsThreads) void heads_fwd_kernel(HeadsFwd p) {
  constexpr int Z2 = 2 * Z, NC0 = 2 * Z + 2 * H, NC1 = Z + 2 * H;
  __shared__ float tile[kHeadsThreads * heads_wp(Z2)];      // the rows of one BatchNorm input, for its column sums
  __shared__ double part[kHeadsThreads], sums[2 * Z2];
  __shared__ float coef[2 * Z2];
  const int t = threadIdx.x, B = p.B;
  const bool live = t < B;
  const int row = live ? t : 0;

  // BatchNorm (training) + LeakyReLU over the rows' values v (raw -> activation, in place); also writes the raw and the activation
  // tensor, the statistics slot (replica 0 of a zeroed slot) and the layer's side effects
  auto batchnorm = [&](auto& v, auto cw, float* raw_out, double* st, const HeadsBn& bn, float* act_out) {
    constexpr int C = decltype(cw)::value;
    constexpr int WP = heads_wp(C);
    if (live) heads_row_store<C>(raw_out, t, v);
    __syncthreads();                                  // (the image's previous readers are done)
    if (live) heads_put_row<C, WP>(tile, t, v);
    __syncthreads();
    heads_colsum<C, 2>(B, part, sums, [&](int r, int c, int s) {
      const double x = (double)tile[r * WP + c];
      return s == 0 ? x : x * x;
    });
    if (t < C) {
      st[t] = sums[t];
      st[C + t] = sums[C + t];
      const BnCoef k = bn_coef_sums(true, B, sums[t], sums[C + t], bn.gamma[t], bn.beta[t], bn.rmean[t], bn.rvar[t], p.eps);
      coef[t] = k.scale; coef[C + t] = k.shift;
      bn_side_effects(k, B, C, t, bn.save, bn.rmean, bn.rvar, p.momentum);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = lrelu(fmaf(v[c], coef[c], coef[C + c]), p.slope);
    if (live) heads_row_store<C>(act_out, t, v);
  };
  using WZ2 = std::integral_constant<int, Z2>;
  using WZ = std::integral_constant<int, Z>;

  // cat(enc, source_emb, class_emb)
  float c0[NC0];
  float semb[H], cemb[H];
  {
    float hrow[Z2];
    heads_row_load<Z2>(p.h, row, hrow);
    const int64_t is = live ? p.src[t] : 0, ic = (live && p.cemb != nullptr) ? p.cls[t] : 0;
    const bool oks = is >= 0 && is < (int64_t)p.n_src, okc = p.cemb != nullptr && ic >= 0 && ic < (int64_t)p.n_cls;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      semb[j] = oks ? p.semb[is * H + j] : 0.f;
      cemb[j] = okc ? p.cemb[ic * H + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < Z2; ++j) c0[j] = hrow[j];
#pragma unroll
    for (int j = 0; j < H; ++j) { c0[Z2 + j] = semb[j]; c0[Z2 + H + j] = cemb[j]; }
  }
  if (live) heads_row_store<NC0>(p.c0, t, c0);

  // encoder_fc: Linear + BatchNorm + LeakyReLU, twice
  float u1[Z2];
  heads_linear<Z2, NC0>(c0, p.w0, p.b0, u1);
  batchnorm(u1, WZ2{}, p.u1, p.st1, p.bn1, p.a1);                 // u1 now holds a1
  float u2[Z];
  heads_linear<Z, Z2>(u1, p.w3, p.b3, u2);
  batchnorm(u2, WZ{}, p.u2, p.st2, p.bn4, p.encv);                // u2 now holds the embedding `enc`

  // z_mean | z_log_var, reparameterize, KL
  float mulv[Z2];
  heads_linear<Z2, Z>(u2, p.wz, p.bz, mulv);
  if (live) heads_row_store<Z2>(p.mulv, t, mulv);
  float c1[NC1];
  double kl = 0.0;
  {
    float e[Z], zrow[Z];
    heads_row_load<Z>(p.epsn, row, e);
#pragma unroll
    for (int j = 0; j < Z; ++j) {
      const float mu = mulv[j], lv = mulv[Z + j];
      zrow[j] = c1[j] = fmaf(e[j], expf(0.5f * lv), mu);
      if (live) kl += -0.5 * (double)(1.f + lv - mu * mu - expf(lv));
    }
#pragma unroll
    for (int j = 0; j < H; ++j) { c1[Z + j] = semb[j]; c1[Z + H + j] = cemb[j]; }
    if (live) heads_row_store<Z>(p.zz, t, zrow);
  }
  // KL sum: per wave, then the waves in order, one fp64 atomic (the loss slot is zeroed with the statistics)
  kl = wave_sum(kl);
  __syncthreads();
  if ((t & 63) == 0) part[t >> 6] = kl;
  __syncthreads();
  if (t == 0) {
    double s = 0.0;
    for (int w = 0; w < kHeadsThreads / 64; ++w) s += part[w];
    atomic_add_f64(p.loss, s);
  }
  if (live) heads_row_store<NC1>(p.c1, t, c1);

  // decoder_fc: Linear + LeakyReLU, Linear + BatchNorm + LeakyReLU
  float u3[Z2];
  heads_linear<Z2, NC1>(c1, p.wf0, p.bf0, u3);
#pragma unroll
  for (int j = 0; j < Z2; ++j) u3[j] = lrelu(u3[j], p.slope);
  if (live) heads_row_store<Z2>(p.u3, t, u3);
  float u4[Z2];
  heads_linear<Z2, Z2>(u3, p.wf2, p.bf2, u4);
  batchnorm(u4, WZ2{}, p.u4, p.st4, p.bn3, p.dv);
}
