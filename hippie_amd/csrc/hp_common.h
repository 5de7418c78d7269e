// Internal helpers shared by the kernel translation units of libhippie_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hippie_hip.h"

namespace hp {

inline int space_of(int64_t ref) { return (int)((uint64_t)ref >> 56); }
inline int64_t offset_of(int64_t ref) { return (int64_t)((uint64_t)ref & 0x00FFFFFFFFFFFFFFull); }

template <typename T>
inline T* ptr(const HpOp& op, int slot, void* const* bases) {
  int64_t r = op.buf[slot];
  if (r == HP_NULL) return nullptr;
  return reinterpret_cast<T*>(reinterpret_cast<char*>(bases[space_of(r)]) + offset_of(r));
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// launchers, one per op family (defined in conv_mfma.hip / ops_small.hip)
hipError_t launch_conv_taps(const HpOp& op, void* const* bases, hipStream_t s);
hipError_t launch_wgrad_taps(const HpOp& op, void* const* bases, hipStream_t s);
hipError_t launch_small(const HpOp& op, void* const* bases, hipStream_t s);
hipError_t launch_conv_pair(const HpOp& a, const HpOp& b, void* const* bases, hipStream_t s);
hipError_t launch_small_pair(const HpOp& a, const HpOp& b, void* const* bases, hipStream_t s);   // BN family
hipError_t build_wgrad_group(const HpOp* members, int count, void* const* bases, void** d_probs, void** d_blocks, int* nblocks);
hipError_t launch_wgrad_group(int ntaps, int bf16, const void* d_probs, const void* d_blocks, int nblocks, hipStream_t s);
// small-leaf group (ops_small.hip): independent LINEAR_BWD_W / EMB_BWD records in one launch
bool groupable(const HpOp& op);
hipError_t build_small_group(const HpOp* members, int count, void* const* bases, void** d_entries);
hipError_t launch_small_group(const HpOp* members, const void* d_entries, int count, hipStream_t s);
// fused heads (ops_small.hip / heads_fused.h): members of one HP_OP_HEADS record; check = the host-side pattern match alone
bool check_heads(const HpOp* members, int count, int kind, const char** why);
hipError_t launch_heads(const HpOp* members, int count, int kind, void* const* bases, hipStream_t s);

}  // namespace hp

// ---- device helpers -------------------------------------------------------------
__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
  // hardware global_atomic_add_f64 (no CAS loop); agent scope, relaxed
  unsafeAtomicAdd(p, v);
}
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
// same, for a pointer that was read from a record in memory (generic to the compiler: it would emit flat_atomic)
__device__ __forceinline__ void atomic_add_f32_global(float* p, float v) {
  __builtin_amdgcn_global_atomic_fadd_f32((float __attribute__((address_space(1)))*)p, v);
}

// Replicated statistics slot double[R][2][C], R = hp_stat_repl(C) (include/hippie_hip.h).
// Both entries (which = 0, 1) of channel c summed over the replicas.  All 2*R loads are issued before anything
// consumes them (sched_barrier: otherwise the scheduler interleaves the adds and the prologue of every BatchNorm
// consumer becomes a chain of dependent L2 round trips, ~5 us), then two pairwise trees.
template <int R>
__device__ __forceinline__ void stat_sum2_r(const double* st, int C, int c, double& s0, double& s1) {
  const double* p = st + c;
  const size_t stride = (size_t)2 * C;
  double a[R], b[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    a[r] = p[r * stride];
    b[r] = p[r * stride + C];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int w = R / 2; w > 0; w >>= 1) {
#pragma unroll
    for (int r = 0; r < w; ++r) { a[r] += a[r + w]; b[r] += b[r + w]; }
  }
  s0 = a[0];
  s1 = b[0];
}
__device__ __forceinline__ void stat_sum2(const double* st, int C, int c, double& s0, double& s1) {
  switch (hp_stat_repl(C)) {          // uniform: one of 2, 4, 8, 16
    case 2: stat_sum2_r<2>(st, C, c, s0, s1); break;
    case 4: stat_sum2_r<4>(st, C, c, s0, s1); break;
    case 8: stat_sum2_r<8>(st, C, c, s0, s1); break;
    default: stat_sum2_r<16>(st, C, c, s0, s1); break;
  }
}
__device__ __forceinline__ double* stat_replica(double* st, int C, int key) {
  return st + (size_t)(key % hp_stat_repl(C)) * 2 * C;
}

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }
// derivative expressed on the OUTPUT of leaky_relu (sign(out) == sign(in) for slope > 0)
__device__ __forceinline__ float lrelu_grad(float out, float slope) { return out > 0.f ? 1.f : slope; }

// ---- BatchNorm coefficients: ONE derivation shared by every kernel that normalises (HP_OP_BN_APPLY, the
// HP_CONV_IN_BN operand loader): identical inputs -> bit-identical (scale, shift) wherever they are re-derived.
struct BnCoef { float mean, invstd, scale, shift, rm, rv; double var; };

// every global load (gamma, beta, running stats, all statistic replicas) is issued before the first use: one
// memory round trip instead of one per operand
// ... from the two sums themselves (a kernel that holds them already: the fused heads kernel) — the same arithmetic as bn_coef
__device__ __forceinline__ BnCoef bn_coef_sums(bool training, int M, double s0, double s1, float g, float b, float rm, float rv, float eps) {
  BnCoef k;
  k.rm = rm; k.rv = rv;
  double mean, var;
  if (training) {
    mean = s0 / (double)M;
    var = s1 / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
  } else {
    mean = (double)k.rm;
    var = (double)k.rv;
  }
#ifdef HP_BN_FAST_RSQ      // timing experiment only (tools/micro/bn_sweep.py): how much of BN_APPLY is the fp64 sqrt + divide?
  const double invstd = (double)rsqrtf((float)(var + (double)eps));
#else
  const double invstd = 1.0 / sqrt(var + (double)eps);
#endif
  const double sc = (double)g * invstd;
  k.mean = (float)mean; k.invstd = (float)invstd; k.var = var;
  k.scale = (float)sc; k.shift = (float)((double)b - mean * sc);
  return k;
}
__device__ __forceinline__ BnCoef bn_coef(bool training, int M, const double* stats, int C, int c, const float* gamma,
                                          const float* beta, const float* rmean, const float* rvar, float eps) {
  const float g = gamma[c], b = beta[c];
  const float rm = rmean[c], rv = rvar[c];
  double s0 = 0.0, s1 = 0.0;
  if (training) stat_sum2(stats, C, c, s0, s1);
  return bn_coef_sums(training, M, s0, s1, g, b, rm, rv, eps);
}

// training-mode side effects of one channel: saved (mean, invstd) for the backward pass, running statistics
// (momentum update, unbiased variance); optionally the (scale, shift) pair for consumers that re-evaluate the
// activation from the raw tensor
__device__ __forceinline__ void bn_side_effects(const BnCoef& k, int M, int C, int c, float* save, float* rmean,
                                                float* rvar, float momentum, float* coef = nullptr) {
  save[c] = k.mean;
  save[C + c] = k.invstd;
  if (coef != nullptr) { coef[c] = k.scale; coef[C + c] = k.shift; }
  const double unb = M > 1 ? k.var * (double)M / (double)(M - 1) : k.var;
  rmean[c] = (float)((1.0 - (double)momentum) * (double)k.rm + (double)momentum * (double)k.mean);
  rvar[c] = (float)((1.0 - (double)momentum) * (double)k.rv + (double)momentum * unb);
}

// BatchNorm backward, input gradient: dr = sc * (g - c1 - xhat * c2) with xhat = (x - mean) * invstd, c1 = sum(g)/M,
// c2 = sum(g*xhat)/M, sc = gamma * invstd — as three per-channel coefficients formed in fp64 and two fused multiply-adds
// per element:  dr = fma(g, A, fma(x, B, C)),  A = sc,  B = -sc*c2*invstd,  C = sc*(c2*invstd*mean - c1).
// One derivation (bn_dr_coef) and one expression (bn_dr) shared by HP_OP_BN_BWD_APPLY and the HP_CONV_IN_DR operand
// loader: the same bits wherever dr is evaluated.
struct BnDrCoef { float A, B, C; };
__device__ __forceinline__ BnDrCoef bn_dr_coef(float mean, float invstd, float gamma, double sg, double sgx, int Mstat) {
  const double sc = (double)gamma * (double)invstd;
  const double c1 = sg / (double)Mstat, c2 = sgx / (double)Mstat;
  BnDrCoef k;
  k.A = (float)sc;
  k.B = (float)(-sc * c2 * (double)invstd);
  k.C = (float)(sc * (c2 * (double)invstd * (double)mean - c1));
  return k;
}
__device__ __forceinline__ float bn_dr(float g, float x, float A, float B, float C) {
  return __fmaf_rn(g, A, __fmaf_rn(x, B, C));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
