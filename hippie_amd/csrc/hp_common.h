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
hipError_t launch_wgrad_group(int ntaps, const void* d_probs, const void* d_blocks, int nblocks, hipStream_t s);

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

// sum of one entry of a replicated statistics slot double[HP_STAT_REPL][2][C]
// Both entries (which = 0, 1) of channel c summed over the replicas.  All 2*HP_STAT_REPL loads are issued
// before anything consumes them (sched_barrier: otherwise the scheduler interleaves the adds and the prologue
// of every BatchNorm kernel becomes a chain of dependent L2 round trips, ~5 us), then two pairwise trees.
__device__ __forceinline__ void stat_sum2(const double* st, int C, int c, double& s0, double& s1) {
  const double* p = st + c;
  const size_t stride = (size_t)2 * C;
  double a[HP_STAT_REPL], b[HP_STAT_REPL];
#pragma unroll
  for (int r = 0; r < HP_STAT_REPL; ++r) {
    a[r] = p[r * stride];
    b[r] = p[r * stride + C];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int w = HP_STAT_REPL / 2; w > 0; w >>= 1) {
#pragma unroll
    for (int r = 0; r < w; ++r) { a[r] += a[r + w]; b[r] += b[r + w]; }
  }
  s0 = a[0];
  s1 = b[0];
}
__device__ __forceinline__ double stat_sum(const double* st, int C, int which, int c) {
  double s0, s1;
  stat_sum2(st, C, c, s0, s1);
  return which ? s1 : s0;
}
__device__ __forceinline__ double* stat_replica(double* st, int C, int key) {
  return st + (size_t)(key % HP_STAT_REPL) * 2 * C;
}

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }
// derivative expressed on the OUTPUT of leaky_relu (sign(out) == sign(in) for slope > 0)
__device__ __forceinline__ float lrelu_grad(float out, float slope) { return out > 0.f ? 1.f : slope; }

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
