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

// sum of one entry of a replicated statistics slot double[HP_STAT_REPL][2][C]
__device__ __forceinline__ double stat_sum(const double* st, int C, int which, int c) {
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < HP_STAT_REPL; ++r) s += st[(size_t)r * 2 * C + which * C + c];
  return s;
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
