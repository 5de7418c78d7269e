// Device helpers shared by the matrix-core kernels (conv_mfma.hip, linear_mfma.h): global-address-space loads / stores, the
// XCD-aware tile order, the epilogue's column-statistics fold.  gfx950 only.
#pragma once
#include "hp_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16 bytes of zeros in device memory: padded / out-of-range rows load from here, so no select is needed
static __device__ __attribute__((aligned(16))) const float hp_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// 16-byte load through the GLOBAL address space.  Pointers that arrive inside a by-value argument struct (or a record in
// memory) are generic to the compiler: it emits flat_load, which (a) also counts against the LDS counter — every
// `s_waitcnt lgkmcnt(0)` in front of an LDS fragment read then waits for the global prefetches as well — and (b) returns
// out of order with LDS traffic, so the compiler can only ever wait with vmcnt(0): a two-slice prefetch degenerates to a
// synchronous load per K-step.  Everything these kernels load lives in device memory, so the cast is always valid.
typedef float hp_v4f __attribute__((ext_vector_type(4)));
typedef float hp_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 gload4(const float* p) {
  const hp_v4f v = *(const hp_v4f __attribute__((address_space(1)))*)(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gstore4(float* p, const float4 v) {
  hp_v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *(hp_v4f __attribute__((address_space(1)))*)(p) = t;
}
__device__ __forceinline__ float2 gload2(const float* p) {
  const hp_v2f v = *(const hp_v2f __attribute__((address_space(1)))*)(p);
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ float gload1(const float* p) { return *(const float __attribute__((address_space(1)))*)(p); }
__device__ __forceinline__ void gstore1(float* p, float v) { *(float __attribute__((address_space(1)))*)(p) = v; }

// ---- activation storage: fp32 or (HP_FLAG_ACT_BF16 on the op record) bfloat16 -------------------------------------------------------
// H = true: the tensor holds bf16 (2 bytes per element).  bf16 -> fp32 is a shift (exact); fp32 -> bf16 rounds to nearest even
// (v_cvt_pk_bf16_f32).  Pointers stay `float*` in the argument structs; element offsets are scaled here.
typedef unsigned hp_v2u __attribute__((ext_vector_type(2)));
typedef unsigned hp_v4u __attribute__((ext_vector_type(4)));
template <bool H>
__device__ __forceinline__ float4 aload4p(const char* p) {          // p -> four consecutive elements
  if (H) {
    const hp_v2u u = *(const hp_v2u __attribute__((address_space(1)))*)(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
  }
  return gload4(reinterpret_cast<const float*>(p));
}
template <bool H>
__device__ __forceinline__ float4 aload4(const float* base, const size_t idx) {
  return aload4p<H>(reinterpret_cast<const char*>(base) + idx * (H ? 2 : 4));
}
template <bool H>
__device__ __forceinline__ float aload1(const float* base, const size_t idx) {
  if (H) {
    const unsigned short b = *(const unsigned short __attribute__((address_space(1)))*)(reinterpret_cast<const char*>(base) + idx * 2);
    return __uint_as_float((unsigned)b << 16);
  }
  return gload1(base + idx);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(const float v) {
  const __bf16 h = (__bf16)v;
  return *reinterpret_cast<const unsigned short*>(&h);
}
template <bool H>
__device__ __forceinline__ void astore1(float* base, const size_t idx, const float v) {
  if (H) *(unsigned short __attribute__((address_space(1)))*)(reinterpret_cast<char*>(base) + idx * 2) = f32_to_bf16_bits(v);
  else gstore1(base + idx, v);
}
template <bool H>
__device__ __forceinline__ void astore4(float* base, const size_t idx, const float4 v) {
  if (H) {
    hp_v2u u;
    u.x = (unsigned)f32_to_bf16_bits(v.x) | ((unsigned)f32_to_bf16_bits(v.y) << 16);
    u.y = (unsigned)f32_to_bf16_bits(v.z) | ((unsigned)f32_to_bf16_bits(v.w) << 16);
    *(hp_v2u __attribute__((address_space(1)))*)(reinterpret_cast<char*>(base) + idx * 2) = u;
  } else {
    gstore4(base + idx, v);
  }
}

// blockIdx -> tile id such that each XCD (blocks are dealt round-robin over the 8 XCDs)
// owns one contiguous run of tile ids: the N-tiles that share an A row-panel then hit the
// same L2.  Bijective for any nblk.  Speed only; correctness never depends on it.
__device__ __forceinline__ int xcd_remap(int id, int nblk) {
  const int xcd = id & 7, q = nblk >> 3, r = nblk & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// Column sums of the four waves that share 32 output columns (two row halves x two K-half owners) are folded in LDS
// and leave the workgroup as ONE fp64 atomic per column and statistic (a quarter of the atomics of per-wave adds).
// `v[k]` = this lane's partial of statistic k (rows of both lane halves already folded), meaningful on lanes < 32.
// (8 waves: wave = quadrant + 4 * K-half, quadrant = 2 * row half + column half.)
template <int NS>
__device__ __forceinline__ void fold_column_stats(double (&v)[NS], double* sred, const int wave, const int lane) {
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k < NS; ++k) sred[(wave * NS + k) * 32 + lane] = v[k];
  }
  __syncthreads();
  // waves 0 and 1 (row half 0, K-half owner 0; column halves 0 and 1) add the partials of waves w, w+2, w+4, w+6
  if (wave < 2 && lane < 32) {
#pragma unroll
    for (int k = 0; k < NS; ++k)
      v[k] = ((sred[(wave * NS + k) * 32 + lane] + sred[((wave + 4) * NS + k) * 32 + lane]) +
              (sred[((wave + 2) * NS + k) * 32 + lane] + sred[((wave + 6) * NS + k) * 32 + lane]));
  }
}
