// Sustained v_mfma_f32_32x32x2_f32 rate probes (GPU box):  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MODE, int RANDOM = 0>   // MODE 0: registers only; 1: + 8 ds_read_b128 per 16 MFMA; 2: + barrier per 16 MFMA
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 36 * 2];
  for (int i = threadIdx.x; i < 64 * 36 * 2; i += 256) {
    unsigned h = (i + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    lds[i] = RANDOM ? ((int)(h & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000) * seed : seed * (i & 15);
  }
  __syncthreads();
  f32x16 acc[NACC];
  for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  const float* As = lds + (threadIdx.x & 31) * 36 + (threadIdx.x & 32 ? 4 : 0);
  for (int it = 0; it < iters; ++it) {
    float4 a4[4], b4[4];
    if (MODE >= 1) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { a4[kk] = *(const float4*)(As + kk * 8); b4[kk] = *(const float4*)(As + 64 * 36 + kk * 8); }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { a4[kk] = make_float4(a, a + 1, a + 2, a + 3); b4[kk] = make_float4(b, b + 1, b + 2, b + 3); }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      acc[kk % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kk].x, b4[kk].x, acc[kk % NACC], 0, 0, 0);
      acc[kk % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kk].y, b4[kk].y, acc[kk % NACC], 0, 0, 0);
      acc[kk % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kk].z, b4[kk].z, acc[kk % NACC], 0, 0, 0);
      acc[kk % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kk].w, b4[kk].w, acc[kk % NACC], 0, 0, 0);
    }
    if (MODE >= 2) __syncthreads();
    a += 1e-6f;
  }
  float s = 0.f;
  for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int MODE, int RANDOM = 0>
void run(const char* name, int blocks, int iters) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<NACC, MODE, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 10; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL((probe<NACC, MODE, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double flop = (double)blocks * 4 * iters * 16 * 32 * 32 * 2 * 2;
  printf("%-34s blocks=%5d iters=%5d  %8.1f us  %6.1f TFLOP/s\n", name, blocks, iters, best * 1e3, flop / (best * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 1024}) {
    run<1, 0>("regs, 1 acc chain", blocks, 48);
    run<4, 0>("regs, 4 acc chains", blocks, 48);
    run<4, 1>("+8 ds_read_b128 / 16 mfma", blocks, 48);
    run<4, 2>("+ds_read + barrier / 16 mfma", blocks, 48);
  }
  run<4, 2, 1>("RANDOM lds+barrier", 256, 48);
  run<4, 2, 1>("RANDOM lds+barrier", 1024, 48);
  run<4, 2, 1>("RANDOM lds+barrier, long", 256, 4800);
  run<4, 0>("regs, 4 acc, long", 256, 4800);
  run<4, 2>("lds+barrier, long", 256, 4800);
  run<4, 2>("lds+barrier, long, 1024 blocks", 1024, 1200);
  return 0;
}
