// Feasibility probe for DESIGN.md section 8: fp32 GEMM emulated on the bf16 matrix pipe by a 3-way bf16 split
// (x = hi + mid + lo, 6 of the 9 cross products kept: dropped terms are O(2^-24) relative).
//   C[M][N] = A[M][K] . B[N][K]^T, fp32 in / fp32 out, v_mfma_f32_32x32x16_bf16, 64x64 tile, 4 waves.
// Prints time, TFLOP/s (algorithmic fp32 FLOPs) and the error against an fp64 host reference, next to the
// error of a plain fp32 host GEMM.      hipcc --offload-arch=gfx950 -O3 bf16x3_gemm.hip -o bf16x3_gemm.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

__device__ __forceinline__ u16 bf16_rne(float x) {
  unsigned b = __float_as_uint(x);
  b += 0x7FFFu + ((b >> 16) & 1u);
  return (u16)(b >> 16);
}
__device__ __forceinline__ float bf16_f(u16 h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, u16& h, u16& m, u16& l) {
  h = bf16_rne(x);
  const float r1 = x - bf16_f(h);
  m = bf16_rne(r1);
  const float r2 = r1 - bf16_f(m);
  l = bf16_rne(r2);
}

constexpr int LDK = 40;   // bf16 per LDS row (32 + 8 pad): 80-byte rows keep ds_read_b128 aligned and spread over banks

template <int NPROD>      // 6 = hh,hm,mh,hl,lh,mm ; 3 = hh,hm,mh ; 1 = hh (plain bf16)
__global__ __launch_bounds__(256) void gemm_bf16x3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) u16 As[3][64][LDK];
  __shared__ __attribute__((aligned(16))) u16 Bs[3][64][LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lg = lane >> 5;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int lr = tid >> 3, lc = (tid & 7) * 4;      // load slot: rows lr, lr+32; 4 consecutive k
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra[2], rb[2];
  auto fetch = [&](int k0) {
    for (int j = 0; j < 2; ++j) {
      ra[j] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + lr + 32 * j) * K + k0 + lc);
      rb[j] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + lr + 32 * j) * K + k0 + lc);
    }
  };
  auto stash = [&]() {
    for (int j = 0; j < 2; ++j) {
      const float a4[4] = {ra[j].x, ra[j].y, ra[j].z, ra[j].w}, b4[4] = {rb[j].x, rb[j].y, rb[j].z, rb[j].w};
      u16 ah[4], am[4], al[4], bh[4], bm[4], bl[4];
      for (int e = 0; e < 4; ++e) { split3(a4[e], ah[e], am[e], al[e]); split3(b4[e], bh[e], bm[e], bl[e]); }
      const int r = lr + 32 * j;
      *reinterpret_cast<uint2*>(&As[0][r][lc]) = make_uint2(ah[0] | (unsigned)ah[1] << 16, ah[2] | (unsigned)ah[3] << 16);
      *reinterpret_cast<uint2*>(&As[1][r][lc]) = make_uint2(am[0] | (unsigned)am[1] << 16, am[2] | (unsigned)am[3] << 16);
      *reinterpret_cast<uint2*>(&As[2][r][lc]) = make_uint2(al[0] | (unsigned)al[1] << 16, al[2] | (unsigned)al[3] << 16);
      *reinterpret_cast<uint2*>(&Bs[0][r][lc]) = make_uint2(bh[0] | (unsigned)bh[1] << 16, bh[2] | (unsigned)bh[3] << 16);
      *reinterpret_cast<uint2*>(&Bs[1][r][lc]) = make_uint2(bm[0] | (unsigned)bm[1] << 16, bm[2] | (unsigned)bm[3] << 16);
      *reinterpret_cast<uint2*>(&Bs[2][r][lc]) = make_uint2(bl[0] | (unsigned)bl[1] << 16, bl[2] | (unsigned)bl[3] << 16);
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    stash();
    __syncthreads();
    if (k0 + 32 < K) fetch(k0 + 32);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 a[3], b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[p] = *reinterpret_cast<const bf16x8*>(&As[p][wm * 32 + li][kb * 16 + lg * 8]);
        b[p] = *reinterpret_cast<const bf16x8*>(&Bs[p][wn * 32 + li][kb * 16 + lg * 8]);
      }
      if (NPROD >= 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
      }
      if (NPROD >= 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn * 32 + li;
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
    C[(size_t)m * N + n] = acc[r];
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 3584, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 1536;
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); };
  for (auto& v : hA) v = rnd() + 0.25f * rnd() * rnd();
  for (auto& v : hB) v = 0.05f * rnd();
  float *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, hC.size() * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  // fp64 reference and plain-fp32 error on a sample of rows
  const int rows = 64;
  std::vector<double> ref((size_t)rows * N);
  std::vector<float> f32((size_t)rows * N);
  for (int i = 0; i < rows; ++i) {
    const int m = (int)((long long)i * (M - 1) / (rows - 1));
    for (int n = 0; n < N; ++n) {
      double acc = 0.0; float accf = 0.f;
      for (int k = 0; k < K; ++k) { acc += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; accf = fmaf(hA[(size_t)m * K + k], hB[(size_t)n * K + k], accf); }
      ref[(size_t)i * N + n] = acc; f32[(size_t)i * N + n] = accf;
    }
  }
  double scale = 0.0, e32 = 0.0;
  for (size_t j = 0; j < ref.size(); ++j) { scale = fmax(scale, fabs(ref[j])); e32 = fmax(e32, fabs(f32[j] - ref[j])); }
  printf("M=%d N=%d K=%d   max|C| %.3f   sequential-fp32 fma chain: max err / max|C| = %.3e\n", M, N, K, scale, e32 / scale);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(M / 64, N / 64);
  auto run = [&](int nprod, const char* name) {
    auto launch = [&]() {
      if (nprod == 6) hipLaunchKernelGGL(gemm_bf16x3<6>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
      else if (nprod == 3) hipLaunchKernelGGL(gemm_bf16x3<3>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
      else hipLaunchKernelGGL(gemm_bf16x3<1>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double err = 0.0;
    for (int i = 0; i < rows; ++i) {
      const int m = (int)((long long)i * (M - 1) / (rows - 1));
      for (int n = 0; n < N; ++n) err = fmax(err, fabs(hC[(size_t)m * N + n] - ref[(size_t)i * N + n]));
    }
    const double us = ms * 1e3 / reps;
    printf("%-34s %8.1f us  %7.1f TFLOP/s (fp32-equivalent)   max err / max|C| = %.3e\n", name, us, 2.0 * M * N * K / (us * 1e-6) / 1e12, err / scale);
  };
  run(6, "bf16 x3 split, 6 products");
  run(3, "bf16 x3 split, 3 products");
  run(1, "plain bf16 (hi only)");
  return 0;
}
