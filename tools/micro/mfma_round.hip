// How does v_mfma_f32_32x32x2_f32 round?  D = C + a0*b0 + a1*b1 per output; compared bit for bit with
//   (A) fma(a1,b1, fma(a0,b0,c))      sequential IEEE fma chain, k ascending
//   (B) fma(a0,b0, fma(a1,b1,c))      k descending
//   (C) round_f32(c + a0*b0 + a1*b1)  one rounding of the exact sum
//   (D) chain A with round-toward-zero
// and the signed error against the exact value, to see whether the rounding is biased.
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_round.hip -o gpurun_out/mfma_round
#include <hip/hip_runtime.h>
#include <cfenv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* a, const float* b, const float* c, float* d, int chain) {
  const int lane = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    acc[r] = c[row * 32 + col];
  }
  for (int s = 0; s < chain; ++s) {
    const float av = a[(s * 64) + lane], bv = b[(s * 64) + lane];     // a[i=lane%32][k=lane/32], b[k=lane/32][j=lane%32]
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    d[row * 32 + col] = acc[r];
  }
}

static float rtz_fma(float x, float y, float z) {
  const int old = fegetround();
  fesetround(FE_TOWARDZERO);
  volatile float r = fmaf(x, y, z);
  fesetround(old);
  return r;
}

int main() {
  const int chain = 96;       // K = 192 like the smallest layer
  std::vector<float> a(chain * 64), b(chain * 64), c(1024), d(1024);
  srand(7);
  auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
  float *da, *db, *dc, *dd;
  hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dc, 4096); hipMalloc(&dd, 4096);
  long same[4] = {0, 0, 0, 0}, total = 0;
  double bias_mfma = 0, bias_chain = 0, abs_mfma = 0, abs_chain = 0;
  for (int trial = 0; trial < 50; ++trial) {
    for (auto& v : a) v = rnd();
    for (auto& v : b) v = rnd();
    for (auto& v : c) v = trial % 2 ? rnd() : 0.f;
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), 4096, hipMemcpyHostToDevice);
    for (int len : {1, chain}) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd, len);
      hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          float A = c[i * 32 + j], B = A, D = A;
          double ex = A;
          float C1 = A;
          for (int s = 0; s < len; ++s) {
            const float a0 = a[s * 64 + i], a1 = a[s * 64 + 32 + i], b0 = b[s * 64 + j], b1 = b[s * 64 + 32 + j];
            A = fmaf(a1, b1, fmaf(a0, b0, A));
            B = fmaf(a0, b0, fmaf(a1, b1, B));
            C1 = (float)((double)C1 + (double)a0 * b0 + (double)a1 * b1);     // (exact in double for |x|<=1 up to 2^-53)
            D = rtz_fma(a1, b1, rtz_fma(a0, b0, D));
            ex += (double)a0 * b0 + (double)a1 * b1;
          }
          const float g = d[i * 32 + j];
          if (len == 1) {
            same[0] += g == A; same[1] += g == B; same[2] += g == C1; same[3] += g == D; ++total;
          } else {
            bias_mfma += (double)g - ex; bias_chain += (double)A - ex;
            abs_mfma += fabs((double)g - ex); abs_chain += fabs((double)A - ex);
          }
        }
    }
  }
  printf("single MFMA (K=2), %ld outputs: == fma chain k-asc %ld, k-desc %ld, single rounding %ld, RTZ chain %ld\n", total, same[0], same[1], same[2], same[3]);
  printf("chain of %d MFMAs (K=%d): mean signed err mfma %.3e vs fma-chain %.3e ; mean |err| mfma %.3e vs fma-chain %.3e\n", chain, 2 * chain,
         bias_mfma / (50.0 * 1024), bias_chain / (50.0 * 1024), abs_mfma / (50.0 * 1024), abs_chain / (50.0 * 1024));
  return 0;
}
