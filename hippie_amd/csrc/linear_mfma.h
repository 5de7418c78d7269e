// nn.Linear forward / input-gradient / weight-gradient on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma
// chains), for the batch sizes at which the heads and the backbones' Linear layers are real GEMMs (M >= 1024 rows: BASELINE
// configs[2] and [4]).  Replaces, like the scalar kernels of ops_small.hip they take over from: encoder.linear / decoder.linear /
// decoder.linear_out (hippie/backbones.py:84,102,111,129,118,138), encoder_fc / fusion_encoder / z_mean / z_log_var / decoder_fc*
// (hippie/model.py:21-41,364-369) and their halves of ATen's linear backward.
//
// Nothing about these operands is tile-friendly: leading dimensions are 2z + 10, 4z + 10, z + 10 floats (rows only 8- or 4-byte
// aligned), widths are 20 ... 512, inputs are column windows of wider tensors.  The loaders therefore take an alignment class per
// operand (16 / 8 / 4 bytes, uniform per launch) and guard every 4-float piece by the number of its elements that exist; inside
// LDS everything is the padded, aligned image the MFMA fragments want.
//
//   lin_mm_body<B_KN>:  C[m][n] (ldc) = epilogue( sum_k A[m][k] (lda) * B(k, n) ),  B(k, n) = B[n*ldb + k] (B_KN false: forward,
//                       B = W) or B[k*ldb + n] (B_KN true: input-gradient, the same W read as its transpose).  64x64 tile per
//                       512-thread workgroup, 8 waves = 4 quadrants x 2 halves of every 32-wide K slice (as conv_body).
//   lin_wgrad_body:     DW[n][k] += sum_{m in split} DY[m][n] * X[m][k],  DB[n] += sum DY[m][n];  64x64 tile per 256-thread
//                       workgroup, contraction over rows (both operands read K-major from LDS), fp32 atomics into the zeroed
//                       gradient.
#pragma once
#include "hp_mfma.h"

struct LinMM {
  const float* A; const float* B; float* C;
  int lda, ldb, ldc;
  int M, N, Kc;                 // output [M][N], contraction length Kc
  int alA, alB;                 // alignment class of every 4-float piece the loaders form: 4 = 16 bytes, 2 = 8, 1 = 4
  // forward epilogue
  const float* bias; double* stats; int act;
  // input-gradient epilogue
  const float* mask; int ldm; int accumulate;
  float slope;
};

// four consecutive floats p[0..3] of which the first nv (<= 0: none, >= 4: all) exist; al: see LinMM
__device__ __forceinline__ float4 lin_load4(const float* p, const int nv, const int al) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nv >= 4) {
    if (al == 4) {
      v = gload4(p);
    } else if (al == 2) {
      const float2 a = gload2(p), b = gload2(p + 2);
      v = make_float4(a.x, a.y, b.x, b.y);
    } else {
      v = make_float4(gload1(p), gload1(p + 1), gload1(p + 2), gload1(p + 3));
    }
  } else if (nv > 0) {
    v.x = gload1(p);
    if (nv > 1) v.y = gload1(p + 1);
    if (nv > 2) v.z = gload1(p + 2);
  }
  return v;
}

constexpr int kLinMMThreads = 512;
constexpr int kLinMMLds = 4 * 64 * 36;      // floats: two double-buffered 64x36 images (the [32][68] K-major image fits too)

template <bool B_KN>
__device__ __forceinline__ void lin_mm_body(const LinMM& p, const int bid, float* smem) {
  constexpr int LDA = 36;      // 32 + 4 floats: ds_read_b128 of 16 rows conflict-free
  constexpr int LDBK = 68;     // [k][n] image row stride
  constexpr int TILE = 64 * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quad = wave & 3, kh = wave >> 2;
  const int wm = quad >> 1, wn = quad & 1, li = lane & 31, lh = lane >> 5;
  const int nt = (p.N + 63) >> 6, mt = (p.M + 63) >> 6;
  const int tile = xcd_remap(bid, mt * nt);
  const int m0 = (tile / nt) << 6, n0 = (tile % nt) << 6;

  // load slots: row ar of the A tile (and of the [n][k] image of B), 4-float piece aq; [k][n] image: row kr, piece nq
  const int ar = tid >> 3, aq = (tid & 7) << 2;
  const int kr = tid >> 4, nq = (tid & 15) << 2;
  const bool a_ok = m0 + ar < p.M;
  const float* pa = p.A + (size_t)(a_ok ? m0 + ar : 0) * p.lda + aq;
  const bool b_ok = B_KN ? true : n0 + ar < p.N;
  const float* pb = B_KN ? p.B + (size_t)kr * p.ldb + n0 + nq : p.B + (size_t)(b_ok ? n0 + ar : 0) * p.ldb + aq;
  const int nvb_kn = p.N - (n0 + nq);          // [k][n] image: columns of this piece that exist
  const int nsteps = (p.Kc + 31) >> 5;

  struct Pref { float4 a, b; };
  auto fetch = [&](const int s) -> Pref {
    Pref r;
    const int k0 = s << 5;
    r.a = lin_load4(pa + k0, a_ok ? p.Kc - (k0 + aq) : 0, p.alA);
    if (!B_KN) r.b = lin_load4(pb + k0, b_ok ? p.Kc - (k0 + aq) : 0, p.alB);
    else       r.b = lin_load4(pb + (size_t)k0 * p.ldb, k0 + kr < p.Kc ? nvb_kn : 0, p.alB);
    return r;
  };
  auto stash = [&](const int buf, const Pref& r) {
    float* As = smem + buf * TILE;
    float* Bs = smem + 2 * TILE + buf * TILE;
    *reinterpret_cast<float4*>(As + ar * LDA + aq) = r.a;
    if (!B_KN) *reinterpret_cast<float4*>(Bs + ar * LDA + aq) = r.b;
    else       *reinterpret_cast<float4*>(Bs + kr * LDBK + nq) = r.b;
  };

  // two independent accumulators per wave (one per 8-wide k group it owns); with the other K-half four partial sums per output
  f32x16 acc2[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[q][r] = 0.f;

  Pref cur = fetch(0);
  stash(0, cur);
  Pref nxt = cur;
  if (nsteps > 1) nxt = fetch(1);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    // lane (li, lh) takes k = 8g + 4lh .. +3 of its row within the wave's 16-wide half; MFMA jj pairs element jj of both
    // operands — a K permutation applied identically to A and B
    const float* As = smem + buf * TILE + (wm * 32 + li) * LDA + lh * 4 + kh * 16;
    const float* Bs = smem + 2 * TILE + buf * TILE;
    float4 a4[2], b4[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      a4[g] = *reinterpret_cast<const float4*>(As + g * 8);
      if (!B_KN) {
        b4[g] = *reinterpret_cast<const float4*>(Bs + (wn * 32 + li) * LDA + kh * 16 + g * 8 + lh * 4);
      } else {
        const float* bk = Bs + (kh * 16 + g * 8 + lh * 4) * LDBK + wn * 32 + li;
        b4[g] = make_float4(bk[0], bk[LDBK], bk[2 * LDBK], bk[3 * LDBK]);
      }
    }
    Pref nn = nxt;
    if (s + 2 < nsteps) nn = fetch(s + 2);        // in flight under this step's MFMAs and the next step's
    if (s + 1 < nsteps) stash(buf ^ 1, nxt);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      acc2[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].x, b4[g].x, acc2[g], 0, 0, 0);
      acc2[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].y, b4[g].y, acc2[g], 0, 0, 0);
      acc2[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].z, b4[g].z, acc2[g], 0, 0, 0);
      acc2[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].w, b4[g].w, acc2[g], 0, 0, 0);
    }
    nxt = nn;
    __syncthreads();
  }

  // ---- epilogue: the two K-half waves of a quadrant each finish eight of its sixteen accumulator rows (as conv_epilogue)
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc2[0][r] + acc2[1][r];
  float* mine = smem + (quad * 2 + kh) * (8 * 64);
  float* theirs = smem + (quad * 2 + (kh ^ 1)) * (8 * 64);
#pragma unroll
  for (int j = 0; j < 8; ++j) theirs[j * 64 + lane] = kh ? acc[j] : acc[8 + j];
  __syncthreads();
  float own[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) own[j] = (kh ? acc[8 + j] : acc[j]) + mine[j * 64 + lane];
  double* sred = reinterpret_cast<double*>(smem + 8 * 8 * 64);

  // C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) with r = 8*kh + j
  const int n = n0 + wn * 32 + li;
  const bool nok = n < p.N;
  const int mrow = m0 + wm * 32 + 16 * kh + 4 * lh;
  const float bv = (p.bias != nullptr && nok) ? gload1(p.bias + n) : 0.f;
  const bool has_mask = p.mask != nullptr;
  float mk[8], old[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {          // the epilogue's loads go out together
    const int m = mrow + (j & 3) + 8 * (j >> 2);
    const bool ok = nok && m < p.M;
    mk[j] = (has_mask && ok) ? gload1(p.mask + (size_t)m * p.ldm + n) : 1.f;
    old[j] = (p.accumulate && ok) ? gload1(p.C + (size_t)m * p.ldc + n) : 0.f;
  }
  double s[2] = {0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int m = mrow + (j & 3) + 8 * (j >> 2);
    if (!(nok && m < p.M)) continue;
    float v = own[j] + bv;
    s[0] += (double)v;
    s[1] += (double)v * (double)v;
    if (p.act) v = lrelu(v, p.slope);
    if (has_mask) v *= lrelu_grad(mk[j], p.slope);
    if (p.accumulate) v = old[j] + v;
    gstore1(p.C + (size_t)m * p.ldc + n, v);
  }
  if (p.stats != nullptr) {              // (uniform) the following BatchNorm's sum / sum of squares of the pre-activation
    s[0] += __shfl_xor(s[0], 32, 64);
    s[1] += __shfl_xor(s[1], 32, 64);
    fold_column_stats<2>(s, sred, wave, lane);
    if (wave < 2 && lane < 32 && nok) {
      double* st = stat_replica(p.stats, p.N, bid);
      atomic_add_f64(st + n, s[0]);
      atomic_add_f64(st + p.N + n, s[1]);
    }
  }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------
struct LinWg {
  const float* DY; const float* X; float* DW; float* DB;
  int M, N, K, ldy, ldx;
  int alY, alX;
  int rows_per_split;           // multiple of 32
};
constexpr int kLinWgLds = 2 * 32 * 64;      // floats: one [32 rows][64 columns] image per operand

__device__ __forceinline__ void lin_wgrad_body(const LinWg& p, const int tile, const int split, float* smem) {
  constexpr int T = 32 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ntc = (p.K + 63) >> 6;
  const int n0 = (tile / ntc) << 6, c0 = (tile % ntc) << 6;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);
  const int lr = tid >> 4, cq = (tid & 15) << 2;
  const int nvy = p.N - (n0 + cq), nvx = p.K - (c0 + cq);
  const bool do_db = c0 == 0 && p.DB != nullptr;      // the first column tile of every row tile also sums DY's columns
  float4 rdy[2], rx[2];
  auto load_regs = [&](const int mb) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = mb + lr + 16 * j;
      const bool mv = m < mend;
      const size_t mm = mv ? m : mbeg;
      rdy[j] = lin_load4(p.DY + mm * p.ldy + n0 + cq, mv ? nvy : 0, p.alY);
      rx[j] = lin_load4(p.X + mm * p.ldx + c0 + cq, mv ? nvx : 0, p.alX);
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<float4*>(smem + (lr + 16 * j) * 64 + cq) = rdy[j];
      *reinterpret_cast<float4*>(smem + T + (lr + 16 * j) * 64 + cq) = rx[j];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float dbs = 0.f;
  __syncthreads();                       // (the body may be called repeatedly by one workgroup)
  if (mbeg < mend) {
    load_regs(mbeg);
    store_lds();
    __syncthreads();
    for (int mb = mbeg; mb < mend; mb += 32) {
      const bool more = mb + 32 < mend;
      if (more) load_regs(mb + 32);
      const float* dys = smem + lh * 64 + wn * 32 + li;
      const float* xs = smem + T + lh * 64 + wc * 32 + li;
      float av[16], bv[16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        av[kk] = dys[kk * 128];
        bv[kk] = xs[kk * 128];
      }
      if (do_db && tid < 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += smem[r * 64 + tid];
        dbs += t;
      }
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
      __syncthreads();
      if (more) {
        store_lds();
        __syncthreads();
      }
    }
  }
  const int c = c0 + wc * 32 + li;
  if (c < p.K) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < p.N) atomic_add_f32_global(p.DW + (size_t)n * p.K + c, acc[r]);      // a half-wave adds one 128-byte row segment
    }
  }
  if (do_db && tid < 64 && n0 + tid < p.N) atomic_add_f32_global(p.DB + n0 + tid, dbs);
}
