// Implicit-GEMM convolution kernels on the gfx950 f32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, 157 TFLOP/s peak).
//
// Every Conv1d variant of the reference's backbones (hippie/backbones.py:13-16,24-33,
// 50-62,78) and both halves of its backward are one of two kernels over a "tap map"
// (include/hippie_hip.h): stride, zero padding, nearest-neighbour upsampling and the
// transposed/flipped forms needed by the input gradient are all row-index arithmetic in
// the A-operand gather, never materialised tensors.  Activations are channels-last
// [B*L][C], so a GEMM row is one contiguous channel vector.
#include "hp_mfma.h"

#include <vector>
#include <algorithm>
#include <cstdlib>

// HP_CONV_BF16: operands rounded to bfloat16 on their way into LDS, products on v_mfma_f32_32x32x16_bf16, fp32 accumulation
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int kLdaH = 40;    // bf16 [row][32 k] image: 80-byte rows, conflict-free ds_read_b128 of 16 rows
constexpr int kLdtH = 96;    // bf16 [32 rows][64 cols] image read with ds_read_b64_tr_b16: 192-byte rows (4 rows x 2 groups hit 64 distinct banks)

__device__ __forceinline__ bf16x4 to_bf16x4(const float4 v) {
  bf16x4 r;
  r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;      // round to nearest even (v_cvt_pk_bf16_f32)
  return r;
}
// HP_CONV_BF16X3: fp32 arithmetic on the bf16 matrix cores.  x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): three
// 8-bit significands cover fp32's 24 bits (both subtractions are exact, l is exact), so the split loses nothing.  A product a * b is
// then formed as the six terms of (ah + am + al)(bh + bm + bl) above 2^-24: ah*bh, ah*bm, am*bh, am*bm, ah*bl, al*bh — every one exact in
// the fp32 accumulator's product width (8 x 8 bits) — and the three dropped ones (am*bl, al*bm, al*bl) are below 2^-25 of |a * b|: the
// result carries fp32's own rounding level (the accumulation is fp32 either way), on a pipe 16x the rate of v_mfma_f32_32x32x2_f32 —
// 2.7x per fp32 product after the six-fold work.
__device__ __forceinline__ float4 widen4(const bf16x4 h) { return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); }
#ifndef HP_ABL
#define HP_ABL 0      // tools/micro/conv_ablate.sh: 1 = return before the epilogue, 2 = every operand load reads the zero page, 4 = no output store,
#endif                // 8 = 100 MHz timestamps of the first / last workgroup's phases into the (otherwise unused) E_BS buffer; three-term path (timing only,
                      // wrong numbers): 16 = only the leading product, 32 = no split arithmetic (m = l = h), 64 = conv_body stores / reads the h image only,
                      // 256 / 512 = the 128-row fragment body skips its A staging / its B fragment loads
__device__ __forceinline__ void split3(const float4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
  h = to_bf16x4(v);
  if (HP_ABL & 32) { m = h; l = h; return; }
  const float4 hf = widen4(h);
  const float4 r = make_float4(v.x - hf.x, v.y - hf.y, v.z - hf.z, v.w - hf.w);
  m = to_bf16x4(r);
  const float4 mf = widen4(m);
  l = to_bf16x4(make_float4(r.x - mf.x, r.y - mf.y, r.z - mf.z, r.w - mf.w));
}
// MFMA 32x32x16 operand whose 8 contraction values per lane are ROWS of a [rows][cols] bf16 image (the operand is the image's
// transpose): two hardware transpose reads.  Lane l = 16g + 4q + p supplies the address of row (r0 + q), columns
// c0 + 16*(g&1) + 4p .. +3 and receives, for its column c0 + (l & 31), rows r0 + 8*(l>>5) + 0..3 (second read: + 4..7).
// EXEC must be all ones.
__device__ __forceinline__ bf16x8 tr_operand(const __bf16* img, int ld, int r0, int c0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, pq = lane & 3;
  const __bf16* a0 = img + (r0 + 8 * (g >> 1) + q) * ld + c0 + 16 * (g & 1) + 4 * pq;
  const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
  const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 4 * ld));
  union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
  u.s.lo = t0; u.s.hi = t1;
  return u.v;
}

struct TapMap {
  int M, N, K, Lout, Lin, P, a, sh, ntaps;
  int tap_o[HP_MAX_TAPS];
  int tap_w[HP_MAX_TAPS];
  int tap_src[HP_MAX_TAPS];     // CONV_TAPS: 0 = (A, W), 1 = (A2, W2)
  int out_Lfull, out_a, out_o;  // CONV_TAPS: output row of GEMM row (b, l) is b*out_Lfull + out_a*l + out_o (Lfull 0: dense)
};

static TapMap tapmap_from(const HpOp& op) {
  TapMap t;
  t.M = op.i[0]; t.N = op.i[1]; t.K = op.i[2]; t.Lout = op.i[3]; t.Lin = op.i[4];
  t.P = op.i[5]; t.a = op.i[6]; t.sh = op.i[7]; t.ntaps = op.i[9];
  for (int j = 0; j < HP_MAX_TAPS; ++j) { t.tap_o[j] = op.i[10 + j]; t.tap_w[j] = op.i[16 + j]; t.tap_src[j] = 0; }
  t.out_Lfull = t.out_a = t.out_o = 0;
  if (op.op == HP_OP_CONV_TAPS) {
    for (int j = 0; j < HP_MAX_TAPS; ++j) t.tap_src[j] = op.i[22 + j];
    t.out_Lfull = op.i[28]; t.out_a = op.i[29]; t.out_o = op.i[30];
  }
  return t;
}

// ------------------------------------------------------------------------------------
// out[M][N] = sum_taps A[src(m,tap)][0:K] . Wslab[tap]   (+bias, +BN statistics)
// 64x64 output tile per 512-thread workgroup (8 waves: 4 quadrants of 32x32 x 2 K-halves),
// K consumed 32 at a time through double-buffered LDS, global->register prefetch two
// K-slices ahead of the MFMAs.
// ------------------------------------------------------------------------------------
struct ConvArgs {
  const float* A; const float* W; float* out; const float* bias; double* stats;
  const float* A2; const float* W2;     // second (source, weight) pair for taps with tap_src == 1
  // eval-mode BatchNorm epilogue (HP_CONV_BN_EVAL): running statistics, optional residual tensor, optional leaky_relu;
  // with HP_CONV_IN_BN the same four pointers are the INPUT BatchNorm's parameters / running statistics
  const float* gamma; const float* beta; float* rmean; float* rvar; const float* res;
  float eps, slope;
  int bn_eval, act;
  // HP_CONV_IN_BN: training-mode BatchNorm + leaky_relu of the input, applied in the A loader
  int in_bn, in_Mstat;
  const double* in_stats; float* in_save; float* in_coef;
  float in_slope, in_eps, in_mom;
  // HP_CONV_EPI_BNRED: BatchNorm-backward reduction of the output, fused into the epilogue
  int epi;
  const float* e_g2; const float* e_act; const float* e_raw; const float* e_save; const float* e_coef;
  const float* e_raw2; const float* e_save2;
  double* e_bs; double* e_bs2;
  float e_slope;
  const char* Wf; const char* Wf2;      // HP_CONV_WFRAG: three-term fragment images of W / W2 (HP_OP_WFRAG, this op's orientation)
  TapMap t;
};

constexpr int kConvLds = 4 * 64 * 36;   // floats of LDS per workgroup (two double-buffered 64x36 images)
// matrix mode MM of a conv / weight-gradient body: 0 = fp32 matrix cores, 1 = HP_CONV_BF16, 2 = HP_CONV_BF16X3 (three bf16 images per operand),
// 3 = HP_CONV_BF16X3 | HP_CONV_WFRAG (conv_body only: the B fragments come ready-made from global memory, LDS holds the A images alone)
constexpr int kSplitA = 64 * 40;        // bf16 elements of one [64 rows][32 k] image (80-byte rows)
constexpr int kSplitBkn = 32 * 96;      // ... of one [32 k][64 cols] image (192-byte rows, hardware-transpose reads)
constexpr int conv_lds(int mm, bool w_kn) {      // floats: MM = 2 holds 2 buffers x 3 images per operand
  return mm == 3 ? (2 * 3 * kSplitA) / 2 : mm == 2 ? (2 * 3 * kSplitA + 2 * 3 * (w_kn ? kSplitBkn : kSplitA)) / 2 : kConvLds;
}
constexpr int kConvCoef = 4 * 512;      // + (scale, shift) of up to 512 input channels, then as many zeros (HP_CONV_IN_BN)
constexpr int conv_extra_lds(int mode) { return mode == 1 ? kConvCoef : 0; }
constexpr int kConvThreads = 512;

// Shared epilogue of the conv bodies: sums the two K-halves of every quadrant through LDS, then bias / BatchNorm
// statistics, the eval-mode BatchNorm fold, or the fused BatchNorm-backward reduction.
#if HP_ABL & 8
#define HP_TS(k)                                                                                  \
  if (threadIdx.x == 0 && p.e_bs != nullptr) {                                                    \
    unsigned long long* q_ = reinterpret_cast<unsigned long long*>(p.e_bs);                       \
    if (bid == 0) { q_[k] = wall_clock64(); q_[16 + (k)] = clock64(); }                           \
    else if (bid == (int)gridDim.x - 1) q_[8 + (k)] = wall_clock64();                             \
  }
#else
#define HP_TS(k)
#endif
template <bool ABF>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc2)[2], float* smem, const int bid, const int m0, const int n0) {
  const TapMap& t = p.t;
  if ((HP_ABL & 1) && t.M >= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quad = wave & 3, kh = wave >> 2;
  const int wm = quad >> 1, wn = quad & 1, li = lane & 31, lh = lane >> 5;
  // The two K-half waves of a quadrant each FINISH eight of its sixteen accumulator rows: wave kh keeps accumulators
  // 8*kh .. 8*kh+7, hands the other eight to its partner through LDS (the staging buffers are free after the last
  // barrier) and adds what the partner left — the epilogue's ~20 dependent instructions per row then run on two waves
  // per SIMD instead of one (a lone wave per SIMD issues them back to back with nothing to hide their latency:
  // measured 3.0 us of a launch, tools/micro/conv_phases.py).  a + b == b + a: the sums are the ones a single owner forms.
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
  double* sred = reinterpret_cast<double*>(smem + 8 * 8 * 64);      // behind the exchange area (fold_column_stats)
  HP_TS(3)

  // C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) with r = 8*kh + j
  const int n = n0 + wn * 32 + li;
  const bool nok = n < t.N;
  const int mrow = m0 + wm * 32 + 16 * kh + 4 * lh;              // + (j&3) + 8*(j>>2)
  // element offset of the output of this lane's row j (-1 = outside the problem).  out_Lfull > 0: the op writes a
  // strided subset of the rows of a taller tensor.  (int: every tensor of a program is < 2^31 elements, checked by
  // hp_program_validate.)  A workgroup whose tile lies inside the problem (all but the last row / column of tiles)
  // takes the unguarded form: no per-row exec-mask juggling.
  const bool full = m0 + 64 <= t.M && n0 + 64 <= t.N && t.out_Lfull == 0;
  int off[8];
  if (full) {
    const int base = mrow * t.N + n;
#pragma unroll
    for (int j = 0; j < 8; ++j) off[j] = base + ((j & 3) + 8 * (j >> 2)) * t.N;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = mrow + (j & 3) + 8 * (j >> 2);
      int o = -1;
      if (nok && m < t.M) {
        o = m;
        if (t.out_Lfull > 0) {
          const int b = m / t.Lout;
          o = b * t.out_Lfull + t.out_a * (m - b * t.Lout) + t.out_o;
        }
        o = o * t.N + n;
      }
      off[j] = o;
    }
  }
  if (p.epi) {
    // HP_CONV_EPI_BNRED: HP_OP_BN_BWD_REDUCE on the accumulators (same expressions as bn_bwd_reduce_body)
    float mean = 0.f, invstd = 0.f, mean2 = 0.f, invstd2 = 0.f, csc = 0.f, csh = 0.f;
    const bool has_act = p.e_act != nullptr, has_g2 = p.e_g2 != nullptr, has_2 = p.e_raw2 != nullptr;
    if (nok) {
      mean = p.e_save[n]; invstd = p.e_save[t.N + n];
      if (has_2) { mean2 = p.e_save2[n]; invstd2 = p.e_save2[t.N + n]; }
      if (!has_act) { csc = p.e_coef[n]; csh = p.e_coef[t.N + n]; }
    }
    double s[3] = {0.0, 0.0, 0.0};
    float xr[8], av[8], g2[8], x2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {          // all loads in flight before the first use
      const int o = off[j] >= 0 ? off[j] : 0;          // clamped, unconditional
      xr[j] = aload1<ABF>(p.e_raw, o);
      av[j] = has_act ? aload1<ABF>(p.e_act, o) : 0.f;
      g2[j] = has_g2 ? aload1<ABF>(p.e_g2, o) : 0.f;
      x2[j] = has_2 ? aload1<ABF>(p.e_raw2, o) : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (!full && off[j] < 0) continue;
      float gv = own[j];
      if (has_g2) gv += g2[j];
      const float pre = has_act ? av[j] : fmaf(xr[j], csc, csh);
      gv *= lrelu_grad(pre, p.e_slope);
      astore1<ABF>(p.out, off[j], gv);
      s[0] += (double)gv;
      s[1] += (double)gv * (double)((xr[j] - mean) * invstd);
      if (has_2) s[2] += (double)gv * (double)((x2[j] - mean2) * invstd2);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] += __shfl_xor(s[k], 32, 64);
    fold_column_stats<3>(s, sred, wave, lane);
    if (wave < 2 && lane < 32 && nok) {
      double* b1 = stat_replica(p.e_bs, t.N, bid);
      atomic_add_f64(b1 + n, s[0]);
      atomic_add_f64(b1 + t.N + n, s[1]);
      if (has_2) {
        double* b2 = stat_replica(p.e_bs2, t.N, bid);
        atomic_add_f64(b2 + n, s[0]);
        atomic_add_f64(b2 + t.N + n, s[2]);
      }
    }
    return;
  }
  const float bv = (p.bias != nullptr && nok) ? p.bias[n] : 0.f;
  if (p.bn_eval) {
    // forward-only path: BatchNorm1d in eval mode (+ residual, + leaky_relu) applied to the accumulators; the
    // coefficients are formed exactly as bn_coef() does for HP_OP_BN_APPLY, so the result is bit-identical
    float sc = 0.f, sh = 0.f;
    if (nok) {
      const double invstd = 1.0 / sqrt((double)p.rvar[n] + (double)p.eps);
      const double scd = (double)p.gamma[n] * invstd;
      sc = (float)scd;
      sh = (float)((double)p.beta[n] - (double)p.rmean[n] * scd);
    }
    float rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rs[j] = p.res != nullptr ? aload1<ABF>(p.res, off[j] >= 0 ? off[j] : 0) : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (!full && off[j] < 0) continue;
      float v = fmaf(own[j] + bv, sc, sh);
      if (p.res != nullptr) v += rs[j];
      if (p.act) v = lrelu(v, p.slope);
      astore1<ABF>(p.out, off[j], v);
    }
    return;
  }
  double s[2] = {0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (!full && off[j] < 0) continue;
    const float v = own[j] + bv;
    if (!((HP_ABL & 4) && t.M >= 0)) astore1<ABF>(p.out, off[j], v);
    s[0] += (double)v;
    s[1] += (double)v * (double)v;
  }
  HP_TS(4)
  if (p.stats != nullptr) {
    s[0] += __shfl_xor(s[0], 32, 64);
    s[1] += __shfl_xor(s[1], 32, 64);
    fold_column_stats<2>(s, sred, wave, lane);
    if (wave < 2 && lane < 32 && nok) {
      double* st = stat_replica(p.stats, t.N, bid);
      atomic_add_f64(st + n, s[0]);
      atomic_add_f64(st + t.N + n, s[1]);
    }
  }
}

// One 64x64 output tile per 512-thread workgroup: 8 waves = 4 tile quadrants (32x32 MFMA tiles) x 2 K-halves.
// The two waves that share a quadrant each take half of every 32-wide K slice (two 8-wide groups) and
// sit on the same SIMD pair-wise, so one wave's global loads / LDS traffic / address arithmetic overlap
// the other's MFMAs — at batch 512 a layer has only ~256 tiles for 1024 SIMDs, so this is the only way to
// get two waves per SIMD.  Their partial accumulators are summed once, through LDS, in the epilogue.
// (Measured and removed, round 3: a dedicated bf16 body with a 64-wide K-step — two 16-byte pieces per operand and thread, 64 KB in flight
// per workgroup, half the barriers — took a K = 512 layer from 19.8 to 18.2 us and the bf16 line from 185.8 k to 188.2 k samples/s
// (+1.3 %): in bf16 mode the loop is bound by the rate at which a CU stages fp32 operands through registers into LDS (~43 GB/s per
// CU), not by bytes in flight or barriers; only bf16-STORED operands change that.)
// (Measured and removed, round 2: a 64-wide K-step — half the barriers and LDS round trips per MFMA, 126-128 VGPRs, 70 KB
// of LDS — is no faster at K = 512 (38.1 us either way) and slower at K = 64 (17-18.6 vs 13.4 us: twice the prologue):
// the per-slice barrier is not what limits the loop.)
// MODE: 0 = the A operand is a stored tensor; 1 = HP_CONV_IN_BN
template <bool W_KN, int MODE, int MM = 0, bool ABF = false>
__device__ __forceinline__ void conv_body(const ConvArgs& p, const int bid, float* smem) {
  constexpr bool BF16 = MM == 1, SPLIT = MM == 2 || MM == 3, FRAG = MM == 3;
  constexpr int BIMG = W_KN ? kSplitBkn : kSplitA;      // SPLIT: bf16 elements of one B image; buffers hold 3 images each, A buffers first
  constexpr int ES = ABF ? 2 : 4;      // bytes per stored activation element (HP_FLAG_ACT_BF16; only with BF16)
  constexpr bool IN_BN = MODE == 1;
  constexpr int LDA = 36;    // 32 + 4 floats: ds_read_b128 of 16 rows conflict-free
  constexpr int LDBK = 68;   // [k][n] image row stride
  constexpr int TILE = 64 * LDA;   // 2304 floats; the [32][68] image (2176) fits too

  const TapMap& t = p.t;
  HP_TS(0)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quad = wave & 3, kh = wave >> 2;
  const int wm = quad >> 1, wn = quad & 1, li = lane & 31, lh = lane >> 5;
  const int nt = (t.N + 63) >> 6, mt = (t.M + 63) >> 6;
  const int tile = xcd_remap(bid, mt * nt);
  const int m0 = (tile / nt) << 6, n0 = (tile % nt) << 6;

  // fixed per-thread load slot: row ar of the A tile (and of the [n][k] weight image), 16-byte column aq
  const int ar = tid >> 3, aq = (tid & 7) << 2;
  const int kr = tid >> 4, nq = (tid & 15) << 2;   // [k][n] weight image slot
  const int m_row = m0 + ar;
  const bool rvalid = m_row < t.M && !((HP_ABL & 2) && t.M >= 0);
  const int b_row = m_row / t.Lout;
  const int rbase = b_row * t.Lin;
  const int rl = t.a * (m_row - b_row * t.Lout);

  const int kper = t.K >> 5;
  const int nsteps = t.ntaps * kper;
  const size_t wslab = (size_t)t.N * t.K;

  // Two register sets (plain structs so they stay in VGPRs): slice s+1 waits in one while slice s+2 is being
  // fetched into the other.  Loads are UNCONDITIONAL (a padded / out-of-range row reads 16 bytes of zeros):
  // branch-free loads let the compiler wait with counted vmcnt(N).
  // (IN_BN: kq = channel offset of the slice in the coefficient table, va = 0 for a padded row: the padding is
  // zeros of the ACTIVATION, not of the raw tensor it is computed from)
  struct Pref { float4 a, b; int kq; };
  // Per-tap load state: pointers advanced by a constant per K-slice, recomputed only at tap boundaries.
  const char* pa; const float* pb;
  int ia, ib;
  int n_tap = 0, kc = 0;
  auto set_tap = [&](int tap) {
    const int to = t.tap_o[tap];
    const bool second = t.tap_src[tap] != 0;
    const float* wp = (second ? p.W2 : p.W) + (size_t)t.tap_w[tap] * wslab;
    const float* ap = second ? p.A2 : p.A;
    const int pos = rl + to;
    const bool oa = rvalid && pos >= 0 && pos < t.P;
    pa = oa ? reinterpret_cast<const char*>(ap) + ((size_t)(rbase + (pos >> t.sh)) * t.K + aq) * ES : reinterpret_cast<const char*>(hp_zero16);
    ia = oa ? 32 * ES : 0;
    bool ob;
    if (!W_KN) {
      ob = n0 + ar < t.N && !((HP_ABL & 2) && t.M >= 0);
      pb = ob ? wp + (size_t)(n0 + ar) * t.K + aq : hp_zero16;
      ib = ob ? 32 : 0;
    } else {
      ob = n0 + nq < t.N && !((HP_ABL & 2) && t.M >= 0);
      pb = ob ? wp + (size_t)kr * t.N + n0 + nq : hp_zero16;
      ib = ob ? 32 * t.N : 0;
    }
  };
  set_tap(0);
  auto advance = [&]() {
    pa += ia; pb += ib;
    if (++kc == kper) {
      kc = 0;
      if (++n_tap < t.ntaps) set_tap(n_tap);
    }
  };
  auto fetch = [&]() -> Pref {
    Pref r;
    r.a = aload4p<ABF>(pa);
    r.b = FRAG ? make_float4(0.f, 0.f, 0.f, 0.f) : gload4(pb);
    r.kq = 0;
    if (IN_BN) r.kq = (ia ? 0 : 2 * t.K) + kc * 32 + aq;      // a padded row reads (scale, shift) = (0, 0): its activation is exactly 0
    advance();
    return r;
  };
  const float* s_coef = smem + conv_lds(MM, W_KN);
  const bool in_bn = IN_BN && p.in_bn;
  const float in_slope = p.in_slope;
  float4 pre_sc = make_float4(0.f, 0.f, 0.f, 0.f), pre_sh = pre_sc;      // IN_BN: (scale, shift) of the slice about to be stored
  auto load_coef = [&](const Pref& r) {
    if (IN_BN && in_bn) {
      pre_sc = *reinterpret_cast<const float4*>(s_coef + r.kq);
      pre_sh = *reinterpret_cast<const float4*>(s_coef + t.K + r.kq);
    }
  };
  auto stash = [&](int buf, Pref r) {
    float* As = smem + buf * TILE;
    float* Bs = smem + 2 * TILE + buf * TILE;
    if (IN_BN) {
      if (in_bn) {
        // a = leaky_relu(fma(x, scale, shift)): the same two operations, on the same operands, as HP_OP_BN_APPLY
        // leaky_relu as max(v, v * slope): for 0 <= slope <= 1 (checked by hp_program_validate) it selects the same one of
        // the same two values as `v > 0 ? v : v * slope`, in two instructions instead of three; the padding mask is the
        // coefficient address (zeros), not a multiply: 12 VALU operations per float4 instead of 20
        const float4 sc = pre_sc, sh = pre_sh;      // (read from LDS at the top of the K-step: HP_KSTEP)
        const float vx = fmaf(r.a.x, sc.x, sh.x), vy = fmaf(r.a.y, sc.y, sh.y), vz = fmaf(r.a.z, sc.z, sh.z), vw = fmaf(r.a.w, sc.w, sh.w);
        r.a.x = fmaxf(vx, vx * in_slope);
        r.a.y = fmaxf(vy, vy * in_slope);
        r.a.z = fmaxf(vz, vz * in_slope);
        r.a.w = fmaxf(vw, vw * in_slope);
      }
    }
    if (SPLIT) {
      __bf16* Ah = reinterpret_cast<__bf16*>(smem) + buf * 3 * kSplitA + ar * kLdaH + aq;
      __bf16* Bh = reinterpret_cast<__bf16*>(smem) + 6 * kSplitA + buf * 3 * BIMG + (W_KN ? kr * kLdtH + nq : ar * kLdaH + aq);
      bf16x4 h, m, l;
      split3(r.a, h, m, l);
      *reinterpret_cast<bf16x4*>(Ah) = h;
      if (!(HP_ABL & 64)) { *reinterpret_cast<bf16x4*>(Ah + kSplitA) = m; *reinterpret_cast<bf16x4*>(Ah + 2 * kSplitA) = l; }
      if (FRAG) return;
      split3(r.b, h, m, l);
      *reinterpret_cast<bf16x4*>(Bh) = h;
      if (!(HP_ABL & 64)) { *reinterpret_cast<bf16x4*>(Bh + BIMG) = m; *reinterpret_cast<bf16x4*>(Bh + 2 * BIMG) = l; }
      return;
    }
    if (BF16) {
      __bf16* Ah = reinterpret_cast<__bf16*>(As);
      __bf16* Bh = reinterpret_cast<__bf16*>(Bs);
      *reinterpret_cast<bf16x4*>(Ah + ar * kLdaH + aq) = to_bf16x4(r.a);
      if (!W_KN) *reinterpret_cast<bf16x4*>(Bh + ar * kLdaH + aq) = to_bf16x4(r.b);
      else       *reinterpret_cast<bf16x4*>(Bh + kr * kLdtH + nq) = to_bf16x4(r.b);
      return;
    }
    *reinterpret_cast<float4*>(As + ar * LDA + aq) = r.a;
    if (!W_KN) *reinterpret_cast<float4*>(Bs + ar * LDA + aq) = r.b;
    else       *reinterpret_cast<float4*>(Bs + kr * LDBK + nq) = r.b;
  };

  // two independent accumulators per wave (one per 8-wide k group it owns): with the other K-half that is
  // four partial sums per output — blocked summation keeps the fp32 rounding of a K=1536 contraction at the
  // level of a vectorised CPU kernel.
  f32x16 acc2[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[q][r] = 0.f;

  // FRAG (HP_CONV_WFRAG): the three-term B fragments of a K step are 3 x 16 bytes per lane, read ready-made from the image HP_OP_WFRAG wrote
  // ([slab][K/16][ceil(N/32)][3][64 lanes][8] bf16), one K step ahead — no weight tile in LDS, no split of the weights in this kernel.
  // This wave's chunk of step (tap j, chunk kc): (slab tap_w[j], k slab 2 kc + kh, column tile n0/32 + wn).
  bf16x8 bfn[3];             // (two steps in flight — a second register set — measured slower: 126.8 k against 131.9 k samples/s without fragments, the
  const char* pf = nullptr;  //  twelve extra registers spill inside the steps at 128; one step ahead: 138.4 k against 135.7 k)
  int f_tap = 0, f_kc = 0;
  const int f_jn = (t.N + 31) >> 5;
  const size_t f_step = (size_t)2 * f_jn * 3072;
  auto set_ftap = [&](int tap) {
    const char* base = t.tap_src[tap] != 0 ? p.Wf2 : p.Wf;
    const int jn = min((n0 >> 5) + wn, f_jn - 1);          // (a column tile past N: its columns are never stored)
    pf = base + (((size_t)t.tap_w[tap] * (t.K >> 4) + kh) * f_jn + jn) * 3072 + lane * 16;
  };
  auto fetch_bfrag = [&]() {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const hp_v4u u = *(const hp_v4u __attribute__((address_space(1)))*)(pf + c * 1024);
      union { hp_v4u u; bf16x8 v; } w; w.u = u; bfn[c] = w.v;
    }
    pf += f_step;
    if (++f_kc == kper) {
      f_kc = 0;
      if (++f_tap < t.ntaps) set_ftap(f_tap);
    }
  };
  if (FRAG) set_ftap(0);
  // One K-step.  LDS[BUF] holds the current slice; ST holds the next one (stored to the other buffer before
  // the second MFMA group); LD receives the slice after that.  Lane (i,h) takes k = 8kk+4h..+3 of its row and
  // MFMA jj pairs element jj of both operands — a K permutation applied identically to A and B.
  // FETCH / STASH are compile-time: the steady-state loop has no conditions, so the compiler's vmcnt
  // bookkeeping stays exact (a merged "maybe pending" path costs a full drain).
  // bf16 K-step: the wave's half (16 k) of the 32-wide slice is ONE v_mfma_f32_32x32x16_bf16
#define HP_KSTEP_H(BUF, ST, LD, FETCH, STASH)                                                           \
  {                                                                                                     \
    const __bf16* Ah = reinterpret_cast<const __bf16*>(smem + (BUF) * TILE);                            \
    const __bf16* Bh = reinterpret_cast<const __bf16*>(smem + 2 * TILE + (BUF) * TILE);                 \
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ah + (wm * 32 + li) * kLdaH + kh * 16 + lh * 8); \
    bf16x8 bf;                                                                                          \
    if (!W_KN) bf = *reinterpret_cast<const bf16x8*>(Bh + (wn * 32 + li) * kLdaH + kh * 16 + lh * 8);   \
    else       bf = tr_operand(Bh, kLdtH, kh * 16, wn * 32, lane);                                      \
    if (FETCH) {                                                                                        \
      LD.a = aload4p<ABF>(pa);                                                                                \
      LD.b = gload4(pb);                                                                                \
      if (IN_BN) LD.kq = (ia ? 0 : 2 * t.K) + kc * 32 + aq;                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                \
      advance();                                                                                        \
    }                                                                                                   \
    if (STASH) { load_coef(ST); stash((BUF) ^ 1, ST); }                                                 \
    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc2[0], 0, 0, 0);                        \
    __syncthreads();                                                                                    \
  }
  // three-term K-step: the wave's half (16 k) of the slice is SIX v_mfma_f32_32x32x16_bf16 — the leading product into one accumulator, the
  // five corrections (smallest first) into the other, so that they are summed among themselves before they meet the large partial sum
#define HP_KSTEP_S(BUF, ST, LD, FETCH, STASH)                                                           \
  {                                                                                                     \
    const __bf16* Ah = reinterpret_cast<const __bf16*>(smem) + (BUF) * 3 * kSplitA + (wm * 32 + li) * kLdaH + kh * 16 + lh * 8; \
    const __bf16* Bh = reinterpret_cast<const __bf16*>(smem) + 6 * kSplitA + (BUF) * 3 * BIMG;          \
    bf16x8 af[3], bf[3];                                                                                \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                     \
      if ((HP_ABL & 64) && c > 0) { af[c] = af[0]; bf[c] = bf[0]; continue; }                           \
      af[c] = *reinterpret_cast<const bf16x8*>(Ah + c * kSplitA);                                       \
      if (FRAG) { bf[c] = bfn[c]; continue; }                                                           \
      if (!W_KN) bf[c] = *reinterpret_cast<const bf16x8*>(Bh + c * BIMG + (wn * 32 + li) * kLdaH + kh * 16 + lh * 8); \
      else       bf[c] = tr_operand(Bh + c * BIMG, kLdtH, kh * 16, wn * 32, lane);                      \
    }                                                                                                   \
    if (STASH) load_coef(ST);                                                                           \
    if (FRAG && (STASH)) fetch_bfrag();         /* (STASH: step s + 1 exists) */                        \
    if (FETCH) {                                                                                        \
      LD.a = aload4p<ABF>(pa);                                                                          \
      if (!FRAG) LD.b = gload4(pb);                                                                     \
      if (IN_BN) LD.kq = (ia ? 0 : 2 * t.K) + kc * 32 + aq;                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                \
      advance();                                                                                        \
    }                                                                                                   \
    if (!(HP_ABL & 16)) acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc2[1], 0, 0, 0); \
    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc2[0], 0, 0, 0);                  \
    if (!(HP_ABL & 16)) acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc2[1], 0, 0, 0); \
    if (STASH) stash((BUF) ^ 1, ST);                                                                    \
    if (!(HP_ABL & 16)) {                                                                               \
      acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc2[1], 0, 0, 0);                \
      acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc2[1], 0, 0, 0);                \
      acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc2[1], 0, 0, 0);                \
    }                                                                                                   \
    __syncthreads();                                                                                    \
  }
#define HP_KSTEP(BUF, ST, LD, FETCH, STASH)                                                             \
  {                                                                                                     \
    const float* As = smem + (BUF) * TILE + (wm * 32 + li) * LDA + lh * 4 + kh * 16;                    \
    const float* Bs = smem + 2 * TILE + (BUF) * TILE;                                                   \
    float4 a4[2], b4[2];                                                                                \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                     \
      a4[g] = *reinterpret_cast<const float4*>(As + g * 8);                                             \
      if (!W_KN) {                                                                                      \
        b4[g] = *reinterpret_cast<const float4*>(Bs + (wn * 32 + li) * LDA + kh * 16 + g * 8 + lh * 4); \
      } else {                                                                                          \
        const float* bk = Bs + (kh * 16 + g * 8 + lh * 4) * LDBK + wn * 32 + li;                        \
        b4[g] = make_float4(bk[0], bk[LDBK], bk[2 * LDBK], bk[3 * LDBK]);                               \
      }                                                                                                 \
    }                                                                                                   \
    if (STASH) load_coef(ST);      /* with the fragment reads: the LDS round trip is over before the store needs it */ \
    if (FETCH) {                                                                                        \
      LD.a = aload4p<ABF>(pa);                                                                                \
      LD.b = gload4(pb);                                                                                \
      if (IN_BN) LD.kq = (ia ? 0 : 2 * t.K) + kc * 32 + aq;                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                \
    }                                                                                                   \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                     \
      if (g == 1 && (FETCH)) { advance(); __builtin_amdgcn_sched_barrier(0); }                          \
      if (g == 1 && (STASH)) { stash((BUF) ^ 1, ST); __builtin_amdgcn_sched_barrier(0); }               \
      acc2[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].x, b4[g].x, acc2[g], 0, 0, 0);               \
      acc2[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].y, b4[g].y, acc2[g], 0, 0, 0);               \
      acc2[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].z, b4[g].z, acc2[g], 0, 0, 0);               \
      acc2[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].w, b4[g].w, acc2[g], 0, 0, 0);               \
    }                                                                                                   \
    __syncthreads();                                                                                    \
  }

  Pref setA = fetch();                       // slice 0 (in flight while the coefficients are derived)
  if (IN_BN) {
    if (in_bn) {
      // every workgroup derives (scale, shift) of the K input channels from the producer's statistics — the same
      // code path as HP_OP_BN_APPLY (bn_coef), hence the same bits in every workgroup; workgroup 0 also performs
      // the BatchNorm's side effects (saved mean / invstd, (scale, shift) for the backward pass, running statistics)
      float* sc_w = smem + conv_lds(MM, W_KN);
      for (int c = tid; c < t.K; c += kConvThreads) {
        const BnCoef k = bn_coef(true, p.in_Mstat, p.in_stats, t.K, c, p.gamma, p.beta, p.rmean, p.rvar, p.in_eps);
        sc_w[c] = k.scale;
        sc_w[t.K + c] = k.shift;
        sc_w[2 * t.K + c] = 0.f;          // what a padded row reads
        sc_w[3 * t.K + c] = 0.f;
        if (bid == 0) bn_side_effects(k, p.in_Mstat, t.K, c, p.in_save, p.rmean, p.rvar, p.in_mom, p.in_coef);
      }
      __syncthreads();
    }
  }
  load_coef(setA);
  stash(0, setA);
  if (nsteps > 1) setA = fetch();            // slice 1 waits in set A
  Pref setB = setA;
  __syncthreads();
  HP_TS(1)
  int s = 0;
  if (SPLIT) {
    if (FRAG) fetch_bfrag();
    for (; s + 3 < nsteps; s += 2) {
      HP_KSTEP_S(0, setA, setB, true, true)
      HP_KSTEP_S(1, setB, setA, true, true)
    }
    const int rems = nsteps - s;
    if (rems == 3) {
      HP_KSTEP_S(0, setA, setB, true, true)
      HP_KSTEP_S(1, setB, setA, false, true)
      HP_KSTEP_S(0, setA, setB, false, false)
    } else if (rems == 2) {
      HP_KSTEP_S(0, setA, setB, false, true)
      HP_KSTEP_S(1, setB, setA, false, false)
    } else {
      HP_KSTEP_S(0, setA, setB, false, false)
    }
  } else if (BF16) {
    for (; s + 3 < nsteps; s += 2) {
      HP_KSTEP_H(0, setA, setB, true, true)
      HP_KSTEP_H(1, setB, setA, true, true)
    }
    const int remh = nsteps - s;
    if (remh == 3) {
      HP_KSTEP_H(0, setA, setB, true, true)
      HP_KSTEP_H(1, setB, setA, false, true)
      HP_KSTEP_H(0, setA, setB, false, false)
    } else if (remh == 2) {
      HP_KSTEP_H(0, setA, setB, false, true)
      HP_KSTEP_H(1, setB, setA, false, false)
    } else {
      HP_KSTEP_H(0, setA, setB, false, false)
    }
  } else {
  for (; s + 3 < nsteps; s += 2) {           // steady state: both steps fetch and stash
    HP_KSTEP(0, setA, setB, true, true)
    HP_KSTEP(1, setB, setA, true, true)
  }
  const int rem = nsteps - s;                // 1 (only when nsteps == 1), 2 or 3
  if (rem == 3) {
    HP_KSTEP(0, setA, setB, true, true)
    HP_KSTEP(1, setB, setA, false, true)
    HP_KSTEP(0, setA, setB, false, false)
  } else if (rem == 2) {
    HP_KSTEP(0, setA, setB, false, true)
    HP_KSTEP(1, setB, setA, false, false)
  } else {
    HP_KSTEP(0, setA, setB, false, false)
  }
  }
#undef HP_KSTEP
#undef HP_KSTEP_H
#undef HP_KSTEP_S

  HP_TS(2)
  conv_epilogue<ABF>(p, acc2, smem, bid, m0, n0);
}


// (Built, measured and removed, round 4: a three-term body for three-tap stride-1 layers that runs the K loop chunk-outer / tap-inner and stages
// ONE A image per 32-wide chunk — the tile's 64 rows plus a halo row on either side, transformed and split once — for all three taps (a tap =
// a row offset of the fragment read, sample boundaries = a per-lane flag), i.e. a third of the A-side loads, BatchNorm transforms, splits and
// LDS stores.  Same results; at batch 512 the plain layers ran 0-7 % faster (K = 512: 50.6 -> 46.9 us), the layers with the BatchNorm input
// transform 10-30 % slower (15.6 -> 20.3 us at K = 64: the second live image piece and the tap pointers spill inside the steps at 128
// registers), the pair-step 131.3 k against 138.8 k samples/s.  The K loop at this batch is not bound by the A-side staging work.
// The same idea IS what runs the weight gradients (wgrad3s_body below), where it removes two of three X images.)
// (amdgpu_waves_per_eu(4): two 512-thread workgroups per CU, i.e. at most 128 VGPRs, for every instantiation)
template <bool W_KN, int MODE, int MM = 0, bool ABF = false>
__global__ __launch_bounds__(kConvThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_taps_kernel(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[conv_lds(MM, W_KN) + conv_extra_lds(MODE)];
  conv_body<W_KN, MODE, MM, ABF>(p, blockIdx.x, smem);
}

// (Built, measured and removed, round 4: a REGISTER-DIRECT form of the fragment body — nothing of the K loop through LDS: every wave loads its own
// A fragment (per lane 8 consecutive k of one row = 32 contiguous bytes, two 16-byte loads a step ahead), applies the input BatchNorm, splits in
// registers and multiplies with the ready-made weight fragments; no staging stores, no fragment reads, no barrier until the epilogue.  Bit-identical
// (the op tests ran on it) and slower: conv launch 22.7 us against 18.6 us back to back, 120.5 k against 140.7 k samples/s.  A wave's load touches
// 32 rows x 64 bytes — 32 cache lines per instruction instead of 8 for the staged form's 128-byte row pieces — and both column-half waves issue it:
// the L1 / address path is the bound, not LDS.)
// (Built, measured and removed, round 4: a 256-register instantiation of the fragment body (amdgpu_waves_per_eu(2, 2): 158 registers, no spills) for
// launches of at most 256 tiles, which never have two workgroups on a CU.  Per launch it IS faster — 18.07 against 18.43 us back to back — and the
// pair-step is slower, 137.4 k against 142.0 k samples/s, twice in a row: two waves per SIMD at 158 registers leave no room for a workgroup of the
// OTHER model's stream (2 x 128) on the same CU, and that overlap is worth more than the spills cost.)
// HP_OP_PAIR: two independent convolutions (e.g. the same layer of the wave and the time model, a block's conv1
// and its shortcut, or the even / odd output phases of a stride-2 input-gradient) in ONE launch: twice the
// workgroups per launch at batch 512, where a single layer only fills each CU with one workgroup.
template <bool W_KN, int MODE, int MM = 0, bool ABF = false>
__global__ __launch_bounds__(kConvThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_taps_pair_kernel(ConvArgs a, ConvArgs b, int nblk_a) {
  __shared__ __attribute__((aligned(16))) float smem[conv_lds(MM, W_KN) + conv_extra_lds(MODE)];
  if ((int)blockIdx.x < nblk_a) conv_body<W_KN, MODE, MM, ABF>(a, blockIdx.x, smem);
  else conv_body<W_KN, MODE, MM, ABF>(b, blockIdx.x - nblk_a, smem);
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Large-batch bf16 body (HP_CONV_BF16 launches with at least two big tiles per CU: BASELINE configs[2] / [4] in their bf16 mode).
// The 64x64 body above gives a wave ONE v_mfma_f32_32x32x16_bf16 (32 cycles) per K-step and barrier: at batch >= 4096 it runs at
// 70-300 TFLOP/s, bound by operand staging.  Here: 128 x (64 | 128) output tile per 256-thread workgroup, 4 waves as 2 x 2, each wave
// 64 rows x (32 | 64) columns = MT x NT = 2 x (1 | 2) MFMA tiles -> 2*MT*NT = 4 | 8 MFMAs per wave and K-step of 32, every A fragment
// reused NT times and every B fragment MT times; operands are converted to bf16 on their way into LDS as in the small body (same
// images: [row][32 k] with 80-byte rows, [32 k][cols + 32] for the transposed reads), fp32 accumulation, the same epilogues.
// Global -> register prefetch one K-step ahead; 2-3 workgroups per CU cover each other's barriers.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int kBigThreads = 256;
constexpr int big_buf_h(int nt) {                     // bf16 elements of one staging buffer: A image [128][40] + the larger of the two B forms
  return 128 * kLdaH + (64 * nt * kLdaH > 32 * (64 * nt + 32) ? 64 * nt * kLdaH : 32 * (64 * nt + 32));
}
constexpr int kBigEpiTile = 32 * 36;                  // one 32 x 32 fp32 tile of the transposed epilogue, 144-byte rows
constexpr int kBigEpiFloats = 4 * 3 * kBigEpiTile;    // per wave: values, xhat, xhat of a second BatchNorm
constexpr int big_stage_floats(int nt, int mm) {      // staging area: two buffers of big_buf_h bf16 (= big_buf_h floats); MM = 2: ONE buffer of three images each (3/2 of that);
  return mm == 4 ? 2 * 3 * 130 * kLdaH / 2 : mm == 3 ? 2 * 3 * 128 * kLdaH / 2 : mm == 2 ? 3 * big_buf_h(nt) / 2 : big_buf_h(nt);      // MM = 3: TWO buffers of the A operand's
}                                                                                                                                   // three images alone; MM = 4: with a halo row either side
constexpr int big_lds_floats(int nt, int mode, int mm = 1) {      // staging + IN_BN coefficients; >= the epilogue's tiles
  return (big_stage_floats(nt, mm) + (mode == 1 ? kConvCoef : 0)) > kBigEpiFloats ? (big_stage_floats(nt, mm) + (mode == 1 ? kConvCoef : 0)) : kBigEpiFloats;
}

// (Built, measured and removed, round 4: a producer / consumer form of the three-term 128 x 128 body — 512 threads, waves 0-3 only read fragments and
// issue the 48 MFMAs of a K-step, waves 4-7 fetch, split and store the next slice into a second staging buffer, one barrier per K-step, two slices
// in flight in the producers' registers: 140-154 TFLOP/s on the K = 512 layers of config 5 against 155-172 for the form below.  One MFMA wave per
// SIMD leaves the pipe idle through every fragment-read round trip; two workgroups per CU, each through its phases, overlap better.)
template <bool W_KN, int MODE, int NT, bool ABF, int MM = 1>
__device__ __forceinline__ void conv_big_body(const ConvArgs& p, const int bid, float* smem) {
  constexpr bool IN_BN = MODE == 1;
  constexpr bool SPLIT = MM >= 2;                       // HP_CONV_BF16X3: three bf16 images per operand and buffer, six products per fragment pair
  constexpr bool FRAG = MM == 3 || MM == 4;             // ... | HP_CONV_WFRAG: the B fragments come ready-made from global memory, a 16-deep slab ahead; LDS holds the A images alone
  constexpr bool SH = MM == 4;                          // ... three taps reading rows m - 1, m, m + 1 of one tensor: ONE A image per 32-wide K chunk serves all three (below)
  constexpr int IMG = SPLIT ? 3 : 1;
  static_assert(!(SPLIT && ABF), "the three-term mode reads fp32-stored tensors");
  constexpr int MT = 2, TM = 128, TN = 64 * NT, WN = 32 * NT;
  constexpr int LDT = TN + 32;                          // [k][n] image row stride (bf16): rows 64 bytes apart mod 256 -> conflict-free tr reads
  constexpr int A_H = TM * kLdaH;                       // bf16 elements of one A image
  constexpr int B_H = W_KN ? 32 * LDT : TN * kLdaH;
  constexpr int BUF_H = FRAG ? 3 * A_H : IMG * big_buf_h(NT);      // one buffer (A + the larger B form; IMG images of each; FRAG: the A images alone)
  constexpr int B_HS = big_buf_h(NT) - A_H;             // distance between the B images of a buffer
  // A pieces per thread: fp32-stored 128 rows x 8 four-float pieces / 256 threads = 4 (16 bytes each); bf16-stored (ABF) 128 rows x 4
  // eight-element pieces = 2 — 16 bytes per lane either way (8-byte bf16 pieces issue twice the loads per byte)
  constexpr int NA = ABF ? 2 : 4;
  constexpr int AROWS = ABF ? 64 : 32;                  // row distance between a thread's A pieces
  constexpr int NB = W_KN ? 2 * NT : 2 * NT;            // B pieces per thread (both forms: TN*32/4/256)
  (void)B_H;
  const TapMap& t = p.t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int nt_ = (t.N + TN - 1) / TN, mt_ = (t.M + TM - 1) / TM;
  const int tile = xcd_remap(bid, mt_ * nt_);
  const int m0 = (tile / nt_) * TM, n0 = (tile % nt_) * TN;
  const int ar = tid >> 3, aq = (tid & 7) << 2;         // [n][k]-B slot (and fp32-stored A): rows ar + 32 j, piece aq
  const int ara = ABF ? tid >> 2 : ar, aqa = ABF ? (tid & 3) << 3 : aq;      // A slot: rows ara + AROWS j, piece aqa (8 elements when ABF)
  const int kr = tid >> 4, nq = (tid & 15) << 2;        // [k][n]-B slot: k rows kr + 16 j, columns nq + 64 jj
  __bf16* const lds = reinterpret_cast<__bf16*>(smem);
  const float* s_coef = smem + big_stage_floats(NT, MM); // (behind the staging area)

  // row geometry of this thread's four A rows
  int rbase[NA], rl[NA];
  bool rvalid[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int m_row = m0 + ara + AROWS * j;
    rvalid[j] = m_row < t.M;
    const int b_row = m_row / t.Lout;
    rbase[j] = b_row * t.Lin;
    rl[j] = t.a * (m_row - b_row * t.Lout);
  }
  const int kper = t.K >> 5;
  const int nsteps = t.ntaps * kper;
  const size_t wslab = (size_t)t.N * t.K;
  constexpr int ES = ABF ? 2 : 4;                       // bytes per stored activation element (HP_FLAG_ACT_BF16)
  const char* pa[NA]; const float* pb[NB];
  int ia[NA], ib[NB];
  int n_tap = 0, kc = 0;
  auto set_tap = [&](int tap) {
    const int to = t.tap_o[tap];
    const bool second = t.tap_src[tap] != 0;
    const float* wp = (second ? p.W2 : p.W) + (size_t)t.tap_w[tap] * wslab;
    const float* ap = second ? p.A2 : p.A;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int pos = rl[j] + to;
      const bool oa = rvalid[j] && pos >= 0 && pos < t.P;
      pa[j] = oa ? reinterpret_cast<const char*>(ap) + ((size_t)(rbase[j] + (pos >> t.sh)) * t.K + aqa) * ES : reinterpret_cast<const char*>(hp_zero16);
      ia[j] = oa ? 32 * ES : 0;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (!W_KN) {
        const int n = n0 + ar + 32 * j;
        const bool ob = n < t.N;
        pb[j] = ob ? wp + (size_t)n * t.K + aq : hp_zero16;
        ib[j] = ob ? 32 : 0;
      } else {
        const int k = kr + 16 * (j & 1), n = n0 + nq + 64 * (j >> 1);
        const bool ob = n < t.N;
        pb[j] = ob ? wp + (size_t)k * t.N + n : hp_zero16;
        ib[j] = ob ? 32 * t.N : 0;
      }
    }
  };
  set_tap(0);
  struct Pref { hp_v4u a[NA]; float4 b[NB]; int kq; };      // a: 16 bytes = 4 fp32 or (ABF) 8 bf16
  auto fetch = [&]() -> Pref {
    Pref r;
#pragma unroll
    for (int j = 0; j < NA; ++j) r.a[j] = *(const hp_v4u __attribute__((address_space(1)))*)(pa[j]);
    if (!FRAG) {
#pragma unroll
      for (int j = 0; j < NB; ++j) r.b[j] = gload4(pb[j]);
    }
    r.kq = kc * 32 + aqa;
#pragma unroll
    for (int j = 0; j < NA; ++j) pa[j] += ia[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) pb[j] += ib[j];
    if (++kc == kper) {
      kc = 0;
      if (++n_tap < t.ntaps) set_tap(n_tap);
    }
    return r;
  };
  const bool in_bn = IN_BN && p.in_bn;
  const float in_slope = p.in_slope;
  // which of the rows of a fetched set were real (IN_BN: a padded row is a zero of the ACTIVATION): recorded with the set
  auto stash = [&](int buf, const Pref& r, const unsigned okmask) {
    __bf16* Ah = lds + buf * BUF_H;
    __bf16* Bh = Ah + IMG * A_H;
    if (ABF) {
      // 8 stored bf16 per piece: straight into the image — or, with the input BatchNorm, widened, transformed and rounded again
      float sc8[8], sh8[8];
      if (IN_BN && in_bn) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4 c = *reinterpret_cast<const float4*>(s_coef + r.kq + 4 * q), d = *reinterpret_cast<const float4*>(s_coef + t.K + r.kq + 4 * q);
          sc8[4 * q] = c.x; sc8[4 * q + 1] = c.y; sc8[4 * q + 2] = c.z; sc8[4 * q + 3] = c.w;
          sh8[4 * q] = d.x; sh8[4 * q + 1] = d.y; sh8[4 * q + 2] = d.z; sh8[4 * q + 3] = d.w;
        }
      }
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        hp_v4u v = r.a[j];
        if (IN_BN && in_bn) {
          const float keep = (okmask >> j) & 1u ? 1.f : 0.f;
          unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float lo = __uint_as_float(w[q] << 16), hi = __uint_as_float(w[q] & 0xffff0000u);
            const float fl = fmaf(lo, sc8[2 * q], sh8[2 * q]), fh = fmaf(hi, sc8[2 * q + 1], sh8[2 * q + 1]);
            w[q] = (unsigned)f32_to_bf16_bits(fmaxf(fl, fl * in_slope) * keep) | ((unsigned)f32_to_bf16_bits(fmaxf(fh, fh * in_slope) * keep) << 16);
          }
          v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        }
        *reinterpret_cast<hp_v4u*>(Ah + (ara + AROWS * j) * kLdaH + aqa) = v;
      }
    } else {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (IN_BN && in_bn) {
      sc = *reinterpret_cast<const float4*>(s_coef + r.kq);
      sh = *reinterpret_cast<const float4*>(s_coef + t.K + r.kq);
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      float4 v = make_float4(__uint_as_float(r.a[j].x), __uint_as_float(r.a[j].y), __uint_as_float(r.a[j].z), __uint_as_float(r.a[j].w));
      if (IN_BN && in_bn) {
        const float keep = (okmask >> j) & 1u ? 1.f : 0.f;
        const float vx = fmaf(v.x, sc.x, sh.x), vy = fmaf(v.y, sc.y, sh.y), vz = fmaf(v.z, sc.z, sh.z), vw = fmaf(v.w, sc.w, sh.w);
        v.x = fmaxf(vx, vx * in_slope) * keep; v.y = fmaxf(vy, vy * in_slope) * keep;
        v.z = fmaxf(vz, vz * in_slope) * keep; v.w = fmaxf(vw, vw * in_slope) * keep;
      }
      if (SPLIT) {
        bf16x4 h, m, l;
        split3(v, h, m, l);
        __bf16* d = Ah + (ar + 32 * j) * kLdaH + aq;
        *reinterpret_cast<bf16x4*>(d) = h; *reinterpret_cast<bf16x4*>(d + A_H) = m; *reinterpret_cast<bf16x4*>(d + 2 * A_H) = l;
      } else {
        *reinterpret_cast<bf16x4*>(Ah + (ar + 32 * j) * kLdaH + aq) = to_bf16x4(v);
      }
    }
    }
#pragma unroll
    for (int j = 0; j < (FRAG ? 0 : NB); ++j) {
      __bf16* d = W_KN ? Bh + (kr + 16 * (j & 1)) * LDT + nq + 64 * (j >> 1) : Bh + (ar + 32 * j) * kLdaH + aq;
      if (SPLIT) {
        bf16x4 h, m, l;
        split3(r.b[j], h, m, l);
        *reinterpret_cast<bf16x4*>(d) = h; *reinterpret_cast<bf16x4*>(d + B_HS) = m; *reinterpret_cast<bf16x4*>(d + 2 * B_HS) = l;
      } else {
        *reinterpret_cast<bf16x4*>(d) = to_bf16x4(r.b[j]);
      }
    }
  };
  auto okbits = [&]() -> unsigned {      // of the set the NEXT fetch() will load
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) m |= (ia[j] ? 1u : 0u) << j;
    return m;
  };

  // FRAG: this wave's B fragments of a 16-deep slab — NT column tiles x 3 terms x 16 bytes per lane — straight from the HP_OP_WFRAG image, one
  // slab ahead: bq[0] holds the first slab of a K step, bq[1] the second; each is refilled for the next use while the other one is multiplied
  bf16x8 bq[FRAG ? 2 : 1][NT][3];
  const char* pfq = nullptr;
  int fq_half = 0, fq_tap = 0;
  const int fq_jn = (t.N + 31) >> 5;
  int fq_off[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) fq_off[j] = min((n0 >> 5) + wn * NT + j, fq_jn - 1) * 3072;      // (a column tile past N: never stored)
  auto set_fq = [&](int tap) {
    pfq = (t.tap_src[tap] != 0 ? p.Wf2 : p.Wf) + (size_t)t.tap_w[tap] * (t.K >> 4) * fq_jn * 3072 + lane * 16;
  };
  auto fetch_bq = [&](const int set) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const hp_v4u u = *(const hp_v4u __attribute__((address_space(1)))*)(pfq + fq_off[j] + c * 1024);
        union { hp_v4u u; bf16x8 v; } w; w.u = u; bq[FRAG ? set : 0][j][c] = w.v;
      }
    pfq += (size_t)fq_jn * 3072;
    if (++fq_half == 2 * kper) {
      fq_half = 0;
      if (++fq_tap < t.ntaps) set_fq(fq_tap);
    }
  };
  if (FRAG && !SH) { set_fq(0); fetch_bq(0); fetch_bq(1); }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (SH) {
    // ---- chunk-outer / tap-inner K loop over ONE staged A image per 32-wide chunk (the tile's 128 rows + a halo row either side, transformed and
    // split once): a tap is a row offset of the fragment read, a sample boundary a per-lane flag; the weights come as fragments, so between two
    // barriers a wave issues the MFMAs of three taps (144 at NT = 2) against one third of the staging work of the tap-outer loop.
    constexpr int AS = 130 * kLdaH;                       // bf16 elements of one image
    const float* pi[5];
    int ii_[5], irow[5];
    const int icol = (tid & 7) << 2;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int pc = j < 4 ? tid + 256 * j : 1024 + tid;  // piece index: 130 rows x 8 four-float pieces = 1040; the last 16 belong to threads 0 .. 15
      irow[j] = pc >> 3;
      const int m = m0 - 1 + irow[j];
      const bool ok = (j < 4 || tid < 16) && m >= 0 && m < t.M;
      pi[j] = ok ? p.A + (size_t)m * t.K + icol : hp_zero16;
      ii_[j] = ok ? 32 : 0;
    }
    float4 img[5];
    auto fetch_img = [&]() {
#pragma unroll
      for (int j = 0; j < 5; ++j) { img[j] = gload4(pi[j]); pi[j] += ii_[j]; }
    };
    auto stash_img = [&](const int buf, const int kc_) {
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (IN_BN && in_bn) {
        sc = *reinterpret_cast<const float4*>(s_coef + kc_ * 32 + icol);
        sh = *reinterpret_cast<const float4*>(s_coef + t.K + kc_ * 32 + icol);
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j == 4 && tid >= 16) break;
        float4 v = img[j];
        if (IN_BN && in_bn) {      // (rows that are padding for a tap are zeroed at the fragment read, whatever stands here)
          const float vx = fmaf(v.x, sc.x, sh.x), vy = fmaf(v.y, sc.y, sh.y), vz = fmaf(v.z, sc.z, sh.z), vw = fmaf(v.w, sc.w, sh.w);
          v.x = fmaxf(vx, vx * in_slope); v.y = fmaxf(vy, vy * in_slope); v.z = fmaxf(vz, vz * in_slope); v.w = fmaxf(vw, vw * in_slope);
        }
        bf16x4 h, m, l;
        split3(v, h, m, l);
        __bf16* d = lds + buf * 3 * AS + irow[j] * kLdaH + icol;
        *reinterpret_cast<bf16x4*>(d) = h; *reinterpret_cast<bf16x4*>(d + AS) = m; *reinterpret_cast<bf16x4*>(d + 2 * AS) = l;
      }
    };
    // per lane and row tile: the fragment's image row, and whether tap tau's source row exists
    int arow[MT];
    bool rok[MT][3];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * 64 + i * 32 + li;
      const int l = m % t.Lout;
      arow[i] = (wm * 64 + i * 32 + li + 1) * kLdaH + lh * 8;
#pragma unroll
      for (int tau = 0; tau < 3; ++tau) rok[i][tau] = m < t.M && l + t.tap_o[tau] >= 0 && l + t.tap_o[tau] < t.Lout;
    }
    // weight fragments: one pointer per tap, advanced a chunk (two 16-deep slabs) once its second slab has been requested
    const char* pfb[3];
#pragma unroll
    for (int tau = 0; tau < 3; ++tau) pfb[tau] = p.Wf + (size_t)t.tap_w[tau] * (t.K >> 4) * fq_jn * 3072 + lane * 16;
    auto fetch_slab = [&](const int set, const int tau, const int kk) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const hp_v4u u = *(const hp_v4u __attribute__((address_space(1)))*)(pfb[tau] + (size_t)kk * fq_jn * 3072 + fq_off[j] + c * 1024);
          union { hp_v4u u; bf16x8 v; } w; w.u = u; bq[set][j][c] = w.v;
        }
      if (kk == 1) pfb[tau] += (size_t)2 * fq_jn * 3072;
    };
    fetch_img();
    if (IN_BN && in_bn) {
      float* sc_w = smem + big_stage_floats(NT, MM);
      for (int c = tid; c < t.K; c += kBigThreads) {
        const BnCoef k = bn_coef(true, p.in_Mstat, p.in_stats, t.K, c, p.gamma, p.beta, p.rmean, p.rvar, p.in_eps);
        sc_w[c] = k.scale;
        sc_w[t.K + c] = k.shift;
        if (bid == 0) bn_side_effects(k, p.in_Mstat, t.K, c, p.in_save, p.rmean, p.rvar, p.in_mom, p.in_coef);
      }
      __syncthreads();
    }
    stash_img(0, 0);
    if (kper > 1) fetch_img();
    fetch_slab(0, 0, 0);
    fetch_slab(1, 0, 1);
    __syncthreads();
    for (int kc_ = 0; kc_ < kper; ++kc_) {
      const bool more = kc_ + 1 < kper;
      const __bf16* Ai = lds + (kc_ & 1) * 3 * AS;
#pragma unroll
      for (int tau = 0; tau < 3; ++tau) {
        const int oshift = t.tap_o[tau] * kLdaH;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 af[MT][3];
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i][c] = *reinterpret_cast<const bf16x8*>(Ai + c * AS + arow[i] + oshift + kk * 16);
#pragma unroll
          for (int i = 0; i < MT; ++i)
            if (!rok[i][tau]) {
              const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
              af[i][0] = z; af[i][1] = z; af[i][2] = z;
            }
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            constexpr int ca[6] = {0, 2, 1, 0, 1, 0}, cb[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ca[q]], bq[kk][j][cb[q]], acc[i][j], 0, 0, 0);
          }
          // this set's next use: the same slab of the next tap, or of the next chunk's first tap
          if (tau < 2) fetch_slab(kk, tau + 1, kk);
          else if (more) fetch_slab(kk, 0, kk);
        }
        if (tau == 1 && more) {
          stash_img((kc_ + 1) & 1, kc_ + 1);
          if (kc_ + 2 < kper) fetch_img();
        }
      }
      __syncthreads();
    }
  } else {
  unsigned ok_cur = okbits();
  Pref cur = fetch();
  if (IN_BN && in_bn) {
    // (scale, shift) of the K input channels, derived exactly as HP_OP_BN_APPLY derives them; workgroup 0 performs the side effects
    float* sc_w = smem + big_stage_floats(NT, MM);
    for (int c = tid; c < t.K; c += kBigThreads) {
      const BnCoef k = bn_coef(true, p.in_Mstat, p.in_stats, t.K, c, p.gamma, p.beta, p.rmean, p.rvar, p.in_eps);
      sc_w[c] = k.scale;
      sc_w[t.K + c] = k.shift;
      if (bid == 0) bn_side_effects(k, p.in_Mstat, t.K, c, p.in_save, p.rmean, p.rvar, p.in_mom, p.in_coef);
    }
    __syncthreads();
  }
  stash(0, cur, ok_cur);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < nsteps;
    unsigned ok_nxt = 0;
    Pref nxt;
    if (more) { ok_nxt = okbits(); nxt = fetch(); }      // in flight under this step's MFMAs
    const __bf16* Ah = lds + ((SPLIT && !FRAG) ? 0 : buf) * BUF_H;
    const __bf16* Bh = Ah + IMG * A_H;
    if (FRAG) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[MT][3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int i = 0; i < MT; ++i) af[i][c] = *reinterpret_cast<const bf16x8*>(Ah + c * A_H + (wm * 64 + i * 32 + li) * kLdaH + kk * 16 + lh * 8);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          constexpr int ca[6] = {0, 2, 1, 0, 1, 0}, cb[6] = {2, 0, 1, 1, 0, 0};
          if ((HP_ABL & 16) && q < 5) continue;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ca[q]], bq[FRAG ? kk : 0][j][cb[q]], acc[i][j], 0, 0, 0);
        }
        // this slab's fragment registers are free once its MFMAs are issued: refill them for the SAME slab of the next K step — a whole step of
        // lead (the other slab's MFMAs, the staging phase, the barrier) instead of one slab's
        if (more && !(HP_ABL & 512)) fetch_bq(kk);
      }
      // (the A images alone are small enough for TWO staging buffers at two workgroups per CU: the next slice goes to the other buffer, one barrier per
      // step.  Staging half of it after each slab's MFMAs — two short phases instead of one — measured no better: 131.4 k against 132.1 k at config 5.)
      if (more && !(HP_ABL & 256)) stash(buf ^ 1, nxt, ok_nxt);
      __syncthreads();
      continue;
    }
    if (SPLIT) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[MT][3], bf[NT][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
          for (int i = 0; i < MT; ++i) af[i][c] = *reinterpret_cast<const bf16x8*>(Ah + c * A_H + (wm * 64 + i * 32 + li) * kLdaH + kk * 16 + lh * 8);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (!W_KN) bf[j][c] = *reinterpret_cast<const bf16x8*>(Bh + c * B_HS + (wn * WN + j * 32 + li) * kLdaH + kk * 16 + lh * 8);
            else       bf[j][c] = tr_operand(Bh + c * B_HS, LDT, kk * 16, wn * WN + j * 32, lane);
          }
        }
        // the five corrections (smallest first), then the leading product: tile after tile, so that consecutive MFMAs hit different accumulators
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          constexpr int ca[6] = {0, 2, 1, 0, 1, 0}, cb[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ca[q]], bf[j][cb[q]], acc[i][j], 0, 0, 0);
        }
      }
      // ONE staging buffer (two would leave room for a single workgroup per CU): the next slice is written once every wave has read this
      // one; the second workgroup of the CU runs its MFMA phase under this one's split-and-store phase
      __syncthreads();
      if (more) {
        stash(0, nxt, ok_nxt);
        __syncthreads();
      }
      continue;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ah + (wm * 64 + i * 32 + li) * kLdaH + kk * 16 + lh * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (!W_KN) bf[j] = *reinterpret_cast<const bf16x8*>(Bh + (wn * WN + j * 32 + li) * kLdaH + kk * 16 + lh * 8);
        else       bf[j] = tr_operand(Bh, LDT, kk * 16, wn * WN + j * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash(buf ^ 1, nxt, ok_nxt);
    __syncthreads();
  }

  }      // (!SH)

  // ---- epilogue (the expressions of conv_epilogue; every wave owns whole accumulator tiles) ------------------------------------------
  // An accumulator tile leaves the registers through LDS, TRANSPOSED: in the MFMA layout a lane holds one column of 16 rows — 2- or 4-byte
  // accesses per lane and tensor, 16 of them per tile; after the transpose a lane holds 8 consecutive columns of one row — one 16-byte load
  // per bf16 operand tensor (two for fp32) and one 16-byte store, 2 per tile.  The fused BatchNorm-backward layers read up to four operand
  // tensors per output element: below 512 channels they are bound by exactly these accesses.  Column statistics are then summed from
  // the LDS tiles (value, xhat) by one lane per column, in fp64, with the per-element expressions of conv_epilogue.
  float* const T0 = smem + wave * 3 * kBigEpiTile;      // this wave's tiles: values | xhat | xhat of the second BatchNorm
  float* const T1 = T0 + kBigEpiTile;
  float* const T2 = T1 + kBigEpiTile;
  double* sred = reinterpret_cast<double*>(smem);       // [statistic][TN] column partials of the wm = 1 waves (used after the tiles are dead)
  const bool plain_full = t.out_Lfull == 0;
  const bool has_act = p.e_act != nullptr, has_g2 = p.e_g2 != nullptr, has_2 = p.e_raw2 != nullptr;
  const int rr = lane >> 2, cg = (lane & 3) << 3;       // transposed slot: rows rr, rr + 16; columns cg .. cg + 7 of the 32 x 32 tile
  // 8 consecutive elements of an activation tensor at element offset o (valid: nv of them, the rest zero) / their store
  auto ld8 = [&](const float* base, const int o, const int nv, float (&v)[8]) {
    if (ABF) {
      if (nv >= 8) {
        const hp_v4u u = *(const hp_v4u __attribute__((address_space(1)))*)(reinterpret_cast<const char*>(base) + (size_t)o * 2);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = e < nv ? aload1<true>(base, (size_t)o + e) : 0.f;
      }
    } else {
      if (nv >= 8) {
        const float4 x0 = gload4(base + o), x1 = gload4(base + o + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = e < nv ? gload1(base + o + e) : 0.f;
      }
    }
  };
  auto st8 = [&](float* base, const int o, const int nv, const float (&v)[8]) {
    if (nv >= 8) {
      astore4<ABF>(base, (size_t)o, make_float4(v[0], v[1], v[2], v[3]));
      astore4<ABF>(base, (size_t)o + 4, make_float4(v[4], v[5], v[6], v[7]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < nv) astore1<ABF>(base, (size_t)o + e, v[e]);
    }
  };
  const bool want = p.epi || (!p.bn_eval && p.stats != nullptr);          // (uniform) column statistics wanted
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ncol0 = n0 + wn * WN + j * 32;            // first column of this tile
    const int n8 = ncol0 + cg;                          // this lane's 8 columns in the transposed pass
    const int nv = min(8, max(0, t.N - n8));
    // per-column constants of the lane's 8 columns
    float mean[8], invstd[8], mean2[8], invstd2[8], csc[8], csh[8], bv[8], esc[8], esh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = e < nv;
      const int n = ok ? n8 + e : 0;
      mean[e] = invstd[e] = mean2[e] = invstd2[e] = csc[e] = csh[e] = bv[e] = esc[e] = esh[e] = 0.f;
      if (p.epi) {
        if (ok) {
          mean[e] = p.e_save[n]; invstd[e] = p.e_save[t.N + n];
          if (has_2) { mean2[e] = p.e_save2[n]; invstd2[e] = p.e_save2[t.N + n]; }
          if (!has_act) { csc[e] = p.e_coef[n]; csh[e] = p.e_coef[t.N + n]; }
        }
      } else if (ok) {
        if (p.bias != nullptr) bv[e] = p.bias[n];
        if (p.bn_eval) {
          const double is = 1.0 / sqrt((double)p.rvar[n] + (double)p.eps);
          const double scd = (double)p.gamma[n] * is;
          esc[e] = (float)scd;
          esh[e] = (float)((double)p.beta[n] - (double)p.rmean[n] * scd);
        }
      }
    }
    double st[3] = {0.0, 0.0, 0.0};                     // column (ncol0 + lane & 31), rows of this lane half, over the MT tiles
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int mtile = m0 + wm * 64 + i * 32;
      __syncthreads();                                  // (the tiles' previous readers are done; first use: the staging buffers are dead)
#pragma unroll
      for (int r = 0; r < 16; ++r) T0[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[i][j][r];
      __syncthreads();
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int row = rr + 16 * ps, m = mtile + row;
        const bool rok = m < t.M && nv > 0;
        int o = 0;
        if (rok) {
          o = m;
          if (!plain_full) {
            const int b_ = m / t.Lout;
            o = b_ * t.out_Lfull + t.out_a * (m - b_ * t.Lout) + t.out_o;
          }
          o = o * t.N + n8;
        }
        const int nvr = rok ? nv : 0;
        float v[8];
        {
          const float4 a0 = *reinterpret_cast<const float4*>(T0 + row * 36 + cg), a1 = *reinterpret_cast<const float4*>(T0 + row * 36 + cg + 4);
          v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
        }
        float xh[8], xh2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[e] = xh2[e] = 0.f;
        if (p.epi) {
          float xr[8], av[8], g2[8], x2[8];
          ld8(p.e_raw, o, nvr, xr);
          if (has_act) ld8(p.e_act, o, nvr, av);
          if (has_g2) ld8(p.e_g2, o, nvr, g2);
          if (has_2) ld8(p.e_raw2, o, nvr, x2);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float gv = v[e];
            if (has_g2) gv += g2[e];
            const float pre = has_act ? av[e] : fmaf(xr[e], csc[e], csh[e]);
            gv *= lrelu_grad(pre, p.e_slope);
            v[e] = e < nvr ? gv : 0.f;
            xh[e] = (xr[e] - mean[e]) * invstd[e];
            if (has_2) xh2[e] = (x2[e] - mean2[e]) * invstd2[e];
          }
        } else if (p.bn_eval) {
          float rs[8];
          if (p.res != nullptr) ld8(p.res, o, nvr, rs);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float w_ = fmaf(v[e] + bv[e], esc[e], esh[e]);
            if (p.res != nullptr) w_ += rs[e];
            v[e] = p.act ? lrelu(w_, p.slope) : w_;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = e < nvr ? v[e] + bv[e] : 0.f;
        }
        if (rok) st8(p.out, o, nvr, v);
        if (want) {                                     // what the column pass sums: the values (zero outside the problem) and xhat
          *reinterpret_cast<float4*>(T0 + row * 36 + cg) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(T0 + row * 36 + cg + 4) = make_float4(v[4], v[5], v[6], v[7]);
          if (p.epi) {
            *reinterpret_cast<float4*>(T1 + row * 36 + cg) = make_float4(xh[0], xh[1], xh[2], xh[3]);
            *reinterpret_cast<float4*>(T1 + row * 36 + cg + 4) = make_float4(xh[4], xh[5], xh[6], xh[7]);
            if (has_2) {
              *reinterpret_cast<float4*>(T2 + row * 36 + cg) = make_float4(xh2[0], xh2[1], xh2[2], xh2[3]);
              *reinterpret_cast<float4*>(T2 + row * 36 + cg + 4) = make_float4(xh2[4], xh2[5], xh2[6], xh2[7]);
            }
          }
        }
      }
      if (want) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = 16 * lh + q;
          const double gv = (double)T0[row * 36 + li];
          st[0] += gv;
          if (p.epi) {
            st[1] += gv * (double)T1[row * 36 + li];
            if (has_2) st[2] += gv * (double)T2[row * 36 + li];
          } else {
            st[1] += gv * gv;
          }
        }
      }
    }
    // column sums: the two lane halves, then the two waves that share these columns (wm = 0, 1), one atomic per column and statistic
    if (want) {
      constexpr int NS = 3;
      const int n = ncol0 + li;
      const bool nok = n < t.N;
#pragma unroll
      for (int k = 0; k < NS; ++k) st[k] += __shfl_xor(st[k], 32, 64);
      const int col = wn * WN + j * 32 + li;
      __syncthreads();                                  // (every wave is done with its tiles: sred aliases them)
      if (wm == 1 && lane < 32) {
#pragma unroll
        for (int k = 0; k < NS; ++k) sred[k * TN + col] = st[k];
      }
      __syncthreads();
      if (wm == 0 && lane < 32 && nok) {
#pragma unroll
        for (int k = 0; k < NS; ++k) st[k] += sred[k * TN + col];
        if (p.epi) {
          double* b1 = stat_replica(p.e_bs, t.N, bid);
          atomic_add_f64(b1 + n, st[0]);
          atomic_add_f64(b1 + t.N + n, st[1]);
          if (has_2) {
            double* b2 = stat_replica(p.e_bs2, t.N, bid);
            atomic_add_f64(b2 + n, st[0]);
            atomic_add_f64(b2 + t.N + n, st[2]);
          }
        } else {
          double* sp = stat_replica(p.stats, t.N, bid);
          atomic_add_f64(sp + n, st[0]);
          atomic_add_f64(sp + t.N + n, st[1]);
        }
      }
    }
  }
}

template <bool W_KN, int MODE, int NT, bool ABF>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_big_kernel(ConvArgs p) {      // two workgroups per CU: <= 256 registers
  __shared__ __attribute__((aligned(16))) float smem[big_lds_floats(NT, MODE)];
  conv_big_body<W_KN, MODE, NT, ABF>(p, blockIdx.x, smem);
}
template <bool W_KN, int MODE, int NT, bool ABF>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_big_pair_kernel(ConvArgs a, ConvArgs b, int nblk_a) {
  __shared__ __attribute__((aligned(16))) float smem[big_lds_floats(NT, MODE)];
  if ((int)blockIdx.x < nblk_a) conv_big_body<W_KN, MODE, NT, ABF>(a, blockIdx.x, smem);
  else conv_big_body<W_KN, MODE, NT, ABF>(b, blockIdx.x - nblk_a, smem);
}
// HP_CONV_BF16X3 form of the big bodies: three images per operand in ONE staging buffer (48 | 60 KB + coefficients): two workgroups per CU
// (one instantiation per weight layout and tile width: the input BatchNorm is a run-time branch of the MODE = 1 loader)
template <bool W_KN, int NT>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_big3_kernel(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[big_lds_floats(NT, 1, 2)];
  conv_big_body<W_KN, 1, NT, false, 2>(p, blockIdx.x, smem);
}
template <bool W_KN, int NT>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_big3_pair_kernel(ConvArgs a, ConvArgs b, int nblk_a) {
  __shared__ __attribute__((aligned(16))) float smem[big_lds_floats(NT, 1, 2)];
  if ((int)blockIdx.x < nblk_a) conv_big_body<W_KN, 1, NT, false, 2>(a, blockIdx.x, smem);
  else conv_big_body<W_KN, 1, NT, false, 2>(b, blockIdx.x - nblk_a, smem);
}
// ... and with the weight fragments (HP_CONV_WFRAG): LDS holds the A images alone
template <int NT>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_big3f_kernel(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[big_lds_floats(NT, 1, 3)];
  conv_big_body<false, 1, NT, false, 3>(p, blockIdx.x, smem);
}
template <int NT>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_big3f_pair_kernel(ConvArgs a, ConvArgs b, int nblk_a) {
  __shared__ __attribute__((aligned(16))) float smem[big_lds_floats(NT, 1, 3)];
  if ((int)blockIdx.x < nblk_a) conv_big_body<false, 1, NT, false, 3>(a, blockIdx.x, smem);
  else conv_big_body<false, 1, NT, false, 3>(b, blockIdx.x - nblk_a, smem);
}
// ... and for three-tap stride-1 launches: one A image per K chunk for all three taps (MM = 4)
template <int NT>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_big3s_kernel(ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[big_lds_floats(NT, 1, 4)];
  conv_big_body<false, 1, NT, false, 4>(p, blockIdx.x, smem);
}
template <int NT>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_big3s_pair_kernel(ConvArgs a, ConvArgs b, int nblk_a) {
  __shared__ __attribute__((aligned(16))) float smem[big_lds_floats(NT, 1, 4)];
  if ((int)blockIdx.x < nblk_a) conv_big_body<false, 1, NT, false, 4>(a, blockIdx.x, smem);
  else conv_big_body<false, 1, NT, false, 4>(b, blockIdx.x - nblk_a, smem);
}
#define HP_BIG3_DISPATCH(KERNEL, KN, NT, ...)                                                                     \
  do {                                                                                                            \
    if (NT == 2) { if (KN) hipLaunchKernelGGL((KERNEL<true, 2>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, 2>), __VA_ARGS__); } \
    else         { if (KN) hipLaunchKernelGGL((KERNEL<true, 1>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, 1>), __VA_ARGS__); } \
  } while (0)

static ConvArgs conv_args_from(const HpOp& op, void* const* bases) {
  ConvArgs a;
  a.t = tapmap_from(op);
  a.A = hp::ptr<const float>(op, 0, bases);
  a.W = hp::ptr<const float>(op, 1, bases);
  a.out = hp::ptr<float>(op, 2, bases);
  a.bias = (op.flags & HP_CONV_BIAS) ? hp::ptr<const float>(op, 3, bases) : nullptr;
  a.stats = (op.flags & HP_CONV_STATS) ? hp::ptr<double>(op, 4, bases) : nullptr;
  a.A2 = hp::ptr<const float>(op, 10, bases);
  a.W2 = hp::ptr<const float>(op, 11, bases);
  a.bn_eval = (op.flags & HP_CONV_BN_EVAL) ? 1 : 0;
  a.act = (op.flags & HP_CONV_ACT) ? 1 : 0;
  a.gamma = a.beta = a.res = nullptr;
  a.rmean = a.rvar = nullptr;
  a.eps = op.f[0]; a.slope = op.f[1];
  if (a.bn_eval || (op.flags & HP_CONV_IN_BN)) {
    a.gamma = hp::ptr<const float>(op, 5, bases); a.beta = hp::ptr<const float>(op, 6, bases);
    a.rmean = hp::ptr<float>(op, 7, bases); a.rvar = hp::ptr<float>(op, 8, bases);
  }
  if (a.bn_eval) a.res = hp::ptr<const float>(op, 9, bases);
  a.in_bn = (op.flags & HP_CONV_IN_BN) ? 1 : 0;
  a.in_Mstat = op.i[31] * (op.i[32] > 1 ? op.i[32] : 1);
  a.in_stats = hp::ptr<const double>(op, 12, bases);
  a.in_save = hp::ptr<float>(op, 13, bases);
  a.in_coef = hp::ptr<float>(op, 14, bases);
  a.in_slope = op.f[2]; a.in_eps = op.f[3]; a.in_mom = op.f[4];
  a.epi = (op.flags & HP_CONV_EPI_BNRED) ? 1 : 0;
  a.e_g2 = hp::ptr<const float>(op, 15, bases); a.e_act = hp::ptr<const float>(op, 16, bases);
  a.e_raw = hp::ptr<const float>(op, 17, bases); a.e_save = hp::ptr<const float>(op, 18, bases);
  a.e_coef = hp::ptr<const float>(op, 19, bases); a.e_bs = hp::ptr<double>(op, 20, bases);
  a.e_raw2 = hp::ptr<const float>(op, 21, bases); a.e_save2 = hp::ptr<const float>(op, 22, bases);
  a.e_bs2 = hp::ptr<double>(op, 23, bases);
  a.e_slope = op.f[5];
  a.Wf = a.Wf2 = nullptr;
  if (op.flags & HP_CONV_WFRAG) { a.Wf = hp::ptr<const char>(op, 24, bases); a.Wf2 = hp::ptr<const char>(op, 25, bases); }
  return a;
}

// operand-loader mode of a launch: 1 = HP_CONV_IN_BN (checked per op at run time inside the instantiation, so a pair
// may mix it with plain members), 0 = plain
static int conv_mode(int flags) { return (flags & HP_CONV_IN_BN) ? 1 : 0; }
// matrix mode: 0 = fp32 matrix cores, 1 = HP_CONV_BF16, 2 = HP_CONV_BF16X3
static int conv_mm(int flags) { return (flags & HP_CONV_BF16) ? 1 : (flags & HP_CONV_BF16X3) ? ((flags & HP_CONV_WFRAG) ? 3 : 2) : 0; }

#define HP_CONV_DISPATCH(KERNEL, KN, MODE, BF, ABF, ...)                                                \
  do {                                                                                                  \
    if (BF == 3)          { hipLaunchKernelGGL((KERNEL<false, 1, 3>), __VA_ARGS__); }      /* (the weight layout only shapes the B loader, which this form does not have) */ \
    else if (BF == 2)     { if (KN) hipLaunchKernelGGL((KERNEL<true, 1, 2>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, 1, 2>), __VA_ARGS__); } \
    else if (BF && ABF)        { if (KN) hipLaunchKernelGGL((KERNEL<true, 1, 1, true>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, 1, 1, true>), __VA_ARGS__); } \
    else if (BF)          { if (KN) hipLaunchKernelGGL((KERNEL<true, 1, 1>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, 1, 1>), __VA_ARGS__); } \
    else if (MODE == 1)   { if (KN) hipLaunchKernelGGL((KERNEL<true, 1, 0>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, 1, 0>), __VA_ARGS__); } \
    else                  { if (KN) hipLaunchKernelGGL((KERNEL<true, 0, 0>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, 0, 0>), __VA_ARGS__); } \
  } while (0)

// Which body runs a bf16 launch: 0 = the 64x64 one, 1 = 128x64 tiles, 2 = 128x128 tiles — the big bodies from two tiles per CU on
// (below that a layer does not fill the chip with 128-row tiles: batch 512).  (the knob: unit tests force the big bodies onto small shapes)
static int conv_big_min_tiles() {
  static const int v = [] {
    const char* on = getenv("HIPPIE_DEBUG_KNOBS");
    const char* e = (on && on[0] == '1') ? getenv("HIPPIE_CONV_BIG_MIN_TILES") : nullptr;
    return e ? atoi(e) : 512;
  }();
  return v;
}
// (A/B knob: HIPPIE_CONV_BIGFRAG=0 keeps the 128-row three-term bodies on the LDS-staged weight tile)
static bool conv_bigfrag_on() {
  static const bool v = [] {
    const char* on = getenv("HIPPIE_DEBUG_KNOBS");
    const char* e = (on && on[0] == '1') ? getenv("HIPPIE_CONV_BIGFRAG") : nullptr;
    return e ? atoi(e) != 0 : true;
  }();
  return v;
}
// three taps reading rows m - 1, m, m + 1 of ONE source tensor at stride 1: the launches conv_big_body's MM = 4 form serves
// (the knob: HIPPIE_CONV_BIGSHARED=0 for the A/B)
static bool conv_shared_taps(const TapMap& t) {
  static const bool on = [] {
    const char* k = getenv("HIPPIE_DEBUG_KNOBS");
    const char* e = (k && k[0] == '1') ? getenv("HIPPIE_CONV_BIGSHARED") : nullptr;
    return e ? atoi(e) != 0 : true;
  }();
  if (!on || t.ntaps != 3 || t.a != 1 || t.sh != 0 || t.Lin != t.Lout || t.P != t.Lout) return false;
  int seen = 0;
  for (int j = 0; j < 3; ++j) {
    if (t.tap_src[j] != 0 || t.tap_o[j] < -1 || t.tap_o[j] > 1) return false;
    seen |= 1 << (t.tap_o[j] + 1);
  }
  return seen == 7;
}
static int conv_big_nt(const TapMap& t) {
  const int rows = hp::cdiv(t.M, 128);
  if (t.N >= 128 && rows * hp::cdiv(t.N, 128) >= conv_big_min_tiles()) return 2;
  if (rows * hp::cdiv(t.N, 64) >= conv_big_min_tiles()) return 1;
  return 0;
}
#define HP_BIG_DISPATCH4(KERNEL, KN, MODE, NTV, ABFV, ...)                                                                                         \
  do {                                                                                                                                            \
    if (MODE == 1) { if (KN) hipLaunchKernelGGL((KERNEL<true, 1, NTV, ABFV>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, 1, NTV, ABFV>), __VA_ARGS__); } \
    else           { if (KN) hipLaunchKernelGGL((KERNEL<true, 0, NTV, ABFV>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<false, 0, NTV, ABFV>), __VA_ARGS__); } \
  } while (0)
#define HP_BIG_DISPATCH(KERNEL, KN, MODE, NT, ABF, ...)                                              \
  do {                                                                                               \
    if (NT == 2) { if (ABF) HP_BIG_DISPATCH4(KERNEL, KN, MODE, 2, true, __VA_ARGS__); else HP_BIG_DISPATCH4(KERNEL, KN, MODE, 2, false, __VA_ARGS__); } \
    else         { if (ABF) HP_BIG_DISPATCH4(KERNEL, KN, MODE, 1, true, __VA_ARGS__); else HP_BIG_DISPATCH4(KERNEL, KN, MODE, 1, false, __VA_ARGS__); } \
  } while (0)

hipError_t hp::launch_conv_pair(const HpOp& opa, const HpOp& opb, void* const* bases, hipStream_t s) {
  if ((opa.flags & 1) != (opb.flags & 1) || (opa.flags & (HP_CONV_BF16 | HP_CONV_BF16X3 | HP_CONV_WFRAG)) != (opb.flags & (HP_CONV_BF16 | HP_CONV_BF16X3 | HP_CONV_WFRAG)) ||
      (opa.flags & HP_FLAG_ACT_BF16) != (opb.flags & HP_FLAG_ACT_BF16)) return hipErrorInvalidValue;
  const bool abf = opa.flags & HP_FLAG_ACT_BF16;
  if (abf && !(opa.flags & HP_CONV_BF16)) return hipErrorInvalidValue;      // bf16-stored activations only with the bf16 matrix path
  const int ma = conv_mode(opa.flags), mb = conv_mode(opb.flags);
  const ConvArgs a = conv_args_from(opa, bases), b = conv_args_from(opb, bases);
  if (opa.flags & (HP_CONV_BF16 | HP_CONV_BF16X3)) {
    const int big = std::min(conv_big_nt(a.t), conv_big_nt(b.t));
    if (big > 0) {
      const int tn = 64 * big;
      const int na = hp::cdiv(a.t.M, 128) * hp::cdiv(a.t.N, tn), nb = hp::cdiv(b.t.M, 128) * hp::cdiv(b.t.N, tn);
      const bool kn = opa.flags & 1;
      const int mode = ma > mb ? ma : mb;
      if ((opa.flags & HP_CONV_BF16X3) && (opa.flags & HP_CONV_WFRAG) && conv_bigfrag_on() && conv_shared_taps(a.t) && conv_shared_taps(b.t)) {
        if (big == 2) hipLaunchKernelGGL((conv_big3s_pair_kernel<2>), dim3(na + nb), dim3(kBigThreads), 0, s, a, b, na);
        else          hipLaunchKernelGGL((conv_big3s_pair_kernel<1>), dim3(na + nb), dim3(kBigThreads), 0, s, a, b, na);
      } else if ((opa.flags & HP_CONV_BF16X3) && (opa.flags & HP_CONV_WFRAG) && conv_bigfrag_on()) {
        if (big == 2) hipLaunchKernelGGL((conv_big3f_pair_kernel<2>), dim3(na + nb), dim3(kBigThreads), 0, s, a, b, na);
        else          hipLaunchKernelGGL((conv_big3f_pair_kernel<1>), dim3(na + nb), dim3(kBigThreads), 0, s, a, b, na);
      } else if (opa.flags & HP_CONV_BF16X3) HP_BIG3_DISPATCH(conv_big3_pair_kernel, kn, big, dim3(na + nb), dim3(kBigThreads), 0, s, a, b, na);
      else HP_BIG_DISPATCH(conv_big_pair_kernel, kn, mode, big, abf, dim3(na + nb), dim3(kBigThreads), 0, s, a, b, na);
      return hipGetLastError();
    }
  }
  const int na = hp::cdiv(a.t.M, 64) * hp::cdiv(a.t.N, 64), nb = hp::cdiv(b.t.M, 64) * hp::cdiv(b.t.N, 64);
  const bool kn = opa.flags & 1;
  const int bf = conv_mm(opa.flags);
  const int mode = ma > mb ? ma : mb;
  const dim3 g(na + nb), th(kConvThreads);
  HP_CONV_DISPATCH(conv_taps_pair_kernel, kn, mode, bf, abf, g, th, 0, s, a, b, na);
  return hipGetLastError();
}

hipError_t hp::launch_conv_taps(const HpOp& op, void* const* bases, hipStream_t s) {
  const ConvArgs a = conv_args_from(op, bases);
  const bool abf = op.flags & HP_FLAG_ACT_BF16;
  if (abf && !(op.flags & HP_CONV_BF16)) return hipErrorInvalidValue;
  if (op.flags & (HP_CONV_BF16 | HP_CONV_BF16X3)) {
    const int big = conv_big_nt(a.t);
    if (big > 0) {
      const bool kn = op.flags & 1;
      const int mode = conv_mode(op.flags);
      const dim3 g(hp::cdiv(a.t.M, 128) * hp::cdiv(a.t.N, 64 * big));
      if ((op.flags & HP_CONV_BF16X3) && (op.flags & HP_CONV_WFRAG) && conv_bigfrag_on() && conv_shared_taps(a.t)) {
        if (big == 2) hipLaunchKernelGGL((conv_big3s_kernel<2>), g, dim3(kBigThreads), 0, s, a);
        else          hipLaunchKernelGGL((conv_big3s_kernel<1>), g, dim3(kBigThreads), 0, s, a);
      } else if ((op.flags & HP_CONV_BF16X3) && (op.flags & HP_CONV_WFRAG) && conv_bigfrag_on()) {
        if (big == 2) hipLaunchKernelGGL((conv_big3f_kernel<2>), g, dim3(kBigThreads), 0, s, a);
        else          hipLaunchKernelGGL((conv_big3f_kernel<1>), g, dim3(kBigThreads), 0, s, a);
      } else if (op.flags & HP_CONV_BF16X3) HP_BIG3_DISPATCH(conv_big3_kernel, kn, big, g, dim3(kBigThreads), 0, s, a);
      else HP_BIG_DISPATCH(conv_big_kernel, kn, mode, big, abf, g, dim3(kBigThreads), 0, s, a);
      return hipGetLastError();
    }
  }
  const dim3 g(hp::cdiv(a.t.M, 64) * hp::cdiv(a.t.N, 64)), th(kConvThreads);
  const bool kn = op.flags & 1;
  const int bf = conv_mm(op.flags);
  const int mode = conv_mode(op.flags);
  HP_CONV_DISPATCH(conv_taps_kernel, kn, mode, bf, abf, g, th, 0, s, a);
  return hipGetLastError();
}
#undef HP_CONV_DISPATCH

// ------------------------------------------------------------------------------------
// slab[split][tap_w][n][k] = sum_{m in split} DY[m][n] * X[src(m,tap)][k]
// Contraction over rows: both MFMA operands are read K-major from LDS with ds_read_b32
// (lane = output row / column, conflict-free).  One DY slice feeds all taps.
// ------------------------------------------------------------------------------------
struct WgradArgs {
  const float* DY; const float* X; float* slab;
  const float* coef;   // HP_CONV_IN_BN: X is a raw BatchNorm input; the operand is leaky_relu(fma(x, scale, shift))
  float slope;
  TapMap t;
  int nsplit, rows_per_split, slab_stride;
  int atomic;   // 1: accumulate into `slab` (= the zeroed gradient tensor) with fp32 atomics, no slabs
  int shared;   // 1 (set by build_wgrad_group): a three-tap stride-1 problem of the three-term mode run by wgrad3s_body (128 x 64 tiles)
};


constexpr int wgrad_lds(int nt, int mm) { return (1 + nt) * (mm == 2 ? 3 * kSplitBkn / 2 : 32 * 64); }      // floats (MM = 2: three bf16 images per operand)
template <int NT, int MM = 0, bool ABF = false>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int tile, const int split, float* smem) {
  constexpr bool BF16 = MM == 1, SPLIT = MM == 2;
  constexpr int T = SPLIT ? 3 * kSplitBkn / 2 : 32 * 64;   // floats of one operand: a [32 rows][64 cols] image (SPLIT: three bf16 images of 192-byte rows)
  const TapMap& t = p.t;
  const float* gDY = p.DY;
  const float* gX = p.X;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ntc = (t.K + 63) >> 6;
  const int n0 = (tile / ntc) << 6, c0 = (tile % ntc) << 6;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(t.M, mbeg + p.rows_per_split);

  const int lr = tid >> 4, cq = (tid & 15) << 2;
  float4 rdy[2], rx[NT][2];
  // (sample, position) of this thread's two rows, advanced by 32 rows per slice without divisions
  int rb_[2], rl_[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = mbeg + lr + 16 * j;
    rb_[j] = m / t.Lout;
    rl_[j] = m - rb_[j] * t.Lout;
  }
  const int q32 = 32 / t.Lout, r32 = 32 - q32 * t.Lout;
  // input BatchNorm coefficients of this thread's four channels (fixed for the whole block)
  const bool x_bn = p.coef != nullptr;
  float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (x_bn && c0 + cq < t.K) {
    xsc = gload4(p.coef + c0 + cq);
    xsh = gload4(p.coef + t.K + c0 + cq);
  }
  const float xslope = p.slope;
  unsigned okmask = 0;     // bit 2*tau + j: rx[tau][j] holds a real row (the transform must not touch padding zeros)
  auto load_regs = [&](int mb) {
    okmask = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = mb + lr + 16 * j;
      const bool mv = m < mend;
      rdy[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mv && n0 + cq < t.N) rdy[j] = aload4<ABF>(gDY, (size_t)m * t.N + n0 + cq);
      const int b = rb_[j];
      const int al = t.a * rl_[j];
      rl_[j] += r32; rb_[j] += q32;
      if (rl_[j] >= t.Lout) { rl_[j] -= t.Lout; ++rb_[j]; }
#pragma unroll
      for (int tau = 0; tau < NT; ++tau) {
        const int pos = al + t.tap_o[tau];
        const bool ok = mv && pos >= 0 && pos < t.P && (c0 + cq < t.K);
        rx[tau][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
          rx[tau][j] = aload4<ABF>(gX, (size_t)(b * t.Lin + (pos >> t.sh)) * t.K + c0 + cq);
          okmask |= 1u << (2 * tau + j);
        }
      }
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (SPLIT) {
        bf16x4 h, m, l;
        split3(rdy[j], h, m, l);
        __bf16* d = reinterpret_cast<__bf16*>(smem) + (lr + 16 * j) * kLdtH + cq;
        *reinterpret_cast<bf16x4*>(d) = h; *reinterpret_cast<bf16x4*>(d + kSplitBkn) = m; *reinterpret_cast<bf16x4*>(d + 2 * kSplitBkn) = l;
      } else if (BF16) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(smem) + (lr + 16 * j) * kLdtH + cq) = to_bf16x4(rdy[j]);
      else *reinterpret_cast<float4*>(smem + (lr + 16 * j) * 64 + cq) = rdy[j];
#pragma unroll
      for (int tau = 0; tau < NT; ++tau) {
        float4 v = rx[tau][j];
        if (x_bn) {
          // the activation the forward conv consumed, re-evaluated bit for bit — HERE, at LDS-store time: the loads
          // were issued one slice earlier and have been in flight under the MFMA block (a transform inside
          // load_regs would wait for them before the MFMAs)
          // (leaky_relu as max(f, f * slope), 0 <= slope <= 1 validated: the same value as the forward conv's loader forms)
          const float keep = (okmask >> (2 * tau + j)) & 1u ? 1.f : 0.f;
          const float fx = fmaf(v.x, xsc.x, xsh.x), fy = fmaf(v.y, xsc.y, xsh.y), fz = fmaf(v.z, xsc.z, xsh.z), fw = fmaf(v.w, xsc.w, xsh.w);
          v.x = fmaxf(fx, fx * xslope) * keep; v.y = fmaxf(fy, fy * xslope) * keep;
          v.z = fmaxf(fz, fz * xslope) * keep; v.w = fmaxf(fw, fw * xslope) * keep;
        }
        if (SPLIT) {
          bf16x4 h, m, l;
          split3(v, h, m, l);
          __bf16* d = reinterpret_cast<__bf16*>(smem + (1 + tau) * T) + (lr + 16 * j) * kLdtH + cq;
          *reinterpret_cast<bf16x4*>(d) = h; *reinterpret_cast<bf16x4*>(d + kSplitBkn) = m; *reinterpret_cast<bf16x4*>(d + 2 * kSplitBkn) = l;
        } else if (BF16) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(smem + (1 + tau) * T) + (lr + 16 * j) * kLdtH + cq) = to_bf16x4(v);
        else *reinterpret_cast<float4*>(smem + (1 + tau) * T + (lr + 16 * j) * 64 + cq) = v;
      }
    }
  };

  // ---- bf16-STORED operands (ABF; HP_FLAG_ACT_BF16): a thread holds 8 consecutive bf16 of ONE row per operand — one 16-byte load and one
  // 16-byte LDS store per operand and slice, no conversion at all unless the input BatchNorm is re-evaluated (the 4-element form above
  // would issue 8-byte loads and unpack to fp32 only to repack: twice the memory instructions per byte, measured 16.8 against 8.3 ms)
  const int lr8 = tid >> 3, c8 = (tid & 7) << 3;
  hp_v4u hdy = {0u, 0u, 0u, 0u}, hx[NT];
  int rb8 = (mbeg + lr8) / t.Lout, rl8 = (mbeg + lr8) - rb8 * t.Lout;
  float xsc8[8], xsh8[8];
  if (ABF) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool in = x_bn && c0 + c8 + e < t.K;
      xsc8[e] = in ? gload1(p.coef + c0 + c8 + e) : 1.f;
      xsh8[e] = in ? gload1(p.coef + t.K + c0 + c8 + e) : 0.f;
    }
  }
  unsigned ok8 = 0;        // bit tau: hx[tau] holds a real row
  auto load_regs_h = [&](int mb) {
    const int m = mb + lr8;
    const bool mv = m < mend;
    const hp_v4u zero = {0u, 0u, 0u, 0u};
    hdy = zero;
    if (mv && n0 + c8 < t.N) hdy = *(const hp_v4u __attribute__((address_space(1)))*)(reinterpret_cast<const char*>(gDY) + ((size_t)m * t.N + n0 + c8) * 2);
    const int b = rb8, al = t.a * rl8;
    rl8 += r32; rb8 += q32;
    if (rl8 >= t.Lout) { rl8 -= t.Lout; ++rb8; }
    ok8 = 0;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
      const int pos = al + t.tap_o[tau];
      const bool ok = mv && pos >= 0 && pos < t.P && (c0 + c8 < t.K);
      hx[tau] = zero;
      if (ok) {
        hx[tau] = *(const hp_v4u __attribute__((address_space(1)))*)(reinterpret_cast<const char*>(gX) + ((size_t)(b * t.Lin + (pos >> t.sh)) * t.K + c0 + c8) * 2);
        ok8 |= 1u << tau;
      }
    }
  };
  auto store_lds_h = [&]() {
    __bf16* dyh = reinterpret_cast<__bf16*>(smem);
    *reinterpret_cast<hp_v4u*>(dyh + lr8 * kLdtH + c8) = hdy;
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
      hp_v4u v = hx[tau];
      if (x_bn) {
        const float keep = (ok8 >> tau) & 1u ? 1.f : 0.f;
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = __uint_as_float(w[q] << 16), hi = __uint_as_float(w[q] & 0xffff0000u);
          const float fl = fmaf(lo, xsc8[2 * q], xsh8[2 * q]), fh = fmaf(hi, xsc8[2 * q + 1], xsh8[2 * q + 1]);
          const float al_ = fmaxf(fl, fl * xslope) * keep, ah = fmaxf(fh, fh * xslope) * keep;
          w[q] = (unsigned)f32_to_bf16_bits(al_) | ((unsigned)f32_to_bf16_bits(ah) << 16);
        }
        v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
      }
      *reinterpret_cast<hp_v4u*>(reinterpret_cast<__bf16*>(smem + (1 + tau) * T) + lr8 * kLdtH + c8) = v;
    }
  };
  auto LOAD = [&](int mb) { if constexpr (ABF) load_regs_h(mb); else load_regs(mb); };
  auto STORE = [&]() { if constexpr (ABF) store_lds_h(); else store_lds(); };

  f32x16 acc[NT];
#pragma unroll
  for (int tau = 0; tau < NT; ++tau)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tau][r] = 0.f;

  if (mbeg < mend) {
    LOAD(mbeg);
    STORE();
    __syncthreads();
    for (int mb = mbeg; mb < mend; mb += 32) {
      const bool more = mb + 32 < mend;
      if (more) LOAD(mb + 32);
      if (SPLIT) {
        // three-term operands (HP_CONV_BF16X3): six products per tap and 16-row half slice, corrections first (smallest first)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          bf16x8 af[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) af[c] = tr_operand(reinterpret_cast<const __bf16*>(smem) + c * kSplitBkn, kLdtH, st * 16, wn * 32, lane);
#pragma unroll
          for (int tau = 0; tau < NT; ++tau) {
            bf16x8 bf[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) bf[c] = tr_operand(reinterpret_cast<const __bf16*>(smem + (1 + tau) * T) + c * kSplitBkn, kLdtH, st * 16, wc * 32, lane);
            acc[tau] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc[tau], 0, 0, 0);
            acc[tau] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc[tau], 0, 0, 0);
            acc[tau] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[tau], 0, 0, 0);
            acc[tau] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[tau], 0, 0, 0);
            acc[tau] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[tau], 0, 0, 0);
            acc[tau] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[tau], 0, 0, 0);
          }
        }
        __syncthreads();
        if (more) {
          STORE();
          __syncthreads();
        }
        continue;
      }
      if (BF16) {
        // both operands are TRANSPOSES of the row-major [32 rows m][64 columns] bf16 images: hardware transpose reads,
        // two v_mfma_f32_32x32x16_bf16 per tap and slice
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x8 af = tr_operand(reinterpret_cast<const __bf16*>(smem), kLdtH, st * 16, wn * 32, lane);
#pragma unroll
          for (int tau = 0; tau < NT; ++tau) {
            const bf16x8 bf = tr_operand(reinterpret_cast<const __bf16*>(smem + (1 + tau) * T), kLdtH, st * 16, wc * 32, lane);
            acc[tau] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[tau], 0, 0, 0);
          }
        }
        __syncthreads();
        if (more) {
          STORE();
          __syncthreads();
        }
        continue;
      }
      const float* dys = smem + lh * 64 + wn * 32 + li;
      const float* xs = smem + T + lh * 64 + wc * 32 + li;
      // all operand reads of the slice go out before the MFMA block (distinct registers), so the
      // matrix pipe never waits on an LDS round trip inside the slice
      float av[16], bv[NT][16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        av[kk] = dys[kk * 128];
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) bv[tau][kk] = xs[tau * T + kk * 128];
      }
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
        for (int tau = 0; tau < NT; ++tau)
          acc[tau] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[tau][kk], acc[tau], 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        STORE();
        __syncthreads();
      }
    }
  }

  const int c = c0 + wc * 32 + li;
  if (c < t.K) {
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
      float* dst = p.slab + (p.atomic ? (size_t)0 : (size_t)split * p.slab_stride) + (size_t)t.tap_w[tau] * t.N * t.K;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < t.N) {
          // a half-wave writes/adds one 128-byte row segment: the full-rate shape for float atomics
          if (p.atomic) atomic_add_f32_global(dst + (size_t)n * t.K + c, acc[tau][r]);
          else dst[(size_t)n * t.K + c] = acc[tau][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Three-tap stride-1 weight gradient in the three-term mode (HP_CONV_BF16X3): the conv layers whose taps read rows m - 1, m, m + 1 of the
// SAME tensor (every 3-wide stride-1 Conv1d: most of a backward pass's weight-gradient work).  The general body above stages one X image
// per tap — three loads, three splits, three LDS images of what is one tensor shifted by a row.  Here ONE image of X (the slice's 32 rows
// plus a halo row on either side) serves all three taps: a tap is a row offset of the hardware-transpose read.  What the shift cannot
// express — a row whose neighbour lies in the next sample contributes nothing to that tap — is a mask on the DY fragment of that tap
// (the lane's 8 contraction rows: zeroed where l(m) + tap offset leaves [0, L)); slabs without a sample boundary skip it.
// 128 (n) x 64 (k) x 3 taps per 256-thread workgroup, 4 waves of 64 x 32 x 3 = 6 accumulator tiles: every DY fragment feeds 3 taps, every X
// fragment 2 row tiles — 30 fragment reads for 36 MFMAs per 16-row slab against 24 for 18 in the general body, and a third of its staging work.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int kW3DyLd = 160, kW3XLd = 96;                 // bf16 row strides: 128 + 32 / 64 + 32 columns (rows 64 bytes apart mod 256: conflict-free transpose reads)
constexpr int kW3DyPlane = 32 * kW3DyLd, kW3XPlane = 34 * kW3XLd;
constexpr int kW3Lds = 3 * (kW3DyPlane + kW3XPlane) / 2;  // floats

__device__ __forceinline__ void wgrad3s_body(const WgradArgs& p, const int tile, const int split, float* smem) {
  const TapMap& t = p.t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ntc = (t.K + 63) >> 6;
  const int n0 = (tile / ntc) << 7, c0 = (tile % ntc) << 6;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(t.M, mbeg + p.rows_per_split);
  const int L = t.Lout;
  __bf16* const dyI = reinterpret_cast<__bf16*>(smem);
  __bf16* const xI = dyI + 3 * kW3DyPlane;
  const int dr = tid >> 5, dc = (tid & 31) << 2;          // DY slots: rows dr + 8 j (j < 4), columns dc .. dc + 3 of the tile's 128
  const int xr = tid >> 4, xc = (tid & 15) << 2;          // X slots: image rows xr + 16 j (j < 3, 34 rows: tensor rows mb - 1 .. mb + 32), columns xc .. + 3 of 64
  const bool x_bn = p.coef != nullptr;
  float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (x_bn && c0 + xc < t.K) {
    xsc = gload4(p.coef + c0 + xc);
    xsh = gload4(p.coef + t.K + c0 + xc);
  }
  const float xslope = p.slope;
  float4 rdy[4], rx[3];
  auto load_regs = [&](const int mb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mb + dr + 8 * j;
      rdy[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < mend && n0 + dc < t.N) rdy[j] = gload4(p.DY + (size_t)m * t.N + n0 + dc);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int r = xr + 16 * j, m = mb - 1 + r;
      rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < 34 && m >= 0 && m < t.M && c0 + xc < t.K) rx[j] = gload4(p.X + (size_t)m * t.K + c0 + xc);      // (rows of other splits / samples are real data: the DY side decides what counts)
    }
  };
  auto store_lds = [&]() {
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      split3(rdy[j], h, m, l);
      __bf16* d = dyI + (dr + 8 * j) * kW3DyLd + dc;
      *reinterpret_cast<bf16x4*>(d) = h; *reinterpret_cast<bf16x4*>(d + kW3DyPlane) = m; *reinterpret_cast<bf16x4*>(d + 2 * kW3DyPlane) = l;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int r = xr + 16 * j;
      if (r < 34) {
        float4 v = rx[j];
        if (x_bn) {      // the activation the forward conv consumed, re-evaluated bit for bit (as wgrad_body)
          const float fx = fmaf(v.x, xsc.x, xsh.x), fy = fmaf(v.y, xsc.y, xsh.y), fz = fmaf(v.z, xsc.z, xsh.z), fw = fmaf(v.w, xsc.w, xsh.w);
          v.x = fmaxf(fx, fx * xslope); v.y = fmaxf(fy, fy * xslope); v.z = fmaxf(fz, fz * xslope); v.w = fmaxf(fw, fw * xslope);
        }
        split3(v, h, m, l);
        __bf16* d = xI + r * kW3XLd + xc;
        *reinterpret_cast<bf16x4*>(d) = h; *reinterpret_cast<bf16x4*>(d + kW3XPlane) = m; *reinterpret_cast<bf16x4*>(d + 2 * kW3XPlane) = l;
      }
    }
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int tau = 0; tau < 3; ++tau)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tau][i][r] = 0.f;

  __syncthreads();                       // (the body may follow another one in the same workgroup)
  if (mbeg < mend) {
    // position within its sample of the first of this lane's 8 contraction rows of a slab, advanced by 16 rows per slab without divisions
    int l0 = (mbeg + 8 * lh) % L;
    const int r16 = 16 % L;
    load_regs(mbeg);
    store_lds();
    __syncthreads();
    for (int mb = mbeg; mb < mend; mb += 32) {
      const bool more = mb + 32 < mend;
      if (more) load_regs(mb + 32);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        // rows of this lane that start (tap offset -1 invalid) / end (offset +1 invalid) a sample
        unsigned first = 0, last = 0;
        {
          int le = l0;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            first |= (le == 0 ? 1u : 0u) << e;
            last |= (le == L - 1 ? 1u : 0u) << e;
            le = le + 1 == L ? 0 : le + 1;
          }
          l0 += r16;
          if (l0 >= L) l0 -= L;
        }
        bf16x8 af[2][3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int i = 0; i < 2; ++i) af[i][c] = tr_operand(dyI + c * kW3DyPlane, kW3DyLd, st * 16, wn * 64 + i * 32, lane);
#pragma unroll
        for (int tau = 0; tau < 3; ++tau) {
          bf16x8 bf[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) bf[c] = tr_operand(xI + c * kW3XPlane, kW3XLd, st * 16 + tau, wc * 32, lane);
          bf16x8 am[2][3];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) am[i][c] = af[i][c];
          const unsigned bad = tau == 0 ? first : (tau == 2 ? last : 0u);
          if (tau != 1 && __builtin_amdgcn_ballot_w64(bad != 0) != 0) {      // (uniform) a sample boundary inside this slab
            hp_v4u keep;
            keep.x = ((bad & 1u) ? 0u : 0x0000ffffu) | ((bad & 2u) ? 0u : 0xffff0000u);
            keep.y = ((bad & 4u) ? 0u : 0x0000ffffu) | ((bad & 8u) ? 0u : 0xffff0000u);
            keep.z = ((bad & 16u) ? 0u : 0x0000ffffu) | ((bad & 32u) ? 0u : 0xffff0000u);
            keep.w = ((bad & 64u) ? 0u : 0x0000ffffu) | ((bad & 128u) ? 0u : 0xffff0000u);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                union { bf16x8 v; hp_v4u u; } w;
                w.v = am[i][c];
                w.u.x &= keep.x; w.u.y &= keep.y; w.u.z &= keep.z; w.u.w &= keep.w;
                am[i][c] = w.v;
              }
          }
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            constexpr int ca[6] = {0, 2, 1, 0, 1, 0}, cb[6] = {2, 0, 1, 1, 0, 0};      // the five corrections (smallest first), then the leading product
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[tau][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i][ca[q]], bf[cb[q]], acc[tau][i], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      if (more) {
        store_lds();
        __syncthreads();
      }
    }
  }

  const int c = c0 + wc * 32 + li;
  if (c < t.K) {
#pragma unroll
    for (int tau = 0; tau < 3; ++tau) {
      float* dst = p.slab + (p.atomic ? (size_t)0 : (size_t)split * p.slab_stride) + (size_t)t.tap_w[tau] * t.N * t.K;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (n < t.N) {
            if (p.atomic) atomic_add_f32_global(dst + (size_t)n * t.K + c, acc[tau][i][r]);
            else dst[(size_t)n * t.K + c] = acc[tau][i][r];
          }
        }
    }
  }
}

template <int NT, int MM = 0, bool ABF = false>
__global__ __launch_bounds__(256) void wgrad_taps_kernel(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[wgrad_lds(NT, MM)];
  wgrad_body<NT, MM, ABF>(p, blockIdx.x, blockIdx.y, smem);
}

// Grouped form: ONE launch runs every weight-gradient GEMM of a backward pass.  They are independent
// leaves whose inputs persist, so their tiles fill the chip together (no per-layer tail, fewer K-splits
// and atomics, one launch instead of ~38).  blocks[b] = (problem, tile, split, -).
#ifndef HP_WGRAD_PAD_FLOATS
#define HP_WGRAD_PAD_FLOATS 0      // experiment (profiles/r03_wgrad_occupancy_ab.txt): extra LDS per workgroup caps the group launch's workgroups per CU
#endif
template <int NT, int MM = 0, bool ABF = false>
__global__ __launch_bounds__(256) void wgrad_group_kernel(const WgradArgs* __restrict__ probs, const int4* __restrict__ blocks) {
  __shared__ __attribute__((aligned(16))) float smem[wgrad_lds(NT, MM) + HP_WGRAD_PAD_FLOATS];
  const int4 bi = blocks[blockIdx.x];
  const int pj = __builtin_amdgcn_readfirstlane(bi.x);
  const int tile = __builtin_amdgcn_readfirstlane(bi.y);
  const int split = __builtin_amdgcn_readfirstlane(bi.z);
  // by value: one scalar load of the problem record up front.  Through a reference into global memory the
  // compiler re-loads the fields (s_load + wait) inside every guarded load of the slice loop.
  const WgradArgs p = probs[pj];
  wgrad_body<NT, MM, ABF>(p, tile, split, smem);
}
// the three-term mode's group: problems flagged `shared` run on wgrad3s_body.  (Two workgroups per CU: at most 256 registers.)
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_group3_kernel(const WgradArgs* __restrict__ probs, const int4* __restrict__ blocks) {
  __shared__ __attribute__((aligned(16))) float smem[wgrad_lds(NT, 2)];
  const int4 bi = blocks[blockIdx.x];
  const int pj = __builtin_amdgcn_readfirstlane(bi.x);
  const int tile = __builtin_amdgcn_readfirstlane(bi.y);
  const int split = __builtin_amdgcn_readfirstlane(bi.z);
  const WgradArgs p = probs[pj];
  if constexpr (NT == 3) {
    if (p.shared) { wgrad3s_body(p, tile, split, smem); return; }      // (uniform per workgroup)
  }
  wgrad_body<NT, 2, false>(p, tile, split, smem);
}

static WgradArgs wgrad_args_from(const HpOp& op, void* const* bases) {
  WgradArgs a;
  a.t = tapmap_from(op);
  a.DY = hp::ptr<const float>(op, 0, bases);
  a.X = hp::ptr<const float>(op, 1, bases);
  a.slab = hp::ptr<float>(op, 2, bases);
  a.coef = (op.flags & HP_CONV_IN_BN) ? hp::ptr<const float>(op, 3, bases) : nullptr;
  a.slope = op.f[0];
  a.nsplit = op.i[22];
  a.rows_per_split = op.i[23];
  a.slab_stride = op.i[24];
  a.atomic = op.flags & 1;
  a.shared = 0;
  return a;
}

// (A/B knob: HIPPIE_WGRAD_SHARED=0 keeps every problem of a group on the general body)
static bool wgrad_shared_on() {
  static const bool v = [] {
    const char* on = getenv("HIPPIE_DEBUG_KNOBS");
    const char* e = (on && on[0] == '1') ? getenv("HIPPIE_WGRAD_SHARED") : nullptr;
    return e ? atoi(e) != 0 : true;
  }();
  return v;
}
// Build the device tables of one WGRAD_GROUP op from its member records ops[first .. first+count).
hipError_t hp::build_wgrad_group(const HpOp* members, int count, void* const* bases, void** d_probs, void** d_blocks, int* nblocks) {
  std::vector<WgradArgs> probs(count);
  std::vector<int4> blocks;
  for (int j = 0; j < count; ++j) {
    probs[j] = wgrad_args_from(members[j], bases);
    WgradArgs& q = probs[j];
    // three-term mode, three taps reading rows m - 1, m, m + 1 of one tensor, at least one full 128-row tile of output channels, summed
    // with atomics (the split count is then the launch's to choose): wgrad3s_body, on 128 x 64 tiles with half the rows per split
    if ((members[j].flags & HP_CONV_BF16X3) && !(members[j].flags & HP_FLAG_ACT_BF16) && q.atomic && q.t.ntaps == 3 && q.t.a == 1 && q.t.sh == 0 &&
        q.t.Lin == q.t.Lout && q.t.P == q.t.Lout && q.t.tap_o[0] == -1 && q.t.tap_o[1] == 0 && q.t.tap_o[2] == 1 && q.t.N >= 128 && wgrad_shared_on()) {
      q.shared = 1;
      q.rows_per_split = std::max(32, hp::cdiv(hp::cdiv(q.rows_per_split, 2), 32) * 32);
      q.nsplit = hp::cdiv(q.t.M, q.rows_per_split);
    }
    const int tiles = hp::cdiv(probs[j].t.N, q.shared ? 128 : 64) * hp::cdiv(probs[j].t.K, 64);
    for (int sp = 0; sp < probs[j].nsplit; ++sp)
      for (int tl = 0; tl < tiles; ++tl) blocks.push_back(make_int4(j, tl, sp, 0));
  }
  // longest blocks first (LPT): the group mixes 12-slice and 100-slice blocks, and a long block that starts
  // last is the tail of the whole launch (measured on the time model's 35-problem group: 757 us -> 675 us)
  std::stable_sort(blocks.begin(), blocks.end(), [&](const int4& a, const int4& b) {
    return probs[a.x].rows_per_split * (1 + probs[a.x].shared) > probs[b.x].rows_per_split * (1 + probs[b.x].shared);
  });
  *d_probs = *d_blocks = nullptr;
  hipError_t e = hipMalloc(d_probs, probs.size() * sizeof(WgradArgs));
  if (e == hipSuccess) e = hipMalloc(d_blocks, blocks.size() * sizeof(int4));
  if (e == hipSuccess) e = hipMemcpy(*d_probs, probs.data(), probs.size() * sizeof(WgradArgs), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(*d_blocks, blocks.data(), blocks.size() * sizeof(int4), hipMemcpyHostToDevice);
  if (e != hipSuccess) {      // nothing half-built survives a failure
    if (*d_probs) hipFree(*d_probs);
    if (*d_blocks) hipFree(*d_blocks);
    *d_probs = *d_blocks = nullptr;
    return e;
  }
  *nblocks = (int)blocks.size();
  return e;
}

hipError_t hp::launch_wgrad_group(int ntaps, int bf16, const void* d_probs, const void* d_blocks, int nblocks, hipStream_t s) {
  // bf16: 0 = fp32 matrix path, 1 = bf16 operands from fp32-stored tensors, 2 = from bf16-stored tensors (HP_FLAG_ACT_BF16),
  // 3 = three-term bf16 operands (HP_CONV_BF16X3)
  const dim3 g(nblocks), th(256);
  const WgradArgs* pr = (const WgradArgs*)d_probs;
  const int4* bl = (const int4*)d_blocks;
  if (bf16 == 2) {
    if (ntaps == 1)      hipLaunchKernelGGL((wgrad_group_kernel<1, 1, true>), g, th, 0, s, pr, bl);
    else if (ntaps == 3) hipLaunchKernelGGL((wgrad_group_kernel<3, 1, true>), g, th, 0, s, pr, bl);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (bf16 == 3) {
    if (ntaps == 1)      hipLaunchKernelGGL((wgrad_group3_kernel<1>), g, th, 0, s, pr, bl);
    else if (ntaps == 3) hipLaunchKernelGGL((wgrad_group3_kernel<3>), g, th, 0, s, pr, bl);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (ntaps == 1 && !bf16)      hipLaunchKernelGGL((wgrad_group_kernel<1, 0>), g, th, 0, s, pr, bl);
  else if (ntaps == 1)          hipLaunchKernelGGL((wgrad_group_kernel<1, 1>), g, th, 0, s, pr, bl);
  else if (ntaps == 3 && !bf16) hipLaunchKernelGGL((wgrad_group_kernel<3, 0>), g, th, 0, s, pr, bl);
  else if (ntaps == 3)          hipLaunchKernelGGL((wgrad_group_kernel<3, 1>), g, th, 0, s, pr, bl);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t hp::launch_wgrad_taps(const HpOp& op, void* const* bases, hipStream_t s) {
  WgradArgs a = wgrad_args_from(op, bases);
  dim3 grid(hp::cdiv(a.t.N, 64) * hp::cdiv(a.t.K, 64), a.nsplit);
  const bool bf16 = op.flags & HP_CONV_BF16;
  if (op.flags & HP_FLAG_ACT_BF16) {
    if (!bf16) return hipErrorInvalidValue;
    if (a.t.ntaps == 1)      hipLaunchKernelGGL((wgrad_taps_kernel<1, 1, true>), grid, dim3(256), 0, s, a);
    else if (a.t.ntaps == 3) hipLaunchKernelGGL((wgrad_taps_kernel<3, 1, true>), grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (op.flags & HP_CONV_BF16X3) {
    if (bf16) return hipErrorInvalidValue;
    if (a.t.ntaps == 1)      hipLaunchKernelGGL((wgrad_taps_kernel<1, 2>), grid, dim3(256), 0, s, a);
    else if (a.t.ntaps == 3) hipLaunchKernelGGL((wgrad_taps_kernel<3, 2>), grid, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (a.t.ntaps == 1 && !bf16)      hipLaunchKernelGGL((wgrad_taps_kernel<1, 0>), grid, dim3(256), 0, s, a);
  else if (a.t.ntaps == 1)          hipLaunchKernelGGL((wgrad_taps_kernel<1, 1>), grid, dim3(256), 0, s, a);
  else if (a.t.ntaps == 3 && !bf16) hipLaunchKernelGGL((wgrad_taps_kernel<3, 0>), grid, dim3(256), 0, s, a);
  else if (a.t.ntaps == 3)          hipLaunchKernelGGL((wgrad_taps_kernel<3, 1>), grid, dim3(256), 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
