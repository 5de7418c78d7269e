/*
 * hippie_hip.h — C ABI of libhippie_hip.so, the MI355X (gfx950) engine behind the
 * HIPPIE cVAE hot path.
 *
 * Boundary.  The reference (aghatpande/HIPPIE) has no FFI: its hot path is the Python
 * class surface of hippie/backbones.py + hippie/model.py executed by stock ATen ops.
 * This library replaces what those modules make ATen do, at the granularity of the
 * reference's own layers.  Each op below cites the reference call site it replaces
 * (paths relative to the reference root).  The host side (hippie_amd/, Python like the
 * reference) lowers one optimisation step of a model into an array of HpOp records — a
 * "program" — which this library executes as HIP kernels on a caller-supplied stream and
 * can capture into a hipGraph.  No torch types cross this boundary: plain device
 * pointers, sizes and POD records only.
 *
 * Memory model.  The caller owns six device arenas and passes their base pointers:
 *   HP_SPACE_WS     workspace (activations, saved tensors, slabs, fp64 statistic slots,
 *                   staged inputs x / labels / eps, scalar outputs)
 *   HP_SPACE_PARAM  fp32 parameters (conv weights stored tap-major [3][Cout][Cin])
 *   HP_SPACE_GRAD   fp32 gradients, same layout as PARAM
 *   HP_SPACE_BUF    BatchNorm running_mean / running_var
 *   HP_SPACE_M/V    AdamW exp_avg / exp_avg_sq, same layout as PARAM
 * A buffer reference inside an op is  (space << 56) | byte_offset ;  HP_NULL = none.
 * Activations are channels-last: tensor [B, C, L] of the reference is stored as the
 * row-major matrix [B*L][C] (row m = b*L + l).
 *
 * Errors: every function returns 0 on success, nonzero otherwise; hp_last_error()
 * returns a static message for the calling thread.  Never aborts.
 * Threading: a program is not thread-safe; all work is stream-ordered and asynchronous.
 */
#ifndef HIPPIE_HIP_H
#define HIPPIE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HP_ABI_VERSION 7

enum {
  HP_SPACE_WS = 0, HP_SPACE_PARAM = 1, HP_SPACE_GRAD = 2, HP_SPACE_BUF = 3,
  HP_SPACE_M = 4, HP_SPACE_V = 5, HP_NUM_SPACES = 6
};
#define HP_NULL ((int64_t)-1)
#define HP_MAX_TAPS 6
/* Every per-channel statistics slot ("STATS", "BS": sum / sum-of-squares style fp64 accumulators) is
 * replicated R = hp_stat_repl(C) times: double[R][2][C].  Producers add into any replica (they pick one by
 * block index, so concurrent same-address atomics spread over R cache lines); consumers use the sum over
 * replicas, taken in replica order.  R shrinks as C grows (R*C ~ 1024): a wide layer has few row tiles per
 * channel, hence little contention, and a consumer that derives the BatchNorm coefficients in its own prologue
 * (HP_CONV_IN_BN) reads R*2*C doubles per workgroup.  The whole statistics region is zeroed at the start of
 * every forward. */
#define HP_STAT_REPL_MAX 16
#if defined(__HIPCC__)
#define HP_HD __host__ __device__
#else
#define HP_HD
#endif
static inline HP_HD int hp_stat_repl(int C) {      /* largest power of two <= 1024 / C, clamped to [2, 16] */
  int r = 2;
  while (r < HP_STAT_REPL_MAX && r * 2 * C <= 1024) r *= 2;
  return r;
}
#define HP_OP_NI 40
#define HP_OP_NF 8
#define HP_OP_NB 26

/* Op flag: the record is a MEMBER of a following HP_OP_WGRAD_GROUP or HP_OP_PAIR op, or of a small-leaf group (below): the
 * program executor skips it (the group launch does its work); hp_run_op and the reference interpreter execute it like any op. */
#define HP_FLAG_MEMBER 0x200
/* Op flag: the op's ACTIVATION-typed buffers — the [rows][channels] tensors between the backbones' layers and their gradients — hold
 * bfloat16 (2 bytes per element, round-to-nearest-even on store, exact widening on load) instead of float32; arithmetic, statistics,
 * coefficients, weights, biases and every other buffer stay as documented.  Which buffers these are, per opcode:
 *   CONV_TAPS A, A2, OUT, RES, E_G2, E_ACT, E_RAW, E_RAW2 (only with HP_CONV_BF16) · WGRAD_TAPS DY, X (only with HP_CONV_BF16) ·
 *   BN_APPLY RAW, OUT, RES · BN_BWD_REDUCE G1, G2, ACT, GOUT, RAW, RAW2 · BN_BWD_APPLY G, RAW, DR · STEM_FWD OUT · STEM_WGRAD DR ·
 *   POOL_FWD IN · POOL_BWD G · REPEAT_FWD OUT · REPEAT_BWD G1, G2 · TAIL_FWD ACT · TAIL_BWD_X DACT · TAIL_BWD_W ACT.
 * The bf16-storage form of BASELINE configs[1] / [2] / [4]'s reduced-precision mode (TrainCfg.act_dtype = "bf16"): at batch >= 4096 every
 * layer below 512 channels is bound by activation traffic, not by the matrix cores.  Never part of the fp32 parity path. */
#define HP_FLAG_ACT_BF16 0x400
/* Small-leaf group: bits 16..23 of `flags` of a record = number n of IMMEDIATELY PRECEDING records (all flagged
 * HP_FLAG_MEMBER) that, together with this record, are INDEPENDENT of each other (no record reads what another writes) and
 * run side by side in ONE launch whose grid is the concatenation of the members' own grids.  Members are
 * HP_OP_LINEAR_BWD_W / HP_OP_EMB_BWD records: the planner defers the small weight-gradient reductions of a backward pass
 * (the heads' Linear dW/db, the embedding tables) into such a group — ten 3 us launches, each mostly launch floor, become
 * one.  hp_run_op and the reference interpreter ignore the field and execute each record on its own. */
#define HP_FLAG_GROUP_SHIFT 16
#define HP_FLAG_GROUP_MASK 0xFF
#define HP_GROUP_MAX 64

/* One op record (POD, 8-byte aligned; numpy dtype mirror in hippie_amd/program.py). */
typedef struct HpOp {
  int32_t op;               /* HP_OP_* */
  int32_t flags;            /* op-specific bit flags */
  int32_t i[HP_OP_NI];      /* integer parameters */
  float   f[HP_OP_NF];      /* float parameters */
  int64_t buf[HP_OP_NB];    /* buffer references */
} HpOp;

/* ---- opcodes ------------------------------------------------------------------
 * Row mapping shared by CONV_TAPS / WGRAD_TAPS (an implicit-GEMM view of every
 * Conv1d variant of the reference):  GEMM row m -> (b = m / Lout, l = m % Lout);
 * for tap j:  pos = a*l + tap_o[j];  the tap contributes iff 0 <= pos < P;
 * source row = b*Lin + (pos >> sh) of source tensor tap_src[j];  weight slab tap_w[j] of weight tensor tap_src[j].
 *   i[0]=M  i[1]=N  i[2]=K  i[3]=Lout  i[4]=Lin  i[5]=P  i[6]=a  i[7]=sh  i[8]=0 (reserved)
 *   i[9]=ntaps  i[10..15]=tap_o  i[16..21]=tap_w  i[22..27]=tap_src (CONV_TAPS only: 0 = (A, W), 1 = (A2, W2))
 * CONV_TAPS output rows:  i[28]=Lfull.  Lfull == 0: GEMM row m is output row m.  Lfull > 0: GEMM row m is output
 * row b*Lfull + i[29]*l + i[30] (a strided subset of a taller tensor: the even / odd output positions of a
 * stride-2 input-gradient are two ops writing interleaved rows of one tensor, each with only its own taps).
 */
/* CONV_TAPS flags */
#define HP_CONV_W_KN     1     /* weight slab is [K][N] (else [N][K]) */
#define HP_CONV_BIAS     2
#define HP_CONV_STATS    4     /* per-column sum / sum of squares of the output into STATS (for the following BatchNorm) */
#define HP_CONV_BN_EVAL  8     /* eval-mode BatchNorm folded into the epilogue */
#define HP_CONV_ACT      16    /* ... followed by leaky_relu */
#define HP_CONV_IN_BN    64    /* training-mode BatchNorm + leaky_relu of the INPUT applied in the operand loader */
#define HP_CONV_EPI_BNRED 128  /* BatchNorm-backward reduction fused into the epilogue (input-gradient convs) */
#define HP_CONV_BF16     256   /* CONV_TAPS / WGRAD_TAPS: both GEMM operands are rounded to bfloat16 (nearest even) when staged into
                                * LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; tensors in HBM, BatchNorm
                                * statistics, epilogues and the optimiser stay fp32.  The separately labelled reduced-precision mode
                                * of BASELINE config 2; never part of the fp32 parity path. */
#define HP_CONV_BF16X3   0x800 /* CONV_TAPS / WGRAD_TAPS (not with HP_CONV_BF16): fp32 ARITHMETIC ON THE bf16 MATRIX CORES.  Each fp32 operand value
                                * is split exactly into three bfloat16 terms (8 + 8 + 8 significand bits) when staged into LDS, and a product
                                * is the six terms of (ah+am+al)(bh+bm+bl) that lie above 2^-24 of it, on v_mfma_f32_32x32x16_bf16 with fp32
                                * accumulation: the error of a dot product is at the level of the fp32 matrix path's own rounding (measured
                                * against fp64: tests/test_gpu_split.py), at 16/6 of its instruction rate.  Same tensors, same epilogues,
                                * same tolerances as the fp32 path — this IS the fp32 parity path when TrainCfg.mfma_dtype = "bf16x3". */
#define HP_CONV_WFRAG    4096   /* CONV_TAPS with HP_CONV_BF16X3: the B operand's three-term fragments are read ready-made from the image an
                                * HP_OP_WFRAG record wrote earlier in the pass — buf[24] for the weights W (buf[1]), buf[25] for W2 (buf[11]) — instead of
                                * being loaded as fp32, split and staged through LDS by every tile: no weight tile in LDS at all.  The image must
                                * have been made for this op's orientation (HP_CONV_W_KN or not) of the same weights; W / W2 stay in the record
                                * (the 128-row bodies and the reference interpreter read them).  Bit-identical results. */
enum {
  /* out[m][n] = sum_taps sum_k A_src[row(m,tap)][k] * W_src[tap_w][..] (+bias[n]);  f32 MFMA.
   * Replaces nn.Conv1d forward (backbones.py:24,26,33,50,55), ResizeConv1d =
   * F.interpolate(nearest)+Conv1d (backbones.py:13-16) with the upsample folded into
   * the row mapping, and their input-gradients (ATen convolution_backward) with
   * transposed weights; a stride-2 block's conv1 and 1x1-shortcut input-gradients (both with respect to the same
   * tensor) are ONE op pair over two sources.  Requires K % 32 == 0 and N % 4 == 0 (every conv of the backbones).
   *
   * HP_CONV_BN_EVAL (forward-only path: no batch-wide dependency, so conv + BN (+ residual) (+ leaky_relu) is one
   * launch):  out = act( (acc + bias) * gamma/sqrt(rvar + f[0]) + (beta - rmean * gamma/sqrt(rvar + f[0])) + RES ),
   * the same arithmetic, in the same order, as HP_OP_BN_APPLY in eval mode on this conv's output.  f[0]=eps f[1]=slope.
   *
   * HP_CONV_IN_BN (training): the A operand is  a = leaky_relu(fma(x, scale[k], shift[k]), f[2])  of the stored
   * tensor x (zero for padded rows), i.e. `F.leaky_relu(bn(conv(...)))` of backbones.py:37,66 evaluated in the
   * consumer: the activation tensor is never written.  (scale, shift) are derived — identically in every workgroup,
   * exactly as HP_OP_BN_APPLY derives them — from IN_STATS (sums over i[31] rows; over i[31]*i[32] rows when
   * i[32] > 1), GAMMA, BETA, f[3]=eps.  Workgroup 0 also performs that BatchNorm's side effects: IN_SAVE :=
   * (mean, invstd), IN_COEF := (scale, shift) for the backward pass, running statistics update with f[4]=momentum.
   * The slope f[2] must lie in [0, 1] (hp_program_validate): the loader forms leaky_relu as max(v, v*slope), which is
   * then the same value as `v > 0 ? v : v*slope`.
   *
   * HP_CONV_EPI_BNRED (input-gradient convs): instead of storing acc,
   *   g = (acc [+ E_G2]) * leaky_relu'(pre),  pre = E_ACT (the activation tensor) or, when E_ACT is NULL,
   *   fma(E_RAW, scale, shift) from E_COEF;  OUT := g;  E_BS[0][n] += sum g;  E_BS[1][n] += sum g*xhat(E_RAW, E_SAVE)
   *   (and the same sums for a second BatchNorm fed by the same g: E_RAW2, E_SAVE2, E_BS2)  —  HP_OP_BN_BWD_REDUCE
   *   on this conv's output, fused.  f[5]=slope of that leaky_relu.
   *
   * buf: 0 A, 1 W, 2 OUT, 3 BIAS, 4 STATS(double[R][2][N]), 5 GAMMA 6 BETA 7 RMEAN 8 RVAR (of the epilogue BN with
   *      BN_EVAL, of the input BN with IN_BN), 9 RES(or NULL), 10 A2, 11 W2,
   *      12 IN_STATS(double[R][2][K]) 13 IN_SAVE(float[2][K]) 14 IN_COEF(float[2][K]),
   *      15 E_G2 16 E_ACT 17 E_RAW 18 E_SAVE 19 E_COEF 20 E_BS 21 E_RAW2 22 E_SAVE2 23 E_BS2 */
  HP_OP_CONV_TAPS = 1,
  /* slab[split][tap_w][n][k] = sum_{m in split} DY[m][n] * X[src(m,tap)][k]; f32 MFMA.
   * Replaces the weight-gradient half of ATen convolution_backward.
   * i[22]=nsplit i[23]=rows_per_split (multiple of 32) i[24]=slab stride per split (floats).
   * flags: 1 = no slabs: every split adds its tile into buf[2] (the zeroed gradient tensor
   * [tap_w][N][K]) with fp32 atomics (faster, summation order not reproducible bit for bit).
   * flags: HP_CONV_IN_BN = X is the raw BatchNorm input of the forward conv: the operand is
   * leaky_relu(fma(x, scale[k], shift[k]), f[0]) with (scale, shift) = COEF (written by the forward conv); f[0] in [0, 1].
   * buf: 0 DY, 1 X, 2 SLAB (or gradient), 3 COEF(float[2][K]) */
  HP_OP_WGRAD_TAPS = 2,
  /* out[j] = sum_s slab[s*stride + j], j < n.  i[0]=n i[1]=nsplit i[2]=stride. buf: 0 SLAB 1 OUT */
  HP_OP_SLAB_REDUCE = 3,
  /* nn.BatchNorm1d forward (+ residual + leaky_relu): backbones.py:36-41,65-70,95;
   * model.py:23-27,39-40.   out = act(scale*raw + shift + res)
   * training: mean/var from STATS (sum, sumsq) over M rows; saves (mean, invstd);
   * updates running stats (momentum, unbiased var).  eval: running stats.
   * i[0]=M i[1]=C i[2]=res_mode(0 none,1 tensor,2 second BN) i[3]=training i[4]=act
   * i[5]=world (sync-BatchNorm: STATS hold the sums over `world` ranks, i.e. over M*world rows; 0/1 = local)
   * f[0]=slope f[1]=eps f[2]=momentum
   * buf: 0 RAW 1 OUT 2 STATS 3 GAMMA 4 BETA 5 RMEAN 6 RVAR 7 SAVE(float[2][C])
   *      8 RES(tensor or second raw) 9 STATS2 10 GAMMA2 11 BETA2 12 RMEAN2 13 RVAR2 14 SAVE2 */
  HP_OP_BN_APPLY = 4,
  /* g = (G1 [+ G2]) * leaky_relu'(ACT);  BS[0][c] += sum g;  BS[1][c] += sum g*xhat
   * (xhat = (raw-mean)*invstd), optionally for a second BN fed by the same g.
   * First half of ATen native_batch_norm_backward + leaky_relu_backward + residual fan-out.
   * ACT NULL: the activation was never stored (HP_CONV_IN_BN consumer); its sign is that of fma(RAW, scale, shift)
   * with (scale, shift) = COEF, bit for bit what the forward consumer evaluated.
   * i[0]=M i[1]=C i[2]=has_g2 i[3]=has_second  f[0]=slope
   * buf: 0 G1 1 G2 2 ACT 3 GOUT 4 RAW 5 SAVE 6 BS(double[R][2][C]) 7 RAW2 8 SAVE2 9 BS2 10 COEF(float[2][C]) */
  HP_OP_BN_BWD_REDUCE = 5,
  /* dr = gamma*invstd*(g - BS0/M - xhat*BS1/M);  dgamma = BS1;  dbeta = BS0.
   * i[0]=M i[1]=C i[2]=world (sync-BatchNorm: BS summed over ranks, divisor M*world, dgamma/dbeta scaled by
   * 1/world so that the data-parallel gradient MEAN restores the sum).
   * buf: 0 G 1 RAW 2 SAVE 3 BS 4 GAMMA 5 DR 6 DGAMMA 7 DBETA */
  HP_OP_BN_BWD_APPLY = 6,
  /* encoder stem Conv1d(1,64,k3,s2,p1) (backbones.py:78,95): raw[b*Lout+l][n] =
   * sum_t x[b][2l+t-1]*W[n][t] (+stats).  i[0]=B i[1]=Lin i[2]=Lout i[3]=C
   * buf: 0 X 1 W 2 OUT 3 STATS */
  HP_OP_STEM_FWD = 7,
  /* dW[n][t] = sum_{b,l} DR[b*Lout+l][n]*x[b][2l+t-1].  buf: 0 DR 1 X 2 DW
   * flags & 1 (also HP_OP_TAIL_BWD_W, HP_OP_LINEAR_BWD_W): one workgroup per output group walks all rows — no
   * cross-workgroup atomics, bit-reproducible sums (TrainCfg.deterministic_wgrad), slower. */
  HP_OP_STEM_WGRAD = 8,
  /* adaptive_avg_pool1d(x,1) (backbones.py:100): out[b][c] = mean_l in[b*L+l][c]
   * i[0]=B i[1]=L i[2]=C.  buf: 0 IN 1 OUT */
  HP_OP_POOL_FWD = 9,
  /* G[b*L+l][c] = d[b][c]/L.  buf: 0 D 1 G */
  HP_OP_POOL_BWD = 10,
  /* F.interpolate(x.unsqueeze(-1), scale_factor=4) (backbones.py:130-131):
   * out[b*R+l][c] = in[b][c].  i[0]=B i[1]=R i[2]=C.  buf: 0 IN 1 OUT */
  HP_OP_REPEAT_FWD = 11,
  /* d[b][c] = sum_l (G1 [+G2])[b*R+l][c].  i[3]=has_g2.  buf: 0 G1 1 G2 2 D */
  HP_OP_REPEAT_BWD = 12,
  /* torch.cat([...], dim=1) with nn.Embedding gathers (model.py:53,60,65-66).
   * up to 4 segments j: kind i[4+3j] (0 dense, 1 embedding rows, 2 zeros), width
   * i[5+3j], ld i[6+3j].  i[0]=B i[1]=nseg i[2]=ldo.  i[16+j] = number of rows of segment j's embedding table:
   * an index outside [0, rows) yields a zero row and never touches memory (the host raises IndexError first, as
   * nn.Embedding does; the device-side guard only makes a bad label harmless).
   * buf: 0 OUT; 1+2j SRC/TABLE; 2+2j IDX(int64) */
  HP_OP_CONCAT = 13,
  /* embedding gradient: DT[idx[b]][k] += D[b*ld + col0 + k] (tables of <= 1024 floats: a wave per table element,
   * fixed-order sums, no atomics; larger tables: fp32 atomics unless flags & 1, which selects the ordered form at any size);
   * rows with idx[b] outside [0, rows) are skipped.  i[0]=B i[1]=w i[2]=ld i[3]=col0 i[4]=rows.  buf: 0 D 1 IDX 2 DT */
  HP_OP_EMB_BWD = 14,
  /* nn.Linear (model.py:21-41, backbones.py:84,102,111,118,129,138):
   * Y[m*ldy+n] = act(sum_k X[m*ldx+k]*W[n*K+k] + b[n]) (+stats on the pre-activation).
   * i[0]=M i[1]=N i[2]=K i[3]=ldx i[4]=ldy i[5]=act i[6]=stats  f[0]=slope
   * buf: 0 X 1 W 2 B 3 Y 4 STATS */
  HP_OP_LINEAR_FWD = 15,
  /* DX[m*ldx+k] (+)= (sum_n DY[m*ldy+n]*W[n*K+k]) * [leaky_relu'(ACT[m*lda+k])]
   * i[0]=M i[1]=N i[2]=K i[3]=ldy i[4]=ldx i[5]=has_mask i[6]=lda i[7]=accumulate
   * f[0]=slope.  buf: 0 DY 1 W 2 DX 3 ACT */
  HP_OP_LINEAR_BWD_X = 16,
  /* DW[n*K+k] = sum_m DY[m*ldy+n]*X[m*ldx+k];  DB[n] = sum_m DY[m*ldy+n].
   * i[0]=M i[1]=N i[2]=K i[3]=ldy i[4]=ldx.  buf: 0 DY 1 X 2 DW 3 DB */
  HP_OP_LINEAR_BWD_W = 17,
  /* reparameterize + KL (model.py:46-49,104): z = mu + eps*exp(0.5*lv);
   * LOSS[0] += sum_i -0.5*sum_j(1+lv-mu^2-exp(lv)).  MULV is [B][2z] = (mu | lv).
   * i[0]=B i[1]=z.  buf: 0 MULV 1 EPS 2 Z 3 LOSS(double[4]) */
  HP_OP_REPARAM_KL_FWD = 18,
  /* dmu = dz + beta*mu/B;  dlv = dz*eps*0.5*exp(0.5 lv) + beta*0.5*(exp(lv)-1)/B.
   * i[0]=B i[1]=z i[2]=ld of DZ.  f[0]=beta.  buf: 0 MULV 1 EPS 2 DZ 3 DMULV */
  HP_OP_REPARAM_KL_BWD = 19,
  /* F.mse_loss(data, dec) (model.py:103) + its gradient:
   * LOSS[slot] += sum (x-rec)^2;  DREC = w*2*(rec-x)/n.  i[0]=n i[1]=slot f[0]=w
   * buf: 0 X 1 REC 2 DREC 3 LOSS */
  HP_OP_MSE_FWD_BWD = 20,
  /* decoder tail ResizeConv1d(64,1,k3,scale2) (backbones.py:117,136):
   * t[b][p] = bias + sum_t sum_c act[b*Lh + ((p+t-1)>>1)][c]*W[c*3+t], 0<=p+t-1<2Lh.
   * i[0]=B i[1]=Lh i[2]=C.  buf: 0 ACT 1 W 2 BIAS 3 OUT */
  HP_OP_TAIL_FWD = 21,
  /* buf: 0 DT 1 W 2 DACT */
  HP_OP_TAIL_BWD_X = 22,
  /* buf: 0 DT 1 ACT 2 DW 3 DB */
  HP_OP_TAIL_BWD_W = 23,
  /* training_step scalars (model.py:103-113, 465-479): OUT[0]=loss OUT[1]=mse1 OUT[2]=mse2
   * OUT[3]=kl_mean.  i[0]=B i[1]=n1 i[2]=n2 f[0]=beta f[1]=w1 f[2]=w2
   * buf: 0 LOSS(double[4]) 1 OUT(float[4]) */
  HP_OP_LOSS_FINALIZE = 24,
  /* NORM2[0] += sum g^2 (fp64).  i[0]=n.  buf: 0 G 1 NORM2(double[1]) */
  HP_OP_GRADNORM = 25,
  /* torch.optim.AdamW step (model.py:93) with optional clip_grad_norm_ scale folded in.
   * step count t is read from STEP (int64, incremented by HP_OP_STEP_INC).
   * i[0]=n  f: 0 lr 1 beta1 2 beta2 3 eps 4 weight_decay 5 clip(0 = off) 6 (1-beta1) 7 (1-beta2)
   * (the complements are formed in fp64 on the host, as torch does, so 1-beta^t stays accurate in fp32 transport)
   * buf: 0 P 1 G 2 M 3 V 4 STEP(int64[1]) 5 NORM2(double[1]) */
  HP_OP_ADAMW = 26,
  /* STEP[0] += 1.  buf: 0 STEP */
  HP_OP_STEP_INC = 27,
  /* memset(dst, 0, i[0] bytes).  buf: 0 DST */
  HP_OP_ZERO = 28,
  /* One launch for i[1] HP_OP_WGRAD_TAPS member records (all with i[2] taps, atomic accumulation) that
   * sit at program indices i[0] .. i[0]+i[1]-1, flagged HP_FLAG_MEMBER.  The library builds the device-side
   * problem and block tables from the member records on first use.  i[0]=first i[1]=count i[2]=ntaps */
  HP_OP_WGRAD_GROUP = 29,
  /* One launch for two independent ops of the same opcode (CONV_TAPS with equal weight-layout flag, or the
   * BatchNorm family BN_APPLY / BN_BWD_REDUCE / BN_BWD_APPLY) at program indices i[0] and i[1], both flagged
   * HP_FLAG_MEMBER: twice the workgroups per launch.  Used to run the same layer of the wave and the time
   * model together. */
  HP_OP_PAIR = 30,
  /* EphysDatasetLabeled.__getitem__ preprocessing, batched (hippie/dataloading.py:74-96): optional
   * log(x + 1), then F.interpolate(size=L, mode="linear", align_corners=False) of each row.
   * out[n][i] = lerp(in[n][x0], in[n][x1], w) with src = max((i + 0.5) * W / L - 0.5, 0), x0 = floor(src),
   * x1 = min(x0 + 1, W - 1), w = src - x0.  i[0]=N i[1]=W i[2]=L  flags: 1 = log(x+1) first
   * buf: 0 IN[N][W] 1 OUT[N][L] */
  HP_OP_RESAMPLE_LINEAR = 31,
  /* Schedule-Free AdamW, per-step scalars (hippie/optimizers.py:118-138), one thread, all in fp64 as the
   * reference's Python floats:  k = STEP[0];  sched = k < warmup ? (k+1)/warmup : 1;
   * lr_t = lr * sched * sqrt(1 - beta2^(k+1));  lr_max = max(lr_t, lr_max);
   * weight = (k+1)^r * lr_max^power;  weight_sum += weight;  ckp1 = weight_sum != 0 ? weight/weight_sum : 0.
   * STATE = {lr_max, weight_sum, lr_t, ckp1} (lr_max, weight_sum persist; a zero-initialised lr_max is
   * equivalent to the reference's -1 because lr_t >= 0).
   * i[0]=warmup_steps  f: 0 lr 1 (1-beta2) 2 r 3 weight_lr_power.  buf: 0 STEP(int64[1]) 1 STATE(double[4]) */
  HP_OP_SF_SCHEDULE = 32,
  /* Schedule-Free AdamW, element update (hippie/optimizers.py:145-207), with the clip_grad_norm_ scale folded
   * in as for HP_OP_ADAMW.  On the first step (STEP[0] == 0) z := y first (:147).
   *   v = beta2*v + (1-beta2)*g^2;  gn = g / (sqrt(v) + eps);  if wd: gn += wd*y;
   *   y = lerp(y, z, ckp1);  y += gn * lr_t*(beta1*(1-ckp1) - 1);  z -= lr_t*gn.
   * i[0]=n  f: 0 beta1 1 beta2 2 eps 3 weight_decay 4 clip(0 = off) 5 (1-beta2)
   * buf: 0 Y(=P) 1 G 2 Z 3 V 4 STEP(int64[1]) 5 STATE(double[4]) 6 NORM2(double[1]) */
  HP_OP_ADAMW_SF = 33,
  /* y = torch.lerp(y, z, w): AdamWScheduleFree.eval() (w = 1 - 1/beta1) and .train() (w = 1 - beta1)
   * (hippie/optimizers.py:82-103).  i[0]=n f[0]=w.  buf: 0 Y 1 Z */
  HP_OP_LERP = 34,
  /* Marker for sync-BatchNorm under data parallelism (torch.nn.SyncBatchNorm semantics; Lightning's
   * sync_batchnorm=True): the i[0] doubles at buf[0] (a replicated statistics slot) must be summed over all
   * ranks before the next op runs.  The library executes it as a no-op; the host splits the range here and
   * issues the collective (hippie_amd/engine.py).  i[0]=count.  buf: 0 SLOT */
  HP_OP_STATS_SYNC = 35,
  /* The loader of the hot path on HBM-resident tables + the reparameterisation noise, as one launch, so that a whole
   * optimisation step is ONE captured graph with nothing of the host in it.  Replaces, for preprocessed tables already in
   * HBM: DataLoader(Subset(ConcatDataset), batch_size, shuffle) index gather + default collate
   * (scripts/train_model_with_multimodal.py:155-166) and `eps = torch.randn_like(std)` (hippie/model.py:48).
   *   j = (CURSOR[0] mod i[4]) * i[5] + i[6]          batch number inside the epoch's permutation (i[5] = world, i[6] = rank:
   *                                                   the DistributedSampler-style interleave of data-parallel ranks)
   *   r = PERM[j*B + b];   X[b][:] = TABLE[r][:]  (and X2[b][:] = TABLE2[r][:]);   SRC[b] = LABELS[r]
   *   EPS[b*z + k] = standard normal: Philox4x32-10(counter = (CURSOR lo, CURSOR hi, (b*z + k) / 4, rank i[6]), key = SEED lo, hi)
   *                  (the rank in the counter: data-parallel replicas loaded with one seed still draw independent noise, as the ranks'
   *                  generators of torch DDP do)
   *                  -> 4 x u32 -> u = ((v >> 8) + 0.5) * 2^-24 -> Box-Muller pairs (sqrt(-2 ln u0) * cos / sin(2 pi u1))
   * The op does NOT advance CURSOR: an HP_OP_STEP_INC on it follows in the program.  A PERM entry outside [0, N) reads row 0.
   * i[0]=B i[1]=L i[2]=L2 (0 = no second table) i[3]=z i[4]=batches per epoch i[5]=world i[6]=rank i[7]=N (table rows)
   * buf: 0 TABLE[N][L] 1 TABLE2[N][L2] 2 LABELS(int64[N]) 3 PERM(int64[>= i[4]*i[5]*B]) 4 CURSOR(int64[1])
   *      5 X 6 X2 7 SRC(int64[B]) 8 EPS 9 SEED(int64[1]) */
  HP_OP_STAGE_BATCH = 36,
  /* One launch of ONE workgroup for the heads between the two backbones in the training forward (hippie/model.py:51-72): the i[1]
   * member records at program indices i[0] .. i[0]+i[1]-1 (all flagged HP_FLAG_MEMBER; hp_run_op and the reference interpreter execute
   * them one by one) are  CONCAT, LINEAR_FWD+stats, BN_APPLY, LINEAR_FWD+stats, BN_APPLY, LINEAR_FWD (z_mean | z_log_var),
   * REPARAM_KL_FWD, CONCAT, LINEAR_FWD+act, LINEAR_FWD+stats, BN_APPLY — each consuming its predecessor's output.  Unimodal model, z_dim 5
   * or 10, class_hidden_dim 5, at most 512 rows; any other chain is refused at program creation.  A thread owns a row and keeps it in
   * registers from layer to layer; BatchNorm statistics are fp64 column sums in a fixed order (csrc/heads_fused.h).  Every tensor the
   * members write is written.  i[0]=first i[1]=count i[2]=0 */
  HP_OP_HEADS = 37,
  /* Three-term MFMA fragments of a conv weight tensor W[T][N][K] (fp32, PARAM): every value split exactly into three bfloat16 terms
   * (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)) and laid out in the order v_mfma_f32_32x32x16_bf16 consumes a B operand, once per
   * pass instead of once per tile.  Image layout (bf16): [T][KK/16][ceil(NN/32)][3 terms][64 lanes][8]; lane (j = lane & 31, h = lane >> 5)
   * of chunk (t, ks, jn) holds B(k = 16 ks + 8 h + e, n = 32 jn + j), e = 0..7 (zero beyond NN).
   *   F (i[3] & 1, buf[1]): the forward orientation, B(k, n) = W[t][n][k]   (NN = N, KK = K; K % 32 == 0);
   *   G (i[3] & 2, buf[2]): the HP_CONV_W_KN orientation, B(k, n) = W[t][k][n] (NN = K, KK = N; N % 32 == 0).
   * One workgroup reads a 32 x 32 block of a slab once and writes its two F and two G chunks.  Groupable (HP_FLAG_GROUP_SHIFT; a
   * group holds only WFRAG records): all conv weights of a model in one launch at the head of a forward pass.
   * i[0]=T i[1]=N i[2]=K i[3]=which   buf[0]=W buf[1]=F buf[2]=G */
  HP_OP_WFRAG = 38,
  HP_OP__COUNT
};

typedef struct HpProgram HpProgram;

/* Library / device info. */
int hp_abi_version(void);
const char* hp_last_error(void);
/* Writes CU count, XCD count (8), wave size and the gfx arch name of the current device. */
int hp_device_info(int* n_cu, int* wave_size, char* arch, int arch_len);

/* Program lifecycle.  ops are copied; bases are the six arena base pointers. */
int hp_program_create(const HpOp* ops, int n_ops, void* const bases[HP_NUM_SPACES],
                      const int64_t sizes[HP_NUM_SPACES], HpProgram** out);
int hp_program_destroy(HpProgram* p);
/* Validate every op (opcode, sizes, buffer ranges against arena sizes).  No GPU needed. */
int hp_program_validate(const HpProgram* p);
/* Enqueue ops [first, first+count) on `stream` (a hipStream_t; NULL = default stream). */
int hp_program_run(HpProgram* p, int first, int count, void* stream);
/* Capture ops [first, first+count) into a hipGraph (segment id returned in *seg);
 * hp_program_replay launches the captured graph on `stream`. */
int hp_program_capture(HpProgram* p, int first, int count, int* seg);
int hp_program_replay(HpProgram* p, int seg, void* stream);
/* Time ops one by one with HIP events on `stream` (ms per op into out_ms[n]); synchronises. */
int hp_program_profile(HpProgram* p, int first, int count, void* stream, float* out_ms);

/* Single-op launch without a program (unit tests, benchmarks). */
int hp_run_op(const HpOp* op, void* const bases[HP_NUM_SPACES], void* stream);

/* ---- serialised models: the level a host WITHOUT Python binds ----------------------------------------------------
 * hp_program_* above executes op records somebody lowered.  The lowering itself (hippie_amd/planner.py) stands for
 * constructing the reference's nn.Module graph — hippieUnimodalCVAE.__init__ / MultiModalCVAE.__init__
 * (hippie/model.py:13-44,352-395) over ResNet18Enc / ResNet18Dec (hippie/backbones.py:74-126) — and is done ONCE per
 * (model configuration, batch size):  `python -m hippie_amd.export ... -o model.hpm`  writes the records, the named
 * segments, the six arena sizes, a table of every parameter / BatchNorm buffer under its reference state_dict key
 * (shape, arena offset, layout) and of every named I/O slot, and optionally initial values.  hp_model_load reads such a
 * file, allocates the arenas ITSELF (hipMalloc, freed by hp_model_destroy: nothing is handed to the caller to free),
 * uploads the initial values and validates the program; the verbs below are what Lightning's automatic optimisation
 * calls around training_step (hippie/model.py:95-116): forward, backward, optimizer.step.
 *
 * I/O slots (hp_model_write / hp_model_read by name):  "x" float[B][1][L] (multimodal: + "x2"), "src" / "cls" int64[B]
 * (source / class labels), "eps" float[B][z] (reparameterisation noise, model.py:46-49), "scalars" float[4] =
 * (loss, mse1, mse2, kl_mean) of the last forward, "enc_train" / "enc_eval" float[B][z] (the embedding), "mulv_*"
 * float[B][2z] = (mu | logvar), "rec_*" / "rec2_*" reconstructions, "adam_step" int64[1].  Parameters are addressed by
 * their reference keys ("encoder.conv1.weight", ...); layout 1 = conv weight stored tap-major [k][Cout][Cin].
 * Segments: "fwd_train", "bwd", "opt", "step" (= the three, one graph), "fwd_eval", "enc_eval" (encoder half only); with resident
 * tables also "stage" (HP_OP_STAGE_BATCH + cursor increment), "step_staged" (= stage + step), "fwd_train_staged". */
typedef struct HpModel HpModel;
typedef struct HpTensorInfo {
  char name[112];           /* reference state_dict key, or I/O slot name */
  int32_t space;            /* HP_SPACE_* */
  int32_t layout;           /* 0 = as `shape` says; 1 = conv weight [Cout][Cin][k] stored as [k][Cout][Cin] */
  int64_t offset_bytes;     /* into the arena */
  int64_t numel;
  int32_t ndim;
  int32_t shape[4];
  int32_t dtype;            /* 0 float32, 1 int64, 2 float64 */
} HpTensorInfo;
#define HP_MODEL_NO_DEVICE 1   /* hp_model_load flag: parse + validate only (no GPU, no allocation) */

int hp_model_load(const char* path, int flags, HpModel** out);
int hp_model_destroy(HpModel* m);
/* Checkpoint from a host without Python: writes the model as an .hpm file again, with the CURRENT parameters, BatchNorm buffers (incl.
 * the AdamW step counter and, for staged models, the batch cursor, which live in the buffer arena), num_batches_tracked (config word 13)
 * and — with_optimizer != 0 — both AdamW moment arenas, so that hp_model_load resumes exactly where training stopped (the reference's
 * .ckpt = {"state_dict", "optimizer_states"}, written by pl.ModelCheckpoint).  `python -m hippie_amd.export --to-ckpt file.hpm out.ckpt`
 * converts such a file into a reference-format .ckpt (reference state_dict keys and layouts; torch.optim.AdamW state). */
int hp_model_save(HpModel* m, const char* path, int with_optimizer);
/* out[0..9] = kind (0 unimodal, 1 multimodal), z_dim, output_size, output_size2, class_hidden_dim, num_sources, num_classes,
 * batch, with_class, number of floats AdamW updates. */
int hp_model_config(const HpModel* m, int32_t out[16]);
/* which: 0 parameters, 1 BatchNorm buffers, 2 I/O slots.  hp_model_find searches slots, then buffers, then parameters. */
int hp_model_tensor_count(const HpModel* m, int which);
int hp_model_tensor_info(const HpModel* m, int which, int index, HpTensorInfo* out);
int hp_model_find(const HpModel* m, const char* name, HpTensorInfo* out);
/* Device base pointer and size of one arena (NULL under HP_MODEL_NO_DEVICE); owned by the model. */
void* hp_model_arena(const HpModel* m, int space, int64_t* nbytes);
HpProgram* hp_model_program(HpModel* m);
int hp_model_segment(const HpModel* m, const char* name, int* first, int* count);
/* Run a named segment on `stream`; use_graph != 0: captured into a hipGraph on first use, replayed afterwards. */
int hp_model_run(HpModel* m, const char* segment, int use_graph, void* stream);
/* The reference's verbs: model(batch) in train / eval mode; loss.backward(); optimizer.step(); all three. */
int hp_model_forward(HpModel* m, int training, int use_graph, void* stream);
int hp_model_backward(HpModel* m, int use_graph, void* stream);
int hp_model_optimizer_step(HpModel* m, int use_graph, void* stream);
int hp_model_train_step(HpModel* m, int use_graph, void* stream);
/* The same step INCLUDING the loader (HP_OP_STAGE_BATCH: batch gather by index from HBM-resident tables + Philox noise), for a model
 * exported with `--resident-units N`: one graph replay per optimisation step, nothing of the host in it.  Before the first step write
 * the slots "data_x" float[N][L] (multimodal: + "data_x2" float[N][L2]), "data_labels" int64[N] (source ids), "perm" int64[N] (the
 * epoch's shuffle: batch j is perm[j*B .. (j+1)*B)), "seed" int64[1]; "cursor" int64[1] counts the steps taken (write 0 to restart;
 * it wraps over N / (B * world) batches).  config[10..12] of hp_model_config = N, data-parallel world, rank.  Replaces
 * DataLoader(..., batch_size, shuffle=True) + training_step + backward + optimizer.step (scripts/train_model_with_multimodal.py:155-166,
 * 200-224).  A model exported for more than one data-parallel rank is REFUSED here (a replica that skipped the gradient all-reduce would
 * silently diverge): use hp_model_train_step_dp, or hp_model_run("stage" | "fwd_train" | "bwd"), the host's all-reduce, then "opt". */
int hp_model_train_step_staged(HpModel* m, int use_graph, void* stream);
/* ---- data parallelism for a host without Python (the "allreduce_init(ncclUniqueId, rank, world) / bucket hooks" entry) ----
 * One process per GPU, replicated parameters, per-rank batches, gradients mean-all-reduced between backward and optimizer.step: what
 * Lightning's default DDP strategy does to the reference's scripts on a multi-GPU host (scripts/train_model_with_multimodal.py:200-207,
 * 694-703).  RCCL is bound at run time (dlopen of librccl.so; HIPPIE_RCCL_LIB overrides the name): nothing to link, nothing loaded
 * on a single-GPU host.
 *   hp_dp_unique_id            rank 0 draws the 128-byte ncclUniqueId and hands it to the other ranks by whatever channel the host has
 *                              (file, socket, MPI); one id PER MODEL — a model owns its communicator and the stream its collectives
 *                              run on, so two models of one process (waveform + timing) never serialise behind each other's all-reduce
 *   hp_model_allreduce_init    collective over all ranks: ncclCommInitRank.  A model with resident tables must have been exported for
 *                              this (world, rank): `python -m hippie_amd.export ... --resident-units N --dp-world W --dp-rank R`
 *   hp_model_train_step_dp     one optimisation step of this replica: [stage +] forward, backward, gradient MEAN over the ranks, AdamW.
 *                              Models exported with `--bucketed-bwd` hold the backward pass as two segments, "bwd_dec" (decoders +
 *                              decoder-side heads and THEIR weight gradients) and "bwd_enc": the all-reduce of the decoder-side gradient
 *                              range [first decoder* parameter, class_embedding) is issued on the communicator's stream as soon as
 *                              "bwd_dec" is queued and runs under "bwd_enc", which stays on the caller's stream; the second bucket
 *                              follows; the caller's stream waits for both (events, no host sync) before "opt".  The communicator's
 *                              stream is picked at the first call (hp_pick_side_stream against `stream`): step a model on ONE stream.  Other models: one
 *                              all-reduce of the whole active arena after "bwd".  Asynchronous like every other verb.
 * At world == 1 everything runs (RCCL copies the bucket onto itself): the structure can be timed and tested on one GPU. */
#define HP_DP_UNIQUE_ID_BYTES 128
int hp_dp_unique_id(void* out128);
int hp_model_allreduce_init(HpModel* m, const void* unique_id, int rank, int world);
int hp_model_allreduce_destroy(HpModel* m);      /* also done by hp_model_destroy */
int hp_model_train_step_dp(HpModel* m, int use_graph, void* stream);

/* New optimiser constants for the next stage without re-exporting: every HP_OP_ADAMW record gets lr and weight_decay, the executor and
 * the captured graphs are rebuilt.  reset_state != 0 also zeroes both moment arenas and the step counter — what the reference does when it
 * wraps the pretrained network in a NEW train module for fine-tuning at a tenth of the learning rate
 * (scripts/train_model_with_multimodal.py:263-268: the module constructor builds a fresh optim.AdamW, hippie/model.py:93).  Gradient
 * clipping on / off changes the program's structure (HP_OP_GRADNORM) and needs a new export. */
int hp_model_set_optimizer(HpModel* m, float lr, float weight_decay, int reset_state);
/* BatchNorm num_batches_tracked (training forwards so far; the reference keeps it as an int64 buffer per layer). */
int64_t hp_model_batches_tracked(const HpModel* m);
/* Copy nbytes (must equal the tensor's size) into / out of a named tensor, stream-ordered; a host destination is
 * complete when hp_model_read returns, a host SOURCE must stay valid until the stream has passed the copy. */
int hp_model_write(HpModel* m, const char* name, const void* src, int64_t nbytes, int src_on_device, void* stream);
int hp_model_read(HpModel* m, const char* name, void* dst, int64_t nbytes, int dst_on_device, void* stream);
int hp_model_synchronize(HpModel* m, void* stream);

/* ---- streams for a host that holds two models -------------------------------------------------------------------------------------
 * The reference fits the waveform model and the timing model one after the other (scripts/train_model_with_multimodal.py:208,224); a
 * host of this library steps them side by side on two HIP streams (4.4 ms per pair of steps instead of 5.9 at batch 512).  `stream`
 * arguments everywhere in this header are hipStream_t values passed as void*; a host without the HIP headers makes them here.
 * Which PAIR of streams really overlaps is decided by how the runtime maps streams onto hardware queues — a pair is concurrent, shares one
 * queue (= back to back) or time-slices (slower than back to back), depending on what else in the process created streams before
 * (DESIGN.md section 5.3).  hp_pick_concurrent_streams creates `candidates` (2..16) streams, times both models' "fwd_eval" graphs (which
 * change no parameter, statistic or counter) each alone and then on candidate pairs — the model whose graph takes longer alone on a
 * HIGH-priority stream, so that the two free-running chains finish together — and returns the first pair at or
 * below accept (0 = 0.85) x the sum of the two alone — else the fastest pair seen; the other candidates are destroyed.  Only the eval-mode output
 * slots ("enc_eval", "rec_eval", ...) are overwritten.  The two streams belong to the caller (hp_stream_destroy).  report (may be NULL): pair us, serial us, pairs tried. */
int hp_stream_create(void** out);
int hp_stream_destroy(void* stream);
int hp_pick_concurrent_streams(HpModel* a, HpModel* b, int candidates, float accept, void** stream_a, void** stream_b, float report[3]);
/* A stream for work that must run BESIDE the `busy` streams (n_busy of them; the gradient all-reduce beside a model's backward pass): the
 * runtime multiplexes streams onto a few hardware queues, and a side stream that shares its queue with a model stream turns each of its
 * event waits into a barrier in front of the model's own kernels.  Creates up to `candidates` (1..32) streams and probes each against every
 * busy stream — a 2 ms single-wave spin kernel on the candidate, a timed empty kernel on the busy stream: on one queue the empty kernel
 * waits its turn — and returns the first that delays none of them (else the least bad); the others are destroyed.  The busy streams must be
 * idle.  The stream belongs to the caller (hp_stream_destroy).  report (may be NULL): worst delay of the chosen one (us), candidates tried. */
int hp_pick_side_stream(void* const* busy, int n_busy, int candidates, void** out, float report[2]);
/* Events (hipEvent_t as void*, timing disabled): record on a stream after step i, synchronize on the event of step i - 2 before queueing step
 * i — a host that stays two steps ahead of the GPU instead of queueing a whole epoch is ~1 % faster. */
int hp_event_create(void** out);
int hp_event_record(void* event, void* stream);
int hp_event_synchronize(void* event);
int hp_event_destroy(void* event);

#ifdef __cplusplus
}
#endif
#endif /* HIPPIE_HIP_H */
