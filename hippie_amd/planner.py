"""Lower one HIPPIE cVAE configuration to an op program for libhippie_hip.so.

What is lowered (citations relative to the reference root):
  * ResNet18Enc / BasicBlockEnc       hippie/backbones.py:19-41,73-103
  * ResNet18Dec / BasicBlockDec / ResizeConv1d   hippie/backbones.py:6-16,44-70,106-141
  * hippieUnimodalCVAE heads + reparameterisation   hippie/model.py:12-72
  * MultiModalCVAE                                   hippie/model.py:350-432
  * training_step loss, AdamW, gradient clipping     hippie/model.py:93-116,454-482
plus the backward pass the reference gets from autograd.

The planner is pure host logic (no GPU, no torch): it assigns every parameter,
BatchNorm buffer and workspace tensor an arena offset and emits HpOp records.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass

import os

import numpy as np

from . import program as P
from .program import Ref, TapMap, OpList

SLOPE_BACKBONE = 0.01
SLOPE_HEADS = 0.2
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
ALIGN = 256


@dataclass
class ModelCfg:
    kind: str = "unimodal"            # "unimodal" | "multimodal"
    z_dim: int = 10
    output_size: int = 50             # also the input length (F.mse_loss(data, dec) needs equal shapes)
    output_size2: int = 100           # multimodal: second modality
    class_hidden_dim: int = 5
    num_sources: int = 5
    num_classes: int = 5


@dataclass
class BackboneCfg:
    """A stand-alone module of hippie/backbones.py lowered on its own (forward only: training-mode BatchNorm with running-
    statistics side effects, and eval mode): "ResizeConv1d" (:6-16), "BasicBlockEnc" (:19-41), "BasicBlockDec" (:44-70),
    "ResNet18Enc" (:73-103), "ResNet18Dec" (:106-141).  Tensors cross the boundary channels-last ([B, L, C])."""
    kind: str = "ResNet18Enc"
    length: int = 50                  # input length L (ResNet18Dec: unused — its input is [B, 2z])
    z_dim: int = 10                   # ResNet18Enc / ResNet18Dec
    output_size: int = 64             # ResNet18Dec
    in_channels: int = 64             # blocks / ResizeConv1d: in_planes / in_channels
    out_channels: int = 64            # ResizeConv1d only
    stride: int = 1                   # blocks: stride; ResizeConv1d: scale_factor


@dataclass
class TrainCfg:
    lr: float = 0.01
    weight_decay: float = 0.01
    beta: float = 1.0
    clip: float = 0.0                 # 0 = no gradient clipping
    w1: float = 1.0
    w2: float = 1.0
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    deterministic_wgrad: bool = False   # True: per-split slabs + ordered reduce instead of fp32 atomics
    grouped_wgrad: bool = True          # one launch for all weight-gradient GEMMs of a backward pass
    intra_pair: bool = True             # HP_OP_PAIR for independent ops inside one model (conv1 + shortcut, ...)
    zip_towers: bool = True             # multimodal: the two towers' same-shaped launches as HP_OP_PAIR units (Builder.zip_towers)
    fold_eval_bn: bool = True           # eval forward: BatchNorm (+residual, +leaky_relu) folded into the producing conv's epilogue
    fuse_bn: bool = True                # training: a block's inner BatchNorm + leaky_relu is evaluated in its consumers' operand
                                        # loaders (HP_CONV_IN_BN: the activation tensor is never written) and the BatchNorm-backward
                                        # reduction runs in the epilogue of the input-gradient conv that produces its operand
                                        # (HP_CONV_EPI_BNRED).  False: one launch per BatchNorm pass (HP_OP_BN_APPLY / BN_BWD_REDUCE)
    sync_bn_world: int = 0              # > 1: sync-BatchNorm over that many data-parallel ranks (HP_OP_STATS_SYNC markers)
    mfma_dtype: str = P.debug_knob("HIPPIE_MFMA_DTYPE", "bf16x3")
                                        # How the conv / weight-gradient GEMMs reach the matrix cores.
                                        # "bf16x3" (default): the reference's fp32 arithmetic ON THE bf16 MATRIX CORES — every fp32 operand value is
                                        # split exactly into three bfloat16 terms in the operand loaders, a product is the six terms above 2^-24 of
                                        # it, accumulation is fp32 (HP_CONV_BF16X3).  Error against fp64 at or below the fp32 matrix path's
                                        # (tests/test_gpu_split.py), same tolerances everywhere, 16/6 of its instruction rate: the parity path.
                                        # "f32": the same arithmetic on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, 157 TFLOP/s): rounds 1-3's
                                        # path, kept selectable and tested.  "bf16": BASELINE config 2's
                                        # reduced-precision mode — conv / weight-gradient operands rounded to bfloat16 in the loaders
                                        # (v_mfma_f32_32x32x16_bf16, fp32 accumulation); tensors in HBM, BatchNorm statistics, master
                                        # weights and AdamW stay fp32.  Own tolerance (tests/test_gpu_bf16.py), own bench line
    act_dtype: str = "f32"              # "bf16" (needs mfma_dtype="bf16"): the backbones' activation tensors and their gradients — every
                                        # [rows][channels] tensor between stem and pool, decoder.linear and tail, of the TRAINING passes — are
                                        # STORED as bfloat16 (HP_FLAG_ACT_BF16 on the ops that touch them); statistics, coefficients, weights,
                                        # the heads and the evaluation forward stay fp32.  At batch >= 4096 every layer below 512 channels is
                                        # bound by activation traffic once the products run on the bf16 matrix cores
    group_small_wgrads: bool = True     # the small weight-gradient reductions of a backward pass (the heads' Linear dW/db, the embedding
                                        # tables) are leaves: deferred to the end of the pass and run side by side in ONE launch
                                        # (HP_FLAG_GROUP) instead of ten launches of ~3 us each
    reuse_workspace: bool = True        # liveness-based packing of the workspace arena (pack_workspace): tensors that only the backward
                                        # pass touches share memory once dead, and so do the eval forward's; training-forward tensors
                                        # (all needed by the backward pass) and named I/O slots keep their own memory
    resident_units: int = 0             # > 0: the training tables ([N][L] per modality + int64 labels) live in the workspace and a
                                        # "stage" segment (HP_OP_STAGE_BATCH + cursor increment) gathers each step's batch by index from a
                                        # resident permutation and draws eps with Philox — "step_staged" = stage + fwd_train + bwd + opt is
                                        # then ONE graph per optimisation step with no host work in it (Engine.load_dataset / train_step_staged)
    fuse_heads: bool = True             # unimodal, z_dim 5 / 10, batch <= 512, training forward: the 11 head ops between the backbones (concat
                                        # ... decoder_fc BatchNorm) run as ONE single-workgroup launch (HP_OP_HEADS, csrc/heads_fused.h: 30 us
                                        # against 39 us); the un-fused records stay in the program as its members.  (A backward twin was built,
                                        # tested and removed: 94 us against 42 us, heads_fused.h.)
    weight_fragments: bool = P.debug_knob("HIPPIE_NO_WFRAG") != "1"      # (the knob: A/B runs of unmodified callers)
                                        # mfma_dtype "bf16x3": every forward pass starts with ONE launch that splits all conv weights into their three bf16
                                        # terms and lays them out in MFMA fragment order (HP_OP_WFRAG, both orientations in the training passes); the
                                        # conv launches served by the 64x64 body then read their B fragments ready-made (HP_CONV_WFRAG) instead of
                                        # loading, splitting and staging the weight tile through LDS per tile and K step — the same products in the
                                        # same order, bit-identical results; half the LDS traffic and half the split arithmetic of a K step
    bucketed_bwd: bool = False          # data parallel: the backward segment in two halves, "bwd_dec" (decoders + decoder-side heads, incl.
                                        # THEIR weight-gradient group) and "bwd_enc" (the rest); Plan.grad_buckets lists, per half, the ranges of
                                        # the gradient arena that are complete when it ends, so that the all-reduce of the decoder-side bucket
                                        # runs on the communicator's stream under the encoder-side kernels — which stay on the model's ONE
                                        # stream (hippie_amd.parallel.backward_allreduce).  "bwd" still names the whole pass.
    dp_world: int = 1                   # data-parallel interleave of the staged batches: rank r of `dp_world` takes batch j*world + r
    dp_rank: int = 0
    optimizer: str = "adamw"            # "adamw" (model.py:93) | "schedulefree" (hippie/optimizers.py:18-209)
    warmup_steps: int = 0               # schedule-free only
    sf_r: float = 0.0
    sf_weight_lr_power: float = 2.0


@dataclass
class PInfo:
    key: str
    shape: tuple
    offset: int        # in floats, into PARAM / GRAD / M / V
    numel: int
    layout: str        # "plain" | "tnc" (conv weight stored [3][Cout][Cin])

    @property
    def ref(self):
        return Ref(P.PARAM, self.offset * 4)

    @property
    def gref(self):
        return Ref(P.GRAD, self.offset * 4)


@dataclass
class BInfo:
    key: str
    numel: int
    offset: int

    @property
    def ref(self):
        return Ref(P.BUF, self.offset * 4)


def _round_up(x, a):
    return (x + a - 1) // a * a


class Plan:
    """Result of lowering: op list + arena layouts + named I/O slots."""

    def __init__(self, cfg: ModelCfg, batch: int, train: TrainCfg, with_class: bool):
        self.cfg, self.B, self.train, self.with_class = cfg, batch, train, with_class
        self.ops = OpList()
        self.params: "OrderedDict[str, PInfo]" = OrderedDict()
        self.bufs: "OrderedDict[str, BInfo]" = OrderedDict()
        self.bn_keys = []                    # BatchNorm prefixes (for num_batches_tracked)
        self.io = {}                         # name -> (Ref, shape, dtype str)
        self._poff = 0
        self._boff = 0
        self._ws = 0
        self.allocs = []                     # workspace allocations [offset, nbytes, pinned] in allocation order (pack_workspace)
        self.ws_unpacked = None              # workspace size before pack_workspace (None = not packed)
        self.stats_bytes = 0
        self.act_sites = []                  # leaky-ReLU sites of the training forward (tests read the branches taken back)
        self.slab_need = 0                   # floats
        self.flops_fwd = 0                   # 2*MAC of conv + linear, forward
        self.grad_buckets = None             # TrainCfg.bucketed_bwd: [[(lo, hi) floats, ...] of "bwd_dec", [...] of "bwd_enc"]

    # ---- arenas -------------------------------------------------------------
    def param(self, key, shape, layout="plain", align=4):
        numel = int(np.prod(shape))
        self._poff = _round_up(self._poff, align)
        info = PInfo(key, tuple(shape), self._poff, numel, layout)
        self.params[key] = info
        self._poff += numel
        return info

    def bn_params(self, prefix, c):
        w = self.param(prefix + ".weight", (c,))
        b = self.param(prefix + ".bias", (c,))
        rm = BInfo(prefix + ".running_mean", c, self._boff); self._boff += _round_up(c, 4)
        rv = BInfo(prefix + ".running_var", c, self._boff); self._boff += _round_up(c, 4)
        self.bufs[rm.key] = rm
        self.bufs[rv.key] = rv
        self.bn_keys.append(prefix)
        return dict(prefix=prefix, C=c, gamma=w, beta=b, rmean=rm, rvar=rv)

    def ws(self, nbytes, name=None, shape=None, dtype="f4"):
        self._ws = _round_up(self._ws, ALIGN)
        r = Ref(P.WS, self._ws)
        self._ws += int(nbytes)
        if nbytes > 0:
            self.allocs.append([r.offset, int(nbytes), name is not None])
        if name is not None:
            self.io[name] = (r, tuple(shape), dtype)
        return r

    def f32(self, n, name=None, shape=None):
        return self.ws(4 * n, name, shape if shape is not None else (n,), "f4")

    def stat(self, ndoubles, replicated=True):
        """fp64 accumulator slots inside the zeroed-every-step statistics region; per-channel slots (ndoubles = 2*C)
        are replicated hp_stat_repl(C) times (double[R][2][C])."""
        if replicated:
            ndoubles *= P.stat_repl(ndoubles // 2)
        off = _round_up(self.stats_bytes, 64)
        self.stats_bytes = off + 8 * ndoubles
        assert self.stats_bytes <= self.stats_cap, "statistics region overflow"
        return Ref(P.WS, self.stats_base + off)

    @property
    def n_param_floats(self):
        return _round_up(self._poff, 4)

    @property
    def n_buf_floats(self):
        return max(4, _round_up(self._boff, 4))

    @property
    def ws_bytes(self):
        return _round_up(self._ws, ALIGN)


# ======================================================================================
class Lowering:
    def __init__(self, cfg: ModelCfg, batch: int, train: TrainCfg = None, with_class=False, wgrad_target_blocks=256, wfrag_needed=None):
        self.cfg, self.B = cfg, batch
        self.train = train or TrainCfg()
        # weight-fragment images (TrainCfg.weight_fragments): which (weight key, HP_CONV_W_KN) pairs the convs of a mode use is only known
        # once the mode has been lowered, and the HP_OP_WFRAG records stand at the HEAD of the forward segment — so `lower` runs the
        # lowering twice: a first pass that only records the uses (wfrag_used), a second one that is told them (wfrag_needed)
        self.wfrag_needed = wfrag_needed
        self.wfrag_used = {"train": set(), "eval": set()}
        self.wfrag_refs = {}                 # (weight key, w_kn) -> Ref of its fragment image
        self.mode = None
        self.with_class = with_class
        self.target_blocks = wgrad_target_blocks
        self.pl = Plan(cfg, batch, self.train, with_class)
        self.o = self.pl.ops
        self.pending_wgrads = []
        self.pending_small = []              # deferred small leaf ops of the backward pass: (op, flags, i, f, buf, note)
        self.conv_rec_of = {}                # encoded OUT ref -> index of the CONV_TAPS record that produced it
        self.count_flops = False          # forward FLOPs (2*MAC, conv + linear) are counted for the training forward only
        if self.train.mfma_dtype not in ("f32", "bf16", "bf16x3"):
            raise ValueError(f"mfma_dtype must be 'f32', 'bf16x3' or 'bf16', not {self.train.mfma_dtype!r}")
        self.mm_flag = {"f32": 0, "bf16": P.CONV_BF16, "bf16x3": P.CONV_BF16X3}[self.train.mfma_dtype]
        if self.train.act_dtype not in ("f32", "bf16") or (self.train.act_dtype == "bf16" and self.train.mfma_dtype != "bf16"):
            raise ValueError("act_dtype must be 'f32' or 'bf16', and 'bf16' needs mfma_dtype='bf16'")
        self.abf = False                     # set per pass by build(): the training passes of an act_dtype="bf16" lowering
        # deterministic_wgrad also covers the small weight-gradient reductions (stem, tail, the heads' linears): flags & 1
        # = one workgroup per output group, no cross-workgroup atomics
        self.det_flag = 1 if self.train.deterministic_wgrad else 0

    # ---- parameter declaration (own order; class_embedding last so that AdamW can skip it) ----
    def declare_enc_block(self, p, cin, stride):
        """BasicBlockEnc(in_planes=cin, stride) (hippie/backbones.py:20-34): planes = cin * stride"""
        pl, planes = self.pl, cin * stride
        blk = dict(prefix=p, cin=cin, cout=planes, stride=stride)
        blk["conv1"] = pl.param(p + "conv1.weight", (planes, cin, 3), "tnc")
        blk["bn1"] = pl.bn_params(p + "bn1", planes)
        blk["conv2"] = pl.param(p + "conv2.weight", (planes, planes, 3), "tnc")
        blk["bn2"] = pl.bn_params(p + "bn2", planes)
        if stride != 1:
            blk["sc"] = pl.param(p + "shortcut.0.weight", (planes, cin, 1))
            blk["scbn"] = pl.bn_params(p + "shortcut.1", planes)
        return blk

    def declare_encoder(self, pre):
        pl = self.pl
        e = dict(prefix=pre)
        e["conv1"] = pl.param(pre + "conv1.weight", (64, 1, 3))
        e["bn1"] = pl.bn_params(pre + "bn1", 64)
        e["blocks"] = []
        cin = 64
        for li, planes in enumerate((64, 128, 256, 512), start=1):
            for bi in (0, 1):
                stride = 2 if (bi == 0 and li > 1) else 1
                e["blocks"].append(self.declare_enc_block(f"{pre}layer{li}.{bi}.", cin, stride))
                cin = planes
        e["lin_w"] = pl.param(pre + "linear.weight", (2 * self.cfg.z_dim, 512))
        e["lin_b"] = pl.param(pre + "linear.bias", (2 * self.cfg.z_dim,))
        return e

    def declare_dec_block(self, p, cin, stride):
        """BasicBlockDec(in_planes=cin, stride) (hippie/backbones.py:45-63): planes = cin / stride"""
        pl, cout = self.pl, cin // stride
        blk = dict(prefix=p, cin=cin, cout=cout, stride=stride)
        blk["conv2"] = pl.param(p + "conv2.weight", (cin, cin, 3), "tnc")
        blk["bn2"] = pl.bn_params(p + "bn2", cin)
        if stride == 1:
            blk["conv1"] = pl.param(p + "conv1.weight", (cout, cin, 3), "tnc")
            blk["bn1"] = pl.bn_params(p + "bn1", cout)
        else:
            blk["conv1"] = pl.param(p + "conv1.conv.weight", (cout, cin, 3), "tnc")
            blk["conv1_b"] = pl.param(p + "conv1.conv.bias", (cout,))
            blk["bn1"] = pl.bn_params(p + "bn1", cout)
            blk["sc"] = pl.param(p + "shortcut.0.conv.weight", (cout, cin, 3), "tnc")
            blk["sc_b"] = pl.param(p + "shortcut.0.conv.bias", (cout,))
            blk["scbn"] = pl.bn_params(p + "shortcut.1", cout)
        return blk

    def declare_decoder(self, pre, output_size):
        pl = self.pl
        d = dict(prefix=pre, output_size=output_size)
        d["lin_w"] = pl.param(pre + "linear.weight", (512, 2 * self.cfg.z_dim))
        d["lin_b"] = pl.param(pre + "linear.bias", (512,))
        d["blocks"] = []
        cin = 512
        for li, planes in ((4, 256), (3, 128), (2, 64), (1, 64)):
            for bi, stride in enumerate((1, 1 if li == 1 else 2)):
                d["blocks"].append(self.declare_dec_block(f"{pre}layer{li}.{bi}.", cin, stride))
            cin = planes
        d["tail_w"] = pl.param(pre + "conv1.conv.weight", (1, 64, 3))
        d["tail_b"] = pl.param(pre + "conv1.conv.bias", (1,))
        d["out_w"] = pl.param(pre + "linear_out.weight", (output_size, 64))
        d["out_b"] = pl.param(pre + "linear_out.bias", (output_size,))
        return d

    def declare_linear(self, key, n, k):
        return dict(w=self.pl.param(key + ".weight", (n, k)), b=self.pl.param(key + ".bias", (n,)), N=n, K=k)

    def declare_zml(self):
        """z_mean and z_log_var stacked into one [2z][z] Linear (their weights/biases are adjacent)."""
        z = self.cfg.z_dim
        pl = self.pl
        wm = pl.param("z_mean.weight", (z, z), align=4)
        wv = pl.param("z_log_var.weight", (z, z), align=1)
        bm = pl.param("z_mean.bias", (z,), align=4)
        bv = pl.param("z_log_var.bias", (z,), align=1)
        assert wv.offset == wm.offset + z * z and bv.offset == bm.offset + z
        return dict(w=wm, b=bm, N=2 * z, K=z)

    # ---- activation storage (TrainCfg.act_dtype) ------------------------------------------------
    @property
    def aflag(self):
        """HP_FLAG_ACT_BF16 while the pass being emitted stores its backbone activations as bfloat16"""
        return P.FLAG_ACT_BF16 if self.abf else 0

    def A(self, n):
        """workspace for an n-element backbone activation / activation-gradient tensor in the pass's storage type"""
        return self.pl.ws(2 * n) if self.abf else self.pl.f32(n)

    # ---- op emitters --------------------------------------------------------------
    def conv(self, tm: TapMap, a, w: PInfo, out, bias=None, stats=None, w_kn=False, note="", a2=None, w2: PInfo = None,
             in_bn=None, epi=None):
        """One HP_OP_CONV_TAPS record.  in_bn = dict(bn, stats, M): the A operand is leaky_relu(bn(a)) evaluated in
        the loader (HP_CONV_IN_BN).  epi = a reduction spec (red_spec): HP_OP_BN_BWD_REDUCE fused into the epilogue,
        `out` must be the spec's g tensor."""
        flags = (P.CONV_W_KN if w_kn else 0) | (P.CONV_BIAS if bias is not None else 0) | (P.CONV_STATS if stats is not None else 0)
        flags |= self.mm_flag | self.aflag
        ii = tm.conv_ints() + [0, 0]
        ff = [0.0] * 6
        bufs = [a, w.ref, out, bias.ref if bias is not None else None, stats] + [None] * 19
        ii += [0] * (40 - len(ii))
        if a2 is not None:
            bufs[10], bufs[11] = a2, w2.ref
        if in_bn is not None:
            bn = in_bn["bn"]
            flags |= P.CONV_IN_BN
            bn["save"], bn["coef"] = self.pl.f32(2 * bn["C"]), self.pl.f32(2 * bn["C"])
            W = self.train.sync_bn_world
            if W > 1:
                self.stats_sync(in_bn["stats"], bn["C"], bn["prefix"])
            ii[31], ii[32] = in_bn["M"], W
            ff[2], ff[3], ff[4] = SLOPE_BACKBONE, BN_EPS, BN_MOMENTUM
            bufs[5:9] = [bn["gamma"].ref, bn["beta"].ref, bn["rmean"].ref, bn["rvar"].ref]
            bufs[12:15] = [in_bn["stats"], bn["save"], bn["coef"]]
            self.pl.act_sites.append(dict(key=bn["prefix"], M=in_bn["M"], C=bn["C"], kind="raw", raw=a, coef=bn["coef"]))
            note += " <- lrelu(" + bn["prefix"] + ") in the loader"
        if epi is not None:
            assert out is epi["g"]
            flags |= P.CONV_EPI_BNRED
            ff[5] = epi["slope"]
            bufs[15:24] = [epi["g2"], epi["act"], epi["raw"], epi["bn"]["save"], epi["coef"], epi["bs"], epi["raw_b"],
                           epi["bn_b"]["save"] if epi["bn_b"] is not None else None, epi["bs_b"]]
            note += " + " + epi["bn"]["prefix"] + " bwd-reduce"
        if (flags & P.CONV_BF16X3) and self.train.weight_fragments and self.mode is not None and tm.K % 32 == 0 and (not w_kn or tm.N % 32 == 0):
            keys = [(w.key, bool(w_kn))] + ([(w2.key, bool(w_kn))] if a2 is not None else [])
            self.wfrag_used[self.mode].update(keys)
            if all(k in self.wfrag_refs for k in keys):
                flags |= P.CONV_WFRAG
                bufs += [None] * (P.NB - len(bufs))
                bufs[24] = self.wfrag_refs[keys[0]]
                if a2 is not None:
                    bufs[25] = self.wfrag_refs[keys[1]]
        self.o.add(P.CONV_TAPS, flags, i=ii, f=ff, buf=bufs, note=note)
        self.conv_rec_of[out.encode()] = len(self.o.recs) - 1
        if self.count_flops and not w_kn:
            self.pl.flops_fwd += 2 * tm.M * tm.N * tm.K * len(tm.taps)

    def pair_last_two(self, note="pair"):
        """Launch the two most recently emitted ops (independent, same opcode) as one HP_OP_PAIR."""
        if not self.train.intra_pair:
            return
        a, b = len(self.o.recs) - 2, len(self.o.recs) - 1
        ra, rb = self.o.recs[a], self.o.recs[b]
        assert int(ra["op"]) == int(rb["op"])
        if int(ra["op"]) == P.CONV_TAPS and (int(ra["flags"]) & 1) != (int(rb["flags"]) & 1):
            return
        ra["flags"] = int(ra["flags"]) | P.FLAG_MEMBER
        rb["flags"] = int(rb["flags"]) | P.FLAG_MEMBER
        self.o.add(P.PAIR, 0, i=[a, b], note=note)

    def zip_towers(self, i0, i1):
        """Multimodal model: recs[i0:i1] and recs[i1:] are the SAME sub-network emitted for the two modality towers (encoder_mod1 /
        encoder_mod2, decoder_mod1 / decoder_mod2; hippie/model.py:352-432), independent of each other.  One after the other they
        are one latency-bound chain twice as long; interleaved, every pair of same-kind launches that HP_OP_PAIR can run together
        (convs, BatchNorm apply / backward reduce / backward apply) becomes ONE launch with both towers' workgroups — at batch 512 a
        single tower's layer does not fill the chip.  Launches that are already intra-tower pairs, and kinds HP_OP_PAIR does not
        take, stay single, tower 1's before tower 2's (the one cross-tower dependence — decoder_fc.0's input gradients accumulate
        into one buffer — keeps its order that way)."""
        if not (self.train.zip_towers and self.train.intra_pair) or P.debug_knob("HIPPIE_NO_ZIP_TOWERS") == "1":     # (the knob: A/B runs of unmodified callers)
            return
        recs, notes = self.o.recs, self.o.notes
        i2 = len(recs)
        # the ordered-slab weight gradients (deterministic_wgrad) are emitted in line and share ONE slab buffer: [wgrad -> slab, slab ->
        # gradient] of tower 1 must not be interleaved with tower 2's.  (The default path defers all weight gradients to one grouped
        # launch at the end of the pass; every other buffer inside the two ranges is the tower's own.)
        if any(int(r["op"]) in (P.SLAB_REDUCE, P.WGRAD_TAPS) for r in recs[i0:i2]):
            return

        def units(lo, hi):
            out, k = [], lo
            while k < hi:
                if int(recs[k]["op"]) != P.PAIR and int(recs[k]["flags"]) & P.FLAG_MEMBER:
                    j = k
                    while int(recs[j]["op"]) != P.PAIR:
                        j += 1
                        if j >= hi:
                            return None                       # a member without its launch record inside the range: leave everything alone
                    out.append(list(range(k, j + 1)))
                    k = j + 1
                else:
                    out.append([k])
                    k += 1
            return out

        U1, U2 = units(i0, i1), units(i1, i2)
        if U1 is None or U2 is None or len(U1) != len(U2):
            return
        pairable = (P.CONV_TAPS, P.BN_APPLY, P.BN_BWD_REDUCE, P.BN_BWD_APPLY)

        def can_pair(a, b):
            ra, rb = recs[a], recs[b]
            op = int(ra["op"])
            if op != int(rb["op"]) or op not in pairable:
                return False
            fa, fb = int(ra["flags"]), int(rb["flags"])
            if op == P.CONV_TAPS:
                return (fa & 1) == (fb & 1) and (fa & (P.CONV_BF16 | P.CONV_BF16X3 | P.CONV_WFRAG)) == (fb & (P.CONV_BF16 | P.CONV_BF16X3 | P.CONV_WFRAG))
            return (int(ra["i"][1]) % 4 == 0) == (int(rb["i"][1]) % 4 == 0)       # both on the same vector width (C % 4)

        order = []                                           # ("o", old index) | ("p", old a, old b)
        for u1, u2 in zip(U1, U2):
            if len(u1) == 1 and len(u2) == 1 and can_pair(u1[0], u2[0]):
                order += [("o", u1[0]), ("o", u2[0]), ("p", u1[0], u2[0])]
            else:
                order += [("o", k) for k in u1] + [("o", k) for k in u2]
        remap, new_recs, new_notes = {}, [], []
        for item in order:
            if item[0] == "o":
                remap[item[1]] = i0 + len(new_recs)
                new_recs.append(recs[item[1]])
                new_notes.append(notes[item[1]])
            else:
                a, b = item[1], item[2]
                for k in (a, b):
                    recs[k]["flags"] = int(recs[k]["flags"]) | P.FLAG_MEMBER
                r = np.zeros((), dtype=P.OP_DTYPE)
                r["op"] = P.PAIR
                r["buf"][:] = P.NULL
                r["i"][0], r["i"][1] = -1 - a, -1 - b         # old indices, marked: resolved below
                new_recs.append(r)
                new_notes.append("pair " + notes[a] + " | " + notes[b])
        for r in new_recs:
            if int(r["op"]) == P.PAIR:
                for q in (0, 1):
                    v = int(r["i"][q])
                    r["i"][q] = remap[-1 - v] if v < 0 else remap[v]
        recs[i0:i2] = new_recs
        notes[i0:i2] = new_notes
        for key, k in list(self.conv_rec_of.items()):
            if k in remap:
                self.conv_rec_of[key] = remap[k]

    def wgrad(self, tm: TapMap, dy, x, w: PInfo, note="", coef=None):
        """coef: x is the raw input of a BatchNorm whose activation was never stored (HP_CONV_IN_BN on the forward
        conv): the kernel re-evaluates leaky_relu(fma(x, scale, shift)) from (scale, shift) = coef."""
        xf = (P.CONV_IN_BN if coef is not None else 0) | self.mm_flag | self.aflag
        tiles = -(-tm.N // 64) * -(-tm.K // 64)
        if self.train.grouped_wgrad and not self.train.deterministic_wgrad:
            # deferred: all wgrads of the backward pass run in one grouped launch at its end.  With the
            # whole chip shared, ~64 workgroups per problem suffice: big-weight layers are not split at all.
            per_problem = int(P.debug_knob("HIPPIE_WGRAD_BLOCKS", "64"))      # (the knob: tools/micro sweeps)
            nsplit = max(1, min(per_problem // tiles, -(-tm.M // 256)))
            if len(tm.taps) == 1:
                # the 1-tap group (three shortcut convs) is too small to fill the chip at 64 blocks per problem
                # (measured: 75 us -> 35 us with <= 512 rows per split)
                nsplit = max(nsplit, -(-tm.M // 512))
            rps = _round_up(-(-tm.M // nsplit), 32)
            nsplit = -(-tm.M // rps)
            self.pending_wgrads.append((tm, nsplit, rps, dy, x, w, note, coef))
            return
        # enough workgroups to fill 256 CUs, but at least 256 rows (8 K-slices) per split so that the
        # tile's atomics / slab traffic stays small next to its MFMA work
        nsplit = max(1, min(self.target_blocks // tiles, -(-tm.M // 256)))
        rps = _round_up(-(-tm.M // nsplit), 32)
        nsplit = -(-tm.M // rps)
        if not self.train.deterministic_wgrad:
            self.o.add(P.WGRAD_TAPS, 1 | xf, i=tm.ints() + [nsplit, rps, w.numel], f=[SLOPE_BACKBONE], buf=[dy, x, w.gref, coef], note=note)
            return
        self.pl.slab_need = max(self.pl.slab_need, nsplit * w.numel)
        self.o.add(P.WGRAD_TAPS, xf, i=tm.ints() + [nsplit, rps, w.numel], f=[SLOPE_BACKBONE], buf=[dy, x, self.slab, coef], note=note)
        self.o.add(P.SLAB_REDUCE, 0, i=[w.numel, nsplit, w.numel], buf=[self.slab, w.gref], note=note + " reduce")

    def foldable(self, raw):
        k = self.conv_rec_of.get(raw.encode())
        return k is not None and int(self.o.recs[k]["op"]) == P.CONV_TAPS and not (int(self.o.recs[k]["flags"]) & P.CONV_BN_EVAL)

    def order_shortcut_first(self, raw, res):
        """The main conv's folded epilogue reads the NORMALISED shortcut, so the shortcut conv must run in an
        earlier launch.  Encoder blocks already have that order (shortcut paired with conv1, BN2 follows conv2).
        Decoder blocks emit [conv1, shortcut, PAIR] immediately before their BN: un-pair and swap those two."""
        k_raw, k_res = self.conv_rec_of[raw.encode()], self.conv_rec_of[res.encode()]
        if k_res < k_raw:
            last = self.o.recs[-1]
            paired_with_raw = int(last["op"]) == P.PAIR and k_raw in (int(last["i"][0]), int(last["i"][1]))
            return not (paired_with_raw and k_res in (int(last["i"][0]), int(last["i"][1])))
        n = len(self.o.recs)
        if k_res != k_raw + 1:
            return False
        if k_res == n - 2 and int(self.o.recs[-1]["op"]) == P.PAIR and \
                sorted((int(self.o.recs[-1]["i"][0]), int(self.o.recs[-1]["i"][1]))) == [k_raw, k_res]:
            self.o.recs.pop()
            self.o.notes.pop()
            for k in (k_raw, k_res):
                self.o.recs[k]["flags"] = int(self.o.recs[k]["flags"]) & ~P.FLAG_MEMBER
        elif k_res != n - 1:
            return False
        self.o.recs[k_raw], self.o.recs[k_res] = self.o.recs[k_res], self.o.recs[k_raw]
        self.o.notes[k_raw], self.o.notes[k_res] = self.o.notes[k_res], self.o.notes[k_raw]
        self.conv_rec_of[raw.encode()], self.conv_rec_of[res.encode()] = k_res, k_raw
        return True

    def fold_bn_into_conv(self, raw, out, bn, act, slope, res):
        """Eval mode: turn `conv -> raw; BN_APPLY(raw) -> out` into one conv launch with a BatchNorm epilogue
        (HP_OP_CONV_TAPS flag 8).  Returns False when `raw` was not produced by a CONV_TAPS record."""
        if not self.foldable(raw):
            return False
        k = self.conv_rec_of[raw.encode()]
        r = self.o.recs[k]
        r["flags"] = int(r["flags"]) | P.CONV_BN_EVAL | (P.CONV_ACT if act else 0)
        r["buf"][2] = out.encode()
        for j, ref in enumerate((bn["gamma"].ref, bn["beta"].ref, bn["rmean"].ref, bn["rvar"].ref)):
            r["buf"][5 + j] = ref.encode()
        r["buf"][9] = res.encode() if res is not None else P.NULL
        r["f"][0], r["f"][1] = BN_EPS, slope
        self.o.notes[k] += " + " + bn["prefix"] + " (eval BN folded)"
        return True

    def bn_apply(self, M, bn, raw, out, stats, training, act, slope, res_mode=0, res=None, bn2=None, stats2=None, heads=False):
        """heads=True: one of the heads' BatchNorms (fp32 tensors whatever the backbones' storage type)"""
        if not training and self.train.fold_eval_bn:
            if res_mode == 2:
                # the shortcut conv normalises its own output in place, then it is a plain residual tensor
                if self.foldable(res) and self.foldable(raw) and self.order_shortcut_first(raw, res):
                    self.fold_bn_into_conv(res, res, bn2, False, slope, None)
                    self.fold_bn_into_conv(raw, out, bn, act, slope, res)
                    return
            elif self.fold_bn_into_conv(raw, out, bn, act, slope, res if res_mode == 1 else None):
                return
        save = self.pl.f32(2 * bn["C"])
        bn["save"] = save
        bufs = [raw, out, stats if training else None, bn["gamma"].ref, bn["beta"].ref, bn["rmean"].ref, bn["rvar"].ref, save]
        if res_mode == 1:
            bufs += [res]
        elif res_mode == 2:
            save2 = self.pl.f32(2 * bn2["C"])
            bn2["save"] = save2
            bufs += [res, stats2 if training else None, bn2["gamma"].ref, bn2["beta"].ref, bn2["rmean"].ref, bn2["rvar"].ref, save2]
        W = self.train.sync_bn_world if training else 0
        if W > 1:
            self.stats_sync(stats, bn["C"], bn["prefix"])
            if res_mode == 2:
                self.stats_sync(stats2, bn2["C"], bn2["prefix"])
        if training and act:
            self.pl.act_sites.append(dict(key=bn["prefix"], M=M, C=bn["C"], kind="tensor", out=out))
        self.o.add(P.BN_APPLY, 0 if heads else self.aflag, i=[M, bn["C"], res_mode, 1 if training else 0, 1 if act else 0, W],
                   f=[slope, BN_EPS, BN_MOMENTUM], buf=bufs, note=bn["prefix"])

    def stats_sync(self, slot, C, prefix):
        """sync-BatchNorm: the replicated fp64 slot must be summed over the data-parallel ranks here."""
        self.o.add(P.STATS_SYNC, 0, i=[P.stat_repl(C) * 2 * C], buf=[slot], note=prefix + " statistics all-reduce")

    # BatchNorm backward = a reduction (mask, sum g, sum g*xhat) + an apply.  The reduction either runs as its own
    # launch (reduce_op) or in the epilogue of the input-gradient conv that produces its operand (conv(epi=spec)).
    def red_spec(self, M, bn, act, raw, g2=None, bn_b=None, raw_b=None, slope=SLOPE_BACKBONE, heads=False):
        """act None: the activation was never stored (HP_CONV_IN_BN consumer); its sign comes from raw + bn["coef"].
        heads: one of the heads' BatchNorms (fp32 tensors whatever the backbones' storage type)."""
        C = bn["C"]
        return dict(M=M, C=C, bn=bn, act=act, raw=raw, g2=g2, bn_b=bn_b, raw_b=raw_b, slope=slope, heads=heads,
                    coef=bn["coef"] if act is None else None,
                    g=self.pl.f32(M * C) if heads else self.A(M * C), bs=self.pl.stat(2 * C), bs_b=self.pl.stat(2 * C) if bn_b is not None else None)

    def reduce_sync(self, sp):
        if self.train.sync_bn_world > 1:
            self.stats_sync(sp["bs"], sp["C"], sp["bn"]["prefix"] + " bwd")
            if sp["bn_b"] is not None:
                self.stats_sync(sp["bs_b"], sp["C"], sp["bn_b"]["prefix"] + " bwd")

    def reduce_op(self, sp, g1):
        bn, bn_b = sp["bn"], sp["bn_b"]
        bufs = [g1, sp["g2"], sp["act"], sp["g"], sp["raw"], bn["save"], sp["bs"],
                sp["raw_b"], bn_b["save"] if bn_b is not None else None, sp["bs_b"], sp["coef"]]
        self.o.add(P.BN_BWD_REDUCE, 0 if sp["heads"] else self.aflag, i=[sp["M"], sp["C"], 1 if sp["g2"] is not None else 0, 1 if bn_b is not None else 0],
                   f=[sp["slope"]], buf=bufs, note=bn["prefix"] + " bwd-reduce")
        self.reduce_sync(sp)

    def apply_op(self, sp):
        """-> (dr, dr_b): gradients of the BatchNorm inputs (conv outputs)."""
        M, C, bn, bn_b, W = sp["M"], sp["C"], sp["bn"], sp["bn_b"], self.train.sync_bn_world
        af = 0 if sp["heads"] else self.aflag
        dr = self.pl.f32(M * C) if sp["heads"] else self.A(M * C)
        self.o.add(P.BN_BWD_APPLY, af, i=[M, C, W], buf=[sp["g"], sp["raw"], bn["save"], sp["bs"], bn["gamma"].ref, dr, bn["gamma"].gref, bn["beta"].gref],
                   note=bn["prefix"] + " bwd-apply")
        dr_b = None
        if bn_b is not None:
            dr_b = self.pl.f32(M * C) if sp["heads"] else self.A(M * C)
            self.o.add(P.BN_BWD_APPLY, af, i=[M, C, W], buf=[sp["g"], sp["raw_b"], bn_b["save"], sp["bs_b"], bn_b["gamma"].ref, dr_b,
                                                         bn_b["gamma"].gref, bn_b["beta"].gref], note=bn_b["prefix"] + " bwd-apply")
            self.pair_last_two("pair " + bn["prefix"] + " + shortcut bwd-apply")
        return dr, dr_b

    def bn_bwd(self, M, bn, g1, g2, act, raw, slope, bn_b=None, raw_b=None, heads=False):
        """standalone reduce + apply; returns (g, dr, dr_b): masked upstream gradient and BN input gradients."""
        sp = self.red_spec(M, bn, act, raw, g2, bn_b, raw_b, slope, heads=heads)
        self.reduce_op(sp, g1)
        dr, dr_b = self.apply_op(sp)
        return sp["g"], dr, dr_b

    def dgrad_reduce(self, tm, dr, w, sp, note):
        """Input-gradient conv whose output feeds the BatchNorm-backward reduction `sp` (fused when fuse_bn)."""
        if self.train.fuse_bn:
            self.conv(tm, dr, w, sp["g"], w_kn=True, epi=sp, note=note)
            self.reduce_sync(sp)
        else:
            tmp = self.A(tm.out_rows * tm.N)
            self.conv(tm, dr, w, tmp, w_kn=True, note=note)
            self.reduce_op(sp, tmp)

    def linear_fwd(self, M, lin, x, ldx, y, ldy, act=False, stats=None, note=""):
        self.o.add(P.LINEAR_FWD, 0, i=[M, lin["N"], lin["K"], ldx, ldy, 1 if act else 0, 1 if stats is not None else 0],
                   f=[SLOPE_HEADS], buf=[x, lin["w"].ref, lin["b"].ref, y, stats], note=note)
        if act and self.count_flops:        # (count_flops marks the training forward)
            self.pl.act_sites.append(dict(key=note.split(" ")[0], M=M, C=lin["N"], kind="tensor", out=y))
        if self.count_flops:
            self.pl.flops_fwd += 2 * M * lin["N"] * lin["K"]

    def linear_bwd(self, M, lin, dy, ldy, x, ldx, dx=None, lddx=None, mask=None, ldmask=0, accumulate=False, note=""):
        self.small_leaf(P.LINEAR_BWD_W, self.det_flag, [M, lin["N"], lin["K"], ldy, ldx], (), [dy, x, lin["w"].gref, lin["b"].gref], note + " dW")
        if dx is not None:
            self.o.add(P.LINEAR_BWD_X, 0, i=[M, lin["N"], lin["K"], ldy, lddx, 1 if mask is not None else 0, ldmask, 1 if accumulate else 0],
                       f=[SLOPE_HEADS], buf=[dy, lin["w"].ref, dx, mask], note=note + " dX")

    # ---- tap maps --------------------------------------------------------------
    def map_fwd(self, Lin, cin, cout, stride, k=3):
        if k == 3:
            Lout = (Lin + 2 - 3) // stride + 1
            taps = [(t - 1, t) for t in range(3)]
        else:
            Lout = (Lin - 1) // stride + 1
            taps = [(0, 0)]
        return TapMap(self.B * Lout, cout, cin, Lout, Lin, Lin, stride, 0, taps), Lout

    def map_fwd_up(self, Lin, cin, cout):
        Lout = 2 * Lin
        return TapMap(self.B * Lout, cout, cin, Lout, Lin, 2 * Lin, 1, 1, [(t - 1, t) for t in range(3)]), Lout

    def map_dgrad(self, Lx, Ly, cin, cout):
        """input-gradient of a stride-1 k=3 conv: dx[p] = sum_t dy[p + 1 - t] . w[t]"""
        return TapMap(self.B * Lx, cin, cout, Lx, Ly, Ly, 1, 0, [(1 - t, t) for t in range(3)])

    def map_dgrad_s2_phases(self, Lx, Ly, cin, cout):
        """Input-gradient of an encoder stride-2 block with respect to its input x (length Lx), both paths at once:
        conv1 (k=3, s=2, p=1: y[l] = sum_t x[2l+t-1] w[t], source 0) and the 1x1 stride-2 shortcut (ys[l] = x[2l] ws,
        source 1).  By output parity, with only the taps that can contribute (no masked MFMA work):
            dx[2l]   = dy[l] w[1] + dys[l] ws            dx[2l+1] = dy[l+1] w[0] + dy[l] w[2]
        Two ops writing the interleaved rows of ONE tensor [B*Lx][cin]; the odd one is absent when Lx == 1."""
        out = []
        for q, n, taps in ((0, (Lx + 1) // 2, [(0, 1, 0), (0, 0, 1)]), (1, Lx // 2, [(1, 0, 0), (0, 2, 0)])):
            if n > 0:
                out.append(TapMap(self.B * n, cin, cout, n, Ly, Ly, 1, 0, taps, out_Lfull=Lx, out_a=2, out_o=q))
        return out

    def map_dgrad_up(self, Lx, cin, cout):
        Ly = 2 * Lx
        return TapMap(self.B * Lx, cin, cout, Lx, Ly, Ly, 2, 0, [(e - t + 1, t) for e in (0, 1) for t in range(3)])

    # ---- encoder ------------------------------------------------------------------
    def encoder_fwd(self, e, x, L, training):
        pl, B = self.pl, self.B
        L1 = (L + 2 - 3) // 2 + 1
        M = B * L1
        raw0 = self.A(M * 64)
        st = pl.stat(128) if training else None
        self.o.add(P.STEM_FWD, self.aflag, i=[B, L, L1, 64], buf=[x, e["conv1"].ref, raw0, st], note=e["prefix"] + "conv1")
        if self.count_flops:
            pl.flops_fwd += 2 * M * 64 * 3
        a0 = self.A(M * 64)
        self.bn_apply(M, e["bn1"], raw0, a0, st, training, True, SLOPE_BACKBONE)
        e.update(x=x, L=L, L1=L1, raw0=raw0, a0=a0)
        cur, Lc = a0, L1
        for blk in e["blocks"]:
            cur, Lc = self.enc_block_fwd(blk, cur, Lc, training)
        pooled = pl.f32(B * 512)
        self.o.add(P.POOL_FWD, self.aflag, i=[B, Lc, 512], buf=[cur, pooled], note=e["prefix"] + "avgpool")
        e.update(pooled=pooled, Llast=Lc, last=cur)
        return pooled

    def enc_block_fwd(self, blk, cur, Lc, training):
        """BasicBlockEnc.forward (hippie/backbones.py:36-41): lrelu(bn1(conv1(x))) -> bn2(conv2(.)) -> += shortcut(x) -> lrelu.
        cur: [B*Lc][cin] channels-last.  -> (out, Lout)"""
        pl, B = self.pl, self.B
        cin, cout, s = blk["cin"], blk["cout"], blk["stride"]
        tm1, Lo = self.map_fwd(Lc, cin, cout, s)
        Mo = B * Lo
        r1 = self.A(Mo * cout)
        st1 = pl.stat(2 * cout) if training else None
        self.conv(tm1, cur, blk["conv1"], r1, stats=st1, note=blk["prefix"] + "conv1")
        if s != 1:
            tms, _ = self.map_fwd(Lc, cin, cout, s, k=1)
            rs = self.A(Mo * cout)
            sts = pl.stat(2 * cout) if training else None
            self.conv(tms, cur, blk["sc"], rs, stats=sts, note=blk["prefix"] + "shortcut.0")
            self.pair_last_two("pair " + blk["prefix"] + "conv1 + shortcut.0")
        tm2, _ = self.map_fwd(Lo, cout, cout, 1)
        r2 = self.A(Mo * cout)
        st2 = pl.stat(2 * cout) if training else None
        if training and self.train.fuse_bn:
            # lrelu(bn1(r1)) has exactly one consumer, conv2: evaluated in its operand loader, never stored
            a1 = None
            self.conv(tm2, r1, blk["conv2"], r2, stats=st2, note=blk["prefix"] + "conv2", in_bn=dict(bn=blk["bn1"], stats=st1, M=Mo))
        else:
            a1 = self.A(Mo * cout)
            self.bn_apply(Mo, blk["bn1"], r1, a1, st1, training, True, SLOPE_BACKBONE)
            self.conv(tm2, a1, blk["conv2"], r2, stats=st2, note=blk["prefix"] + "conv2")
        out = self.A(Mo * cout)
        blk.update(x=cur, Lin=Lc, Lout=Lo, r1=r1, a1=a1, r2=r2, out=out, tm1=tm1, tm2=tm2)
        if s == 1:
            self.bn_apply(Mo, blk["bn2"], r2, out, st2, training, True, SLOPE_BACKBONE, 1, cur)
        else:
            self.bn_apply(Mo, blk["bn2"], r2, out, st2, training, True, SLOPE_BACKBONE, 2, rs, blk["scbn"], sts)
            blk.update(rs=rs, tms=tms)
        return out, Lo

    def enc_out_spec(self, e, bi, g2):
        """reduction spec of the BatchNorm that produced the INPUT of encoder block bi (block bi-1's bn2 [+ shortcut
        BN], or the stem's bn1 for bi == 0); g2 = the gradient arriving over the identity shortcut, if any."""
        B = self.B
        if bi == 0:
            return self.red_spec(B * e["L1"], e["bn1"], e["a0"], e["raw0"], g2)
        blk = e["blocks"][bi - 1]
        M = B * blk["Lout"]
        if blk["stride"] == 1:
            return self.red_spec(M, blk["bn2"], blk["out"], blk["r2"], g2)
        return self.red_spec(M, blk["bn2"], blk["out"], blk["r2"], g2, blk["scbn"], blk["rs"])

    def encoder_bwd(self, e, dpooled):
        """dpooled: [B][512] gradient of the pooled features."""
        pl, B = self.pl, self.B
        Lc = e["Llast"]
        G1 = self.A(B * Lc * 512)
        self.o.add(P.POOL_BWD, self.aflag, i=[B, Lc, 512], buf=[dpooled, G1], note=e["prefix"] + "avgpool bwd")
        blocks = e["blocks"]
        sp = self.enc_out_spec(e, len(blocks), None)          # the last block's output BatchNorm
        self.reduce_op(sp, G1)
        for bi in range(len(blocks) - 1, -1, -1):
            blk = blocks[bi]
            cin, cout, s = blk["cin"], blk["cout"], blk["stride"]
            Lo, Li = blk["Lout"], blk["Lin"]
            Mo = B * Lo
            p = blk["prefix"]
            dr2, drs = self.apply_op(sp)
            stored = blk["a1"] is not None
            self.wgrad(blk["tm2"], dr2, blk["a1"] if stored else blk["r1"], blk["conv2"], note=p + "conv2 wgrad",
                       coef=None if stored else blk["bn1"]["coef"])
            sp1 = self.red_spec(Mo, blk["bn1"], blk["a1"], blk["r1"])
            self.dgrad_reduce(self.map_dgrad(Lo, Lo, cout, cout), dr2, blk["conv2"], sp1, p + "conv2 dgrad")
            dr1, _ = self.apply_op(sp1)
            self.wgrad(blk["tm1"], dr1, blk["x"], blk["conv1"], note=p + "conv1 wgrad")
            if s == 1:
                # d/dx = conv1 path + identity shortcut (this block's masked gradient)
                spp = self.enc_out_spec(e, bi, sp["g"])
                self.dgrad_reduce(self.map_dgrad(Li, Lo, cin, cout), dr1, blk["conv1"], spp, p + "conv1 dgrad")
            else:
                self.wgrad(blk["tms"], drs, blk["x"], blk["sc"], note=p + "shortcut wgrad")
                # conv1 and the 1x1 shortcut both differentiate w.r.t. the same x: one tensor, written by an even-row
                # and an odd-row op (paired into one launch), each with only the taps that contribute
                spp = self.enc_out_spec(e, bi, None)
                fuse = self.train.fuse_bn
                dst = spp["g"] if fuse else self.A(B * Li * cin)
                tms = self.map_dgrad_s2_phases(Li, Lo, cin, cout)
                for q, tm in enumerate(tms):
                    self.conv(tm, dr1, blk["conv1"], dst, w_kn=True, a2=drs, w2=blk["sc"], epi=spp if fuse else None,
                              note=p + "conv1 + shortcut dgrad, " + ("even" if tm.out_o == 0 else "odd") + " rows")
                if len(tms) == 2:
                    self.pair_last_two("pair " + p + "conv1 + shortcut dgrad (even | odd rows)")
                if fuse:
                    self.reduce_sync(spp)
                else:
                    self.reduce_op(spp, dst)
            sp = spp
        dr0, _ = self.apply_op(sp)
        self.o.add(P.STEM_WGRAD, self.det_flag | self.aflag, i=[B, e["L"], e["L1"], 64], buf=[dr0, e["x"], e["conv1"].gref], note=e["prefix"] + "conv1 wgrad")

    # ---- decoder ------------------------------------------------------------------
    def decoder_fwd(self, d, din, training):
        """din: [B][2z] -> rec [B][output_size]"""
        pl, B, z = self.pl, self.B, self.cfg.z_dim
        lin = dict(w=d["lin_w"], b=d["lin_b"], N=512, K=2 * z)
        y = pl.f32(B * 512)
        self.linear_fwd(B, lin, din, 2 * z, y, 512, note=d["prefix"] + "linear")
        act0 = self.A(B * 4 * 512)
        self.o.add(P.REPEAT_FWD, self.aflag, i=[B, 4, 512], buf=[y, act0], note=d["prefix"] + "interpolate x4")
        d.update(din=din, lin=lin, y=y, act0=act0)
        cur, Lc = act0, 4
        for blk in d["blocks"]:
            cur, Lc = self.dec_block_fwd(blk, cur, Lc, training)
        assert Lc == 32
        t = pl.f32(B * 64)
        self.o.add(P.TAIL_FWD, self.aflag, i=[B, 32, 64], buf=[cur, d["tail_w"].ref, d["tail_b"].ref, t], note=d["prefix"] + "conv1 (resize 64->1)")
        if self.count_flops:
            pl.flops_fwd += 2 * B * 64 * 64 * 3
        lo = dict(w=d["out_w"], b=d["out_b"], N=d["output_size"], K=64)
        rec = pl.f32(B * d["output_size"])
        self.linear_fwd(B, lo, t, 64, rec, d["output_size"], note=d["prefix"] + "linear_out")
        d.update(last=cur, t=t, lo=lo, rec=rec)
        return rec

    def dec_block_fwd(self, blk, cur, Lc, training):
        """BasicBlockDec.forward (hippie/backbones.py:65-70): lrelu(bn2(conv2(x))) -> bn1(conv1(.)) (conv1 = ResizeConv1d when the
        block up-samples) -> += shortcut(x) (identity, or ResizeConv1d + BN) -> lrelu.  cur: [B*Lc][cin].  -> (out, Lout)"""
        pl, B = self.pl, self.B
        cin, cout, s = blk["cin"], blk["cout"], blk["stride"]
        Mi = B * Lc
        tm2, _ = self.map_fwd(Lc, cin, cin, 1)
        r2 = self.A(Mi * cin)
        st2 = pl.stat(2 * cin) if training else None
        self.conv(tm2, cur, blk["conv2"], r2, stats=st2, note=blk["prefix"] + "conv2")
        if training and self.train.fuse_bn:
            a2, src2 = None, r2                 # lrelu(bn2(r2)) is evaluated in conv1's operand loader
            ib = dict(bn=blk["bn2"], stats=st2, M=Mi)
        else:
            a2 = src2 = self.A(Mi * cin)
            ib = None
            self.bn_apply(Mi, blk["bn2"], r2, a2, st2, training, True, SLOPE_BACKBONE)
        blk.update(x=cur, Lin=Lc, r2=r2, a2=a2, tm2=tm2)
        if s == 1:
            tm1, Lo = self.map_fwd(Lc, cin, cout, 1)
            r1 = self.A(Mi * cout)
            st1 = pl.stat(2 * cout) if training else None
            self.conv(tm1, src2, blk["conv1"], r1, stats=st1, note=blk["prefix"] + "conv1", in_bn=ib)
            out = self.A(Mi * cout)
            self.bn_apply(Mi, blk["bn1"], r1, out, st1, training, True, SLOPE_BACKBONE, 1, cur)
        else:
            tm1, Lo = self.map_fwd_up(Lc, cin, cout)
            Mo = B * Lo
            r1 = self.A(Mo * cout)
            st1 = pl.stat(2 * cout) if training else None
            self.conv(tm1, src2, blk["conv1"], r1, bias=blk["conv1_b"], stats=st1, note=blk["prefix"] + "conv1 (resize)", in_bn=ib)
            rs = self.A(Mo * cout)
            sts = pl.stat(2 * cout) if training else None
            self.conv(tm1, cur, blk["sc"], rs, bias=blk["sc_b"], stats=sts, note=blk["prefix"] + "shortcut (resize)")
            self.pair_last_two("pair " + blk["prefix"] + "conv1 + shortcut (resize)")
            out = self.A(Mo * cout)
            self.bn_apply(Mo, blk["bn1"], r1, out, st1, training, True, SLOPE_BACKBONE, 2, rs, blk["scbn"], sts)
            blk.update(rs=rs)
        blk.update(r1=r1, out=out, tm1=tm1, Lout=Lo)
        return out, Lo

    def decoder_bwd(self, d, drec, ddin, accumulate=False):
        """drec: [B][out] ; writes d(din) [B][2z] into ddin."""
        pl, B, z = self.pl, self.B, self.cfg.z_dim
        dt = pl.f32(B * 64)
        self.linear_bwd(B, d["lo"], drec, d["output_size"], d["t"], 64, dt, 64, note=d["prefix"] + "linear_out")
        self.o.add(P.TAIL_BWD_W, self.det_flag | self.aflag, i=[B, 32, 64], buf=[dt, d["last"], d["tail_w"].gref, d["tail_b"].gref], note=d["prefix"] + "tail dW")
        G1 = self.A(B * 32 * 64)
        self.o.add(P.TAIL_BWD_X, self.aflag, i=[B, 32, 64], buf=[dt, d["tail_w"].ref, G1], note=d["prefix"] + "tail dX")
        blocks = d["blocks"]

        def out_spec(bi, g2):
            blk = blocks[bi]
            M = B * blk["Lout"]
            if blk["stride"] == 1:
                return self.red_spec(M, blk["bn1"], blk["out"], blk["r1"], g2)
            return self.red_spec(M, blk["bn1"], blk["out"], blk["r1"], g2, blk["scbn"], blk["rs"])

        sp = out_spec(len(blocks) - 1, None)
        self.reduce_op(sp, G1)
        G2 = None
        for bi in range(len(blocks) - 1, -1, -1):
            blk = blocks[bi]
            cin, cout, s = blk["cin"], blk["cout"], blk["stride"]
            Li, Lo = blk["Lin"], blk["Lout"]
            Mi = B * Li
            p = blk["prefix"]
            dr1, drs = self.apply_op(sp)
            stored = blk["a2"] is not None
            x1, c1 = (blk["a2"], None) if stored else (blk["r2"], blk["bn2"]["coef"])
            sp2 = self.red_spec(Mi, blk["bn2"], blk["a2"], blk["r2"])
            if s == 1:
                self.wgrad(blk["tm1"], dr1, x1, blk["conv1"], note=p + "conv1 wgrad", coef=c1)
                self.dgrad_reduce(self.map_dgrad(Li, Lo, cin, cout), dr1, blk["conv1"], sp2, p + "conv1 dgrad")
                side = sp["g"]
            else:
                self.wgrad(blk["tm1"], dr1, x1, blk["conv1"], note=p + "conv1 (resize) wgrad", coef=c1)
                self.wgrad(blk["tm1"], drs, blk["x"], blk["sc"], note=p + "shortcut (resize) wgrad")
                fuse = self.train.fuse_bn
                da2 = sp2["g"] if fuse else self.A(Mi * cin)
                self.conv(self.map_dgrad_up(Li, cin, cout), dr1, blk["conv1"], da2, w_kn=True, epi=sp2 if fuse else None,
                          note=p + "conv1 (resize) dgrad")
                side = self.A(Mi * cin)
                self.conv(self.map_dgrad_up(Li, cin, cout), drs, blk["sc"], side, w_kn=True, note=p + "shortcut (resize) dgrad")
                self.pair_last_two("pair " + p + "resize dgrads")
                if fuse:
                    self.reduce_sync(sp2)
                else:
                    self.reduce_op(sp2, da2)
            dr2, _ = self.apply_op(sp2)
            self.wgrad(blk["tm2"], dr2, blk["x"], blk["conv2"], note=p + "conv2 wgrad")
            tm = self.map_dgrad(Li, Li, cin, cin)
            if bi > 0:
                # the block's input is the previous block's output: its BatchNorm-backward reduction takes this
                # conv's result plus the gradient arriving over the shortcut path
                spp = out_spec(bi - 1, side)
                self.dgrad_reduce(tm, dr2, blk["conv2"], spp, p + "conv2 dgrad")
                sp = spp
            else:
                G1 = self.A(Mi * cin)
                self.conv(tm, dr2, blk["conv2"], G1, w_kn=True, note=p + "conv2 dgrad")
                G2 = side
        dy = pl.f32(B * 512)
        self.o.add(P.REPEAT_BWD, self.aflag, i=[B, 4, 512, 1], buf=[G1, G2, dy], note=d["prefix"] + "interpolate x4 bwd")
        self.linear_bwd(B, d["lin"], dy, 512, d["din"], 2 * z, ddin, 2 * z, accumulate=accumulate, note=d["prefix"] + "linear")

    # ---- heads: shared pieces --------------------------------------------------
    def concat(self, out, ldo, segs, note):
        """segs: list of (kind, width, ld, src_ref, idx_ref[, table rows])"""
        ii = [self.B, len(segs), ldo, 0]
        bufs = [out]
        rows = []
        for sg in segs:
            kind, w, ld, src, idx = sg[:5]
            ii += [kind, w, ld]
            bufs += [src, idx]
            rows.append(sg[5] if len(sg) > 5 else 0)
        ii += [0] * (16 - len(ii))
        ii += rows + [0] * (4 - len(rows))
        self.o.add(P.CONCAT, 0, i=ii, buf=bufs, note=note)

    def emb_segs(self):
        H = self.cfg.class_hidden_dim
        segs = [(1, H, H, self.semb.ref, self.src, self.cfg.num_sources)]
        segs.append((1, H, H, self.cemb.ref, self.cls, self.cfg.num_classes) if self.with_class else (2, H, 0, None, None))
        return segs

    def emb_bwd(self, dcat, ld, col0):
        H = self.cfg.class_hidden_dim
        self.small_leaf(P.EMB_BWD, self.det_flag, [self.B, H, ld, col0, self.cfg.num_sources], (), [dcat, self.src, self.semb.gref], "source_embedding grad")
        if self.with_class:
            self.small_leaf(P.EMB_BWD, self.det_flag, [self.B, H, ld, col0 + H, self.cfg.num_classes], (), [dcat, self.cls, self.cemb.gref], "class_embedding grad")

    def small_leaf(self, op, flags, i, f, buf, note):
        """A small op of the backward pass whose result nothing else in the pass reads (a weight / bias / embedding-table
        gradient) and whose operands stay untouched to the end of the pass: emitted where it stands, or — with
        group_small_wgrads — deferred to flush_wgrads and run side by side with the other leaves in one launch."""
        if self.train.group_small_wgrads:
            self.pending_small.append((op, flags, list(i), list(f), list(buf), note))
        else:
            self.o.add(op, flags, i=i, f=f, buf=buf, note=note)

    def close_heads(self, first, kind, note):
        """recs[first:] are the heads chain of one pass: flag them as members and close them with an HP_OP_HEADS record (one launch)."""
        n = len(self.o.recs) - first
        for r in self.o.recs[first:]:
            r["flags"] = int(r["flags"]) | P.FLAG_MEMBER
        self.o.add(P.HEADS, 0, i=[first, n, kind], note=note)

    def emit_wfrags(self, part=None):
        """The HP_OP_WFRAG records of the pass that is about to be lowered (self.mode): one per conv weight tensor its convs were seen to
        use on the recording pass, forward (F) and / or HP_CONV_W_KN (G) orientation, as small-leaf groups of at most 64 records — one or
        two launches for all conv weights of a model.  Images are allocated once per lowering and shared by the train and eval passes.
        part: None = every weight; "enc" / "dec" = all but the decoders' / the decoders' (the eval forward fragments the decoders' weights
        where its encoder-only prefix ends, so that the embedding path does not pay for them)."""
        need = (self.wfrag_needed or {}).get(self.mode)
        if not need:
            return
        recs = []
        for key in sorted({k for k, _ in need}):
            if part is not None and key.startswith("decoder") != (part == "dec"):
                continue
            w = self.pl.params[key]
            N, K, T = w.shape                      # reference shape (C_out, C_in, taps); stored tap-major [taps][C_out][C_in] ("tnc"; a 1-tap weight: the same bytes)
            assert w.layout == "tnc" or T == 1, key
            which = (1 if (key, False) in need else 0) | (2 if (key, True) in need else 0)
            for w_kn, bit in ((False, 1), (True, 2)):
                if (which & bit) and (key, w_kn) not in self.wfrag_refs:
                    chunks = T * (K // 16) * (-(-N // 32)) if not w_kn else T * ((-(-N // 32)) * 2) * (K // 32)
                    self.wfrag_refs[(key, w_kn)] = self.pl.ws(chunks * 3072)
            recs.append(([T, N, K, which], [w.ref, self.wfrag_refs.get((key, False)) if which & 1 else None,
                                            self.wfrag_refs.get((key, True)) if which & 2 else None], key))
        for g0 in range(0, len(recs), 64):
            grp = recs[g0: g0 + 64]
            n = len(grp)
            for j, (ii, bufs, key) in enumerate(grp):
                fl = 0
                if n > 1:
                    fl = P.FLAG_MEMBER if j < n - 1 else ((n - 1) << P.FLAG_GROUP_SHIFT)
                self.o.add(P.WFRAG, fl, i=ii, buf=bufs, note="three-term fragments of " + key + (f" [group of {n}]" if n > 1 and j == n - 1 else ""))

    def flush_wgrads(self):
        """Emit the deferred weight-gradient GEMMs as one grouped launch per tap count, then the deferred small leaves as one
        small-leaf group (inside the open segment)."""
        for ntaps in (3, 1):
            mem = [w_ for w_ in self.pending_wgrads if len(w_[0].taps) == ntaps]
            if not mem:
                continue
            first = len(self.o.recs)
            for (tm, nsplit, rps, dy, x, w, note, coef) in mem:
                self.o.add(P.WGRAD_TAPS, 1 | P.FLAG_MEMBER | (P.CONV_IN_BN if coef is not None else 0) | self.mm_flag | self.aflag, i=tm.ints() + [nsplit, rps, w.numel],
                           f=[SLOPE_BACKBONE], buf=[dy, x, w.gref, coef], note=note)
            self.o.add(P.WGRAD_GROUP, 0, i=[first, len(mem), ntaps], note=f"grouped wgrad x{len(mem)} ({ntaps} taps)")
        if self.pending_small:
            n = len(self.pending_small)
            for j, (op, flags, i, f, buf, note) in enumerate(self.pending_small):
                if n > 1:
                    flags |= P.FLAG_MEMBER if j < n - 1 else ((n - 1) << P.FLAG_GROUP_SHIFT)
                    note += " [grouped]" if j < n - 1 else f" [group of {n} small weight gradients]"
                self.o.add(op, flags, i=i, f=f, buf=buf, note=note)
            self.pending_small = []
        self.pending_wgrads = []

    # ---- whole model ---------------------------------------------------------------
    def build(self):
        cfg, pl, B, z, H = self.cfg, self.pl, self.B, self.cfg.z_dim, self.cfg.class_hidden_dim
        multi = cfg.kind == "multimodal"
        # ---------------- parameters ----------------
        if not multi:
            enc = [self.declare_encoder("encoder.")]
            fc0 = self.declare_linear("encoder_fc.0", 2 * z, 2 * z + 2 * H)
            bn_e1 = pl.bn_params("encoder_fc.1", 2 * z)
            fc3 = self.declare_linear("encoder_fc.3", z, 2 * z)
            bn_e4 = pl.bn_params("encoder_fc.4", z)
        else:
            enc = [self.declare_encoder("encoder_mod1."), self.declare_encoder("encoder_mod2.")]
            fc0 = self.declare_linear("fusion_encoder.0", 2 * z, 4 * z + 2 * H)
            bn_e1 = pl.bn_params("fusion_encoder.1", 2 * z)
            fc3 = self.declare_linear("fusion_encoder.3", z, 2 * z)
            bn_e4 = None
        self.semb = pl.param("source_embedding.weight", (cfg.num_sources, H))
        zml = self.declare_zml()
        decs, dfc = [], []
        names = [("decoder_fc", "decoder.", cfg.output_size)] if not multi else [
            ("decoder_fc_mod1", "decoder_mod1.", cfg.output_size), ("decoder_fc_mod2", "decoder_mod2.", cfg.output_size2)]
        for fcname, dpre, osz in names:
            f0 = self.declare_linear(fcname + ".0", 2 * z, z + 2 * H)
            f2 = self.declare_linear(fcname + ".2", 2 * z, 2 * z)
            bn3 = pl.bn_params(fcname + ".3", 2 * z)
            dfc.append(dict(f0=f0, f2=f2, bn3=bn3))
            decs.append(self.declare_decoder(dpre, osz))
        self.cemb = pl.param("class_embedding.weight", (cfg.num_classes, H))    # LAST: skipped by AdamW without class labels
        pl.n_active = pl.n_param_floats if self.with_class else _round_up(self.cemb.offset, 4)
        # the floats between n_active and cemb.offset (alignment gap) are zero padding

        # ---------------- workspace: persistent + I/O ----------------
        pl.stats_cap = 32 << 20
        pl.stats_base = pl.ws(pl.stats_cap).offset
        # AdamW step counter: in the BUF arena, so that engines lowered for other batch sizes share it
        step_off = _round_up(pl._boff, 4)
        pl._boff = step_off + 4
        self.step_ref = Ref(P.BUF, step_off * 4)
        pl.io["adam_step"] = (self.step_ref, (1,), "i8")
        # schedule-free AdamW scalars {lr_max, weight_sum, lr_t, ckp1} (double[4]); always present so that
        # the BUF layout does not depend on the optimiser choice
        sf_off = _round_up(pl._boff, 4)
        pl._boff = sf_off + 8
        self.sf_ref = Ref(P.BUF, sf_off * 4)
        pl.io["sf_state"] = (self.sf_ref, (4,), "f8")
        # batch cursor + noise seed of the staged steps (int64 each); always present, like the scalars above
        cur_off = _round_up(pl._boff, 4)
        pl._boff = cur_off + 4
        cursor, seed = Ref(P.BUF, cur_off * 4), Ref(P.BUF, cur_off * 4 + 8)
        pl.io["cursor"], pl.io["seed"] = (cursor, (1,), "i8"), (seed, (1,), "i8")
        lens = [cfg.output_size] if not multi else [cfg.output_size, cfg.output_size2]
        xs = [pl.f32(B * L, "x" if i == 0 else "x2", (B, 1, L)) for i, L in enumerate(lens)]
        self.src = pl.ws(8 * B, "src", (B,), "i8")
        self.cls = pl.ws(8 * B, "cls", (B,), "i8")
        eps = pl.f32(B * z, "eps", (B, z))
        scal = pl.f32(4, "scalars", (4,))
        self.slab = pl.ws(0)                 # sized at the end (placed last)
        loss = pl.stat(4, replicated=False)
        norm2 = pl.stat(1, replicated=False)
        # ---------------- resident training tables + the staging segment ----------------
        N = int(self.train.resident_units)
        if N > 0:
            W, R = max(1, int(self.train.dp_world)), int(self.train.dp_rank)
            spe = N // (B * W)               # whole batches per epoch and rank (the ragged tail of a permutation is dropped)
            if spe < 1 or not 0 <= R < W:
                raise ValueError(f"resident_units={N}: need at least batch * dp_world = {B * W} units and 0 <= dp_rank < dp_world")
            tabs = [pl.f32(N * L, "data_x" if i == 0 else "data_x2", (N, L)) for i, L in enumerate(lens)]
            dlab = pl.ws(8 * N, "data_labels", (N,), "i8")
            perm = pl.ws(8 * N, "perm", (N,), "i8")
            self.o.begin("stage")
            self.o.add(P.STAGE_BATCH, 0, i=[B, lens[0], lens[1] if multi else 0, z, spe, W, R, N],
                       buf=[tabs[0], tabs[1] if multi else None, dlab, perm, cursor, xs[0], xs[1] if multi else None, self.src, eps, seed],
                       note="gather the batch by index + Philox eps")
            self.o.add(P.STEP_INC, 0, buf=[cursor], note="batch cursor += 1")
            self.o.end()

        segs = {}
        for mode in ("train", "eval"):
            training = mode == "train"
            self.abf = training and self.train.act_dtype == "bf16"      # (stays set through the backward pass emitted below)
            self.count_flops = training
            self.mode = mode
            self.o.begin("fwd_" + mode)
            zero_idx = self.o.add(P.ZERO, 0, i=[0, 0], buf=[Ref(P.WS, pl.stats_base)], note="zero statistics")
            self.emit_wfrags(None if training else "enc")
            marks = [len(self.o.recs)]
            pooled = []
            for e, x, L in zip(enc, xs, lens):
                pooled.append(self.encoder_fwd(e, x, L, training))
                marks.append(len(self.o.recs))
            if multi:
                self.zip_towers(marks[0], marks[1])
            ncat = (2 * z) * len(enc) + 2 * H
            fuse_heads = (training and self.train.fuse_heads and not multi and z in (5, 10) and H == 5 and B <= 512 and
                          self.train.sync_bn_world <= 1 and self.train.group_small_wgrads and
                          P.debug_knob("HIPPIE_NO_FUSE_HEADS") != "1")     # (the knob: A/B runs of unmodified callers)
            c0 = pl.f32(B * ncat)
            hsegs = []
            hs = []
            for e, pz in zip(enc, pooled):
                h = pl.f32(B * 2 * z)
                self.linear_fwd(B, dict(w=e["lin_w"], b=e["lin_b"], N=2 * z, K=512), pz, 512, h, 2 * z, note=e["prefix"] + "linear")
                hs.append(h)
                hsegs.append((0, 2 * z, 2 * z, h, None))
            heads_first = len(self.o.recs)
            self.concat(c0, ncat, hsegs + self.emb_segs(), "cat(enc, source_emb, class_emb)")
            u1 = pl.f32(B * 2 * z)
            st = pl.stat(4 * z) if training else None
            self.linear_fwd(B, fc0, c0, ncat, u1, 2 * z, stats=st, note="encoder_fc.0")
            a1 = pl.f32(B * 2 * z)
            self.bn_apply(B, bn_e1, u1, a1, st, training, True, SLOPE_HEADS, heads=True)
            if not multi:
                u2 = pl.f32(B * z)
                st2 = pl.stat(2 * z) if training else None
                self.linear_fwd(B, fc3, a1, 2 * z, u2, z, stats=st2, note="encoder_fc.3")
                encv = pl.f32(B * z, "enc_" + mode, (B, z))
                self.bn_apply(B, bn_e4, u2, encv, st2, training, True, SLOPE_HEADS, heads=True)
            else:
                u2 = None
                encv = pl.f32(B * z, "enc_" + mode, (B, z))
                self.linear_fwd(B, fc3, a1, 2 * z, encv, z, note="fusion_encoder.3")
            mulv = pl.f32(B * 2 * z, "mulv_" + mode, (B, 2 * z))
            self.linear_fwd(B, zml, encv, z, mulv, 2 * z, note="z_mean | z_log_var")
            if not training:
                # encoder-only prefix of the eval forward: (enc, mu, logvar) are complete here.  The embedding path
                # (scripts/utils.py:75-101) keeps only `enc`, so the decoder (44-59 % of the FLOPs) can be skipped.
                enc_eval_end = len(self.o.recs)
                self.emit_wfrags("dec")
            zz = pl.f32(B * z)
            self.o.add(P.REPARAM_KL_FWD, 0, i=[B, z], buf=[mulv, eps, zz, loss], note="reparameterize + KL")
            ncat1 = z + 2 * H
            c1 = pl.f32(B * ncat1)
            self.concat(c1, ncat1, [(0, z, z, zz, None)] + self.emb_segs(), "cat(z, source_emb, class_emb)")
            heads = []
            recs = []
            marks = [len(self.o.recs)]
            for k, (fc, dd) in enumerate(zip(dfc, decs)):
                fcname = names[k][0]
                u3 = pl.f32(B * 2 * z)
                self.linear_fwd(B, fc["f0"], c1, ncat1, u3, 2 * z, act=True, note=fcname + ".0 + LeakyReLU")
                u4 = pl.f32(B * 2 * z)
                st4 = pl.stat(4 * z) if training else None
                self.linear_fwd(B, fc["f2"], u3, 2 * z, u4, 2 * z, stats=st4, note=fcname + ".2")
                dv = pl.f32(B * 2 * z)
                self.bn_apply(B, fc["bn3"], u4, dv, st4, training, True, SLOPE_HEADS, heads=True)
                if fuse_heads:
                    self.close_heads(heads_first, 0, "heads forward: cat ... decoder_fc (one workgroup)")
                rec = self.decoder_fwd(dd, dv, training)
                pl.io[("rec_" if k == 0 else "rec2_") + mode] = (rec, (B, 1, dd["output_size"]), "f4")
                drec = pl.f32(B * dd["output_size"])
                n = B * dd["output_size"]
                w = self.train.w1 if k == 0 else self.train.w2
                self.o.add(P.MSE_FWD_BWD, 0, i=[n, 1 + k], f=[w if multi else 1.0], buf=[xs[k], rec, drec, loss], note="mse")
                heads.append(dict(u3=u3, u4=u4, dv=dv, drec=drec))
                recs.append(rec)
                marks.append(len(self.o.recs))
            if multi:
                self.zip_towers(marks[0], marks[1])
            n1 = B * lens[0]
            n2 = B * lens[1] if multi else 0
            self.o.add(P.LOSS_FINALIZE, 0, i=[B, n1, n2], f=[self.train.beta, self.train.w1 if multi else 1.0,
                                                             self.train.w2 if multi else 0.0], buf=[loss, scal], note="loss scalars")
            self.o.end()
            self.count_flops = False
            if not training:
                segs["eval_zero"] = zero_idx
                f_eval = self.o.segments["fwd_eval"][0]
                self.o.segments["enc_eval"] = (f_eval, enc_eval_end - f_eval)      # alias: prefix of fwd_eval
                continue
            segs["train_zero"] = zero_idx

            # ---------------- backward ----------------
            # decoder side first, then heads and encoder(s); the weight-gradient GEMMs and the small leaf reductions are deferred
            # to the end of the segment (flush_wgrads): one grouped launch each.  (Round 2 could split the pass in two halves and run
            # the decoder-side group on a second stream under the encoder-side chain — measured slower, removed: DESIGN.md.)
            self.o.begin("bwd")
            nb = pl.n_param_floats * 4
            self.o.add(P.ZERO, 0, i=[nb & 0xFFFFFFFF, nb >> 32], buf=[Ref(P.GRAD, 0)], note="zero gradients")
            dc1 = pl.f32(B * ncat1)
            marks = [len(self.o.recs)]
            for k, (fc, dd, hd) in enumerate(zip(dfc, decs, heads)):
                ddv = pl.f32(B * 2 * z)
                self.decoder_bwd(dd, hd["drec"], ddv)
                _, du4, _ = self.bn_bwd(B, fc["bn3"], ddv, None, hd["dv"], hd["u4"], SLOPE_HEADS, heads=True)
                du3 = pl.f32(B * 2 * z)
                self.linear_bwd(B, fc["f2"], du4, 2 * z, hd["u3"], 2 * z, du3, 2 * z, mask=hd["u3"], ldmask=2 * z, note="decoder_fc.2")
                self.linear_bwd(B, fc["f0"], du3, 2 * z, c1, ncat1, dc1, ncat1, accumulate=(k > 0), note="decoder_fc.0")
                marks.append(len(self.o.recs))
            if multi:
                self.zip_towers(marks[0], marks[1])
            bwd_split = None
            if self.train.bucketed_bwd:
                # everything decoder-side is done: its weight-gradient group and leaf group run HERE, so that the gradient range
                # [decoder_fc*.0.weight, class_embedding) is complete at the end of "bwd_dec"
                self.flush_wgrads()
                bwd_split = len(self.o.recs)
            self.emb_bwd(dc1, ncat1, z)
            dmulv = pl.f32(B * 2 * z)
            self.o.add(P.REPARAM_KL_BWD, 0, i=[B, z, ncat1], f=[self.train.beta], buf=[mulv, eps, dc1, dmulv], note="reparameterize + KL bwd")
            denc = pl.f32(B * z)
            self.linear_bwd(B, zml, dmulv, 2 * z, encv, z, denc, z, note="z_mean | z_log_var")
            if not multi:
                _, du2, _ = self.bn_bwd(B, bn_e4, denc, None, encv, u2, SLOPE_HEADS, heads=True)
            else:
                du2 = denc
            da1 = pl.f32(B * 2 * z)
            self.linear_bwd(B, fc3, du2, z, a1, 2 * z, da1, 2 * z, note="encoder_fc.3")
            _, du1, _ = self.bn_bwd(B, bn_e1, da1, None, a1, u1, SLOPE_HEADS, heads=True)
            dc0 = pl.f32(B * ncat)
            self.linear_bwd(B, fc0, du1, 2 * z, c0, ncat, dc0, ncat, note="encoder_fc.0")
            self.emb_bwd(dc0, ncat, 2 * z * len(enc))
            marks = [len(self.o.recs)]
            for k, e in enumerate(enc):
                dpooled = pl.f32(B * 512)
                lin = dict(w=e["lin_w"], b=e["lin_b"], N=2 * z, K=512)
                dh = dc0 + 4 * (2 * z * k)      # column window of dc0, leading dimension ncat
                self.linear_bwd(B, lin, dh, ncat, e["pooled"], 512, dpooled, 512, note=e["prefix"] + "linear")
                self.encoder_bwd(e, dpooled)
                marks.append(len(self.o.recs))
            if multi:
                self.zip_towers(marks[0], marks[1])
            self.flush_wgrads()
            self.o.end()
            if bwd_split is not None:
                b0_, bc_ = self.o.segments["bwd"]
                self.o.segments["bwd_dec"] = (b0_, bwd_split - b0_)
                self.o.segments["bwd_enc"] = (bwd_split, b0_ + bc_ - bwd_split)
                dec_off, cemb_off = dfc[0]["f0"]["w"].offset, self.cemb.offset
                assert all(q.offset >= dec_off for q in pl.params.values() if q.key.startswith("decoder")) and \
                    all(q.offset < dec_off or q.offset >= cemb_off for q in pl.params.values() if not q.key.startswith("decoder"))
                # (float ranges of the gradient arena, per half; the class-embedding table — last in the arena, fed by both halves'
                #  concatenations — belongs to the second)
                pl.grad_buckets = [[(dec_off, cemb_off)], [(0, dec_off)] + ([(cemb_off, pl.n_active)] if pl.n_active > cemb_off else [])]

            # ---------------- optimiser ----------------
            self.o.begin("opt")
            n = pl.n_active
            t = self.train
            if t.optimizer == "adamw":
                self.o.add(P.STEP_INC, 0, buf=[self.step_ref], note="adam step += 1")
            if self.train.clip > 0:
                # the accumulator is zeroed HERE (not only by the forward's statistics memset): optimizer.step() may
                # run more than once per forward (a closure, a second step()) and must not double-count the norm
                self.o.add(P.ZERO, 0, i=[8, 0], buf=[norm2], note="zero the gradient-norm accumulator")
                self.o.add(P.GRADNORM, 0, i=[n], buf=[Ref(P.GRAD, 0), norm2], note="clip_grad_norm: total norm")
            arenas = [Ref(P.PARAM, 0), Ref(P.GRAD, 0), Ref(P.ADAM_M, 0), Ref(P.ADAM_V, 0)]
            if t.optimizer == "adamw":
                self.o.add(P.ADAMW, 0, i=[n], f=[t.lr, t.beta1, t.beta2, t.adam_eps, t.weight_decay, t.clip, 1.0 - t.beta1, 1.0 - t.beta2],
                           buf=arenas + [self.step_ref, norm2], note="AdamW")
            elif t.optimizer == "schedulefree":
                # z lives in the ADAM_M arena, exp_avg_sq in ADAM_V; k (= STEP) is incremented after the update
                self.o.add(P.SF_SCHEDULE, 0, i=[t.warmup_steps], f=[t.lr, 1.0 - t.beta2, t.sf_r, t.sf_weight_lr_power],
                           buf=[self.step_ref, self.sf_ref], note="schedule-free: lr_t, ckp1")
                self.o.add(P.ADAMW_SF, 0, i=[n], f=[t.beta1, t.beta2, t.adam_eps, t.weight_decay, t.clip, 1.0 - t.beta2],
                           buf=arenas + [self.step_ref, self.sf_ref, norm2], note="AdamWScheduleFree")
                self.o.add(P.STEP_INC, 0, buf=[self.step_ref], note="k += 1")
            else:
                raise ValueError(f"unknown optimizer {t.optimizer!r}")
            self.o.end()
            # "step" names fwd_train + bwd + opt when they are one contiguous range: a single-process step (no gradient
            # all-reduce between bwd and opt) replays ONE graph instead of three (15 us per model-step)
            f0, fc = self.o.segments["fwd_train"]
            b0, bc = self.o.segments["bwd"]
            o0, oc = self.o.segments["opt"]
            if f0 + fc == b0 and b0 + bc == o0:
                self.o.segments["step"] = (f0, fc + bc + oc)
                if "stage" in self.o.segments and sum(self.o.segments["stage"]) == f0:
                    # gather + eps + forward + backward + optimiser: one graph per optimisation step, nothing of the host in it
                    self.o.segments["step_staged"] = (self.o.segments["stage"][0], self.o.segments["stage"][1] + fc + bc + oc)
                    # ... and the data-parallel form (an all-reduce sits between bwd and opt): gather + eps + forward as one graph
                    self.o.segments["fwd_train_staged"] = (self.o.segments["stage"][0], self.o.segments["stage"][1] + fc)
            if t.optimizer == "schedulefree":
                # AdamWScheduleFree.eval() / .train(): y <-> x swaps (hippie/optimizers.py:82-103)
                for seg, w in (("sf_eval", 1.0 - 1.0 / t.beta1), ("sf_train", 1.0 - t.beta1)):
                    self.o.begin(seg)
                    self.o.add(P.LERP, 0, i=[n], f=[w], buf=[Ref(P.PARAM, 0), Ref(P.ADAM_M, 0)], note=seg)
                    self.o.end()
        # finalize: statistics region size into both ZERO ops; slab at the end of the workspace
        used = _round_up(pl.stats_bytes, 256)
        for key in ("train_zero", "eval_zero"):
            self.o.recs[segs[key]]["i"][0] = used
            self.o.recs[segs[key]]["i"][1] = 0
        slab = pl.ws(4 * max(pl.slab_need, 4))
        for r in self.o.recs:
            if int(r["op"]) == P.WGRAD_TAPS and not (int(r["flags"]) & 1):
                r["buf"][2] = slab.encode()
            elif int(r["op"]) == P.SLAB_REDUCE:
                r["buf"][0] = slab.encode()
        if self.train.reuse_workspace and not P.debug_knob("HIPPIE_NO_WS_REUSE"):     # (the knob: A/B runs of unmodified callers)
            pack_workspace(pl)
        return pl


def pack_workspace(pl):
    """Liveness-based reuse of the workspace arena (arena colouring over op intervals).

    Two classes of allocations are packed, each into a region of its own: those that only records of the backward pass
    touch, and those that only records of the eval forward touch.  Inside a class an allocation is live from the first
    to the last record that names it — counted at the position where the record EXECUTES: members of a WGRAD_GROUP /
    PAIR / small-leaf group launch run at their group record, so the operands of the grouped weight-gradient GEMMs stay live to
    the end of the backward pass — and two allocations share memory only when one is dead strictly before the other is
    born.  Everything else (training-forward tensors, which the backward pass reads; named I/O slots; the statistics
    region; the slabs) keeps memory of its own, so training and eval passes may be interleaved in any order.
    Rewrites the records, plan.io and plan.act_sites."""
    import bisect
    recs = pl.ops.recs
    segs = pl.ops.segments
    allocs = sorted(a for a in pl.allocs if a[1] > 0)
    starts = [a[0] for a in allocs]
    mask = (1 << 56) - 1

    def alloc_of(off):
        j = bisect.bisect_right(starts, off) - 1
        assert j >= 0 and off < allocs[j][0] + allocs[j][1], f"workspace ref {off} outside every allocation"
        return j

    # where each record executes
    at = list(range(len(recs)))
    for g, r in enumerate(recs):
        op = int(r["op"])
        if op in (P.WGRAD_GROUP, P.HEADS):
            for k in range(int(r["i"][0]), int(r["i"][0]) + int(r["i"][1])):
                at[k] = g
        elif op == P.PAIR:
            at[int(r["i"][0])] = at[int(r["i"][1])] = g
        n_group = (int(r["flags"]) >> P.FLAG_GROUP_SHIFT) & P.FLAG_GROUP_MASK
        for k in range(g - n_group, g):
            at[k] = g

    def seg_of(k):
        for name in ("fwd_train", "bwd", "opt", "fwd_eval"):
            if name in segs and segs[name][0] <= k < segs[name][0] + segs[name][1]:
                return name
        return "other"

    users = [set() for _ in allocs]
    span = [[None, None] for _ in allocs]
    for k, r in enumerate(recs):
        sname = seg_of(k)
        for b in r["buf"]:
            b = int(b)
            if b == P.NULL or (b >> 56) != P.WS:
                continue
            j = alloc_of(b & mask)
            users[j].add(sname)
            span[j][0] = at[k] if span[j][0] is None else min(span[j][0], at[k])
            span[j][1] = at[k] if span[j][1] is None else max(span[j][1], at[k])
    pinned = [a[2] for a in allocs]
    for ref, _, _ in pl.io.values():        # named slots are read by the host after the pass
        if ref.space == P.WS:
            pinned[alloc_of(ref.offset)] = True
    for site in pl.act_sites:
        for key in ("raw", "out", "coef"):
            ref = site.get(key)
            if isinstance(ref, Ref) and ref.space == P.WS:
                pinned[alloc_of(ref.offset)] = True
    classes = {"bwd": [], "fwd_eval": []}
    for j in range(len(allocs)):
        if pinned[j] or len(users[j]) != 1:
            continue
        (u,) = users[j]
        if u in ("fwd_eval", "bwd"):
            classes[u].append(j)

    new_off = {}
    top = 0
    packed = {j for js in classes.values() for j in js}
    for j, a in enumerate(allocs):          # everything else: one after the other, allocation order
        if j in packed:
            continue
        top = _round_up(top, ALIGN)
        new_off[j] = top
        top += a[1]
    for js in classes.values():
        base = _round_up(top, ALIGN)
        live = []                           # (last record, offset, size) of allocations placed so far that may still be live
        hi = 0
        for j in sorted(js, key=lambda j: (span[j][0], -allocs[j][1])):
            live = [x for x in live if x[0] >= span[j][0]]
            size = _round_up(allocs[j][1], ALIGN)
            off = 0
            for _, o, sz in sorted(live, key=lambda x: x[1]):     # first fit over the address-ordered live set
                if off + size <= o:
                    break
                off = max(off, o + sz)
            new_off[j] = base + off
            live.append((span[j][1], off, size))
            hi = max(hi, off + size)
        top = base + hi

    def remap(off):
        j = alloc_of(off)
        return new_off[j] + (off - allocs[j][0])

    for r in recs:
        for q in range(P.NB):
            b = int(r["buf"][q])
            if b != P.NULL and (b >> 56) == P.WS:
                r["buf"][q] = (P.WS << 56) | remap(b & mask)
    pl.io = {k: ((Ref(P.WS, remap(ref.offset)) if ref.space == P.WS else ref), shp, dt) for k, (ref, shp, dt) in pl.io.items()}
    for site in pl.act_sites:
        for key in ("raw", "out", "coef"):
            ref = site.get(key)
            if isinstance(ref, Ref) and ref.space == P.WS:
                site[key] = Ref(P.WS, remap(ref.offset))
    pl.stats_base = remap(pl.stats_base)
    pl.ws_unpacked = pl.ws_bytes
    pl.allocs = [[new_off[j], a[1], a[2]] for j, a in enumerate(allocs)]
    pl._ws = top


def lower_backbone(spec: BackboneCfg, batch: int, train: TrainCfg = None) -> Plan:
    """Forward programs ("fwd_train", "fwd_eval") of one stand-alone backbone module.  I/O slots: "x" (channels-last
    [B, L, Cin]; ResNet18Enc [B, 1, L]; ResNet18Dec [B, 2z]) and "out_train" / "out_eval" (channels-last [B, Lout, Cout];
    ResNet18Enc [B, 2z]; ResNet18Dec [B, 1, output_size]).  Parameter keys are the module's own state_dict keys."""
    kind, L, B = spec.kind, spec.length, batch
    lw = Lowering(ModelCfg(kind="unimodal", z_dim=spec.z_dim, output_size=spec.output_size), batch, train)
    pl, z = lw.pl, spec.z_dim
    pl.cfg = spec
    cin, s_ = spec.in_channels, spec.stride
    if kind in ("BasicBlockEnc", "BasicBlockDec", "ResizeConv1d"):
        if cin % 32 != 0:
            raise ValueError(f"{kind}: in_channels must be a multiple of 32 (the MFMA K-step); got {cin}")
        if s_ not in (1, 2):
            raise ValueError(f"{kind}: stride / scale_factor must be 1 or 2; got {s_}")
    if kind == "ResNet18Enc":
        mod = lw.declare_encoder("")
        xshape, xn = (B, 1, L), B * L
    elif kind == "ResNet18Dec":
        mod = lw.declare_decoder("", spec.output_size)
        xshape, xn = (B, 2 * z), B * 2 * z
    elif kind == "BasicBlockEnc":
        mod = lw.declare_enc_block("", cin, s_)
        xshape, xn = (B, L, cin), B * L * cin
    elif kind == "BasicBlockDec":
        if cin % (32 * s_) != 0:
            raise ValueError(f"BasicBlockDec: in_planes / stride must be a multiple of 32; got {cin} / {s_}")
        mod = lw.declare_dec_block("", cin, s_)
        xshape, xn = (B, L, cin), B * L * cin
    elif kind == "ResizeConv1d":
        if spec.out_channels % 4 != 0:
            raise ValueError(f"ResizeConv1d: out_channels must be a multiple of 4; got {spec.out_channels}")
        mod = dict(w=pl.param("conv.weight", (spec.out_channels, cin, 3), "tnc"), b=pl.param("conv.bias", (spec.out_channels,)))
        xshape, xn = (B, L, cin), B * L * cin
    else:
        raise ValueError(f"unknown backbone module {kind!r}")
    pl.n_active = pl.n_param_floats
    pl.stats_cap = 32 << 20
    pl.stats_base = pl.ws(pl.stats_cap).offset
    x = pl.f32(xn, "x", xshape)
    lw.slab = pl.ws(0)
    zeros = []
    for mode in ("train", "eval"):
        training = mode == "train"
        lw.o.begin("fwd_" + mode)
        zeros.append(lw.o.add(P.ZERO, 0, i=[0, 0], buf=[Ref(P.WS, pl.stats_base)], note="zero statistics"))
        if kind == "ResNet18Enc":
            pooled = lw.encoder_fwd(mod, x, L, training)
            out = pl.f32(B * 2 * z, "out_" + mode, (B, 2 * z))
            lw.linear_fwd(B, dict(w=mod["lin_w"], b=mod["lin_b"], N=2 * z, K=512), pooled, 512, out, 2 * z, note="linear")
        elif kind == "ResNet18Dec":
            rec = lw.decoder_fwd(mod, x, training)
            pl.io["out_" + mode] = (rec, (B, 1, spec.output_size), "f4")
        elif kind in ("BasicBlockEnc", "BasicBlockDec"):
            out, Lo = (lw.enc_block_fwd if kind == "BasicBlockEnc" else lw.dec_block_fwd)(mod, x, L, training)
            pl.io["out_" + mode] = (out, (B, Lo, mod["cout"]), "f4")
        else:
            tm = lw.map_fwd_up(L, cin, spec.out_channels)[0] if s_ == 2 else lw.map_fwd(L, cin, spec.out_channels, 1)[0]
            out = pl.f32(tm.M * tm.N, "out_" + mode, (B, tm.Lout, tm.N))
            lw.conv(tm, x, mod["w"], out, bias=mod["b"], note="conv")
        lw.o.end()
    used = _round_up(max(pl.stats_bytes, 16), 256)
    for k in zeros:
        lw.o.recs[k]["i"][0] = used
    pl.ws(16)          # (the slab slot of a full model: nothing here needs one)
    if lw.train.reuse_workspace and not P.debug_knob("HIPPIE_NO_WS_REUSE"):
        pack_workspace(pl)
    return pl


def lower(cfg, batch: int, train: TrainCfg = None, with_class=False, **kw) -> Plan:
    if isinstance(cfg, BackboneCfg):
        return lower_backbone(cfg, batch, train)
    first = Lowering(cfg, batch, train, with_class, **kw)
    plan = first.build()
    if not any(first.wfrag_used.values()):
        return plan
    # (second pass: the same lowering with the HP_OP_WFRAG records at the head of each forward segment and HP_CONV_WFRAG on the convs they serve)
    return Lowering(cfg, batch, train, with_class, wfrag_needed=first.wfrag_used, **kw).build()
