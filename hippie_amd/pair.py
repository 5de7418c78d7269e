"""Joint program for two independent models trained in lockstep (the reference's default pipeline trains a
waveform cVAE and a spike-timing cVAE on the same units, scripts/train_model_with_multimodal.py:200-224).

At batch 512 one layer of one model only gives each CU a single workgroup.  The two models have the same
op sequence (only L differs), so their programs are zipped: every heavy op (implicit-GEMM convolution,
BatchNorm passes) of model A is launched together with the same op of model B (HP_OP_PAIR), and all
weight-gradient GEMMs of both backward passes go into one grouped launch.  Arithmetic per model is
unchanged; each model keeps its own arenas slice, optimiser settings and state_dict."""
from __future__ import annotations

import numpy as np
import torch

from . import planner, program as P
from .engine import Engine
from .program import DeviceProgram

PAIRABLE = {P.CONV_TAPS, P.BN_APPLY, P.BN_BWD_REDUCE, P.BN_BWD_APPLY}
_MASK = (1 << 56) - 1


def _up(x, a=256):
    return (x + a - 1) // a * a


def arena_sizes(plan):
    n = plan.n_param_floats * 4
    return [_up(plan.ws_bytes), _up(n), _up(n), _up(plan.n_buf_floats * 4), _up(n), _up(n)]


def _shift(rec, bases):
    r = rec.copy()
    for k in range(P.NB):
        ref = int(r["buf"][k])
        if ref == P.NULL:
            continue
        sp, off = ref >> 56, ref & _MASK
        r["buf"][k] = (sp << 56) | (off + bases[sp])
    return r


def _pair_ok(a, b):
    op = int(a["op"])
    if op != int(b["op"]) or op not in PAIRABLE:
        return False
    if (int(a["flags"]) | int(b["flags"])) & P.FLAG_MEMBER:
        return False
    if op == P.CONV_TAPS:
        return (int(a["flags"]) & 1) == (int(b["flags"]) & 1)
    return (int(a["i"][1]) % 4 == 0) == (int(b["i"][1]) % 4 == 0)


def zip_programs(plan_a, plan_b):
    """-> (joint op array, segments, notes, B's arena base offsets in bytes)"""
    ops_a, ops_b = plan_a.ops.array(), plan_b.ops.array()
    bases_b = arena_sizes(plan_a)
    out, notes, segments = [], [], {}
    new_a, new_b = {}, {}
    for seg, (fa, ca) in plan_a.ops.segments.items():
        if seg in ("bwd", "enc_eval", "step"):
            continue                      # aliases (bwd = bwd_a + wg_a + bwd_b + wg_b; enc_eval = prefix of fwd_eval)
        fb, cb = plan_b.ops.segments[seg]
        start = len(out)
        same = ca == cb and all(int(ops_a[fa + k]["op"]) == int(ops_b[fb + k]["op"]) for k in range(ca))
        if not same:
            # e.g. the optimiser segment when only one model clips gradients: run A's ops, then B's
            for k in range(ca):
                new_a[fa + k] = len(out)
                out.append(ops_a[fa + k].copy())
                notes.append("A:" + plan_a.ops.notes[fa + k])
            for k in range(cb):
                new_b[fb + k] = len(out)
                out.append(_shift(ops_b[fb + k], bases_b))
                notes.append("B:" + plan_b.ops.notes[fb + k])
            segments[seg] = (start, len(out) - start)
            continue
        for k in range(ca):
            ra, rb = ops_a[fa + k].copy(), _shift(ops_b[fb + k], bases_b)
            opc = int(ra["op"])
            if opc != int(rb["op"]):
                raise ValueError(f"segment {seg} op {k}: opcode mismatch {opc} vs {int(rb['op'])}")
            na, nb = plan_a.ops.notes[fa + k], plan_b.ops.notes[fb + k]
            if opc == P.WGRAD_GROUP:
                first = new_a[int(ra["i"][0])]
                assert new_b[int(rb["i"][0])] == first + 1, "group members must interleave contiguously"
                g = ra.copy()
                g["i"][0], g["i"][1] = first, int(ra["i"][1]) + int(rb["i"][1])
                out.append(g)
                notes.append(na + " (both models)")
                continue
            if opc == P.PAIR:
                raise ValueError("nested PAIR ops are not supported")
            new_a[fa + k], new_b[fb + k] = len(out), len(out) + 1
            if _pair_ok(ra, rb):
                ra["flags"] = int(ra["flags"]) | P.FLAG_MEMBER
                rb["flags"] = int(rb["flags"]) | P.FLAG_MEMBER
                out += [ra, rb]
                notes += ["A:" + na, "B:" + nb]
                pr = np.zeros((), dtype=P.OP_DTYPE)
                pr["op"] = P.PAIR
                pr["i"][0], pr["i"][1] = len(out) - 2, len(out) - 1
                pr["buf"][:] = P.NULL
                out.append(pr)
                notes.append("pair " + na)
            else:
                out += [ra, rb]
                notes += ["A:" + na, "B:" + nb]
        segments[seg] = (start, len(out) - start)
    first = segments["bwd_a"][0]
    segments["bwd"] = (first, segments["wg_b"][0] + segments["wg_b"][1] - first)
    return np.array(out, dtype=P.OP_DTYPE), segments, notes, bases_b


class PairEngine:
    """Two models, one program, one stream.  `models[k]` is an Engine view (state_dict, io, scalars ...)."""

    def __init__(self, cfg_a, cfg_b, batch, train_a=None, train_b=None, device=None):
        if not torch.cuda.is_available():
            raise P.HipEngineError("hippie_amd.PairEngine needs an MI355X; no CPU fallback")
        P.load_library()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.B = batch
        import dataclasses
        # the zipped program pairs A with B; pairs inside one model would nest
        # (and no chained launches: the zip interleaves the two models' records, a chain needs consecutive ones)
        ta = dataclasses.replace(train_a or planner.TrainCfg(), intra_pair=False, chain_small=False, group_small_wgrads=False, fuse_heads=False)
        tb = dataclasses.replace(train_b or planner.TrainCfg(), intra_pair=False, chain_small=False, group_small_wgrads=False, fuse_heads=False)
        plans = [planner.lower(cfg_a, batch, ta), planner.lower(cfg_b, batch, tb)]
        self.ops, self.segments, self.notes, bases_b = zip_programs(*plans)
        sa, sb = arena_sizes(plans[0]), arena_sizes(plans[1])
        with torch.cuda.device(self.device):
            self.arenas = [torch.zeros(sa[k] + sb[k], dtype=torch.uint8, device=self.device) for k in range(P.NUM_SPACES)]
        self.models = []
        for plan, base, size, cfg, tc in ((plans[0], [0] * 6, sa, cfg_a, train_a), (plans[1], bases_b, sb, cfg_b, train_b)):
            sl = [self.arenas[k][base[k]: base[k] + size[k]] for k in range(P.NUM_SPACES)]
            nfl = plan.n_param_floats
            view = (plan, (sl[P.WS], sl[P.PARAM].view(torch.float32)[:nfl], sl[P.GRAD].view(torch.float32)[:nfl],
                           sl[P.BUF].view(torch.float32)[: plan.n_buf_floats], sl[P.ADAM_M].view(torch.float32)[:nfl],
                           sl[P.ADAM_V].view(torch.float32)[:nfl]))
            self.models.append(Engine(cfg, batch, tc, device=self.device, _view=view))
        self.grads = self.arenas[P.GRAD].view(torch.float32)      # both models: one all-reduce buffer
        self.prog = DeviceProgram(self.ops, [a.data_ptr() for a in self.arenas], [a.numel() for a in self.arenas])
        self._graphs = {}

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def run(self, seg, use_graph=True):
        first, count = self.segments[seg]
        if use_graph:
            g = self._graphs.get(seg)
            if g is None:
                g = self._graphs[seg] = self.prog.capture(first, count)
            self.prog.replay(g, self._stream())
        else:
            self.prog.run(first, count, self._stream())

    def forward(self, training=True, use_graph=True):
        self.run("fwd_train" if training else "fwd_eval", use_graph)
        if training:
            for m in self.models:
                for p in m.num_batches_tracked:
                    m.num_batches_tracked[p] += 1

    def backward(self, use_graph=True):
        self.run("bwd", use_graph)

    def optimizer_step(self, use_graph=True):
        self.run("opt", use_graph)

    def train_step(self, use_graph=True):
        self.forward(True, use_graph)
        self.backward(use_graph)
        self.optimizer_step(use_graph)

    def profile(self, seg):
        first, count = self.segments[seg]
        return self.prog.profile(first, count, self._stream())
