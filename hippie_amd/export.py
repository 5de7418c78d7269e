"""Serialised lowered models (".hpm"): the planner's output as ONE file a host without Python can load.

    python -m hippie_amd.export --kind unimodal --z-dim 10 --output-size 50 --batch 512 --lr 1e-3 -o wave_b512.hpm

The file holds the HpOp records, the named segments ("fwd_train", "bwd", "opt", "step", "fwd_eval", "enc_eval"), the
six arena sizes, a table of every parameter / BatchNorm buffer (reference state_dict key, shape, offset, layout) and of
every named I/O slot ("x", "src", "cls", "eps", "scalars", "enc_train", ...), and optionally initial parameter and
buffer values.  libhippie_hip.so loads it with hp_model_load (include/hippie_hip.h, hippie_amd/csrc/model.hip), allocates
the arenas itself and exposes the reference's verbs: hp_model_forward / backward / optimizer_step / train_step.  What
the file replaces for such a host is planner.py — i.e. constructing the reference's nn.Module graph
(hippie/model.py:13-44,352-395; hippie/backbones.py:74-126) — not any arithmetic.
"""
from __future__ import annotations

import argparse
import struct

import numpy as np

from . import planner, program as P

MAGIC = b"HPMODEL\0"
VERSION = 1
NO_DEVICE = 1            # HP_MODEL_NO_DEVICE
SEGMENT_DT = np.dtype([("name", "S32"), ("first", "<i4"), ("count", "<i4")])
TENSOR_DT = np.dtype([("name", "S112"), ("space", "<i4"), ("layout", "<i4"), ("offset_bytes", "<i8"), ("numel", "<i8"),
                      ("ndim", "<i4"), ("shape", "<i4", (4,)), ("dtype", "<i4")], align=True)
assert SEGMENT_DT.itemsize == 40 and TENSOR_DT.itemsize == 160
_DT_CODE = {"f4": 0, "i8": 1, "f8": 2}


def arena_sizes(plan):
    n = plan.n_param_floats * 4
    return [plan.ws_bytes, n, n, plan.n_buf_floats * 4, n, n]


def _tensor(name, space, layout, offset_bytes, shape, dtype):
    t = np.zeros((), dtype=TENSOR_DT)
    t["name"] = name.encode()
    t["space"], t["layout"], t["offset_bytes"] = space, layout, offset_bytes
    t["numel"] = int(np.prod(shape))
    shp = list(shape)[:4]
    t["ndim"] = len(shp)
    t["shape"][: len(shp)] = shp
    t["dtype"] = _DT_CODE[dtype]
    return t


def tables(plan):
    """(params, bufs, io) structured arrays of a Plan."""
    params = [_tensor(k, P.PARAM, 1 if i.layout == "tnc" else 0, i.offset * 4, i.shape, "f4") for k, i in plan.params.items()]
    bufs = [_tensor(k, P.BUF, 0, i.offset * 4, (i.numel,), "f4") for k, i in plan.bufs.items()]
    io = [_tensor(k, ref.space, 0, ref.offset, shape, dt) for k, (ref, shape, dt) in plan.io.items()]
    return [np.array(v, dtype=TENSOR_DT) for v in (params, bufs, io)]


def save_model(plan, path, param_values=None, buf_values=None):
    """Write `plan` to `path`.  param_values / buf_values: float32 arrays in ARENA layout (engine.params / engine.bufs, or
    hippie_amd.export.arena_values(plan, state_dict)); omitted = the loader leaves the arenas zeroed."""
    ops = plan.ops.array()
    segs = np.array([(k.encode(), f, c) for k, (f, c) in plan.ops.segments.items()], dtype=SEGMENT_DT)
    params, bufs, io = tables(plan)
    sizes = arena_sizes(plan)
    cfg = plan.cfg
    config = [1 if cfg.kind == "multimodal" else 0, cfg.z_dim, cfg.output_size, cfg.output_size2, cfg.class_hidden_dim, cfg.num_sources,
              cfg.num_classes, plan.B, 1 if plan.with_class else 0, plan.n_active,
              plan.train.resident_units, plan.train.dp_world, plan.train.dp_rank] + [0] * 3
    has_init = (1 if param_values is not None else 0) | (2 if buf_values is not None else 0)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<8i", VERSION, P.ABI_VERSION, len(ops), len(segs), len(params), len(bufs), len(io), has_init))
        f.write(struct.pack("<6q", *sizes))
        f.write(struct.pack("<16i", *config))
        f.write(ops.tobytes())
        f.write(segs.tobytes())
        for t in (params, bufs, io):
            f.write(t.tobytes())
        if param_values is not None:
            v = np.ascontiguousarray(param_values, dtype=np.float32).reshape(-1)
            assert v.size * 4 == sizes[P.PARAM], (v.size, sizes[P.PARAM])
            f.write(v.tobytes())
        if buf_values is not None:
            v = np.ascontiguousarray(buf_values, dtype=np.float32).reshape(-1)
            assert v.size * 4 == sizes[P.BUF], (v.size, sizes[P.BUF])
            f.write(v.tobytes())
    return path


def arena_values(plan, state_dict):
    """(param arena, buffer arena) as float32 numpy arrays in arena layout from a reference-keyed state_dict (torch tensors or
    numpy arrays; missing BatchNorm buffers default to running_mean 0 / running_var 1)."""
    pv = np.zeros(plan.n_param_floats, dtype=np.float32)
    for k, info in plan.params.items():
        v = np.asarray(state_dict[k].detach().cpu().numpy() if hasattr(state_dict[k], "detach") else state_dict[k], dtype=np.float32)
        if info.layout == "tnc":
            v = np.transpose(v, (2, 0, 1))
        pv[info.offset: info.offset + info.numel] = np.ascontiguousarray(v).reshape(-1)
    bv = np.zeros(plan.n_buf_floats, dtype=np.float32)
    for k, info in plan.bufs.items():
        if k in state_dict:
            v = state_dict[k]
            v = v.detach().cpu().numpy() if hasattr(v, "detach") else v
            bv[info.offset: info.offset + info.numel] = np.asarray(v, dtype=np.float32).reshape(-1)
        elif k.endswith("running_var"):
            bv[info.offset: info.offset + info.numel] = 1.0
    return pv, bv


def read_model(path):
    """Pure-Python reader (tests, tooling): dict(header fields, ops, segments, params, bufs, io, param_values, buf_values)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != MAGIC:
        raise ValueError("not an .hpm file")
    version, abi, n_ops, n_seg, n_par, n_buf, n_io, has_init = struct.unpack_from("<8i", raw, 8)
    sizes = list(struct.unpack_from("<6q", raw, 40))
    config = list(struct.unpack_from("<16i", raw, 88))
    off = 152
    ops = np.frombuffer(raw, dtype=P.OP_DTYPE, count=n_ops, offset=off)
    off += n_ops * P.OP_DTYPE.itemsize
    segs = np.frombuffer(raw, dtype=SEGMENT_DT, count=n_seg, offset=off)
    off += n_seg * SEGMENT_DT.itemsize
    tabs = []
    for n in (n_par, n_buf, n_io):
        tabs.append(np.frombuffer(raw, dtype=TENSOR_DT, count=n, offset=off))
        off += n * TENSOR_DT.itemsize
    pv = bv = None
    if has_init & 1:
        pv = np.frombuffer(raw, dtype=np.float32, count=sizes[P.PARAM] // 4, offset=off)
        off += sizes[P.PARAM]
    if has_init & 2:
        bv = np.frombuffer(raw, dtype=np.float32, count=sizes[P.BUF] // 4, offset=off)
        off += sizes[P.BUF]
    mv = vv = None
    if has_init & 4:           # AdamW moments (hp_model_save(..., with_optimizer=1))
        mv = np.frombuffer(raw, dtype=np.float32, count=sizes[P.ADAM_M] // 4, offset=off)
        off += sizes[P.ADAM_M]
        vv = np.frombuffer(raw, dtype=np.float32, count=sizes[P.ADAM_V] // 4, offset=off)
        off += sizes[P.ADAM_V]
    assert off == len(raw), (off, len(raw))
    return dict(version=version, abi=abi, arena_bytes=sizes, config=config, ops=ops,
                segments={s["name"].decode(): (int(s["first"]), int(s["count"])) for s in segs},
                params=tabs[0], bufs=tabs[1], io=tabs[2], param_values=pv, buf_values=bv, m_values=mv, v_values=vv)


def _param_tensor(t, arena):
    """one parameter out of an arena image, in the reference's layout (conv weights are stored tap-major)"""
    import torch
    shape = [int(v) for v in t["shape"][: int(t["ndim"])]]
    flat = np.array(arena[int(t["offset_bytes"]) // 4: int(t["offset_bytes"]) // 4 + int(t["numel"])])
    if int(t["layout"]) == 1:
        co, ci, k = shape
        flat = flat.reshape(k, co, ci).transpose(1, 2, 0)
    return torch.from_numpy(np.ascontiguousarray(flat.reshape(shape)))


def checkpoint_from_file(path):
    """A model file written by hp_model_save (a C host's checkpoint) as the reference's checkpoint dict: {"state_dict": parameters,
    BatchNorm running statistics and num_batches_tracked under "model.<reference key>" (what pl.ModelCheckpoint stores for the train
    module, hippie/model.py:76-93), "optimizer_states": [torch.optim.AdamW.state_dict() layout]} — loadable by
    hippieUnimodalEmbeddingModelCVAE.load_state_dict here and in the reference."""
    import torch
    from collections import OrderedDict
    d = read_model(path)
    if d["param_values"] is None or d["buf_values"] is None:
        raise ValueError(f"{path} holds no parameter / buffer values (export --seed, or hp_model_save)")
    sd = OrderedDict()
    for t in d["params"]:
        sd["model." + t["name"].decode()] = _param_tensor(t, d["param_values"])
    tracked = (int(d["config"][13]) & 0xFFFFFFFF) | ((int(d["config"][14]) & 0xFFFFFFFF) << 32)          # int64: low / high word
    for t in d["bufs"]:
        k = t["name"].decode()
        o = int(t["offset_bytes"]) // 4
        sd["model." + k] = torch.from_numpy(np.array(d["buf_values"][o: o + int(t["numel"])]))
        if k.endswith("running_var"):
            sd["model." + k[: -len("running_var")] + "num_batches_tracked"] = torch.tensor(tracked, dtype=torch.int64)
    out = {"state_dict": sd}
    if d["m_values"] is not None:
        step = 0
        for t in d["io"]:
            if t["name"].decode() == "adam_step":
                o = int(t["offset_bytes"])
                step = int(np.frombuffer(d["buf_values"].tobytes()[o: o + 8], dtype=np.int64)[0])
        adam = next(r for r in d["ops"] if int(r["op"]) == P.ADAMW)
        state = OrderedDict()
        for i, t in enumerate(d["params"]):
            state[i] = dict(step=torch.tensor(float(step)), exp_avg=_param_tensor(t, d["m_values"]), exp_avg_sq=_param_tensor(t, d["v_values"]))
        group = dict(lr=float(adam["f"][0]), betas=(float(adam["f"][1]), float(adam["f"][2])), eps=float(adam["f"][3]), weight_decay=float(adam["f"][4]),
                     amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(state))))
        out["optimizer_states"] = [dict(state=state, param_groups=[group], param_names=[t["name"].decode() for t in d["params"]])]
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--kind", choices=["unimodal", "multimodal"], default="unimodal")
    ap.add_argument("--z-dim", type=int, default=10)
    ap.add_argument("--output-size", type=int, default=50)
    ap.add_argument("--output-size2", type=int, default=100)
    ap.add_argument("--num-sources", type=int, default=5)
    ap.add_argument("--num-classes", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--with-class", action="store_true")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--weight-decay", type=float, default=0.01)
    ap.add_argument("--beta", type=float, default=1.0)
    ap.add_argument("--clip", type=float, default=0.0)
    ap.add_argument("--resident-units", type=int, default=0, help="N > 0: the training tables live in the workspace and the file also holds the "
                    "'stage' / 'step_staged' segments (hp_model_train_step_staged: loader + step as one graph)")
    ap.add_argument("--dp-world", type=int, default=1)
    ap.add_argument("--dp-rank", type=int, default=0)
    ap.add_argument("--bucketed-bwd", action="store_true", help="data parallel: the backward pass as two segments (bwd_dec | bwd_enc) so that "
                    "hp_model_train_step_dp reduces the decoder-side gradient bucket while the encoder-side half runs")
    ap.add_argument("--seed", type=int, default=None, help="also store the reference constructor's random initialisation under this torch seed")
    ap.add_argument("--from-ckpt", default=None, metavar="FILE.ckpt",
                    help="initial parameters / BatchNorm buffers from a reference-format checkpoint (torch.save'd {'state_dict': {'model.<key>': tensor}}, "
                         "as pl.ModelCheckpoint writes for the reference's train modules): weights trained by the reference, served from a C host")
    ap.add_argument("--to-ckpt", nargs=2, metavar=("MODEL.hpm", "OUT.ckpt"), default=None,
                    help="convert a file written by hp_model_save into a reference-format .ckpt (torch.save of checkpoint_from_file) and exit")
    ap.add_argument("-o", "--output", default=None)
    a = ap.parse_args(argv)
    if a.to_ckpt:
        import torch
        ck = checkpoint_from_file(a.to_ckpt[0])
        torch.save(ck, a.to_ckpt[1])
        print(f"{a.to_ckpt[1]}: {len(ck['state_dict'])} state_dict entries" + (", AdamW state" if "optimizer_states" in ck else ""))
        return
    if not a.output:
        ap.error("-o / --output is required")
    cfg = planner.ModelCfg(a.kind, a.z_dim, a.output_size, a.output_size2, 5, a.num_sources, a.num_classes)
    plan = planner.lower(cfg, a.batch, planner.TrainCfg(lr=a.lr, weight_decay=a.weight_decay, beta=a.beta, clip=a.clip, resident_units=a.resident_units,
                                                        dp_world=a.dp_world, dp_rank=a.dp_rank, bucketed_bwd=a.bucketed_bwd), with_class=a.with_class)
    pv = bv = None
    if a.from_ckpt:
        import torch
        ck = torch.load(a.from_ckpt, map_location="cpu", weights_only=False)
        sd = ck.get("state_dict", ck)
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
        missing = [k for k in plan.params if k not in sd]
        if missing:
            raise SystemExit(f"{a.from_ckpt}: no entry for {missing[:3]}{' ...' if len(missing) > 3 else ''}")
        pv, bv = arena_values(plan, sd)
    elif a.seed is not None:
        import torch
        from .model import reference_init_state
        sd = reference_init_state(cfg, torch.Generator().manual_seed(a.seed))
        pv, bv = arena_values(plan, sd)
    save_model(plan, a.output, pv, bv)
    print(f"{a.output}: {len(plan.ops.recs)} ops, {len(plan.params)} parameter tensors, workspace {plan.ws_bytes / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
