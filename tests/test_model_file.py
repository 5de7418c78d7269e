"""Serialised lowered models (.hpm, hippie_amd/export.py) and the hp_model_* level of the C ABI (hippie_amd/csrc/model.hip).

CPU: the file round-trips, hp_model_load(HP_MODEL_NO_DEVICE) parses and validates it without a GPU and reports the same
tables as the planner; damaged files are refused.  GPU: a plain-C host (tests/c_host/host_step.c — no Python, no torch, no HIP
headers) loads the file, takes optimisation steps through hp_model_forward / backward / optimizer_step / train_step, and lands
on the Python engine's and the oracle's numbers."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from hippie_amd import export, planner, program as P
from oracle import cvae_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TensorInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 112), ("space", ctypes.c_int32), ("layout", ctypes.c_int32), ("offset_bytes", ctypes.c_int64),
                ("numel", ctypes.c_int64), ("ndim", ctypes.c_int32), ("shape", ctypes.c_int32 * 4), ("dtype", ctypes.c_int32)]


assert ctypes.sizeof(TensorInfo) == export.TENSOR_DT.itemsize == 160


def _export(tmp_path, kind="unimodal", z=10, L=50, B=16, L2=100, salt=0, clip=0.0, lr=1e-3, bucketed=False):
    cfg = planner.ModelCfg(kind, z, L, L2)
    plan = planner.lower(cfg, B, planner.TrainCfg(lr=lr, clip=clip, bucketed_bwd=bucketed))
    om = O.OracleModel(kind, z, L, output_size2=L2 if kind == "multimodal" else None, salt=salt)
    pv, bv = export.arena_values(plan, {k: v.detach() for k, v in om.state.items()})
    path = str(tmp_path / f"{kind}.hpm")
    export.save_model(plan, path, pv, bv)
    return plan, om, path, pv, bv


@pytest.mark.parametrize("kind", ["unimodal", "multimodal"])
def test_file_round_trip_and_host_only_load(tmp_path, kind):
    plan, om, path, pv, bv = _export(tmp_path, kind, B=6)
    d = export.read_model(path)
    assert d["abi"] == P.ABI_VERSION and d["arena_bytes"] == export.arena_sizes(plan)
    assert np.array_equal(d["ops"], plan.ops.array()) and d["segments"] == dict(plan.ops.segments)
    assert [t["name"].decode() for t in d["params"]] == list(plan.params)
    assert np.array_equal(d["param_values"], pv) and np.array_equal(d["buf_values"], bv)
    lib = P.load_library()
    m = ctypes.c_void_p()
    assert lib.hp_model_load(path.encode(), export.NO_DEVICE, ctypes.byref(m)) == 0, lib.hp_last_error()
    cfg = (ctypes.c_int32 * 16)()
    assert lib.hp_model_config(m, cfg) == 0
    assert list(cfg[:9]) == [1 if kind == "multimodal" else 0, 10, 50, 100, 5, 5, 5, 6, 0] and cfg[9] == plan.n_active
    assert lib.hp_model_tensor_count(m, 0) == len(plan.params) and lib.hp_model_tensor_count(m, 1) == len(plan.bufs)
    assert lib.hp_model_tensor_count(m, 2) == len(plan.io)
    t = TensorInfo()
    for i, (k, info) in enumerate(plan.params.items()):
        assert lib.hp_model_tensor_info(m, 0, i, ctypes.byref(t)) == 0
        assert t.name.decode() == k and t.offset_bytes == info.offset * 4 and t.numel == info.numel
        assert list(t.shape[: t.ndim]) == list(info.shape) and t.layout == (1 if info.layout == "tnc" else 0)
    assert lib.hp_model_find(m, b"scalars", ctypes.byref(t)) == 0 and t.numel == 4 and t.space == P.WS
    assert lib.hp_model_find(m, b"no_such_tensor", ctypes.byref(t)) != 0 and b"no_such_tensor" in lib.hp_last_error()
    first, count = ctypes.c_int(), ctypes.c_int()
    for seg, (f, c) in plan.ops.segments.items():
        assert lib.hp_model_segment(m, seg.encode(), ctypes.byref(first), ctypes.byref(count)) == 0
        assert (first.value, count.value) == (f, c)
    assert lib.hp_model_run(m, b"fwd_train", 0, None) != 0            # host-only load: running is refused, not crashed
    assert lib.hp_model_destroy(m) == 0


def test_damaged_files_are_refused(tmp_path):
    plan, om, path, pv, bv = _export(tmp_path, B=4)
    raw = open(path, "rb").read()
    lib = P.load_library()
    m = ctypes.c_void_p()

    def load(data):
        bad = str(tmp_path / "bad.hpm")
        with open(bad, "wb") as f:
            f.write(data)
        return lib.hp_model_load(bad.encode(), export.NO_DEVICE, ctypes.byref(m))

    assert load(b"NOTMODEL" + raw[8:]) != 0 and b"magic" in lib.hp_last_error()
    assert load(raw[: len(raw) // 2]) != 0 and b"short file" in lib.hp_last_error()
    wrong_abi = bytearray(raw)
    wrong_abi[12:16] = (P.ABI_VERSION + 1).to_bytes(4, "little")
    assert load(bytes(wrong_abi)) != 0 and b"ABI" in lib.hp_last_error()
    # an op whose buffer reference points outside its arena: caught by the same validation as hp_program_create
    d = export.read_model(path)
    ops = d["ops"].copy()
    k = next(i for i, r in enumerate(ops) if int(r["op"]) == P.CONV_TAPS)
    ops[k]["buf"][2] = (P.WS << 56) | (d["arena_bytes"][P.WS] - 64)
    off = 152
    assert load(raw[:off] + ops.tobytes() + raw[off + ops.nbytes:]) != 0 and b"out of range" in lib.hp_last_error()
    assert lib.hp_model_load(str(tmp_path / "missing.hpm").encode(), 0, ctypes.byref(m)) != 0
    # header fields a damaged file could carry: an absurd arena size, a segment range that overflows int32
    huge = bytearray(raw)
    huge[40:48] = (1 << 50).to_bytes(8, "little")
    assert load(bytes(huge)) != 0 and b"arena size" in lib.hp_last_error()
    seg_off = 152 + d["ops"].nbytes
    wild = bytearray(raw)
    wild[seg_off + 32: seg_off + 40] = (2**31 - 1).to_bytes(4, "little") * 2
    assert load(bytes(wild)) != 0 and b"segment out of range" in lib.hp_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,use_graph,dp", [("unimodal", 1, ""), ("unimodal", 0, ""), ("multimodal", 1, ""),
                                               ("unimodal", 1, "dp"), ("multimodal", 1, "dp-bucketed"), ("unimodal", 0, "dp-bucketed")])
def test_plain_c_host_steps_the_model(tmp_path, kind, use_graph, dp):
    """tests/c_host/host_step.c — C99, only include/hippie_hip.h — against the Python engine on the same file and inputs.
    dp: the host draws an ncclUniqueId, creates a 1-rank RCCL communicator (hp_model_allreduce_init) and takes every step with
    hp_model_train_step_dp — one all-reduce after the pass, or ("dp-bucketed": a file exported with bucketed_bwd) the two-half / two-bucket
    form; at one rank the mean over ranks is the identity, so the numbers are the single-process engine's."""
    from hippie_amd.engine import Engine
    z, L, L2, B, lr, clip = 10, 50, 100, 16, 1e-6, 1.0      # (a small lr: Adam turns last-bit gradient differences — atomic sums — into +-lr moves)
    plan, om, path, pv, bv = _export(tmp_path, kind, z, L, B, L2, salt=4, clip=clip, lr=lr, bucketed=dp == "dp-bucketed")
    assert ("bwd_dec" in plan.ops.segments) == (dp == "dp-bucketed")
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=4, name="x1" if kind == "multimodal" else "x")
    parts = [x.numpy().astype(np.float32).tobytes()]
    x2 = None
    if kind == "multimodal":
        x2 = O.synth_inputs(B, L2, z, salt=4, name="x2")[0]
        parts.append(x2.numpy().astype(np.float32).tobytes())
    parts += [src.numpy().astype(np.int64).tobytes(), eps.numpy().astype(np.float32).tobytes()]
    inputs = str(tmp_path / "inputs.bin")
    with open(inputs, "wb") as f:
        f.write(b"".join(parts))
    exe = str(tmp_path / "host_step")
    libdir = os.path.join(ROOT, "hippie_amd")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_host", "host_step.c"), "-o", exe, "-L", libdir, "-lhippie_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    steps = 3
    out = subprocess.run([exe, path, inputs, str(steps), str(use_graph)] + (["dp"] if dp else []), check=True, capture_output=True, text=True, timeout=300).stdout
    lines = out.strip().splitlines()
    assert ("data parallel: rank 0 of 1" in out) == bool(dp)
    got = np.array([[float(v) for v in ln.split()[2:]] for ln in lines if ln.startswith("step ")])
    assert got.shape == (steps, 4), out
    # the Python engine on the same program and inputs
    eng = Engine(planner.ModelCfg(kind, z, L, L2), B, planner.TrainCfg(lr=lr, clip=clip))
    eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
    eng.set_inputs(x.cuda(), src.cuda(), None, eps.cuda(), x2=x2.cuda() if x2 is not None else None)
    want = []
    for _ in range(steps):
        eng.train_step(use_graph=False)
        want.append(eng.scalars())
    np.testing.assert_allclose(got, np.array(want), rtol=1e-4)        # same kernels; only the order of atomic sums differs (amplified step by step: see the pair test below)
    enc = eng.io("enc_train").double()
    tail = dict(ln.split(" sum ") for ln in lines if " sum " in ln)
    s, s2 = (float(v) for v in tail["enc_train"].replace("sumsq ", "").split())
    np.testing.assert_allclose([s, s2], [float(enc.sum()), float((enc * enc).sum())], rtol=1e-5, atol=1e-6)
    key = ("encoder_mod1." if kind == "multimodal" else "encoder.") + "conv1.weight"
    wsum, rest = tail[key].split(" numel ")
    np.testing.assert_allclose(float(wsum), float(eng.state_dict()[key].double().sum()), rtol=1e-5, atol=1e-6)
    assert rest.split() == ["192", "batches_tracked", str(steps)]
    # and the oracle: the first step's scalars (before any parameter moved)
    outs = om.forward((x, src, None) if kind == "unimodal" else (x, x2, src, None), eps, True)
    ls = [float(v) for v in om.losses((x, src, None) if kind == "unimodal" else (x, x2, src, None), outs, 1.0)]
    mine = [got[0][0], got[0][1], got[0][3]] if kind == "unimodal" else list(got[0])
    np.testing.assert_allclose(mine, ls, rtol=1e-4)


@pytest.mark.gpu
def test_plain_c_host_steps_two_models_on_picked_streams(tmp_path):
    """tests/c_host/host_pair.c: the wave and the time model side by side on the two streams hp_pick_concurrent_streams returns,
    against the Python engines stepping one after the other on the default stream."""
    from hippie_amd.engine import Engine
    z, B, lr, steps = 10, 16, 1e-6, 3
    args, wants = [], []
    for j, (L, clip, salt) in enumerate(((50, 0.0, 4), (100, 1.0, 5))):
        d = tmp_path / f"m{j}"
        d.mkdir()
        plan, om, path, pv, bv = _export(d, "unimodal", z, L, B, salt=salt, clip=clip, lr=lr)
        x, src, cls, eps = O.synth_inputs(B, L, z, salt=salt)
        inputs = str(d / "inputs.bin")
        with open(inputs, "wb") as f:
            f.write(x.numpy().astype(np.float32).tobytes() + src.numpy().astype(np.int64).tobytes() + eps.numpy().astype(np.float32).tobytes())
        args += [path, inputs]
        eng = Engine(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(lr=lr, clip=clip))
        eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
        eng.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
        want = []
        for _ in range(steps):
            eng.train_step(use_graph=False)
            want.append(eng.scalars())
        wants.append(np.array(want))
    exe = str(tmp_path / "host_pair")
    libdir = os.path.join(ROOT, "hippie_amd")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_host", "host_pair.c"), "-o", exe, "-L", libdir, "-lhippie_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe] + args + [str(steps)], check=True, capture_output=True, text=True, timeout=300).stdout
    lines = out.strip().splitlines()
    pick = dict(zip(lines[0].split()[1::2], lines[0].split()[2::2]))
    assert lines[0].startswith("pick ") and pick["distinct"] == "1" and float(pick["pair_us"]) > 0 and 1 <= int(pick["tried"]) <= 36, out
    for j in range(2):
        got = np.array([[float(v) for v in ln.split()[4:]] for ln in lines if ln.startswith(f"model {j} step ")])
        assert got.shape == (steps, 4), out
        np.testing.assert_allclose(got, wants[j], rtol=1e-4)        # same kernels; only the order of the weight gradients' fp32 atomic sums differs from run to run,
            # and three AdamW steps amplify it: 1e-7 after the first step, up to 2.5e-5 after the third (measured)
        assert f"model {j} batches_tracked {steps}" in out


def test_staged_model_file_config_and_segments(tmp_path):
    """--resident-units: the file carries the loader segments and says so in its config words (no GPU needed)."""
    cfg = planner.ModelCfg("unimodal", 10, 50)
    plan = planner.lower(cfg, 8, planner.TrainCfg(lr=1e-3, resident_units=40, dp_world=2, dp_rank=1))
    path = str(tmp_path / "staged.hpm")
    export.save_model(plan, path)
    d = export.read_model(path)
    assert d["config"][10:13] == [40, 2, 1]
    assert {"stage", "step_staged", "fwd_train_staged"} <= set(d["segments"])
    assert {"data_x", "data_labels", "perm", "seed", "cursor"} <= {t["name"].decode() for t in d["io"]}
    lib = P.load_library()
    m = ctypes.c_void_p()
    assert lib.hp_model_load(path.encode(), export.NO_DEVICE, ctypes.byref(m)) == 0, lib.hp_last_error()
    out = (ctypes.c_int32 * 16)()
    assert lib.hp_model_config(m, out) == 0 and list(out)[10:13] == [40, 2, 1]
    # exported for two data-parallel ranks: the fused staged step (which holds no gradient all-reduce) is refused by name
    assert lib.hp_model_train_step_staged(m, 1, None) != 0 and b"hp_model_train_step_dp" in lib.hp_last_error()
    assert lib.hp_model_train_step_dp(m, 1, None) != 0 and b"hp_model_allreduce_init" in lib.hp_last_error()
    uid = (ctypes.c_char * 128)()
    assert lib.hp_model_allreduce_init(m, uid, 1, 2) != 0 and b"HP_MODEL_NO_DEVICE" in lib.hp_last_error()
    lib.hp_model_destroy(m)
    plan1 = planner.lower(cfg, 8, planner.TrainCfg(lr=1e-3, resident_units=40))
    export.save_model(plan1, path)
    assert lib.hp_model_load(path.encode(), export.NO_DEVICE, ctypes.byref(m)) == 0, lib.hp_last_error()
    assert lib.hp_model_train_step_staged(m, 1, None) != 0 and b"HP_MODEL_NO_DEVICE" in lib.hp_last_error()
    lib.hp_model_destroy(m)
    # a file without resident tables refuses the staged verb by name
    plan2, om, path2, pv, bv = _export(tmp_path, B=4)
    assert lib.hp_model_load(path2.encode(), export.NO_DEVICE, ctypes.byref(m)) == 0
    assert lib.hp_model_train_step_staged(m, 1, None) != 0 and b"resident" in lib.hp_last_error()
    lib.hp_model_destroy(m)


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [1, 0])
def test_c_level_staged_steps_match_the_python_engine(tmp_path, use_graph):
    """hp_model_train_step_staged (loader + step in one graph) through ctypes against Engine.train_step_staged on the same tables,
    permutation and seed: same kernels, same Philox noise -> the same scalars step by step."""
    from hippie_amd.engine import Engine
    z, L, B, N, lr, steps = 10, 50, 16, 80, 1e-6, 4
    cfg = planner.ModelCfg("unimodal", z, L)
    tc = planner.TrainCfg(lr=lr, clip=1.0, resident_units=N)
    plan = planner.lower(cfg, B, tc)
    om = O.OracleModel("unimodal", z, L, salt=7)
    pv, bv = export.arena_values(plan, {k: v.detach() for k, v in om.state.items()})
    path = str(tmp_path / "staged.hpm")
    export.save_model(plan, path, pv, bv)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, L, generator=g)
    labels = torch.randint(0, 5, (N,), generator=g)
    perm = torch.randperm(N, generator=g)
    seed = 99

    eng = Engine(cfg, B, tc)
    eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
    eng.load_dataset(x.cuda(), labels.cuda(), perm=perm.cuda(), seed=seed)
    want = []
    for _ in range(steps):
        eng.train_step_staged(use_graph=bool(use_graph))
        want.append(eng.scalars())

    lib = P.load_library()
    m = ctypes.c_void_p()
    assert lib.hp_model_load(path.encode(), 0, ctypes.byref(m)) == 0, lib.hp_last_error()

    def write(name, arr):
        a = np.ascontiguousarray(arr)
        assert lib.hp_model_write(m, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.nbytes, 0, None) == 0, lib.hp_last_error()
        assert lib.hp_model_synchronize(m, None) == 0

    write("data_x", x.numpy().astype(np.float32))
    write("data_labels", labels.numpy().astype(np.int64))
    write("perm", perm.numpy().astype(np.int64))
    write("seed", np.array([seed], dtype=np.int64))
    write("cursor", np.array([0], dtype=np.int64))
    got = []
    for _ in range(steps):
        assert lib.hp_model_train_step_staged(m, use_graph, None) == 0, lib.hp_last_error()
        sc = (ctypes.c_float * 4)()
        assert lib.hp_model_read(m, b"scalars", sc, 16, 0, None) == 0
        got.append(list(sc))
    cur = (ctypes.c_int64 * 1)()
    assert lib.hp_model_read(m, b"cursor", cur, 8, 0, None) == 0 and cur[0] == steps
    assert lib.hp_model_batches_tracked(m) == steps
    np.testing.assert_allclose(np.array(got), np.array(want), rtol=1e-4)      # (the order of fp32 atomic sums differs from run to run; AdamW steps amplify it)
    # the batch the last step trained on is the permutation's: row b of "x" is table row perm[(steps-1)*B + b]
    xb = np.zeros((B, L), dtype=np.float32)
    assert lib.hp_model_read(m, b"x", xb.ctypes.data_as(ctypes.c_void_p), xb.nbytes, 0, None) == 0
    np.testing.assert_array_equal(xb, x.numpy()[perm.numpy()[(steps - 1) * B: steps * B]])
    lib.hp_model_destroy(m)


def test_set_optimizer_host_only(tmp_path):
    plan, om, path, pv, bv = _export(tmp_path, B=4)
    lib = P.load_library()
    m = ctypes.c_void_p()
    assert lib.hp_model_load(path.encode(), export.NO_DEVICE, ctypes.byref(m)) == 0
    assert lib.hp_model_set_optimizer(m, 1e-4, 0.01, 1) == 0, lib.hp_last_error()
    assert lib.hp_model_set_optimizer(m, -1.0, 0.01, 0) != 0 and b"lr" in lib.hp_last_error()
    lib.hp_model_destroy(m)


@pytest.mark.gpu
def test_c_level_fine_tune_stage_new_lr_fresh_adamw(tmp_path):
    """hp_model_set_optimizer(lr / 10, wd, reset) = the reference re-wrapping the pretrained network in a new train module
    (scripts/...:263-268): parameters and BatchNorm buffers kept, AdamW moments and step count zeroed, new learning rate — against the
    Python surface doing exactly that."""
    from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE
    z, L, B, lr = 10, 50, 16, 1e-5
    plan, om, path, pv, bv = _export(tmp_path, "unimodal", z, L, B, salt=6, clip=0.0, lr=lr)
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=6)
    # Python surface: two steps, new module at lr / 10 (fresh AdamW), two more steps
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    net.load_state_dict({k: v.detach() for k, v in om.state.items()})
    net.set_eps_source(lambda eng: eps.cuda())
    want = []
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=lr, weight_decay=0.01)
    for stage in range(2):
        for i in range(2):
            mod.optimizer.zero_grad()
            loss = mod.training_step((x.cuda().view(B, 1, L), src.cuda()), i)
            loss.backward()
            mod.optimizer.step()
            want.append(float(loss.item()))
        mod = hippieUnimodalEmbeddingModelCVAE(mod.model, learning_rate=lr / 10, weight_decay=0.01)
    w_want = net.state_dict()["encoder.conv1.weight"].double().sum().item()

    lib = P.load_library()
    m = ctypes.c_void_p()
    assert lib.hp_model_load(path.encode(), 0, ctypes.byref(m)) == 0, lib.hp_last_error()
    for name, arr in (("x", x.numpy().astype(np.float32)), ("src", src.numpy().astype(np.int64)), ("eps", eps.numpy().astype(np.float32))):
        a = np.ascontiguousarray(arr)
        assert lib.hp_model_write(m, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.nbytes, 0, None) == 0, lib.hp_last_error()
        lib.hp_model_synchronize(m, None)
    got = []
    for stage in range(2):
        for i in range(2):
            assert lib.hp_model_train_step(m, 1, None) == 0, lib.hp_last_error()
            sc = (ctypes.c_float * 4)()
            assert lib.hp_model_read(m, b"scalars", sc, 16, 0, None) == 0
            got.append(sc[0])
        assert lib.hp_model_set_optimizer(m, lr / 10, 0.01, 1) == 0, lib.hp_last_error()
        step = (ctypes.c_int64 * 1)()
        assert lib.hp_model_read(m, b"adam_step", step, 8, 0, None) == 0 and step[0] == 0
    np.testing.assert_allclose(got, want, rtol=2e-6)
    t = TensorInfo()
    assert lib.hp_model_find(m, b"encoder.conv1.weight", ctypes.byref(t)) == 0
    w = np.zeros(t.numel, dtype=np.float32)
    assert lib.hp_model_read(m, b"encoder.conv1.weight", w.ctypes.data_as(ctypes.c_void_p), w.nbytes, 0, None) == 0
    np.testing.assert_allclose(w.astype(np.float64).sum(), w_want, rtol=1e-6)
    lib.hp_model_destroy(m)


@pytest.mark.gpu
def test_c_level_checkpoint_save_resume_and_ckpt_conversion(tmp_path):
    """hp_model_save after two steps -> (a) hp_model_load + two more steps == four uninterrupted steps; (b) export.checkpoint_from_file is
    the reference's checkpoint dict: it loads into the Python train module (state_dict keys of tests/golden/manifest.json, optimiser
    state) which then takes the same next step."""
    import json
    from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE
    z, L, B, lr = 10, 50, 16, 1e-5
    plan, om, path, pv, bv = _export(tmp_path, "unimodal", z, L, B, salt=8, clip=0.0, lr=lr)
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=8)
    lib = P.load_library()

    def load(p):
        m = ctypes.c_void_p()
        assert lib.hp_model_load(p.encode(), 0, ctypes.byref(m)) == 0, lib.hp_last_error()
        for name, arr in (("x", x.numpy().astype(np.float32)), ("src", src.numpy().astype(np.int64)), ("eps", eps.numpy().astype(np.float32))):
            a = np.ascontiguousarray(arr)
            assert lib.hp_model_write(m, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.nbytes, 0, None) == 0
            lib.hp_model_synchronize(m, None)
        return m

    def steps(m, n):
        out = []
        for _ in range(n):
            assert lib.hp_model_train_step(m, 1, None) == 0, lib.hp_last_error()
            sc = (ctypes.c_float * 4)()
            assert lib.hp_model_read(m, b"scalars", sc, 16, 0, None) == 0
            out.append(sc[0])
        return out

    ref = load(path)
    want = steps(ref, 4)
    lib.hp_model_destroy(ref)
    a = load(path)
    got = steps(a, 2)
    saved = str(tmp_path / "ckpt.hpm")
    assert lib.hp_model_save(a, saved.encode(), 1) == 0, lib.hp_last_error()
    lib.hp_model_destroy(a)
    b = load(saved)
    assert lib.hp_model_batches_tracked(b) == 2
    got += steps(b, 2)
    lib.hp_model_destroy(b)
    np.testing.assert_allclose(got, want, rtol=2e-6)          # resumed == uninterrupted (moments, step count, running statistics all restored)

    ck = export.checkpoint_from_file(saved)
    manifest = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    keys = {"model." + k: tuple(shape) for k, shape, _ in manifest["unimodal_z10_o50"]}      # the reference module's own state_dict() listing
    assert set(ck["state_dict"]) == set(keys), set(ck["state_dict"]) ^ set(keys)
    for k, shp in keys.items():
        assert tuple(ck["state_dict"][k].shape) == shp, k
    assert int(ck["state_dict"]["model.encoder.bn1.num_batches_tracked"]) == 2
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    net.set_eps_source(lambda eng: eps.cuda())
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=lr, weight_decay=0.01)
    mod.load_state_dict(ck["state_dict"])
    mod.optimizer.zero_grad()
    loss = mod.training_step((x.cuda().view(B, 1, L), src.cuda()), 0)         # lowers the engine the optimiser state is loaded into
    mod.load_state_dict(ck["state_dict"])
    mod.optimizer.load_state_dict(ck["optimizer_states"][0])
    nxt = []
    for i in range(2):
        mod.optimizer.zero_grad()
        loss = mod.training_step((x.cuda().view(B, 1, L), src.cuda()), i)
        loss.backward()
        mod.optimizer.step()
        nxt.append(float(loss.item()))
    np.testing.assert_allclose(nxt, want[2:], rtol=2e-6)


def test_checkpoint_conversion_round_trip_on_the_host(tmp_path):
    """model file with values -> reference-format .ckpt (--to-ckpt) -> model file again (--from-ckpt): the same arena images."""
    plan, om, path, pv, bv = _export(tmp_path, "unimodal", 10, 50, 4, salt=2)
    ck = str(tmp_path / "m.ckpt")
    export.main(["--to-ckpt", path, ck])
    sd = torch.load(ck, weights_only=False)["state_dict"]
    assert sd["model.encoder.conv1.weight"].shape == (64, 1, 3) and sd["model.encoder.layer1.0.conv1.weight"].shape == (64, 64, 3)
    np.testing.assert_array_equal(sd["model.encoder.layer1.0.conv1.weight"].numpy(), om.state["encoder.layer1.0.conv1.weight"].detach().numpy())
    back = str(tmp_path / "back.hpm")
    export.main(["--kind", "unimodal", "--z-dim", "10", "--output-size", "50", "--batch", "4", "--from-ckpt", ck, "-o", back])
    d = export.read_model(back)
    np.testing.assert_array_equal(d["param_values"], np.asarray(pv, dtype=np.float32).reshape(-1))
    np.testing.assert_array_equal(d["buf_values"][: len(bv)], np.asarray(bv, dtype=np.float32).reshape(-1))
