"""CPU: the C-ABI library loads and exports every symbol include/hippie_hip.h declares; program
validation (no GPU needed) accepts the planner's output and rejects malformed records."""
import ctypes
import os
import re

import numpy as np
import pytest

from hippie_amd import planner, program as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "hippie_hip.h")).read()
    declared = set(re.findall(r"\b(hp_[a-z_]+)\s*\(", hdr)) - {"hp_stat_repl"}       # a static inline helper, not an export
    assert declared == set(P.EXPORTS), declared ^ set(P.EXPORTS)
    lib = P.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hp_abi_version() == P.ABI_VERSION == 7


def test_op_record_layout_matches_header():
    hdr = open(os.path.join(ROOT, "include", "hippie_hip.h")).read()
    ni = int(re.search(r"#define HP_OP_NI (\d+)", hdr).group(1))
    nf = int(re.search(r"#define HP_OP_NF (\d+)", hdr).group(1))
    nb = int(re.search(r"#define HP_OP_NB (\d+)", hdr).group(1))
    assert (ni, nf, nb) == (P.NI, P.NF, P.NB)
    assert P.OP_DTYPE.itemsize == 8 + 4 * ni + 4 * nf + 8 * nb
    assert int(re.search(r"#define HP_STAT_REPL_MAX (\d+)", hdr).group(1)) == P.STAT_REPL_MAX
    for name, val in re.findall(r"#define HP_CONV_([A-Z_]+)\s+(\d+)", hdr):
        assert getattr(P, "CONV_" + name) == int(val), name
    assert [P.stat_repl(c) for c in (5, 10, 64, 128, 256, 512, 1024)] == [16, 16, 16, 8, 4, 2, 2]
    for name, val in re.findall(r"HP_OP_([A-Z_]+) = (\d+)", hdr):
        assert getattr(P, name) == int(val), name


def _create(ops, sizes):
    lib = P.load_library()
    bases = (ctypes.c_void_p * 6)(*[ctypes.c_void_p(0x1000)] * 6)     # never dereferenced by create/validate
    sz = (ctypes.c_int64 * 6)(*sizes)
    h = ctypes.c_void_p()
    rc = lib.hp_program_create(ops.ctypes.data_as(ctypes.c_void_p), len(ops), bases, sz, ctypes.byref(h))
    msg = lib.hp_last_error().decode()
    if rc == 0:
        lib.hp_program_destroy(h)
    return rc, msg


@pytest.mark.parametrize("kind", ["unimodal", "multimodal"])
def test_planner_programs_validate_without_gpu(kind):
    plan = planner.lower(planner.ModelCfg(kind=kind, z_dim=10, output_size=50, output_size2=100), 512,
                         planner.TrainCfg(lr=1e-3, clip=1.0))
    ops = plan.ops.array()
    n = plan.n_param_floats * 4
    sizes = [plan.ws_bytes, n, n, plan.n_buf_floats * 4, n, n]
    rc, msg = _create(ops, sizes)
    assert rc == 0, msg
    # an arena that is too small, an unknown opcode and a bad tap map are all refused with a message
    rc, msg = _create(ops, [plan.ws_bytes // 2] + sizes[1:])
    assert rc != 0 and "out of range" in msg
    bad = ops.copy()
    bad[3]["op"] = 99
    rc, msg = _create(bad, sizes)
    assert rc != 0 and "unknown opcode" in msg
    bad = ops.copy()
    k = [i for i, r in enumerate(bad) if int(r["op"]) == P.CONV_TAPS][0]
    bad[k]["i"][2] = 30          # K not a multiple of 32
    rc, msg = _create(bad, sizes)
    assert rc != 0 and "tap-map" in msg
    # extents, not only start offsets: an output tensor that starts inside the arena but runs past its end
    bad = ops.copy()
    k = [i for i, r in enumerate(bad) if int(r["op"]) == P.CONV_TAPS][0]
    bad[k]["buf"][2] = (P.WS << 56) | (plan.ws_bytes - 256)
    rc, msg = _create(bad, sizes)
    assert rc != 0 and "out of range" in msg and "extent" in msg
    bad = ops.copy()
    k = [i for i, r in enumerate(bad) if int(r["op"]) == P.ADAMW][0]
    bad[k]["i"][0] = plan.n_param_floats + 4          # optimiser arena overrun
    rc, msg = _create(bad, sizes)
    assert rc != 0 and "out of range" in msg
    bad = ops.copy()
    k = [i for i, r in enumerate(bad) if int(r["op"]) == P.BN_APPLY][0]
    bad[k]["buf"][0] = P.NULL                          # a required operand missing
    rc, msg = _create(bad, sizes)
    assert rc != 0 and "required" in msg
    segs = plan.ops.segments
    assert set(segs) == {"fwd_train", "bwd", "opt", "fwd_eval", "enc_eval", "step"}
    assert sum(c for k, (_, c) in segs.items() if k not in ("enc_eval", "step")) == len(ops)
    assert segs["step"] == (segs["fwd_train"][0], segs["fwd_train"][1] + segs["bwd"][1] + segs["opt"][1])      # alias: the whole step
    assert segs["enc_eval"][0] == segs["fwd_eval"][0] and 0 < segs["enc_eval"][1] < segs["fwd_eval"][1]


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: on a machine without an MI355X the engine refuses to construct."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hippie_amd.engine import Engine
    from hippie_amd.program import HipEngineError
    with pytest.raises(HipEngineError):
        Engine(planner.ModelCfg("unimodal", 10, 50), 8)
    from hippie_amd.dataloading import resample_on_device
    with pytest.raises(HipEngineError):
        resample_on_device(torch.zeros(4, 40), 50)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under hippie_amd/ or scripts/ may import it."""
    import glob
    for path in glob.glob(os.path.join(ROOT, "hippie_amd", "*.py")) + glob.glob(os.path.join(ROOT, "scripts", "*.py")):
        src = open(path).read()
        assert "import oracle" not in src and "from oracle" not in src, path


def test_graft_entry_build_runs():
    """The driver's build check: __graft_entry__.build() must compile, load the library and accept its ABI."""
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    g = importlib.import_module("__graft_entry__")
    g.build()
