"""GPU: the fp32 matrix-core path (TrainCfg.mfma_dtype="f32", v_mfma_f32_32x32x2_f32) — rounds 1-3's parity path, now the selectable
alternative to the default three-term bf16 path ("bf16x3", tests/test_gpu_split.py).  The op-level tests of tests/test_gpu_ops.py build
their records without a matrix-mode flag and therefore always run THIS path; here the end-to-end parity tests are re-run with it as the
lowering default (oracle parity, reference trajectory, flip budget, batch-512 step, module surface), at their unchanged tolerances — and
the two paths are compared with each other."""
import numpy as np
import pytest
import torch

from hippie_amd import planner, program as P
from hippie_amd.engine import Engine
from oracle import cvae_oracle as O
from tests import test_gpu_e2e as E
from tests import test_gpu_model as M
from tests import test_backbones as BB

pytestmark = pytest.mark.gpu


@pytest.fixture
def native_f32(monkeypatch):
    """planner.TrainCfg(...) without an explicit mfma_dtype lowers for the fp32 matrix cores (the modules' fp32_matrix_path follows it)"""
    orig = planner.TrainCfg

    def make(*a, **k):
        k.setdefault("mfma_dtype", "f32")
        return orig(*a, **k)
    monkeypatch.setattr(planner, "TrainCfg", make)


E2E = ([(E, "test_forward_grads_and_step_vs_oracle", (n,)) for n in E.CASES]
       + [(E, "test_training_trajectory_vs_reference_golden", (n, True)) for n in E.TRAJ]
       + [(E, "test_full_batch_512_step_matches_oracle", (n,)) for n in E.FULL]
       + [(E, "test_masked_trajectory_is_tight", ("wave",)),
          (E, "test_graph_replay_equals_eager_and_is_repeatable", ()),
          (E, "test_grouped_atomic_wgrad_equals_ordered_slab_reduction_at_full_batch", ()),
          (E, "test_config3_training_properties_at_batch_4096", ()),
          (M, "test_multimodal_module_step_and_metrics", ()),
          (M, "test_get_embeddings_matches_reference_fixture", ())]
       + [(BB, "test_backbone_class_matches_the_reference_fixture", (n,)) for n in BB.CASES])


@pytest.mark.parametrize("mod,fn,args", E2E, ids=[f"{f[5:]}-{'-'.join(map(str, a))}" for _, f, a in E2E])
def test_parity_suite_on_the_fp32_matrix_cores(native_f32, mod, fn, args):
    getattr(mod, fn)(*args)


def test_the_fixture_selects_the_fp32_matrix_cores(native_f32):
    eng = Engine(planner.ModelCfg("unimodal", 10, 50), 8, planner.TrainCfg(lr=1e-3))
    convs = [r for r in eng.ops if int(r["op"]) in (P.CONV_TAPS, P.WGRAD_TAPS)]
    assert convs and not any(int(r["flags"]) & (P.CONV_BF16 | P.CONV_BF16X3) for r in convs)


def test_default_lowering_uses_the_three_term_path():
    eng = Engine(planner.ModelCfg("unimodal", 10, 50), 8, planner.TrainCfg(lr=1e-3))
    convs = [r for r in eng.ops if int(r["op"]) in (P.CONV_TAPS, P.WGRAD_TAPS)]
    assert convs and all(int(r["flags"]) & P.CONV_BF16X3 for r in convs) and not any(int(r["flags"]) & P.CONV_BF16 for r in convs)


@pytest.mark.parametrize("B,L,clip", [(64, 50, 0.0), (512, 100, 1.0)])
def test_the_two_fp32_paths_agree_to_rounding(B, L, clip):
    """Same parameters, batch and noise through both lowerings, one optimisation step: outputs, loss scalars, every gradient and every
    gradient agree to fp32 rounding accumulated over the 40 conv layers: outputs 5-7e-6 of the tensor's max, loss scalars < 1e-6.  Both runs
    are FREE-RUNNING, so an activation within rounding of zero may take the other leaky-ReLU branch in one of them (tests/helpers.py's flip
    budget: <= 3e-6 of the activations) and moves the gradient elements downstream of it: gradients agree to 0.7-2.1e-2 of the tensor's max in
    the worst tensor (bounds 6e-2; l2 3e-2).  Each path meets the 1e-4 of the parity tests against the oracle evaluated on ITS OWN branches."""
    z = 10
    cfg = planner.ModelCfg("unimodal", z, L)
    om = O.OracleModel("unimodal", z, L, salt=3)
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=3)
    res = {}
    for path in ("f32", "bf16x3"):
        eng = Engine(cfg, B, planner.TrainCfg(lr=1e-3, clip=clip, mfma_dtype=path))
        eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
        eng.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
        outs = [o.double().cpu() for o in eng.forward(True)]
        eng.backward()
        torch.cuda.synchronize()
        res[path] = (outs, np.array(eng.scalars(), dtype=np.float64), {k: v.double().cpu() for k, v in eng.grad_dict().items()})
    worst_out = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res["f32"][0], res["bf16x3"][0]))
    worst_sc = float(np.max(np.abs(res["f32"][1] - res["bf16x3"][1]) / np.maximum(np.abs(res["f32"][1]), 1e-12)))
    import re
    from tests import helpers as H
    # (gradients that are analytically zero — a bias in front of a BatchNorm — are rounding noise on both paths: not compared)
    worst_g = max(float((res["f32"][2][k] - g).abs().max() / max(float(g.abs().max()), 1e-30)) for k, g in res["bf16x3"][2].items()
                  if not re.search(H.ZERO_GRAD_RE, k))
    l2_g = max(float((res["f32"][2][k] - g).norm() / max(float(g.norm()), 1e-30)) for k, g in res["bf16x3"][2].items() if not re.search(H.ZERO_GRAD_RE, k))
    print(f"B={B} L={L}: outputs {worst_out:.2e}  scalars {worst_sc:.2e}  gradients max-norm {worst_g:.2e} l2 {l2_g:.2e}")
    assert worst_out < 5e-5 and worst_sc < 2e-5 and worst_g < 6e-2 and l2_g < 3e-2
