"""GPU: the SHIPPED tables of the target dataset of BASELINE configs[0] / [1] (datasets/cellexplorer-celltype: 392 units, 47 waveform and
100 ISI columns incl. the unnamed index column the scripts read as a feature) through the GPU path once: preprocessing against what the
reference's own EphysDatasetLabeled yields for every row, then the label-free fine-tune stage's shape — the 39 / 353 split of
`random_split` under seed 42 (scripts/train_model_with_multimodal.py:234-268), a few optimisation steps at the fine-tune learning rate
on the 39 training units, eval-mode embeddings of all 392 — against the CPU oracle walking the same steps.
Fixture: tests/golden/cellexplorer_celltype_tables.npz (tests/golden/make_golden_cellexplorer.py)."""
import os

import numpy as np
import pytest
import torch

from hippie_amd import planner
from hippie_amd.engine import Engine
from oracle import cvae_oracle as O
from oracle import preproc
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_shipped_cellexplorer_tables_preprocess_finetune_embed():
    from hippie_amd.dataloading import EphysDatasetLabeled
    g = np.load(os.path.join(G, "cellexplorer_celltype_tables.npz"))
    split = np.load(os.path.join(G, "random_split_seed42.npz"))
    tr_idx = split["n392_train"]
    assert g["wave_in"].shape == (392, 47) and g["isi_in"].shape == (392, 100) and len(tr_idx) == 39 and len(split["n392_test"]) == 353
    n = len(g["wave_in"])
    src_id = 3                                                    # scripts/...:81-89: cellexplorer-* is source 3
    labels = np.full(n, src_id, dtype=np.int64)
    # --- preprocessing: GPU kernel and CPU oracle against the reference's dataset class, all 392 rows
    dsw = EphysDatasetLabeled(g["wave_in"], g["isi_in"], labels, mode="wave", normalize=False)
    dst = EphysDatasetLabeled(g["wave_in"], g["isi_in"], labels, mode="time", normalize=False)
    np.testing.assert_allclose(dsw.data.cpu().numpy()[:, None, :], g["wave_out"], rtol=2e-6, atol=1e-5)       # (values up to 391: the index column)
    np.testing.assert_allclose(dst.data.cpu().numpy()[:, None, :], g["isi_out"], rtol=2e-6, atol=2e-7)
    ow, ot = preproc.preprocess(g["wave_in"], g["isi_in"])
    np.testing.assert_allclose(ow, g["wave_out"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(ot, g["isi_out"], rtol=1e-6, atol=1e-7)
    # --- fine-tune on the 39 training units, embeddings of all 392: engine against the float32 / float64 oracles
    z, steps, lr = 10, 3, 1e-4                                     # (lr = the scripts' learning rate / 10, :263-268)
    for kind, L, ds, clip in (("wave", 50, dsw, None), ("time", 100, dst, 1.0)):
        cfg = planner.ModelCfg(kind="unimodal", z_dim=z, output_size=L)
        tc = planner.TrainCfg(lr=lr, weight_decay=0.01, beta=1.0, clip=clip or 0.0)
        eng = Engine(cfg, len(tr_idx), tc)
        emb = Engine(cfg, n, tc, share_params_from=eng)
        oms = [O.OracleModel("unimodal", z, L, salt=21, dtype=dt) for dt in (torch.float32, torch.float64)]
        eng.load_state_dict({k: v.detach() for k, v in oms[0].state.items()})
        x_all = ds.data.view(n, 1, L)
        x_cpu = torch.from_numpy(g["wave_out" if kind == "wave" else "isi_out"])          # the reference's own preprocessed rows for the oracle
        src = torch.from_numpy(labels)
        ti = torch.from_numpy(tr_idx)
        for step in range(steps):
            eps = O.synth_inputs(len(tr_idx), L, z, salt=40 + step)[3]
            eng.set_inputs(x_all[ti.cuda()], src[ti].cuda(), None, eps.cuda())
            eng.train_step()
            for om, dt in zip(oms, (torch.float32, torch.float64)):
                om.train_step((x_cpu[ti].to(dt), src[ti], None), eps.to(dt), lr=lr, weight_decay=0.01, beta=1.0, clip=clip)
        torch.cuda.synchronize()
        eps_all = O.synth_inputs(n, L, z, salt=50)[3]
        emb.set_inputs(x_all, src.cuda(), None, eps_all.cuda())
        enc = emb.forward(training=False)[0].cpu().numpy()
        refs = []
        for om, dt in zip(oms, (torch.float32, torch.float64)):
            with torch.no_grad():
                refs.append(om.forward((x_cpu.to(dt), src, None), eps_all.to(dt), training=False)[0].numpy())
        H.parity(enc, refs[0], refs[1], f"{kind}: embeddings of the 392 shipped units after {steps} fine-tune steps")
        # ... and row-standardised, as get_embeddings writes them (scripts/utils.py:86-89)
        st = lambda e: (e - e.mean(1, keepdims=True)) / e.std(1, ddof=1, keepdims=True)
        H.parity(st(enc.astype(np.float64)), st(refs[0].astype(np.float64)), st(refs[1]), f"{kind}: row-standardised embeddings")
        scal = eng.scalars()
        assert np.isfinite(scal).all() and scal[0] > 0
