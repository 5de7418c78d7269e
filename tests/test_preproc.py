"""Preprocessing row (f1): oracle vs the reference's own EphysDatasetLabeled outputs (CPU), HIP kernel vs
both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import preproc

G = os.path.join(os.path.dirname(__file__), "golden", "datasets_first8.npz")


def cases():
    g = np.load(G)
    return g, [str(n) for n in g["names"]]


def test_oracle_matches_reference_dataset_items():
    g, names = cases()
    assert len(names) == 6
    for n in names:
        w, t = preproc.preprocess(g[n + ".wave_in"], g[n + ".isi_in"])
        np.testing.assert_allclose(w, g[n + ".wave_out"], rtol=1e-6, atol=1e-6, err_msg=n)
        np.testing.assert_allclose(t, g[n + ".isi_out"], rtol=1e-6, atol=1e-7, err_msg=n)
    # index-column quirk is part of the fixture: cellexplorer-celltype waveforms have 47 columns (46 + index)
    assert g["cellexplorer-celltype.wave_in"].shape[1] == 47


@pytest.mark.gpu
def test_hip_resample_matches_reference_and_dataset_contract():
    from hippie_amd.dataloading import EphysDatasetLabeled, resample_on_device
    g, names = cases()
    for n in names:
        w_in, t_in = g[n + ".wave_in"], g[n + ".isi_in"]
        labels = np.arange(len(w_in))
        dw = EphysDatasetLabeled(w_in, t_in, labels, mode="wave", normalize=False)
        dt = EphysDatasetLabeled(w_in, t_in, labels, mode="time", normalize=False)
        np.testing.assert_allclose(dw.data.cpu().numpy()[:, None, :], g[n + ".wave_out"], rtol=2e-6, atol=1e-6, err_msg=n)
        np.testing.assert_allclose(dt.data.cpu().numpy()[:, None, :], g[n + ".isi_out"], rtol=2e-6, atol=2e-7, err_msg=n)
        x, lab = dw[3]
        assert x.shape == (1, 50) and int(lab) == 3 and len(dw) == len(w_in)
        bx, bl = next(dt.batches(5))
        assert bx.shape == (5, 1, 100) and bl.tolist() == [0, 1, 2, 3, 4]
    with pytest.raises(TypeError):
        EphysDatasetLabeled(w_in, t_in, labels, mode="wave", normalize=True)
    # ragged / large: 10 000 x 352 -> 50 against the numpy oracle
    rng = np.random.default_rng(0)
    big = rng.standard_normal((10000, 352)).astype(np.float32)
    out = resample_on_device(torch.from_numpy(big).cuda(), 50).cpu().numpy()
    np.testing.assert_allclose(out, preproc.resample_linear(big, 50), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_get_embeddings_contract():
    from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE
    from hippie_amd.utils import get_embeddings
    from oracle import cvae_oracle as O
    z = 10
    mods, loaders, oms, xs = [], [], [], []
    src = None
    for k, L in enumerate((50, 100)):
        net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
        om = O.OracleModel("unimodal", z, L, salt=30 + k)
        net.load_state_dict({kk: v.detach() for kk, v in om.state.items()})
        mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3)
        mod.eval()
        x, s_, _, _ = O.synth_inputs(20, L, z, salt=30)
        src = s_
        loaders.append([(x[i: i + 8].cuda(), s_[i: i + 8].cuda()) for i in range(0, 20, 8)])
        mods.append(mod); oms.append(om); xs.append(x)
    ew, et, joint = get_embeddings(loaders[0], loaders[1], mods[0], mods[1])
    assert ew.shape == (20, z) and et.shape == (20, z) and joint.shape == (20, 2 * z)
    np.testing.assert_allclose(ew.mean(1), 0, atol=1e-5)
    np.testing.assert_allclose(ew.std(1, ddof=1), 1, rtol=1e-4)
    # enc does not depend on eps: compare with the oracle's eval-mode enc per loader batch
    want = []
    for i in range(0, 20, 8):
        with torch.no_grad():
            e = oms[0].forward((xs[0][i: i + 8], src[i: i + 8], None), torch.zeros(len(xs[0][i: i + 8]), z), training=False)[0]
        want.append(((e - e.mean(1, keepdim=True)) / e.std(1, keepdim=True)).numpy())
    np.testing.assert_allclose(ew, np.concatenate(want), rtol=2e-3, atol=2e-4)


def test_train_val_split_matches_reference_seed42():
    """The pipeline's split is the reference's: same torch function, same seed, same call order."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "scripts"))
    from torch.utils.data import random_split
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "random_split_seed42.npz"))
    for n, prop in ((2975, 0.8), (3797, 0.8), (392, 0.1)):
        torch.manual_seed(42)
        tr, te = random_split(list(range(n)), [int(prop * n), n - int(prop * n)])
        np.testing.assert_array_equal(np.array(tr.indices), g[f"n{n}_train"])
        np.testing.assert_array_equal(np.array(te.indices), g[f"n{n}_test"])
