"""Pin the CPU oracle against golden vectors produced by the reference's own modules.

The fixtures under tests/golden/ were written by tests/golden/make_golden.py,
which imports /root/reference (build container only).  These tests never touch
the reference: they re-run the oracle on the same closed-form inputs.
"""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import cvae_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
torch.set_num_threads(4)


def load(name):
    return dict(np.load(os.path.join(G, name), allow_pickle=False))


def test_manifest_matches_reference_state_dict():
    man = json.load(open(os.path.join(G, "manifest.json")))
    mine = O.unimodal_manifest(10, 50, 5, 5, 5)
    ref = man["unimodal_z10_o50"]
    assert [k for k, _, _ in ref] == list(mine.keys())
    assert [tuple(s) for _, s, _ in ref] == [tuple(v) for v in mine.values()]
    assert len(ref) == 277
    mine = O.multimodal_manifest(10, 50, 100, 5, 5, 5)
    ref = man["multimodal_z10_o50_100"]
    assert [k for k, _, _ in ref] == list(mine.keys())
    assert [tuple(s) for _, s, _ in ref] == [tuple(v) for v in mine.values()]
    assert len(ref) == 529
    n = sum(int(np.prod(s)) for k, s in O.unimodal_manifest(10, 50, 5, 5, 5).items() if not O.is_buffer(k))
    assert n == man["n_params_unimodal"] == 8056639


CASES = [
    ("wave_z10_L50_B16", dict(z=10, L=50, B=16, cls=False, beta=1.0, clip=None, steps=3, lr=1e-3, salt=0)),
    ("time_z10_L100_B16_clip", dict(z=10, L=100, B=16, cls=False, beta=1.0, clip=1.0, steps=3, lr=1e-3, salt=0)),
    ("wave_z5_L50_B12_cls", dict(z=5, L=50, B=12, cls=True, beta=0.5, clip=1.0, steps=2, lr=1e-4, salt=3)),
    ("wave_z32_L256_B8", dict(z=32, L=256, B=8, cls=False, beta=1.0, clip=None, steps=1, lr=1e-3, salt=5)),
    ("time_z32_L32_B8", dict(z=32, L=32, B=8, cls=False, beta=1.0, clip=None, steps=1, lr=1e-3, salt=6)),
]


def zero_grad_key(k):
    """Biases whose output feeds (Linear ->) BatchNorm with nothing nonlinear in between: their true
    gradient is exactly zero, the reference's value is rounding noise, and Adam turns that noise into
    updates of up to lr per step -- not reproducible even by the reference itself."""
    return bool(re.search(r"(encoder(_mod\d)?\.linear\.bias|encoder_fc\.[03]\.bias|fusion_encoder\.0\.bias|"
                          r"decoder_fc(_mod\d)?\.2\.bias|layer\d\.1\.(conv1|shortcut\.0)\.conv\.bias)$", k))


def stats(t):
    t = t.detach().double()
    return np.array([float(t.sum()), float(t.norm()), float(t.abs().max())])


@pytest.mark.parametrize("tag,c", CASES, ids=[c[0] for c in CASES])
def test_unimodal_oracle_equals_reference(tag, c):
    g = load(f"unimodal_{tag}.npz")
    m = O.OracleModel("unimodal", c["z"], c["L"], salt=c["salt"])
    x, src, cls, eps = O.synth_inputs(c["B"], c["L"], c["z"], salt=c["salt"])
    batch = (x, src, cls if c["cls"] else None)

    with torch.no_grad():
        enc, mu, lv, dec = m.forward(batch, eps, training=False)
    # same ATen ops on the same inputs: bitwise is expected, allow last-ulp noise
    for k, v in (("eval_enc", enc), ("eval_mu", mu), ("eval_logvar", lv), ("eval_dec", dec)):
        np.testing.assert_allclose(v.numpy(), g[k], rtol=1e-6, atol=1e-6, err_msg=k)

    taps = {}
    with torch.no_grad():
        enc, mu, lv, dec = m.forward(batch, eps, training=True, taps=taps)
    for k, v in (("enc", enc), ("mu", mu), ("logvar", lv), ("dec", dec)):
        np.testing.assert_allclose(v.numpy(), g[k], rtol=1e-6, atol=1e-6, err_msg=k)
    for name, ref in zip(g["tap_names"], g["tap_stats"]):
        key = str(name)
        if key == "enc_h":
            mine = taps["enc_h"]
        else:
            mine = taps[key[:-3]+"out"] if key.endswith("out") else taps[key]
        np.testing.assert_allclose(stats(mine), ref, rtol=1e-5, atol=1e-5, err_msg=key)

    # restore buffers, then real training steps
    m2 = O.OracleModel("unimodal", c["z"], c["L"], salt=c["salt"])
    for s in range(1, c["steps"] + 1):
        outs, ls, norm = m2.train_step(batch, eps, lr=c["lr"], weight_decay=0.01, beta=c["beta"], clip=c["clip"])
        if s == 1:
            np.testing.assert_allclose([float(v) for v in ls], g["scalars"], rtol=1e-6)
            if c["clip"] is not None:
                np.testing.assert_allclose(float(norm), g["grad_total_norm"][0], rtol=1e-5)
        if s in (1, c["steps"]):
            ref = g[f"state_stats_step{s}"]
            for name, r in zip(g["state_names"], ref):
                name = str(name)
                if zero_grad_key(name):
                    assert abs(stats(m2.state[name])[2] - r[2]) <= 1.1 * s * c["lr"], name
                    continue
                np.testing.assert_allclose(stats(m2.state[name]), r, rtol=3e-5, atol=3e-6, err_msg=f"step{s} {name}")
            for k in g:
                if k.startswith(f"param_step{s}.") and not zero_grad_key(k):
                    np.testing.assert_allclose(m2.state[k.split(".", 1)[1]].detach().numpy(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)
    assert int(g["n_adam_states"][0]) == len(m2.exp_avg)
    # gradients of step 1 (pre-clip): recompute on a fresh model
    m3 = O.OracleModel("unimodal", c["z"], c["L"], salt=c["salt"])
    outs = m3.forward(batch, eps, True)
    m3.losses(batch, outs, c["beta"])[0].backward()
    gr = m3.grads()
    none = set(str(s) for s in g["grad_none"])
    assert none == {k for k, v in gr.items() if v is None}
    for name, r in zip(g["grad_names"], g["grad_stats"]):
        name = str(name)
        if name in none:
            continue
        if zero_grad_key(name):
            assert stats(gr[name])[2] < 1e-5 and r[2] < 1e-5, name
            continue
        np.testing.assert_allclose(stats(gr[name]), r, rtol=1e-4, atol=1e-6, err_msg=name)
    for k in g:
        if k.startswith("grad_full.") and not zero_grad_key(k):
            np.testing.assert_allclose(gr[k.split(".", 1)[1]].numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("fname,z,L1,L2,B,salt,w2,steps", [
    ("multimodal_z10_B12.npz", 10, 50, 100, 12, 7, 0.5, 2),
    ("multimodal_z64_L256_32_B8.npz", 64, 256, 32, 8, 8, 1.0, 1),       # BASELINE config 5's shape at a tiny batch
])
def test_multimodal_oracle_equals_reference(fname, z, L1, L2, B, salt, w2, steps):
    g = load(fname)
    m = O.OracleModel("multimodal", z, L1, output_size2=L2, salt=salt)
    x1, src, cls, eps = O.synth_inputs(B, L1, z, salt=salt, name="x1")
    x2, _, _, _ = O.synth_inputs(B, L2, z, salt=salt, name="x2")
    batch = (x1, x2, src, None)
    for s in range(1, steps + 1):
        outs, ls, norm = m.train_step(batch, eps, lr=1e-3, beta=1.0, clip=1.0, w1=1.0, w2=w2)
        if s == 1:
            np.testing.assert_allclose([float(v) for v in ls], g["scalars"], rtol=1e-6)
    for name, r in zip(g["state_names"], g[f"state_stats_step{steps}"]):
        name = str(name)
        if zero_grad_key(name):
            assert abs(stats(m.state[name])[2] - r[2]) <= 2.2e-3, name
            continue
        np.testing.assert_allclose(stats(m.state[name]), r, rtol=3e-5, atol=3e-6, err_msg=name)
    with torch.no_grad():
        enc, mu, lv, d1, d2 = m.forward(batch, eps, training=False)
    for k, v in (("eval_enc", enc), ("eval_mu", mu), ("eval_logvar", lv), ("eval_dec1", d1), ("eval_dec2", d2)):
        np.testing.assert_allclose(v.numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_get_embeddings_oracle_equals_reference():
    """Fixture (11): scripts/utils.py:get_embeddings run on the reference's modules (make_golden_embeddings.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "get_embeddings_z10_B6x2.npz"))
    z, B, nb, sw, st = (int(v) for v in g["meta"])
    xw, src, _, eps = O.synth_inputs(B * nb, 50, z, salt=sw)
    xt = O.synth_inputs(B * nb, 100, z, salt=st)[0]
    out = []
    for L, salt, x in ((50, 0, xw), (100, 1, xt)):
        om = O.OracleModel("unimodal", z, L, salt=salt)
        rows = []
        with torch.no_grad():
            for i in range(nb):
                sl = slice(i * B, (i + 1) * B)
                enc = om.forward((x[sl], src[sl], None), eps[sl], training=False)[0]
                rows.append((enc - enc.mean(dim=1)[:, None]) / enc.std(dim=1)[:, None])
        out.append(torch.cat(rows).numpy())
    np.testing.assert_allclose(out[0], g["waveform"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[1], g["isi"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.concatenate(out, axis=1), g["joint"], rtol=1e-5, atol=1e-6)
