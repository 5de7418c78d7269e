"""GPU: the reference's class surface (hippie_amd.model) and the Lightning-stand-in Trainer."""
import json
import os

import numpy as np
import pytest
import torch

from hippie_amd.model import (hippieUnimodalCVAE, MultiModalCVAE, hippieUnimodalEmbeddingModelCVAE,
                              MultiModalCVAETrainModule)
from hippie_amd.trainer import Trainer
from oracle import cvae_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def batches(n, B, L, z, two_col=False, seed=0):
    x, src, cls, _ = O.synth_inputs(n, L, z, salt=seed)
    labels = torch.stack([cls, src], 1) if two_col else src
    return [(x[i: i + B], labels[i: i + B]) for i in range(0, n, B)]


def test_unimodal_module_surface_fit_checkpoint_reload(tmp_path):
    z, L = 10, 50
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    om = O.OracleModel("unimodal", z, L, salt=4)
    net.load_state_dict({k: v.detach() for k, v in om.state.items()})
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-4, weight_decay=0.01)
    assert mod.model is net and mod.configure_optimizers() is mod.optimizer
    # eval forward through the module == oracle (eps injected through the net's forward)
    x, src, cls, eps = O.synth_inputs(24, L, z, salt=4)
    net.eval()
    enc, mu, lv, dec = net(x.cuda(), source_labels=src.cuda(), eps=eps.cuda())
    with torch.no_grad():
        o = om.forward((x, src, None), eps, training=False)
    for a, b, nm in zip((enc, mu, lv, dec), o, ("enc", "mu", "logvar", "dec")):
        H.assert_close(a.cpu().numpy().reshape(b.shape), b.numpy(), 1e-4, nm)
    assert dec.shape == (24, 1, L)
    # fit: 70 samples at batch 32 -> a ragged last batch of 6 goes through a second engine sharing the arenas
    train = batches(70, 32, L, z, seed=1)
    val = batches(40, 32, L, z, seed=2)
    log = tmp_path / "log.jsonl"
    tr = Trainer(max_epochs=3, gradient_clip_val=1.0, default_root_dir=str(tmp_path / "ckpt"), logger_path=str(log))
    tr.fit(mod, train, val)
    assert tr.global_step == 9 and mod.model._any_engine().adam_step == 9
    assert len(net._engines) >= 2
    assert set(mod.logged) >= {"train_loss", "train_mse_loss", "train_kl_loss", "val_loss", "val_mse_loss", "val_kl_loss"}
    recs = [json.loads(l) for l in open(log)]
    assert len(recs) == 3 and all(np.isfinite(r["val_loss"]) for r in recs)
    assert all(np.isfinite(r["train_loss"]) for r in recs)
    # checkpoint format the reference's consumers read (scripts/...:229-230,473-477)
    ck = torch.load(tr.best_model_path, weights_only=False)
    man = json.load(open(os.path.join(G, "manifest.json")))["unimodal_z10_o50"]
    assert sorted(ck["state_dict"]) == sorted("model." + k for k, _, _ in man)
    assert len(ck["optimizer_states"][0]["state"]) == 151
    # reload into a fresh model: identical eval outputs; class-embedding size mismatch tolerated as the scripts do
    net2 = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=7)
    mod2 = hippieUnimodalEmbeddingModelCVAE(net2, learning_rate=1e-4)
    sd = dict(ck["state_dict"])
    sd.pop("model.class_embedding.weight")
    missing, unexpected = mod2.load_state_dict(sd, strict=False)
    assert missing == ["class_embedding.weight"] and not unexpected
    mod2.optimizer.load_state_dict(ck["optimizer_states"][0])
    best = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    best.load_state_dict({k[len("model."):]: v for k, v in ck["state_dict"].items()})
    best.eval(); net2.eval()
    a = best(x.cuda(), source_labels=src.cuda(), eps=eps.cuda())
    b = net2(x.cuda(), source_labels=src.cuda(), eps=eps.cuda())
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u.cpu().numpy(), v.cpu().numpy())


def test_two_column_labels_and_lr_rewrap():
    z, L = 5, 50
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3, beta=0.5)
    b = batches(16, 16, L, z, two_col=True, seed=3)[0]
    b = (b[0].cuda(), b[1].cuda())
    l0 = mod.training_step(b, 0)
    l0.backward()
    mod.optimizer.step()
    before = net.state_dict()["class_embedding.weight"].clone()
    # the scripts re-wrap the same nn.Module with lr/10 for fine-tuning (scripts/...:263-268)
    mod2 = hippieUnimodalEmbeddingModelCVAE(mod.model, learning_rate=1e-4)
    assert mod2.optimizer.param_groups[0]["lr"] == 1e-4
    l1 = mod2.training_step(b, 0)
    l1.backward()
    mod2.optimizer.step()
    after = net.state_dict()["class_embedding.weight"]
    assert not torch.equal(before, after)          # class embedding trains when class labels are given
    assert net._any_engine().adam_step == 2
    enc, mu, lv, dec = mod2((b[0], b[1]))
    assert enc.shape == (16, z) and dec.shape == (16, 1, L)
    with pytest.raises(ValueError):
        net(torch.zeros(4, 1, 60).cuda(), source_labels=torch.ones(4, dtype=torch.int64).cuda())
    net.train()
    with pytest.raises(ValueError):          # BatchNorm cannot train on a single row (torch raises too)
        net(torch.zeros(1, 1, L).cuda(), source_labels=torch.ones(1, dtype=torch.int64).cuda())
    net.eval()
    one = net(torch.zeros(1, 1, L).cuda(), source_labels=torch.ones(1, dtype=torch.int64).cuda())
    assert one[0].shape == (1, z) and torch.isfinite(one[3]).all()


def test_multimodal_module_step_and_metrics():
    z = 10
    net = MultiModalCVAE(z_dim=z, output_size_wave=50, output_size_isi=100, class_hidden_dim=5, num_sources=5, num_classes=5)
    mod = MultiModalCVAETrainModule(net, learning_rate=1e-4, beta=1.0, mod1_weight=1.0, mod2_weight=0.5)
    x1, src, cls, _ = O.synth_inputs(20, 50, z, salt=1, name="x1")
    x2, _, _, _ = O.synth_inputs(20, 100, z, salt=1, name="x2")
    batch = (x1.cuda(), x2.cuda(), src.cuda())
    losses = []
    for i in range(4):
        loss = mod.training_step(batch, i)
        loss.backward()
        mod.optimizer.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0] and np.isfinite(losses).all()
    assert set(mod.logged) >= {"train_loss", "train_mse_loss1", "train_mse_loss2", "train_kl_loss"}
    v = mod.validation_step(batch, 0)
    assert np.isfinite(v.item()) and "val_mse_loss2" in mod.logged
    enc, mu, lv, d1, d2 = mod(batch)
    assert d1.shape == (20, 1, 50) and d2.shape == (20, 1, 100) and enc.shape == (20, z)
    assert len(mod.state_dict()) == 529


def test_get_embeddings_matches_reference_fixture():
    """hippie_amd.utils.get_embeddings on the GPU modules against the reference's own function output
    (tests/golden/get_embeddings_z10_B6x2.npz); 1e-4 relative to the row-standardised scale (~1)."""
    from hippie_amd.utils import get_embeddings
    g = np.load(os.path.join(G, "get_embeddings_z10_B6x2.npz"))
    z, B, nb, sw, st = (int(v) for v in g["meta"])
    xw, src, _, _ = O.synth_inputs(B * nb, 50, z, salt=sw)
    xt = O.synth_inputs(B * nb, 100, z, salt=st)[0]
    mods = []
    for L, salt in ((50, 0), (100, 1)):
        net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
        net.load_state_dict({k: v.detach() for k, v in O.OracleModel("unimodal", z, L, salt=salt).state.items()})
        mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3)
        mod.eval()
        mods.append(mod)
    lw = [(xw[i * B:(i + 1) * B].cuda(), src[i * B:(i + 1) * B].cuda()) for i in range(nb)]
    lt = [(xt[i * B:(i + 1) * B].cuda(), src[i * B:(i + 1) * B].cuda()) for i in range(nb)]
    ew, et, joint = get_embeddings(lw, lt, mods[0], mods[1])
    np.testing.assert_allclose(ew, g["waveform"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(et, g["isi"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(joint, g["joint"], rtol=1e-4, atol=1e-4)


def test_encoder_only_path_equals_full_forward():
    z, L, B = 10, 100, 37
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    net.load_state_dict({k: v.detach() for k, v in O.OracleModel("unimodal", z, L, salt=2).state.items()})
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3)
    x, src, cls, _ = O.synth_inputs(B, L, z, salt=2)
    batch = (x.cuda(), torch.stack([cls, src], 1).cuda())
    with pytest.raises(RuntimeError, match="eval"):
        net.encode_labels(batch[0], src.cuda())
    mod.eval()
    enc, mu, lv, _ = [t.clone() for t in mod(batch)]
    e2, m2, l2 = [t.clone() for t in net.encode_labels(batch[0], src.cuda(), cls.cuda())]
    assert torch.equal(enc, e2) and torch.equal(mu, m2) and torch.equal(lv, l2)
    assert torch.equal(mod.embed(batch), enc)
    seg = net.engine(B, True).plan.ops.segments
    assert seg["enc_eval"][1] < 0.6 * seg["fwd_eval"][1]
