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
    # a new train module owns a FRESH AdamW (hippie/model.py:93): step counter and moments restart
    assert net._any_engine().adam_step == 1
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


def test_rewrapped_module_starts_from_a_fresh_adamw():
    """scripts/train_model_with_multimodal.py:263-268 re-wraps the pretrained network in a new LightningModule, whose
    constructor builds a new torch.optim.AdamW (hippie/model.py:93): the first fine-tune step must be Adam's step 1
    on zero moments, whatever the pretraining left behind.  Checked against the oracle's first AdamW step."""
    z, L, B = 10, 50, 16
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    om = O.OracleModel("unimodal", z, L, salt=6)
    net.load_state_dict({k: v.detach() for k, v in om.state.items()})
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=6)
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3)
    for _ in range(3):                                  # "pretraining": leaves non-zero moments and step = 3
        mod.training_step((x.cuda(), src.cuda()), 0).backward()
        mod.optimizer.step()
    eng = net._any_engine()
    assert eng.adam_step == 3 and float(eng.m.abs().max()) > 0
    mod2 = hippieUnimodalEmbeddingModelCVAE(mod.model, learning_rate=1e-4)
    eng = net._any_engine()
    assert eng.adam_step == 0 and float(eng.m.abs().max()) == 0 and float(eng.v.abs().max()) == 0
    # set_gradient_clip re-lowers inside ONE module: state must survive that
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    e = net.engine(B, False)
    net.train()
    e.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
    e.forward(True)
    e.backward()
    e.optimizer_step()
    torch.cuda.synchronize()
    assert e.adam_step == 1
    mod2.set_gradient_clip(1.0)
    assert net._any_engine().adam_step == 1 and float(net._any_engine().m.abs().max()) > 0
    # oracle: same parameters, same batch, masked to the engine's branches, ONE AdamW step from zero state at lr 1e-4
    om2 = O.OracleModel("unimodal", z, L, salt=6)
    om2.load({k: v.cpu() for k, v in sd0.items()})
    masks = H.engine_masks(e)
    outs = om2.forward((x, src, None), eps, True, masks=masks)
    om2.losses((x, src, None), outs)[0].backward()
    with torch.no_grad():
        g = om2.grads()
        for k in om2.param_keys:
            if g[k] is not None:
                om2.exp_avg[k] = torch.zeros_like(om2.state[k])
                om2.exp_avg_sq[k] = torch.zeros_like(om2.state[k])
        O.adamw_step({k: om2.state[k] for k in om2.param_keys}, g, om2.exp_avg, om2.exp_avg_sq, 1, 1e-4, 0.01)
    sd = net.state_dict()
    import re
    for k in om2.param_keys:
        if g[k] is None or re.search(H.ZERO_GRAD_RE, k):
            continue
        H.assert_adam_close(sd[k].cpu().numpy(), om2.state[k].detach().numpy(), 1e-4, k, grad=g[k].numpy())


def test_out_of_range_labels_raise_index_error():
    """nn.Embedding semantics at the boundary (hippie/model.py:65-66): IndexError, nothing launched."""
    z, L = 10, 50
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=3)
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3)
    x = torch.zeros(8, 1, L).cuda()
    ok = torch.tensor([1, 2, 3, 4, 1, 2, 3, 4]).cuda()
    with pytest.raises(IndexError, match="source label"):
        mod.training_step((x, torch.tensor([1, 2, 3, 5, 1, 2, 3, 4]).cuda()), 0)
    with pytest.raises(IndexError, match="source label"):
        net(x, source_labels=ok - 2)
    with pytest.raises(IndexError, match="class label"):
        mod.training_step((x, torch.stack([torch.full((8,), 3), ok.cpu()], 1).cuda()), 0)       # class id 3 of 3
    loss = mod.training_step((x + 0.1, torch.stack([torch.full((8,), 2), ok.cpu()], 1).cuda()), 0)
    assert np.isfinite(loss.item())


def test_second_optimizer_step_does_not_double_count_the_gradient_norm():
    """clip_grad_norm_'s accumulator is zeroed inside the optimiser segment: stepping twice on the same gradients
    applies the same clip coefficient twice (= two torch AdamW steps on an unchanged .grad)."""
    z, L, B = 10, 50, 16
    from hippie_amd import planner
    from hippie_amd.engine import Engine
    om = O.OracleModel("unimodal", z, L, salt=8)
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=8)
    res = []
    for steps in (1, 2):
        eng = Engine(planner.ModelCfg("unimodal", z, L), B, planner.TrainCfg(lr=1e-3, clip=0.05))
        eng.load_state_dict({k: v.detach() for k, v in om.state.items()})
        eng.set_inputs(x.cuda(), src.cuda(), None, eps.cuda())
        eng.forward(True)
        eng.backward()
        for _ in range(steps):
            eng.optimizer_step()
        torch.cuda.synchronize()
        res.append((eng.m.clone(), eng.grads.clone()))
    # exp_avg after two steps on the same clipped gradient g_c: (1 - 0.9^2) g_c = 1.9 x the one-step value 0.1 g_c
    m1, m2 = res[0][0], res[1][0]
    big = m1.abs() > 1e-3 * m1.abs().max()
    ratio = (m2[big] / m1[big])
    assert float((ratio - 1.9).abs().max()) < 1e-3, float((ratio - 1.9).abs().max())


@pytest.mark.parametrize("deterministic", [True, False])
def test_trainer_async_path_equals_per_step_sync_path(tmp_path, deterministic):
    """Trainer.fit default (hipGraph replay, losses kept on the device, label checks deferred to the epoch end) against
    sync_every_step=True (the reference's per-step loss.item(), host-side label check every step) with the module
    running eagerly.  deterministic=True (Lightning's flag: ordered weight-gradient sums instead of fp32 atomics): the
    two paths launch the same kernels in the same order, so epoch losses and every parameter are EQUAL, bit for bit.
    deterministic=False: the order of the fp32 atomic sums differs from run to run; Adam turns a rounding-level
    gradient into a +-lr update and the ragged 6-sample batch (BatchNorm over 6 rows) amplifies that to a few 1e-3 of
    its loss (tools/debug/trainer_noise.py: same spread between two identical runs) — only a loose bound holds."""
    z, L = 10, 50
    om = O.OracleModel("unimodal", z, L, salt=4)
    train = batches(70, 32, L, z, seed=1)          # 32 + 32 + 6: a ragged last batch
    val = batches(40, 32, L, z, seed=2)
    res = []
    for sync in (True, False):
        net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
        net.load_state_dict({k: v.detach() for k, v in om.state.items()})
        net.use_graph = not sync
        mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-5, weight_decay=0.01)
        torch.manual_seed(123)                      # the reparameterisation noise comes from the device generator
        tr = Trainer(max_epochs=2, gradient_clip_val=1.0, enable_checkpointing=False, sync_every_step=sync, deterministic=deterministic)
        tr.fit(mod, train, val)
        assert net._train_cfg.deterministic_wgrad == deterministic
        res.append((tr.history, {k: v.cpu() for k, v in net.state_dict().items()}, mod.last_train_mean, mod.last_val_mean))
    (h0, sd0, t0, v0), (h1, sd1, t1, v1) = res
    assert len(h0) == len(h1) == 2
    if deterministic:
        for a, b in zip(h0, h1):
            # (the per-step values are identical; the sync path averages Python floats, the async path a float64 tensor)
            np.testing.assert_allclose([a["val_loss"], a["train_loss"]], [b["val_loss"], b["train_loss"]], rtol=1e-12)
        np.testing.assert_allclose([t0, v0], [t1, v1], rtol=1e-12)
        for k in sd0:
            assert torch.equal(sd0[k], sd1[k]), k
        return
    for a, b in zip(h0, h1):
        np.testing.assert_allclose(a["val_loss"], b["val_loss"], rtol=2e-2)
        np.testing.assert_allclose(a["train_loss"], b["train_loss"], rtol=2e-2)
    np.testing.assert_allclose([t0, v0], [t1, v1], rtol=2e-2)
    import re
    for k in sd0:
        if "running_" in k:
            H.assert_close(sd1[k].numpy(), sd0[k].numpy(), 5e-2, k)          # follow the +-lr parameter noise, amplified by the 6-row batch
        elif sd0[k].dtype.is_floating_point and not re.search(H.ZERO_GRAD_RE, k):
            # only Adam's hard bound (every element within 2.2 * lr * steps): how many elements sit beyond rounding noise
            # depends on the order of the atomic sums in the two runs
            H.assert_adam_close(sd1[k].numpy(), sd0[k].numpy(), 1e-5, k, steps=6, frac=1.0)


def test_concurrent_fits_with_self_drawn_noise_are_reproducible():
    """fit_concurrently with NO prescribed noise source: the wave and the time fit run on two threads; each network draws its
    reparameterisation noise from a device generator of its own, seeded before the threads start (two threads on the one global
    device generator consumed it in a scheduling-dependent order).  Two seeded runs end with identical parameters (deterministic=True)."""
    from hippie_amd.trainer import fit_concurrently
    z = 10
    res = []
    for _ in range(2):
        torch.manual_seed(321)
        jobs, nets = [], []
        for k, L in enumerate((50, 100)):
            om = O.OracleModel("unimodal", z, L, salt=6 + k)
            net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
            net.load_state_dict({kk: v.detach() for kk, v in om.state.items()})
            mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-4, weight_decay=0.01)
            tr = Trainer(max_epochs=2, gradient_clip_val=1.0, enable_checkpointing=False, deterministic=True)
            jobs.append((tr, mod, batches(96, 32, L, z, seed=3 + k), batches(32, 32, L, z, seed=5 + k)))
            nets.append(net)
        fit_concurrently(jobs)
        torch.cuda.synchronize()
        assert all(n.eps_generator is None for n in nets)                       # handed back after the fit
        res.append([{k: v.cpu() for k, v in n.state_dict().items()} for n in nets])
    for sd0, sd1 in zip(*res):
        for k in sd0:
            assert torch.equal(sd0[k], sd1[k]), k


def test_trainer_reports_bad_labels_at_epoch_end():
    z, L = 10, 50
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-4)
    train = batches(64, 32, L, z, seed=1)
    x, lab = train[1]
    lab = lab.clone()
    lab[3] = 9                                       # source id 9 of 5
    train[1] = (x, lab)
    with pytest.raises(IndexError):
        Trainer(max_epochs=1, enable_checkpointing=False).fit(mod, train)
    assert all(torch.isfinite(v).all() for v in net.state_dict().values() if v.dtype.is_floating_point)   # the kernels stayed safe


def test_label_checks_are_immediate_again_after_fit():
    """Trainer.fit runs with a deferred (device-side) label check; once it returns, `module(batch)` and the embedding path
    must raise IndexError at the call, like nn.Embedding (hippie/model.py:65-66) — nobody polls the deferred flag there."""
    z, L = 10, 50
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-4)
    train = batches(64, 32, L, z, seed=2)
    Trainer(max_epochs=1, enable_checkpointing=False).fit(mod, train)
    assert net.label_check == "sync" and mod.sync_every_step is True
    x, lab = train[0]
    bad = lab.clone()
    bad[5] = 7
    mod.eval()
    with pytest.raises(IndexError):
        mod((x.cuda(), bad.cuda()))
    with pytest.raises(IndexError):
        mod.embed((x.cuda(), bad.cuda()))
    # a flag left pending on an engine is reported before configure_training drops that engine
    net.label_check = "deferred"
    mod((x.cuda(), bad.cuda()))
    with pytest.raises(IndexError):
        mod.set_gradient_clip(0.5)
    net.label_check = "sync"


def test_class_embedding_reassignment_keeps_everything_else():
    """`model.class_embedding = nn.Embedding(n, class_hidden_dim)` as the reference's supervised stage does
    (scripts/train_model_with_multimodal.py:378-379): a new class table, all other parameters / buffers / AdamW moments kept."""
    z, L, B = 10, 50, 16
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    om = O.OracleModel("unimodal", z, L, salt=9)
    net.load_state_dict({k: v.detach() for k, v in om.state.items()})
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-4, weight_decay=0.01)
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=9)
    net.set_eps_source(lambda eng: eps.cuda())
    labels = torch.stack([cls % 3, src], 1).cuda()
    for i in range(2):
        mod.optimizer.zero_grad()
        loss = mod.training_step((x.cuda().view(B, 1, L), labels), i)
        loss.backward()
        mod.optimizer.step()
    assert net.class_embedding.num_embeddings == 5 and net.class_embedding.embedding_dim == 5 and net.source_embedding.num_embeddings == 5
    before = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    opt_before = mod.optimizer.state_dict()
    torch.manual_seed(123)
    new = torch.nn.Embedding(3, net.class_hidden_dim)
    net.class_embedding = new
    assert net.class_embedding.num_embeddings == 3
    after = net.state_dict()
    for k, v in before.items():
        if k == "class_embedding.weight":
            np.testing.assert_array_equal(after[k].cpu().numpy(), new.weight.detach().numpy())
        else:
            assert torch.equal(after[k].cpu(), v), k
    opt_after = mod.optimizer.state_dict()
    names = opt_after["param_names"]
    for i, k in enumerate(names):
        a, b = opt_after["state"][i], opt_before["state"][opt_before["param_names"].index(k)]
        if k == "class_embedding.weight":
            assert tuple(a["exp_avg"].shape) == (3, 5) and float(a["exp_avg"].abs().sum()) == 0.0
        else:
            assert torch.equal(a["exp_avg"].cpu(), b["exp_avg"].cpu()) and torch.equal(a["exp_avg_sq"].cpu(), b["exp_avg_sq"].cpu()), k
        assert float(a["step"]) == 2.0
    # the re-lowered network computes what a freshly built 3-class network with the same weights computes, and keeps training
    ref = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=3)
    ref.load_state_dict({k: v.detach() for k, v in after.items()})
    ref.set_eps_source(lambda eng: eps.cuda())
    net.eval(), ref.eval()
    a = net(x.cuda().view(B, 1, L), source_labels=src.cuda(), class_labels=(cls % 3).cuda())
    b = ref(x.cuda().view(B, 1, L), source_labels=src.cuda(), class_labels=(cls % 3).cuda())
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    with pytest.raises(IndexError):
        net(x.cuda().view(B, 1, L), source_labels=src.cuda(), class_labels=torch.full((B,), 4, device="cuda"))
    net.train()
    mod.optimizer.zero_grad()
    loss = mod.training_step((x.cuda().view(B, 1, L), labels), 2)
    loss.backward()
    mod.optimizer.step()
    assert np.isfinite(float(loss.item())) and mod.optimizer.state_dict()["state"][0]["step"] == 3.0
    with pytest.raises(ValueError):
        net.class_embedding = torch.nn.Embedding(3, 7)


def test_get_embeddings_two_streams_equal_one_stream(monkeypatch):
    """the wave and time encoder passes of get_embeddings run side by side on two streams: the same bits as one after the other, over
    several batches with a ragged last one (inputs handed over from the caller's stream, results consumed on it)."""
    from hippie_amd import utils
    z, B, N = 10, 64, 64 * 5 + 23
    mods, loaders = [], []
    g = torch.Generator().manual_seed(11)
    labels = torch.randint(1, 5, (N,), generator=g).cuda()
    for L, salt in ((50, 3), (100, 4)):
        net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
        om = O.OracleModel("unimodal", z, L, salt=salt)
        net.load_state_dict({k: v.detach() for k, v in om.state.items()})
        mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-4, weight_decay=0.01)
        mod.eval()
        mods.append(mod)
        x = torch.randn(N, 1, L, generator=g).cuda()
        loaders.append([(x[i: i + B], labels[i: i + B]) for i in range(0, N, B)])
    seen = []
    real = utils._side_streams

    def spy(*a):
        seen.append(real(*a))
        return seen[-1]
    monkeypatch.setattr(utils, "_side_streams", spy)
    two = [utils.get_embeddings(loaders[0], loaders[1], mods[0], mods[1]) for _ in range(3)]
    assert seen and all(s is not None and len(s) == 2 and s[0] != s[1] for s in seen)
    monkeypatch.setattr(utils, "_side_streams", lambda *a: None)
    one = utils.get_embeddings(loaders[0], loaders[1], mods[0], mods[1])
    for run in two:
        for a, b in zip(run, one):
            assert a.shape == b.shape and a.shape[0] == N
            np.testing.assert_array_equal(a, b)
