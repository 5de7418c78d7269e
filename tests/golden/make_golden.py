"""Generate golden vectors from the REAL reference modules (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

Imports /root/reference/hippie/{backbones,model}.py (read-only, no bytecode
written) behind a tiny in-memory stand-in for the absent ``pytorch_lightning``
package, fills every parameter/buffer with the closed-form recipe of
``oracle/cvae_oracle.fill_value`` through ``load_state_dict``, runs the
reference's own forward / training_step / AdamW on closed-form inputs with the
reparameterisation noise injected by patching ``torch.randn_like``, and stores
inputs-free fixtures (everything is re-derivable from the closed forms) as
small ``.npz`` files next to this script.  The reference itself never travels;
only these data files do.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

# ---- in-memory pytorch_lightning stand-in (model.py:5,7 need these names) ----
pl = types.ModuleType("pytorch_lightning")


class _Trainer:
    max_epochs = 1


class LightningModule(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.trainer = _Trainer()
        self.current_epoch = 0
        self.logged = {}

    def log(self, name, value, *a, **k):
        self.logged[name] = float(value)


pl.LightningModule = LightningModule
util = types.ModuleType("pytorch_lightning.utilities")
util.grad_norm = lambda *a, **k: {}
pl.utilities = util
sys.modules["pytorch_lightning"] = pl
sys.modules["pytorch_lightning.utilities"] = util

from hippie import model as ref_model            # noqa: E402  (the reference)
from hippie import backbones as ref_backbones    # noqa: E402
from oracle import cvae_oracle as O              # noqa: E402

torch.set_num_threads(4)


class inject_eps:
    """Patch torch.randn_like so reparameterize() (model.py:48) uses our eps."""

    def __init__(self, eps):
        self.eps = eps

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda t, *a, **k: self.eps.to(t.dtype)
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


def filled(module, salt=0):
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        val = O.fill_value(k, tuple(v.shape), salt)
        new[k] = torch.from_numpy(np.ascontiguousarray(val)).to(v.dtype)
    module.load_state_dict(new)
    return module


def tstats(t):
    t = t.detach().double()
    return np.array([float(t.sum()), float(t.norm()), float(t.abs().max())])


def summarize(named):
    """per-tensor (sum, l2, maxabs) table + names."""
    names = list(named.keys())
    return names, np.stack([tstats(named[k]) for k in names])


SMALL_FULL = ("encoder.conv1.weight", "encoder.bn1.weight", "encoder.layer1.0.conv1.weight",
              "encoder.linear.bias", "encoder_fc.0.weight", "z_log_var.weight", "source_embedding.weight",
              "decoder.conv1.conv.weight", "decoder.conv1.conv.bias", "decoder.linear_out.bias",
              "decoder.layer2.1.shortcut.0.conv.bias", "decoder_fc.3.weight")


def unimodal_case(tag, z, L, B, with_class, beta, clip, steps, lr, wd=0.01, salt=0):
    net = filled(ref_model.hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5), salt)
    mod = ref_model.hippieUnimodalEmbeddingModelCVAE(net, learning_rate=lr, weight_decay=wd, beta=beta)
    out = {}
    x, src, cls, eps = O.synth_inputs(B, L, z, salt=salt)
    labels = torch.stack([cls, src], dim=1) if with_class else src

    # taps through forward hooks: every block output + stem
    taps = {}
    hooks = []
    for name, m in net.named_modules():
        if isinstance(m, (ref_backbones.BasicBlockEnc, ref_backbones.BasicBlockDec)):
            hooks.append(m.register_forward_hook(lambda mod_, i, o, name=name: taps.__setitem__(name + ".out", o.detach().clone())))
    hooks.append(net.encoder.register_forward_hook(lambda m_, i, o: taps.__setitem__("enc_h", o.detach().clone())))

    # eval-mode forward first (does not touch running stats)
    mod.eval()
    with torch.no_grad(), inject_eps(eps):
        e_enc, e_mu, e_lv, e_dec = mod((x, labels))
    out.update(eval_enc=e_enc.numpy(), eval_mu=e_mu.numpy(), eval_logvar=e_lv.numpy(), eval_dec=e_dec.numpy())
    taps.clear()

    mod.train()
    opt = mod.configure_optimizers()
    traj = []
    for s in range(1, steps + 1):
        opt.zero_grad()
        with inject_eps(eps):
            if s == 1:
                enc, mu, lv, dec = mod((x, labels))
                out.update(enc=enc.detach().numpy(), mu=mu.detach().numpy(), logvar=lv.detach().numpy(), dec=dec.detach().numpy())
                tn, tv = summarize(taps)
                out["tap_names"] = np.array(tn)
                out["tap_stats"] = tv
                # redo as a training step from the same pre-step state: restore BN buffers
                filled_sd = {k: torch.from_numpy(np.ascontiguousarray(O.fill_value(k, tuple(v.shape), salt))).to(v.dtype)
                             for k, v in net.state_dict().items() if O.is_buffer(k)}
                net.load_state_dict(filled_sd, strict=False)
            for h in hooks:
                h.remove()
            hooks = []
            loss = mod.training_step((x, labels), 0)
        loss.backward()
        if s == 1:
            out["scalars"] = np.array([mod.logged["train_loss"], mod.logged["train_mse_loss"], mod.logged["train_kl_loss"]])
            g = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters()}
            gn, gv = summarize(g)
            out["grad_names"] = np.array(gn)
            out["grad_stats"] = gv
            out["grad_none"] = np.array([k for k, p in net.named_parameters() if p.grad is None])
            for k in SMALL_FULL:
                out["grad_full." + k] = g[k].detach().numpy().copy()
        if clip is not None:
            norm = torch.nn.utils.clip_grad_norm_(net.parameters(), clip)
            if s == 1:
                out["grad_total_norm"] = np.array([float(norm)])
        opt.step()
        if s in (1, steps):
            sd = net.state_dict()
            pn, pv = summarize({k: v for k, v in sd.items() if v.dtype.is_floating_point})
            out[f"state_names"] = np.array(pn)
            out[f"state_stats_step{s}"] = pv
            for k in SMALL_FULL:
                out[f"param_step{s}." + k] = sd[k].detach().numpy().copy()
            out[f"scalars_step{s}"] = np.array([mod.logged["train_loss"], mod.logged["train_mse_loss"], mod.logged["train_kl_loss"]])
        traj.append([mod.logged["train_loss"], mod.logged["train_mse_loss"], mod.logged["train_kl_loss"]])
    out["scalars_traj"] = np.array(traj)
    st = opt.state_dict()["state"]
    out["adam_steps"] = np.array([float(v["step"]) for v in st.values()])
    out["n_adam_states"] = np.array([len(st)])
    np.savez_compressed(os.path.join(HERE, f"unimodal_{tag}.npz"), **out)
    print("wrote", tag, {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.size < 8})


def multimodal_case(tag, z, L1, L2, B, beta, w1, w2, steps, lr, salt=0):
    net = filled(ref_model.MultiModalCVAE(z_dim=z, output_size_wave=L1, output_size_isi=L2, class_hidden_dim=5,
                                           num_sources=5, num_classes=5), salt)
    mod = ref_model.MultiModalCVAETrainModule(net, learning_rate=lr, weight_decay=0.01, beta=beta, mod1_weight=w1, mod2_weight=w2)
    x1, src, cls, eps = O.synth_inputs(B, L1, z, salt=salt, name="x1")
    x2, _, _, _ = O.synth_inputs(B, L2, z, salt=salt, name="x2")
    out = {}
    mod.train()
    opt = mod.configure_optimizers()
    for s in range(1, steps + 1):
        opt.zero_grad()
        with inject_eps(eps):
            loss = mod.training_step((x1, x2, src), 0)
        loss.backward()
        if s == 1:
            out["scalars"] = np.array([mod.logged["train_loss"], mod.logged["train_mse_loss1"], mod.logged["train_mse_loss2"], mod.logged["train_kl_loss"]])
            g = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters()}
            gn, gv = summarize(g)
            out["grad_names"] = np.array(gn)
            out["grad_stats"] = gv
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)   # multimodal trainer clips (scripts/...:701)
        opt.step()
    sd = net.state_dict()
    pn, pv = summarize({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    out["state_names"] = np.array(pn)
    out[f"state_stats_step{steps}"] = pv
    mod.eval()
    with torch.no_grad(), inject_eps(eps):
        enc, mu, lv, d1, d2 = mod((x1, x2, src))
    out.update(eval_enc=enc.numpy(), eval_mu=mu.numpy(), eval_logvar=lv.numpy(), eval_dec1=d1.numpy(), eval_dec2=d2.numpy())
    np.savez_compressed(os.path.join(HERE, f"multimodal_{tag}.npz"), **out)
    print("wrote multimodal", tag)


def manifests():
    man = {}
    net = ref_model.hippieUnimodalCVAE(z_dim=10, output_size=50, class_hidden_dim=5, num_sources=5, num_classes=5)
    man["unimodal_z10_o50"] = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
    net = ref_model.MultiModalCVAE(z_dim=10, output_size_wave=50, output_size_isi=100, class_hidden_dim=5, num_sources=5, num_classes=5)
    man["multimodal_z10_o50_100"] = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
    man["n_params_unimodal"] = sum(p.numel() for p in ref_model.hippieUnimodalCVAE(10, 50, 5, 5, 5).parameters())
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f)
    # shape-only self test of the reference (backbones.py:156-165)
    ref_backbones.test_decoder()
    print("manifest:", len(man["unimodal_z10_o50"]), len(man["multimodal_z10_o50_100"]), man["n_params_unimodal"])


C5_CASE = dict(tag="z64_L256_32_B8", z=64, L1=256, L2=32, B=8, beta=1.0, w1=1.0, w2=1.0, steps=1, lr=1e-3, salt=8)

if __name__ == "__main__":
    if sys.argv[1:] == ["c5"]:          # only the BASELINE config-5 shape (multimodal, z=64, wave 256 + time 32) at a tiny batch
        multimodal_case(**C5_CASE)
        sys.exit(0)
    manifests()
    # wave model: pretrain style (1-D labels -> source only), no clip (scripts/...:200-207)
    unimodal_case("wave_z10_L50_B16", z=10, L=50, B=16, with_class=False, beta=1.0, clip=None, steps=3, lr=1e-3)
    # time model: clip 1.0 (scripts/...:215-223)
    unimodal_case("time_z10_L100_B16_clip", z=10, L=100, B=16, with_class=False, beta=1.0, clip=1.0, steps=3, lr=1e-3)
    # supervised style: [B,2] labels (class, source), beta 0.5, script-default z=5
    unimodal_case("wave_z5_L50_B12_cls", z=5, L=50, B=12, with_class=True, beta=0.5, clip=1.0, steps=2, lr=1e-4, salt=3)
    # synthetic widths (config 3 shapes at tiny batch)
    unimodal_case("wave_z32_L256_B8", z=32, L=256, B=8, with_class=False, beta=1.0, clip=None, steps=1, lr=1e-3, salt=5)
    unimodal_case("time_z32_L32_B8", z=32, L=32, B=8, with_class=False, beta=1.0, clip=None, steps=1, lr=1e-3, salt=6)
    # trajectories at a small learning rate: Adam moves every element by ~lr per step whatever its gradient,
    # so at lr=1e-3 elements whose gradient is rounding noise make the loss path chaotic; at 1e-6 it is not
    unimodal_case("wave_z10_L50_B32_traj", z=10, L=50, B=32, with_class=False, beta=1.0, clip=None, steps=6, lr=1e-6, salt=9)
    unimodal_case("time_z10_L100_B32_traj_clip", z=10, L=100, B=32, with_class=False, beta=1.0, clip=1.0, steps=6, lr=1e-6, salt=10)
    multimodal_case("z10_B12", z=10, L1=50, L2=100, B=12, beta=1.0, w1=1.0, w2=0.5, steps=2, lr=1e-3, salt=7)
    multimodal_case(**C5_CASE)
