"""Golden fixture for constructor initialisation under the reference's seeding (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_init.py

scripts/train_model_with_multimodal.py seeds once (`torch.manual_seed(42)`, :78), splits the pretrain pool with
`random_split` (:144-147, consumes the global generator), then builds the wave and the time `hippieUnimodalCVAE`
(:169-176, their constructors draw every Conv/Linear/Embedding parameter from the same generator, in construction
order).  This script does exactly that with the REAL reference classes (behind the usual in-memory stand-in for the
absent pytorch_lightning) and stores per-tensor checksums — sum, sum of |.|, the first three and the last value — of
both models, for the pool sizes the tests use, plus the generator's next draw after construction (pins the number of
draws consumed).  Data only; the reference itself never travels."""
import os
import sys
import types

import numpy as np
import torch
from torch.utils.data import random_split

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
pl = types.ModuleType("pytorch_lightning")


class LightningModule(torch.nn.Module):
    pass


pl.LightningModule = LightningModule
util = types.ModuleType("pytorch_lightning.utilities")
util.grad_norm = lambda *a, **k: {}
pl.utilities = util
sys.modules["pytorch_lightning"] = pl
sys.modules["pytorch_lightning.utilities"] = util
from hippie import model as ref_model            # noqa: E402  (the reference)


def checks(t):
    f = t.detach().double().reshape(-1)
    return np.array([float(f.sum()), float(f.abs().sum()), float(f[0]), float(f[min(1, len(f) - 1)]), float(f[min(2, len(f) - 1)]), float(f[-1])])


def record(out, tag, net):
    names = [k for k, _ in net.named_parameters()]
    out[tag + ".names"] = np.array(names)
    out[tag + ".checks"] = np.stack([checks(p) for _, p in net.named_parameters()])


out = {}
for n_pool, z in ((280, 5), (3797, 10), (15631, 10)):
    torch.manual_seed(42)
    n_tr = int(0.8 * n_pool)
    tr, te = random_split(list(range(n_pool)), [n_tr, n_pool - n_tr])
    wave = ref_model.hippieUnimodalCVAE(z_dim=z, output_size=50, class_hidden_dim=5, num_sources=5, num_classes=5)
    time = ref_model.hippieUnimodalCVAE(z_dim=z, output_size=100, class_hidden_dim=5, num_sources=5, num_classes=5)
    tag = f"pool{n_pool}_z{z}"
    out[tag + ".train_idx_head"] = np.array(tr.indices[:16])
    record(out, tag + ".wave", wave)
    record(out, tag + ".time", time)
    out[tag + ".next_draw"] = np.array(int(torch.empty((), dtype=torch.int64).random_().item()))
torch.manual_seed(42)
mm = ref_model.MultiModalCVAE(z_dim=10, output_size_wave=50, output_size_isi=100, class_hidden_dim=5, num_sources=5, num_classes=5)
record(out, "multimodal_z10", mm)
out["multimodal_z10.next_draw"] = np.array(int(torch.empty((), dtype=torch.int64).random_().item()))
np.savez_compressed(os.path.join(HERE, "init_seed42.npz"), **out)
print("wrote init_seed42.npz", {k: v.shape for k, v in out.items() if k.endswith("checks")})
