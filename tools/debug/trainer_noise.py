"""Spread of repeated identical Trainer.fit runs (hipGraph replay path): python tools/debug/trainer_noise.py REPS MODE
MODE: cpu = host batches (pageable), gpu = device batches, pinned = pinned host batches, eager = host batches, no graphs."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_gpu_model import batches, O, hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE, Trainer

import functools
from hippie_amd import planner
if len(sys.argv) > 3 and sys.argv[3] == "det":       # ordered weight-gradient slabs instead of fp32 atomics
    _T = planner.TrainCfg
    planner.TrainCfg = functools.partial(_T, deterministic_wgrad=True)

z, L = 10, 50
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mode = sys.argv[2] if len(sys.argv) > 2 else "cpu"
om = O.OracleModel("unimodal", z, L, salt=4)
train = batches(70, 32, L, z, seed=1)
val = batches(40, 32, L, z, seed=2)
if mode == "gpu":
    train = [(a.cuda(), b.cuda()) for a, b in train]
    val = [(a.cuda(), b.cuda()) for a, b in val]
elif mode == "pinned":
    train = [(a.pin_memory(), b.pin_memory()) for a, b in train]
    val = [(a.pin_memory(), b.pin_memory()) for a, b in val]
ref = None
for rep in range(reps):
    net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
    net.load_state_dict({k: v.detach() for k, v in om.state.items()})
    net.use_graph = mode != "eager"
    mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-5, weight_decay=0.01)
    steps = []
    orig = mod._record
    def rec(store, loss, orig=orig):
        steps.append(loss.value)
        orig(store, loss)
    mod._record = rec
    torch.manual_seed(123)
    tr = Trainer(max_epochs=2, gradient_clip_val=1.0, enable_checkpointing=False, sync_every_step=False)
    tr.fit(mod, train, val)
    v = torch.stack(steps).double().cpu().numpy()
    if ref is None:
        ref = v
    print(f"{mode} rep {rep}: per-step rel dev vs rep 0: " + " ".join(f"{abs(a / b - 1):.1e}" for a, b in zip(v, ref)), flush=True)
