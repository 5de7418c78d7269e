"""scripts/pretrain_pipeline.py on a synthetic data root over a set of flag combinations (unimodal / multimodal, bf16, sequential fits,
limited batches, early stopping, z 32, another target): prints OK / FAIL per combination.  python tools/pipeline_combo_smoke.py"""
import os, sys, tempfile, pathlib, traceback
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "scripts"); sys.path.insert(0, "tests")
import pretrain_pipeline as pp
from test_gpu_pipeline import make_root
combos = {
  "unimodal default": [],
  "multimodal": ["--model-type", "multimodal"],
  "multimodal bf16": ["--model-type", "multimodal", "--precision", "bf16"],
  "unimodal bf16 sequential": ["--precision", "bf16", "--sequential-fits"],
  "unimodal limit batches": ["--limit-train-batches", "0.5", "--limit-val-batches", "0.5"],
  "multimodal weights": ["--model-type", "multimodal", "--mod1-weight", "0.7", "--mod2-weight", "1.3", "--beta", "0.5"],
  "unimodal 3 epochs patience 1": ["--pretrain-max-epochs", "3", "--early-stopping-patience", "1"],
  "z 32": ["--z_dim", "32"],
  "dataset juxta": ["--dataset", "juxtacellular-mouse-s1-area"],
}
for name, extra in combos.items():
    with tempfile.TemporaryDirectory() as t:
        t = pathlib.Path(t); data = t / "datasets"; data.mkdir()
        make_root(data, np.random.default_rng(0))
        base = ["--dataset", "cellexplorer-celltype", "--data-root", str(data), "--output-dir", str(t / "out"), "--batch-size", "64", "--z_dim", "5"]
        try:
            paths = pp.main(base + extra)
            print("OK  ", name, sorted(paths)[:4], flush=True)
        except BaseException as ex:
            print("FAIL", name, repr(ex)[:300], flush=True)
            traceback.print_exc()
