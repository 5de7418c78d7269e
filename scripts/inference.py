#!/usr/bin/env python3
"""Counterpart of the reference's scripts/inference_from_trained_model.py:58-163 (embedding export; the
UMAP plots of :165-220 are visualisation only and not reproduced).

Same CLI (--z_dim default 64, --dataset, --wave-checkpoint, --time-checkpoint, --output-dir), same table
reading (dropna(axis=1), no index_col), batch 128 loaders without shuffling, `num_sources = 5`, class
embedding dropped from the checkpoint when its row count differs, eval-mode forward, row-standardised `enc`
embeddings, and `{output_dir}/{dataset}_{waveform,isi,joint}_embeddings.csv` with columns
`0..z-1,label,label_name` written with index=False.  Labels come from metadata.csv's `label` column when
present; otherwise the dummy label 0 / "unknown" (the reference indexes a list with float zeros there and
raises TypeError, :159 — integer zeros are used instead)."""
import argparse
import os
import sys

import numpy as np
import pandas as pd
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hippie_amd.dataloading import EphysDatasetLabeled                                # noqa: E402
from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE     # noqa: E402
from hippie_amd.utils import get_embeddings                                            # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--z_dim", type=int, default=64, required=False)
    ap.add_argument("--dataset", type=str, default="cellexplorer-celltype")
    ap.add_argument("--wave-checkpoint", type=str, required=True)
    ap.add_argument("--time-checkpoint", type=str, required=True)
    ap.add_argument("--output-dir", type=str, default="./embeddings")
    ap.add_argument("--data-root", type=str, default="datasets")
    args = ap.parse_args(argv)
    os.makedirs(args.output_dir, exist_ok=True)
    torch.manual_seed(42)
    base = os.path.join(args.data_root, args.dataset)
    wf = pd.read_csv(os.path.join(base, "waveforms.csv")).dropna(axis=1).to_numpy()
    isi = pd.read_csv(os.path.join(base, "isi_dist.csv")).dropna(axis=1).to_numpy()
    labels, label_names = None, None
    meta = os.path.join(base, "metadata.csv")
    if os.path.exists(meta):
        md = pd.read_csv(meta)
        if "label" in md.columns:
            label_names = list(md["label"].unique())
            labels = md["label"].map({n: i for i, n in enumerate(label_names)}).to_numpy()
    if labels is None:
        labels = np.zeros(wf.shape[0], dtype=np.int64)
        label_names = ["unknown"]
    ds_w = EphysDatasetLabeled(wf, isi, labels, mode="wave", normalize=False)
    ds_t = EphysDatasetLabeled(wf, isi, labels, mode="time", normalize=False)
    num_sources, num_classes = 5, len(np.unique(labels))
    mods = []
    for L, path in ((50, args.wave_checkpoint), (100, args.time_checkpoint)):
        net = hippieUnimodalCVAE(z_dim=args.z_dim, output_size=L, class_hidden_dim=5, num_sources=num_sources, num_classes=num_classes)
        mod = hippieUnimodalEmbeddingModelCVAE(net)
        sd = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
        key = "model.class_embedding.weight"
        if key in sd and sd[key].size(0) != num_classes:
            print("Warning: Class embedding size mismatch. Removing from checkpoint.")
            sd.pop(key)
        mod.load_state_dict(sd, strict=False)
        mod.eval()
        mods.append(mod)
    # the reference feeds the (dummy or real) labels as source ids: 1-D labels -> source_labels (model.py:100)
    ew, et, joint = get_embeddings(ds_w.batches(128), ds_t.batches(128), mods[0], mods[1])
    out = {}
    for name, emb in (("waveform", ew), ("isi", et), ("joint", joint)):
        df = pd.DataFrame(emb)
        df["label"] = labels
        df["label_name"] = pd.Categorical([label_names[int(i)] for i in labels])
        path = os.path.join(args.output_dir, f"{args.dataset}_{name}_embeddings.csv")
        df.to_csv(path, index=False)
        out[name] = path
        print(f"Saved {name} embeddings to {path}")
    return out


if __name__ == "__main__":
    main()
