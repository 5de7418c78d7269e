#!/usr/bin/env python3
"""Counterpart of the reference's scripts/train_model_with_multimodal.py (unimodal branch, :36-347):
pretrain the wave and time cVAEs on every *other* dataset -> label-free fine-tune on the target dataset
(lr/10, 10 %/90 % split) -> `pretraining_{dataset}_{waveform,isi,joint}_embeddings.csv`.

Kept from the reference: the 22 CLI flags and defaults (:40-67), the dataset -> source-id map and pool
exclusion rules including the "justacellular" typo that never matches (:81-101), `pd.read_csv` WITHOUT
index_col (:117-121, the index column of some CSVs becomes feature 0), `dropna(axis=1)` on the fine-tune
tables (:234-236), `torch.manual_seed(42)` then `random_split` of the index list (:78,136-147), loaders
shuffle=True for pretrain / False for fine-tune, the wave trainer without and the time trainer with
gradient clipping (:200-224), `val_loss`-monitored top-1 checkpoints reloaded before fine-tuning
(:227-230), embeddings = row-standardised `enc` of the *train* fine-tune subset (:313-315), CSV layout
(:329-343).  wandb is replaced by JSONL logs.

The supervised stage (:349-616) is behind `--supervised` (the reference runs it unconditionally and dies with
KeyError("label") on every shipped labels.csv, whose only column is "0"; `--label-column 0` selects that one):
LabelEncoder + `random_split` at --train-val-split, `[class, source]` label pairs, class-balanced oversampling
(BalancedBatchSampler) at --supervised-batch-size, fresh models with `num_classes = #train classes` loaded from the
PRETRAIN checkpoints minus `class_embedding` (strict=False), lr/10, gradient clipping on BOTH trainers, best
checkpoint (+ optimiser state) reloaded, embeddings at batch 128, the 5..19-neighbour kNN sweep (scikit-learn, CPU)
and the `{dataset}_{waveform,isi,joint}_{knn,embeddings}.csv` files.

`--model-type multimodal` (:618-790): ONE MultiModalCVAE over (waveform, isi) pairs, pretrain -> label-free fine-tune at lr / 10 ->
`pretraining_{dataset}_joint_embeddings.csv`.  The reference's own branch cannot get past its first dataset — it asks
EphysDatasetLabeled for mode="both", which the class asserts away (hippie/dataloading.py:67) — so this is the branch with the dataset
its training_step expects, (waveform, isi, label) rows (main_multimodal); `--supervised` runs its supervised stage (:790-960) too.
"""
import argparse
import json
import os
import sys

import numpy as np
import pandas as pd
import torch
import torch.distributed as dist
from torch.utils.data import random_split

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hippie_amd.dataloading import EphysDatasetLabeled                     # noqa: E402
from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE   # noqa: E402
from hippie_amd.trainer import Trainer, fit_concurrently                    # noqa: E402
from hippie_amd.utils import get_embeddings                                 # noqa: E402

DATASET_FILES = {
    "extracellular-mouse-a1": 1,
    "cellexplorer-celltype": 3,
    "cellexplorer-area": 3,
    "juxtacellular-mouse-s1-celltype": 4,
    "juxtacellular-mouse-s1-area": 4,
    "allenscope-neuropixel": 3,
    "neonatal-mouse-brain-slice": 2,
}


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--z_dim", type=int, default=5)
    p.add_argument("--weight-decay", type=float, default=0.01)
    p.add_argument("--learning-rate", type=float, default=0.001)
    p.add_argument("--beta", type=float, default=1)
    p.add_argument("--dataset", type=str, default="cellexplorer-celltype")
    p.add_argument("--upload-model", action="store_true")
    p.add_argument("--wandb-tag", type=str, default="no_curr_sup_pretrain_data")
    p.add_argument("--project", type=str, default="HIPPIE")
    p.add_argument("--finetune-without-labels", type=bool, default=True)
    p.add_argument("--pretrain-max-epochs", type=int, default=1)
    p.add_argument("--finetune-max-epochs", type=int, default=1)
    p.add_argument("--supervised-max-epochs", type=int, default=1)
    p.add_argument("--batch-size", type=int, default=512)
    p.add_argument("--supervised-batch-size", type=int, default=64)
    p.add_argument("--early-stopping-patience", type=int, default=30)
    p.add_argument("--gradient-clip-val", type=float, default=1.0)
    p.add_argument("--train-val-split", type=float, default=0.8)
    p.add_argument("--finetune-split", type=float, default=0.1)
    p.add_argument("--limit-train-batches", type=float, default=None)
    p.add_argument("--limit-val-batches", type=float, default=None)
    p.add_argument("--model-type", type=str, choices=["unimodal", "multimodal"], default="unimodal")
    p.add_argument("--mod1-weight", type=float, default=1.0)
    p.add_argument("--mod2-weight", type=float, default=1.0)
    # additions of this build
    p.add_argument("--data-root", type=str, default="datasets")
    p.add_argument("--output-dir", type=str, default=".")
    p.add_argument("--supervised", action="store_true", help="run the supervised fine-tune + kNN stage (:349-616)")
    p.add_argument("--label-column", type=str, default="label", help='column of labels.csv (the reference reads "label")')
    p.add_argument("--precision", type=str, default="32", choices=["32", "bf16"],
                   help="Trainer(precision=...): 32 = the reference's fp32 arithmetic; bf16 = BASELINE config 2's reduced-precision mode "
                        "(bf16 MFMA operands, fp32 accumulation / statistics / master weights)")
    p.add_argument("--strategy", type=str, default="auto", choices=["auto", "ddp", "single_device"],
                   help="as pl.Trainer(strategy=...): under torchrun with more than one rank 'auto' and 'ddp' train data-parallel")
    p.add_argument("--sync-batchnorm", action="store_true", help="pl.Trainer(sync_batchnorm=True)")
    p.add_argument("--sequential-fits", action="store_true",
                   help="fit the wave model, then the time model, as the reference does (default: both at once on two HIP streams, "
                        "with the same random draws and therefore the same numbers)")
    return p


def pretrain_pool(dataset):
    files = dict(DATASET_FILES)
    if "justacellular" in dataset:      # sic: the reference's typo, never true for the shipped names
        files.pop("justacellular-mouse-s1-celltype", None)
        files.pop("juxtacellular-mouse-s1-area", None)
    if "cellexplorer" in dataset:
        files.pop("cellexplorer-celltype", None)
        files.pop("cellexplorer-area", None)
    return {k: v for k, v in files.items() if k != dataset}


class _Concat:
    """torch.utils.data.ConcatDataset of preprocessed GPU tables, addressed by global index lists."""

    def __init__(self, parts):
        self.data = torch.cat([p.data for p in parts], dim=0)
        self.labels = torch.cat([p.labels for p in parts], dim=0)

    def loader(self, indices, batch_size, shuffle):
        """DataLoader(Subset(ConcatDataset, indices), batch_size, shuffle) of the reference (:155-166), with torch's own
        DataLoader producing the INDEX batches — so the global generator is consumed exactly as there (one base-seed
        draw per iterator, one more by RandomSampler when shuffling) — and the rows gathered from the HBM tables."""
        indices = list(indices)
        index_loader = torch.utils.data.DataLoader(indices, batch_size=batch_size, shuffle=shuffle)
        table = self

        class _L:
            def __iter__(s):
                for j in index_loader:
                    j = j.to(table.data.device)
                    yield table.data.index_select(0, j).unsqueeze(1), table.labels.index_select(0, j)

            def __len__(s):
                return len(index_loader)

            def frozen(s):
                """one pass with its index draws made now (hippie_amd.trainer.fit_concurrently): the torch DataLoader iterator
                over the index list is created and exhausted here — consuming the global generator exactly as iter(self)
                would — and the rows are gathered later, on the consumer's stream"""
                batches = [j for j in index_loader]

                class _Pass:
                    def __iter__(p):
                        for j in batches:
                            j = j.to(table.data.device, non_blocking=True)
                            yield table.data.index_select(0, j).unsqueeze(1), table.labels.index_select(0, j)

                    def __len__(p):
                        return len(batches)
                return _Pass()

            def shard(s, rank, world, epoch, seed=0):
                """What Lightning's DDP strategy does to this DataLoader: its sampler becomes DistributedSampler(dataset,
                shuffle=<as given>, seed) with set_epoch(epoch) — a seeded permutation padded to a multiple of `world`,
                rank r taking positions r::world — and batch_size stays PER RANK (hippie_amd.parallel.shard_indices)."""
                from hippie_amd.parallel import shard_indices
                pos = shard_indices(len(indices), rank, world, epoch=epoch, seed=seed, shuffle=shuffle)
                mine = torch.as_tensor(indices, dtype=torch.int64)[pos].to(table.data.device)
                for i in range(0, len(mine), batch_size):
                    j = mine[i: i + batch_size]
                    yield table.data.index_select(0, j).unsqueeze(1), table.labels.index_select(0, j)
        return _L()


class _ConcatJoint:
    """The dataset the reference's multimodal branch MEANT to build (scripts/...:638 asks EphysDatasetLabeled for mode="both", which
    its constructor refuses, dataloading.py:67): rows of (waveform[1,50], isi[1,100], label) — what MultiModalCVAETrainModule's
    training_step unpacks (hippie/model.py:454-458) and what EphysDataset's own mode "both" returns (dataloading.py:55-56)."""

    def __init__(self, wave_parts, time_parts):
        self.wave, self.time = _Concat(wave_parts), _Concat(time_parts)
        assert torch.equal(self.wave.labels, self.time.labels)
        self.labels = self.wave.labels

    def loader(self, indices, batch_size, shuffle):
        indices = list(indices)
        index_loader = torch.utils.data.DataLoader(indices, batch_size=batch_size, shuffle=shuffle)
        t = self

        def rows(j):
            j = j.to(t.wave.data.device)
            return t.wave.data.index_select(0, j).unsqueeze(1), t.time.data.index_select(0, j).unsqueeze(1), t.labels.index_select(0, j)

        class _L:
            def __iter__(s):
                for j in index_loader:
                    yield rows(j)

            def __len__(s):
                return len(index_loader)

            def shard(s, rank, world, epoch, seed=0):
                from hippie_amd.parallel import shard_indices
                pos = shard_indices(len(indices), rank, world, epoch=epoch, seed=seed, shuffle=shuffle)
                mine = torch.as_tensor(indices, dtype=torch.int64)[pos]
                for i in range(0, len(mine), batch_size):
                    yield rows(mine[i: i + batch_size])
        return _L()


def _write_csv_rank0(df, path):
    """Under torchrun every rank walks the supervised stage (same split, same sampler stream; BatchNorm running statistics are rank-local
    without --sync-batchnorm, so the ranks' embeddings differ in the last digits): rank 0's file is THE file — like the checkpoints and the
    pretraining CSVs — and a barrier keeps any rank from reading it half-written."""
    if not dist.is_initialized() or dist.get_rank() == 0:
        df.to_csv(path)
    if dist.is_initialized():
        dist.barrier()


def main_multimodal(args, eps_source, rank0):
    """The multimodal branch of the reference script (scripts/train_model_with_multimodal.py:618-790): ONE MultiModalCVAE over
    (waveform, isi) pairs — pretraining on the pool, label-free fine-tuning on the target at a tenth of the learning rate, joint
    embeddings CSV.  As written the reference stops at its first dataset (mode="both" is asserted away, dataloading.py:67); this is
    the branch with that dataset returning (waveform, isi, label), everything else as there: gradient clipping on, --beta and the
    modality weights honoured (:670-677), a single pretraining_{dataset}_joint_embeddings.csv standardised with np.std (:22-35)."""
    from hippie_amd.model import MultiModalCVAE, MultiModalCVAETrainModule
    from hippie_amd.utils import get_embeddings_multimodal
    num_sources = max(DATASET_FILES.values()) + 1
    wave_parts, time_parts = [], []
    for folder, sid in pretrain_pool(args.dataset).items():
        wf = pd.read_csv(os.path.join(args.data_root, folder, "waveforms.csv")).to_numpy()
        isi = pd.read_csv(os.path.join(args.data_root, folder, "isi_dist.csv")).to_numpy()
        source = np.full((wf.shape[0]), sid)
        print(f"Folder {folder} has shapes {wf.shape} and {isi.shape}")
        wave_parts.append(EphysDatasetLabeled(wf, isi, source, mode="wave", normalize=False))
        time_parts.append(EphysDatasetLabeled(wf, isi, source, mode="time", normalize=False))
    joint = _ConcatJoint(wave_parts, time_parts)
    n = len(joint.labels)
    prop = args.train_val_split
    train_idx, test_idx = random_split(list(range(n)), [int(prop * n), n - int(prop * n)])
    bs = args.batch_size

    def trainer(epochs, tag):
        return Trainer(max_epochs=epochs, gradient_clip_val=args.gradient_clip_val, patience=args.early_stopping_patience,
                       default_root_dir=os.path.join(args.output_dir, "checkpoints", f"joint_{tag}"),
                       logger_path=os.path.join(args.output_dir, f"joint_{tag}_log.jsonl"),
                       precision=args.precision, strategy=args.strategy, sync_batchnorm=args.sync_batchnorm)

    net = MultiModalCVAE(z_dim=args.z_dim, output_size_wave=50, output_size_isi=100, class_hidden_dim=5, num_sources=num_sources, num_classes=5)
    net.set_eps_source(eps_source)
    mod = MultiModalCVAETrainModule(net, learning_rate=args.learning_rate, weight_decay=args.weight_decay, beta=args.beta,
                                    mod1_weight=args.mod1_weight, mod2_weight=args.mod2_weight)
    tr = trainer(args.pretrain_max_epochs, "pretrain")
    tr.fit(mod, joint.loader(train_idx, bs, True), joint.loader(test_idx, bs, False))
    if tr.best_model_path:
        mod.load_state_dict(torch.load(tr.best_model_path, weights_only=False)["state_dict"])
    wf_ft = pd.read_csv(os.path.join(args.data_root, args.dataset, "waveforms.csv")).dropna(axis=1).to_numpy()
    isi_ft = pd.read_csv(os.path.join(args.data_root, args.dataset, "isi_dist.csv")).dropna(axis=1).to_numpy()
    label_ft = np.full((wf_ft.shape[0]), DATASET_FILES[args.dataset])
    ft = _ConcatJoint([EphysDatasetLabeled(wf_ft, isi_ft, label_ft, mode="wave", normalize=False)],
                      [EphysDatasetLabeled(wf_ft, isi_ft, label_ft, mode="time", normalize=False)])
    m = len(label_ft)
    if args.finetune_without_labels:
        p2 = args.finetune_split
        tr_i, te_i = random_split(list(range(m)), [int(p2 * m), m - int(p2 * m)])
        mod = MultiModalCVAETrainModule(mod.model, learning_rate=(1 / 10) * args.learning_rate, weight_decay=args.weight_decay, beta=args.beta,
                                        mod1_weight=args.mod1_weight, mod2_weight=args.mod2_weight)
        tr2 = trainer(args.finetune_max_epochs, "finetune")
        tr2.fit(mod, ft.loader(tr_i, bs, False), ft.loader(te_i, bs, False))
        if tr2.best_model_path:
            mod.load_state_dict(torch.load(tr2.best_model_path, weights_only=False)["state_dict"])
        emb = get_embeddings_multimodal(ft.loader(te_i, bs, False), mod)       # the held-out 90 % (:775)
    else:
        emb = get_embeddings_multimodal(ft.loader(range(m), bs, False), mod)
    path = os.path.join(args.output_dir, f"pretraining_{args.dataset}_joint_embeddings.csv")
    if rank0:
        pd.DataFrame({"embeddings": list(emb)}).to_csv(path)
        with open(os.path.join(args.output_dir, "run_config.json"), "w") as f:
            json.dump(vars(args), f)
    paths = {"joint": path}
    if args.supervised:
        paths.update(supervised_stage_multimodal(args, num_sources, tr.best_model_path, trainer, eps_source))
    return paths


def supervised_stage_multimodal(args, num_sources, joint_path, trainer, eps_source=None):
    """scripts/train_model_with_multimodal.py:790-960 (multimodal branch; the same stage as supervised_stage below for ONE joint model):
    LabelEncoder + random_split, [class, source] label pairs, BalancedBatchSampler at --supervised-batch-size, a fresh MultiModalCVAE with
    num_classes = #train classes loaded from the PRETRAIN checkpoint minus class_embedding (strict=False), lr / 10, gradient clipping,
    best checkpoint reloaded, embeddings at batch 128, the 5..19-neighbour kNN sweep, {dataset}_joint_{knn,embeddings}.csv."""
    from sklearn.metrics import balanced_accuracy_score, confusion_matrix
    from sklearn.neighbors import KNeighborsClassifier
    from sklearn.preprocessing import LabelEncoder
    from hippie_amd.dataloading import BalancedBatchSampler
    from hippie_amd.model import MultiModalCVAE, MultiModalCVAETrainModule
    from hippie_amd.utils import get_embeddings_multimodal
    dataset = args.dataset
    root = os.path.join(args.data_root, dataset)
    sup_wf = pd.read_csv(os.path.join(root, "waveforms.csv")).to_numpy()
    sup_isi = pd.read_csv(os.path.join(root, "isi_dist.csv")).to_numpy()
    if os.path.exists(os.path.join(root, "labels.csv")):
        raw = pd.read_csv(os.path.join(root, "labels.csv"))[args.label_column].values
        le = LabelEncoder().fit(raw)
        sup_labels = le.transform(raw)
    else:
        print(f"No labels.csv found for {dataset}")
        sup_labels = np.zeros(len(sup_wf))
        le = LabelEncoder().fit(sup_labels)
    n = len(sup_wf)
    train_size = int(args.train_val_split * n)
    tr_i, va_i = random_split(list(range(n)), [train_size, n - train_size])
    tr_i, va_i = list(tr_i), list(va_i)
    label_train, label_val = sup_labels[tr_i], sup_labels[va_i]
    num_class_labels = len(np.unique(label_train))
    sid = DATASET_FILES[dataset]

    def table(idx, lab):
        pair = np.vstack((lab, sid * np.ones_like(lab))).T             # [class, source], model.py:456-458
        return _ConcatJoint([EphysDatasetLabeled(sup_wf[idx], sup_isi[idx], pair, mode="wave", normalize=False)],
                            [EphysDatasetLabeled(sup_wf[idx], sup_isi[idx], pair, mode="time", normalize=False)])
    tr_t, va_t = table(tr_i, label_train), table(va_i, label_val)
    net = MultiModalCVAE(z_dim=args.z_dim, output_size_wave=50, output_size_isi=100, class_hidden_dim=5, num_sources=num_sources,
                         num_classes=num_class_labels)
    net.set_eps_source(eps_source)
    mod = MultiModalCVAETrainModule(net, learning_rate=(1 / 10) * args.learning_rate, weight_decay=args.weight_decay, beta=args.beta,
                                    mod1_weight=args.mod1_weight, mod2_weight=args.mod2_weight)
    if joint_path:
        sd = torch.load(joint_path, weights_only=False)["state_dict"]
        sd.pop("model.class_embedding.weight")
        mod.load_state_dict(sd, strict=False)
    order = list(BalancedBatchSampler(range(len(tr_i)), torch.as_tensor(label_train)))
    sb = args.supervised_batch_size
    tr = trainer(args.supervised_max_epochs, "supervised")
    tr.fit(mod, tr_t.loader(order, sb, False), va_t.loader(range(len(va_i)), sb, False))
    if tr.best_model_path:
        mod.load_state_dict(torch.load(tr.best_model_path, weights_only=False)["state_dict"])
    mod.eval()
    e_tr = get_embeddings_multimodal(tr_t.loader(range(len(tr_i)), 128, False), mod)
    e_va = get_embeddings_multimodal(va_t.loader(range(len(va_i)), sb, False), mod)
    neighbor_options = list(range(5, 20))
    acc = []
    for k in neighbor_options:
        knn = KNeighborsClassifier(n_neighbors=min(k, len(e_tr))).fit(e_tr, label_train)
        acc.append(balanced_accuracy_score(label_val, knn.predict(e_va)))
    best_k = neighbor_options[int(np.argmax(acc))]
    pred = KNeighborsClassifier(n_neighbors=min(best_k, len(e_tr))).fit(e_tr, label_train).predict(e_va)
    confusion_matrix(label_val, pred)
    out = {"joint_balanced_accuracy": acc}
    out["joint_knn"] = os.path.join(args.output_dir, f"{dataset}_joint_knn.csv")
    _write_csv_rank0(pd.DataFrame({"pred": le.inverse_transform(pred.astype(int)), "true": le.inverse_transform(label_val.astype(int))}), out["joint_knn"])
    e_all = get_embeddings_multimodal(table(list(range(n)), sup_labels).loader(range(n), 128, False), mod)
    df = pd.DataFrame(e_all)
    df["label"] = le.inverse_transform(sup_labels.astype(int))
    out["joint_supervised_embeddings"] = os.path.join(args.output_dir, f"{dataset}_joint_embeddings.csv")
    _write_csv_rank0(df, out["joint_supervised_embeddings"])
    return out


def main(argv=None, eps_source=None):
    """eps_source: callable(engine) -> [B, z] noise for every forward of the networks built here (parity tests run the whole
    pipeline on a prescribed sequence); None = torch's device generator, as the reference's torch.randn_like."""
    args = build_parser().parse_args(argv)
    os.makedirs(args.output_dir, exist_ok=True)
    # under a launcher (torchrun: one process per GPU) this is a DDP run, as Lightning would make it (scripts/...:200-207):
    # RCCL ("nccl") over xGMI, rank r on GPU LOCAL_RANK; rank 0 writes checkpoints, logs and CSVs
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and args.strategy != "single_device" and not dist.is_initialized():
        # (HIPPIE_SINGLE_DEVICE: rehearsal of a multi-rank run with every rank on GPU 0 — the test suite's one-GPU box)
        torch.cuda.set_device(0 if os.environ.get("HIPPIE_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("HIPPIE_DIST_BACKEND", "nccl"))
    rank0 = not dist.is_initialized() or dist.get_rank() == 0
    torch.manual_seed(42)
    if args.model_type == "multimodal":
        return main_multimodal(args, eps_source, rank0)
    num_sources = max(DATASET_FILES.values()) + 1
    wave_parts, time_parts = [], []
    for folder, sid in pretrain_pool(args.dataset).items():
        wf = pd.read_csv(os.path.join(args.data_root, folder, "waveforms.csv")).to_numpy()
        isi = pd.read_csv(os.path.join(args.data_root, folder, "isi_dist.csv")).to_numpy()
        source = np.full((wf.shape[0]), sid)
        print(f"Folder {folder} has shapes {wf.shape} and {isi.shape}")
        wave_parts.append(EphysDatasetLabeled(wf, isi, source, mode="wave", normalize=False))
        time_parts.append(EphysDatasetLabeled(wf, isi, source, mode="time", normalize=False))
    all_wave, all_time = _Concat(wave_parts), _Concat(time_parts)
    n = len(all_wave.labels)
    print(f"Total waveforms {n} and total isi {len(all_time.labels)}")
    prop = args.train_val_split
    train_idx, test_idx = random_split(list(range(n)), [int(prop * n), n - int(prop * n)])
    bs = args.batch_size

    def trainer(kind, epochs, clip, tag):
        return Trainer(max_epochs=epochs, gradient_clip_val=clip, patience=args.early_stopping_patience,
                       default_root_dir=os.path.join(args.output_dir, "checkpoints", f"{kind}_{tag}"),
                       logger_path=os.path.join(args.output_dir, f"{kind}_{tag}_log.jsonl"),
                       precision=args.precision, strategy=args.strategy, sync_batchnorm=args.sync_batchnorm)

    def fit(kind, module, train_loader, val_loader, epochs, clip, tag):
        tr = trainer(kind, epochs, clip, tag)
        tr.fit(module, train_loader, val_loader)
        return tr

    def fit_both(wave_job, time_job, epochs, tag):
        """the wave fit (no clipping) and the time fit (gradient clipping) of one stage (scripts/...:200-224, :290-309): the
        reference runs them one after the other; here concurrently on two streams with the same random draws, unless
        --sequential-fits (or a multi-rank run, whose collectives must be issued in one order on every rank)"""
        trw, trt = trainer("wave", epochs, None, tag), trainer("time", epochs, args.gradient_clip_val, tag)
        jobs = [(trw,) + wave_job, (trt,) + time_job]
        if args.sequential_fits or world > 1:
            for tr, mod, tl, vl in jobs:
                tr.fit(mod, tl, vl)
        else:
            fit_concurrently(jobs)
        return trw, trt

    wave_net = hippieUnimodalCVAE(z_dim=args.z_dim, output_size=50, class_hidden_dim=5, num_sources=num_sources, num_classes=5)
    time_net = hippieUnimodalCVAE(z_dim=args.z_dim, output_size=100, class_hidden_dim=5, num_sources=num_sources, num_classes=5)
    wave_net.set_eps_source(eps_source)
    time_net.set_eps_source(eps_source)
    # the unimodal branch builds its modules WITHOUT beta= (scripts/...:178-183), so --beta is ignored here too
    wave_mod = hippieUnimodalEmbeddingModelCVAE(wave_net, learning_rate=args.learning_rate, weight_decay=args.weight_decay)
    time_mod = hippieUnimodalEmbeddingModelCVAE(time_net, learning_rate=args.learning_rate, weight_decay=args.weight_decay)
    trw, trt = fit_both((wave_mod, all_wave.loader(train_idx, bs, True), all_wave.loader(test_idx, bs, False)),
                        (time_mod, all_time.loader(train_idx, bs, True), all_time.loader(test_idx, bs, False)),
                        args.pretrain_max_epochs, "pretrain")
    if trw.best_model_path:
        wave_mod.load_state_dict(torch.load(trw.best_model_path, weights_only=False)["state_dict"])
    if trt.best_model_path:
        time_mod.load_state_dict(torch.load(trt.best_model_path, weights_only=False)["state_dict"])

    # ---- label-free fine-tuning on the target dataset ----
    wf_ft = pd.read_csv(os.path.join(args.data_root, args.dataset, "waveforms.csv")).dropna(axis=1).to_numpy()
    isi_ft = pd.read_csv(os.path.join(args.data_root, args.dataset, "isi_dist.csv")).dropna(axis=1).to_numpy()
    label_ft = np.full((wf_ft.shape[0]), DATASET_FILES[args.dataset])
    ft_wave = _Concat([EphysDatasetLabeled(wf_ft, isi_ft, label_ft, mode="wave", normalize=False)])
    ft_time = _Concat([EphysDatasetLabeled(wf_ft, isi_ft, label_ft, mode="time", normalize=False)])
    m = len(label_ft)
    if args.finetune_without_labels:
        p2 = args.finetune_split
        tr_i, te_i = random_split(list(range(m)), [int(p2 * m), m - int(p2 * m)])
        wave_mod = hippieUnimodalEmbeddingModelCVAE(wave_mod.model, learning_rate=(1 / 10) * args.learning_rate, weight_decay=args.weight_decay)
        time_mod = hippieUnimodalEmbeddingModelCVAE(time_mod.model, learning_rate=(1 / 10) * args.learning_rate, weight_decay=args.weight_decay)
        lw, lt = ft_wave.loader(tr_i, bs, False), ft_time.loader(tr_i, bs, False)
        fit_both((wave_mod, lw, ft_wave.loader(te_i, bs, False)), (time_mod, lt, ft_time.loader(te_i, bs, False)),
                 args.finetune_max_epochs, "finetune")
    else:
        lw, lt = ft_wave.loader(range(m), bs, False), ft_time.loader(range(m), bs, False)
    wave_mod.eval()
    time_mod.eval()
    ew, et, joint = get_embeddings(lw, lt, wave_mod, time_mod)
    paths = {}
    for name, emb in (("waveform", ew), ("isi", et), ("joint", joint)):
        path = os.path.join(args.output_dir, f"pretraining_{args.dataset}_{name}_embeddings.csv")
        if rank0:
            pd.DataFrame({"embeddings": list(emb)}).to_csv(path)
        paths[name] = path
    if rank0:
        with open(os.path.join(args.output_dir, "run_config.json"), "w") as f:
            json.dump(vars(args), f)
    if args.supervised:
        paths.update(supervised_stage(args, num_sources, trw.best_model_path, trt.best_model_path, fit, eps_source))
    return paths


def supervised_stage(args, num_sources, wave_path, time_path, fit, eps_source=None):
    """scripts/train_model_with_multimodal.py:349-616 (unimodal branch)."""
    from sklearn.metrics import balanced_accuracy_score, confusion_matrix
    from sklearn.neighbors import KNeighborsClassifier
    from sklearn.preprocessing import LabelEncoder
    from hippie_amd.dataloading import BalancedBatchSampler
    dataset = args.dataset
    root = os.path.join(args.data_root, dataset)
    sup_wf = pd.read_csv(os.path.join(root, "waveforms.csv")).to_numpy()
    sup_isi = pd.read_csv(os.path.join(root, "isi_dist.csv")).to_numpy()
    if os.path.exists(os.path.join(root, "labels.csv")):
        labels = pd.read_csv(os.path.join(root, "labels.csv"))
        raw = labels[args.label_column].values            # KeyError on the shipped files unless --label-column 0, as in the reference
        le = LabelEncoder().fit(raw)
        sup_labels = le.transform(raw)
    else:
        print(f"No labels.csv found for {dataset}")
        sup_labels = np.zeros(len(sup_wf))
        le = LabelEncoder().fit(sup_labels)
    n = len(sup_wf)
    train_size = int(args.train_val_split * n)
    tr_i, va_i = random_split(list(range(n)), [train_size, n - train_size])
    tr_i, va_i = list(tr_i), list(va_i)
    label_train, label_val = sup_labels[tr_i], sup_labels[va_i]
    num_class_labels = len(np.unique(label_train))
    sid = DATASET_FILES[dataset]

    def tables(idx, lab):
        pair = np.vstack((lab, sid * np.ones_like(lab))).T             # [class, source], model.py:97-99
        return (_Concat([EphysDatasetLabeled(sup_wf[idx], sup_isi[idx], pair, mode="wave", normalize=False)]),
                _Concat([EphysDatasetLabeled(sup_wf[idx], sup_isi[idx], pair, mode="time", normalize=False)]))
    tr_wave, tr_time = tables(tr_i, label_train)
    va_wave, va_time = tables(va_i, label_val)
    sampler = BalancedBatchSampler(range(len(tr_i)), torch.as_tensor(label_train))
    order = list(sampler)                                                # the same stream every epoch, shared by both loaders
    sb = args.supervised_batch_size
    mods = {}
    for kind, L, ckpt, tr_t, va_t in (("wave", 50, wave_path, tr_wave, va_wave), ("time", 100, time_path, tr_time, va_time)):
        net = hippieUnimodalCVAE(z_dim=args.z_dim, output_size=L, class_hidden_dim=5, num_sources=num_sources,
                                 num_classes=num_class_labels)
        net.set_eps_source(eps_source)
        mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=(1 / 10) * args.learning_rate, weight_decay=args.weight_decay)
        if ckpt:
            sd = torch.load(ckpt, weights_only=False)["state_dict"]
            sd.pop("model.class_embedding.weight")
            mod.load_state_dict(sd, strict=False)
        tr = fit(kind, mod, tr_t.loader(order, sb, False), va_t.loader(range(len(va_i)), sb, False),
                 args.supervised_max_epochs, args.gradient_clip_val, "supervised")
        if tr.best_model_path:
            best = torch.load(tr.best_model_path, weights_only=False)
            mod.load_state_dict(best["state_dict"])
            mod.optimizer.load_state_dict(best["optimizer_states"][0])
        mod.eval()
        mods[kind] = mod
    emb_tr = get_embeddings(tr_wave.loader(range(len(tr_i)), 128, False), tr_time.loader(range(len(tr_i)), 128, False),
                            mods["wave"], mods["time"])
    emb_va = get_embeddings(va_wave.loader(range(len(va_i)), sb, False), va_time.loader(range(len(va_i)), sb, False),
                            mods["wave"], mods["time"])
    out = {}
    neighbor_options = list(range(5, 20))
    for name, e_tr, e_va in zip(("waveform", "isi", "joint"), emb_tr, emb_va):
        acc = []
        for k in neighbor_options:
            knn = KNeighborsClassifier(n_neighbors=min(k, len(e_tr))).fit(e_tr, label_train)
            acc.append(balanced_accuracy_score(label_val, knn.predict(e_va)))
        best_k = neighbor_options[int(np.argmax(acc))]
        pred = KNeighborsClassifier(n_neighbors=min(best_k, len(e_tr))).fit(e_tr, label_train).predict(e_va)
        confusion_matrix(label_val, pred)
        path = os.path.join(args.output_dir, f"{dataset}_{name}_knn.csv")
        _write_csv_rank0(pd.DataFrame({"pred": le.inverse_transform(pred.astype(int)), "true": le.inverse_transform(label_val.astype(int))}), path)
        out[name + "_knn"] = path
        out[name + "_balanced_accuracy"] = acc
    all_wave, all_time = tables(list(range(n)), sup_labels)
    embs = get_embeddings(all_wave.loader(range(n), 128, False), all_time.loader(range(n), 128, False), mods["wave"], mods["time"])
    for name, e in zip(("waveform", "isi", "joint"), embs):
        df = pd.DataFrame(e)
        df["label"] = le.inverse_transform(sup_labels.astype(int))
        path = os.path.join(args.output_dir, f"{dataset}_{name}_embeddings.csv")
        _write_csv_rank0(df, path)
        out[name + "_supervised_embeddings"] = path
    return out


if __name__ == "__main__":
    main()
