"""GPU: the pipeline script counterpart on a synthetic data root (same folder / CSV conventions as the
reference's datasets/, incl. a CSV with an unnamed index column)."""
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_root(tmp, rng):
    spec = {"extracellular-mouse-a1": (60, 40, 51, False), "neonatal-mouse-brain-slice": (90, 50, 100, False),
            "cellexplorer-celltype": (50, 46, 100, True), "cellexplorer-area": (40, 46, 100, False),
            "juxtacellular-mouse-s1-celltype": (30, 351, 100, False), "juxtacellular-mouse-s1-area": (30, 351, 100, True),
            "allenscope-neuropixel": (70, 60, 100, False)}
    for name, (n, w, h, with_index) in spec.items():
        d = tmp / name
        d.mkdir()
        wf = pd.DataFrame(rng.standard_normal((n, w)))
        isi = pd.DataFrame(np.abs(rng.standard_normal((n, h))) * 0.01)
        wf.to_csv(d / "waveforms.csv", index=with_index)
        isi.to_csv(d / "isi_dist.csv", index=with_index)
    return spec


def test_pipeline_end_to_end(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pretrain_pipeline as pp
    assert set(pp.pretrain_pool("cellexplorer-celltype")) == {"extracellular-mouse-a1", "juxtacellular-mouse-s1-celltype",
                                                               "juxtacellular-mouse-s1-area", "allenscope-neuropixel",
                                                               "neonatal-mouse-brain-slice"}
    # the reference's typo: a juxtacellular target only removes itself from the pool
    assert "juxtacellular-mouse-s1-area" in pp.pretrain_pool("juxtacellular-mouse-s1-celltype")
    a = pp.build_parser().parse_args([])
    assert (a.z_dim, a.batch_size, a.learning_rate, a.gradient_clip_val, a.finetune_split) == (5, 512, 0.001, 1.0, 0.1)
    rng = np.random.default_rng(0)
    data = tmp_path / "datasets"
    data.mkdir()
    spec = make_root(data, rng)
    out = tmp_path / "out"
    paths = pp.main(["--dataset", "cellexplorer-celltype", "--data-root", str(data), "--output-dir", str(out),
                     "--batch-size", "64", "--pretrain-max-epochs", "2", "--z_dim", "5"])
    n_ft = int(0.1 * spec["cellexplorer-celltype"][0])
    for name, width in (("waveform", 5), ("isi", 5), ("joint", 10)):
        df = pd.read_csv(paths[name])
        assert list(df.columns) == ["Unnamed: 0", "embeddings"] and len(df) == n_ft
        vec = np.array(df["embeddings"][0].strip("[]").split(), dtype=float)
        assert vec.shape == (width,) and np.isfinite(vec).all()
    logs = [l for l in os.listdir(out) if l.endswith("_log.jsonl")]
    assert sorted(logs) == ["time_finetune_log.jsonl", "time_pretrain_log.jsonl", "wave_finetune_log.jsonl", "wave_pretrain_log.jsonl"]
    assert any(f.endswith(".ckpt") for f in os.listdir(out / "checkpoints" / "wave_pretrain"))
    # random_split right after manual_seed(42) is the reference's split (same torch function, same seed)
    torch.manual_seed(42)
    from torch.utils.data import random_split
    n = sum(v[0] for k, v in spec.items() if k in pp.pretrain_pool("cellexplorer-celltype"))
    tr, te = random_split(list(range(n)), [int(0.8 * n), n - int(0.8 * n)])
    assert len(tr) == int(0.8 * n) and len(set(tr.indices) & set(te.indices)) == 0


@pytest.mark.parametrize("extra", [[], ["--precision", "bf16", "--mod1-weight", "0.7", "--mod2-weight", "1.3", "--beta", "0.5"]])
def test_multimodal_pipeline_branch(tmp_path, extra):
    """--model-type multimodal: the branch the reference script means to run (:618-790; as shipped it stops at
    EphysDatasetLabeled(mode="both"), dataloading.py:67): one MultiModalCVAE pretrained on the pool, fine-tuned without labels at lr / 10,
    the held-out 90 % of the target embedded and row-standardised with np.std (get_embeddings_multimodal, :22-35)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pretrain_pipeline as pp
    rng = np.random.default_rng(1)
    data = tmp_path / "datasets"
    data.mkdir()
    spec = make_root(data, rng)
    out = tmp_path / "out"
    paths = pp.main(["--dataset", "cellexplorer-area", "--model-type", "multimodal", "--data-root", str(data), "--output-dir", str(out),
                     "--batch-size", "64", "--pretrain-max-epochs", "2", "--z_dim", "6"] + extra)
    assert set(paths) == {"joint"}
    m = spec["cellexplorer-area"][0]
    df = pd.read_csv(paths["joint"])
    assert list(df.columns) == ["Unnamed: 0", "embeddings"] and len(df) == m - int(0.1 * m)
    emb = np.array([np.array(v.strip("[]").split(), dtype=float) for v in df["embeddings"]])
    assert emb.shape == (m - int(0.1 * m), 6) and np.isfinite(emb).all()
    np.testing.assert_allclose(emb.mean(1), 0, atol=1e-6)
    np.testing.assert_allclose(emb.std(1), 1, atol=1e-5)          # population std, as np.std in the reference's helper
    logs = sorted(l for l in os.listdir(out) if l.endswith("_log.jsonl"))
    assert logs == ["joint_finetune_log.jsonl", "joint_pretrain_log.jsonl"]
    import json
    recs = [json.loads(l) for l in open(out / "joint_pretrain_log.jsonl")]
    assert len(recs) == 2 and all(np.isfinite(r["val_loss"]) for r in recs) and "train_mse_loss1" in recs[0] and "train_mse_loss2" in recs[0]
    assert any(f.endswith(".ckpt") for f in os.listdir(out / "checkpoints" / "joint_pretrain"))


def test_supervised_stage(tmp_path):
    """--supervised: class labels, balanced sampler, class_embedding re-created, kNN + embedding CSVs."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pretrain_pipeline as pp
    rng = np.random.default_rng(3)
    data = tmp_path / "datasets"
    data.mkdir()
    spec = make_root(data, rng)
    n = spec["extracellular-mouse-a1"][0]
    names = np.array(["PV", "SST", "PYR"])[rng.choice(3, size=n, p=[0.6, 0.3, 0.1])]
    pd.DataFrame({"0": names}).to_csv(data / "extracellular-mouse-a1" / "labels.csv")
    out = tmp_path / "out"
    common = ["--dataset", "extracellular-mouse-a1", "--data-root", str(data), "--output-dir", str(out), "--batch-size", "64",
              "--supervised-batch-size", "16", "--z_dim", "5", "--supervised"]
    with pytest.raises(KeyError):                       # the reference's behaviour on the shipped column name
        pp.main(common)
    paths = pp.main(common + ["--label-column", "0"])
    for name, width in (("waveform", 5), ("isi", 5), ("joint", 10)):
        knn = pd.read_csv(paths[name + "_knn"])
        assert list(knn.columns) == ["Unnamed: 0", "pred", "true"] and len(knn) == n - int(0.8 * n)
        assert set(knn["pred"]) <= {"PV", "SST", "PYR"}
        emb = pd.read_csv(paths[name + "_supervised_embeddings"])
        assert len(emb) == n and list(emb.columns)[-1] == "label" and emb.shape[1] == width + 2
        assert np.isfinite(emb[[str(i) for i in range(width)]].to_numpy()).all()
        assert len(paths[name + "_balanced_accuracy"]) == 15
    assert any(f.endswith(".ckpt") for f in os.listdir(out / "checkpoints" / "wave_supervised"))


def test_supervised_stage_multimodal(tmp_path):
    """--model-type multimodal --supervised (scripts/...:790-960): one joint model, joint kNN + embedding CSVs."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pretrain_pipeline as pp
    rng = np.random.default_rng(4)
    data = tmp_path / "datasets"
    data.mkdir()
    spec = make_root(data, rng)
    n = spec["extracellular-mouse-a1"][0]
    names = np.array(["PV", "SST", "PYR"])[rng.choice(3, size=n, p=[0.5, 0.3, 0.2])]
    pd.DataFrame({"0": names}).to_csv(data / "extracellular-mouse-a1" / "labels.csv")
    out = tmp_path / "out"
    paths = pp.main(["--dataset", "extracellular-mouse-a1", "--model-type", "multimodal", "--data-root", str(data), "--output-dir", str(out),
                     "--batch-size", "64", "--supervised-batch-size", "16", "--z_dim", "5", "--supervised", "--label-column", "0"])
    knn = pd.read_csv(paths["joint_knn"])
    assert list(knn.columns) == ["Unnamed: 0", "pred", "true"] and len(knn) == n - int(0.8 * n) and set(knn["pred"]) <= {"PV", "SST", "PYR"}
    emb = pd.read_csv(paths["joint_supervised_embeddings"])
    assert len(emb) == n and list(emb.columns)[-1] == "label" and emb.shape[1] == 5 + 2
    assert np.isfinite(emb[[str(i) for i in range(5)]].to_numpy()).all() and len(paths["joint_balanced_accuracy"]) == 15
    assert any(f.endswith(".ckpt") for f in os.listdir(out / "checkpoints" / "joint_supervised"))


def test_inference_script_roundtrip(tmp_path):
    """checkpoints written by the Trainer -> scripts/inference.py -> embedding CSVs in the reference's layout"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import inference
    from hippie_amd.model import hippieUnimodalCVAE, hippieUnimodalEmbeddingModelCVAE
    from hippie_amd.trainer import Trainer
    rng = np.random.default_rng(1)
    data = tmp_path / "datasets"
    data.mkdir()
    make_root(data, rng)
    z = 8
    ck = {}
    for kind, L in (("wave", 50), ("time", 100)):
        net = hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5)
        mod = hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-4)
        tr = Trainer(max_epochs=1, default_root_dir=str(tmp_path / kind))
        tr.current_epoch = 0
        net.engine(4, False)                       # materialise parameters
        path = str(tmp_path / f"{kind}.ckpt")
        tr.save_checkpoint(mod, path)
        ck[kind] = path
    out = inference.main(["--z_dim", str(z), "--dataset", "cellexplorer-area", "--wave-checkpoint", ck["wave"],
                          "--time-checkpoint", ck["time"], "--output-dir", str(tmp_path / "emb"), "--data-root", str(data)])
    for name, width in (("waveform", z), ("isi", z), ("joint", 2 * z)):
        df = pd.read_csv(out[name])
        assert list(df.columns) == [str(i) for i in range(width)] + ["label", "label_name"]
        assert len(df) == 40 and (df["label_name"] == "unknown").all()
        v = df[[str(i) for i in range(width)]].to_numpy()
        assert np.isfinite(v).all()
    w = pd.read_csv(out["waveform"])[[str(i) for i in range(z)]].to_numpy()
    np.testing.assert_allclose(w.mean(1), 0, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------------
# Numerical pin of the pipeline: the script counterpart against the CPU oracle walking through the SAME
# steps — seed 42, reference-order initialisation, torch's DataLoader index streams, sanity validation, epochs of
# training steps with validation + top-1 checkpoint reload, label-free fine-tune with a fresh AdamW at lr/10,
# row-standardised `enc` embeddings — on a prescribed reparameterisation-noise sequence.
class _Noise:
    """closed-form noise for the k-th forward of one MODEL of a run (same values on both sides).  One stream per model: the
    pipeline fits the wave and the time model concurrently (hippie_amd.trainer.fit_concurrently), so a stream shared by both
    would be consumed in a scheduling-dependent order."""

    def __init__(self, name=""):
        self.k, self.name = 0, name

    def draw(self, B, z):
        from oracle import cvae_oracle as O
        e = sum(O.unit_noise(f"pipe.{self.name}.eps{i}", B * z, salt=self.k) for i in range(4)) * (3.0 / 4.0) ** 0.5
        self.k += 1
        return torch.from_numpy(e.reshape(B, z)).float()


def keyed_noise():
    """({"wave": _Noise, "time": _Noise}, eps_source for pretrain_pipeline.main): the engine's input length tells the model"""
    noise = {"wave": _Noise("wave"), "time": _Noise("time")}
    return noise, (lambda eng: noise["wave" if eng.cfg.output_size == 50 else "time"].draw(eng.B, eng.cfg.z_dim).to(eng.device))


def _oracle_fit(om, noise, train_batches, val_batches, epochs, lr, clip):
    """hippie_amd.trainer.Trainer.fit + hippieUnimodalEmbeddingModelCVAE on an OracleModel: returns the state of the best
    validation epoch (what the pipeline reloads)."""
    import copy

    def val(limit=None):
        losses = []
        for i, (x, lab) in enumerate(val_batches()):
            if limit is not None and i >= limit:
                break
            eps = noise.draw(x.shape[0], om_z(om))
            with torch.no_grad():
                outs = om.forward((x, lab, None), eps, training=False)
                losses.append(float(om.losses((x, lab, None), outs)[0]))
        return sum(losses) / max(1, len(losses))

    val(2)                                     # sanity check: two validation batches
    best, best_state = float("inf"), None
    for _ in range(epochs):
        for x, lab in train_batches():
            eps = noise.draw(x.shape[0], om_z(om))
            om.train_step((x, lab, None), eps, lr=lr, weight_decay=0.01, beta=1.0, clip=clip)
        v = val()
        if v < best:
            best, best_state = v, {k: t.detach().clone() for k, t in om.state.items()}
    return best_state


def om_z(om):
    return om.state["z_mean.bias"].shape[0]


def test_pipeline_numbers_match_the_oracle_walking_the_same_steps(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pretrain_pipeline as pp
    from torch.utils.data import DataLoader, random_split
    from hippie_amd.engine import Engine
    from hippie_amd.model import reference_init_state
    from hippie_amd import planner
    from oracle import cvae_oracle as O
    from oracle import preproc
    rng = np.random.default_rng(1)
    data = tmp_path / "datasets"
    data.mkdir()
    spec = make_root(data, rng)
    out = tmp_path / "out"
    z, bs, lr, epochs = 5, 64, 1e-6, 2        # a small lr: Adam moves every element by +-lr per step, noise-gradient elements included
    argv = ["--dataset", "cellexplorer-celltype", "--data-root", str(data), "--output-dir", str(out), "--batch-size", str(bs),
            "--pretrain-max-epochs", str(epochs), "--finetune-max-epochs", "2", "--z_dim", str(z), "--learning-rate", str(lr)]
    # ---- the product pipeline on the prescribed noise
    noise, eps_source = keyed_noise()
    paths = pp.main(argv, eps_source=eps_source)
    n_forwards = {k: v.k for k, v in noise.items()}
    # ---- the oracle, same steps
    torch.manual_seed(42)
    pool = pp.pretrain_pool("cellexplorer-celltype")
    waves, times, labels = [], [], []
    for folder, sid in pool.items():
        wf = pd.read_csv(data / folder / "waveforms.csv").to_numpy()
        isi = pd.read_csv(data / folder / "isi_dist.csv").to_numpy()
        w, t = preproc.preprocess(wf, isi)
        waves.append(w), times.append(t), labels.append(np.full(len(w), sid))
    tabs = {"wave": torch.from_numpy(np.concatenate(waves)), "time": torch.from_numpy(np.concatenate(times))}
    lab = torch.from_numpy(np.concatenate(labels)).long()
    n = len(lab)
    tr_idx, te_idx = random_split(list(range(n)), [int(0.8 * n), n - int(0.8 * n)])
    oms = {}
    for kind, L in (("wave", 50), ("time", 100)):         # construction order = RNG order
        om = O.OracleModel("unimodal", z, L)
        om.load(reference_init_state(planner.ModelCfg("unimodal", z, L, 0, 5, 5, 5)))
        for k in om.state:                                 # BatchNorm buffers of a fresh module
            if k.endswith("running_mean"):
                om.state[k].zero_()
            elif k.endswith("running_var"):
                om.state[k].fill_(1.0)
            elif k.endswith("num_batches_tracked"):
                om.state[k].zero_()
        oms[kind] = om
    noise2 = {"wave": _Noise("wave"), "time": _Noise("time")}

    def batches_of(tab, idx, shuffle):
        loader = DataLoader(list(idx), batch_size=bs, shuffle=shuffle)
        return lambda: ((tab[j], lab_of[j]) for j in loader)

    lab_of = lab
    for kind, clip in (("wave", None), ("time", 1.0)):
        best = _oracle_fit(oms[kind], noise2[kind], batches_of(tabs[kind], tr_idx, True), batches_of(tabs[kind], te_idx, False), epochs, lr, clip)
        oms[kind].load(best)
        for k in ("num_batches_tracked",):
            pass
    # label-free fine-tune on the target dataset: fresh AdamW at lr/10, loaders without shuffling
    wf = pd.read_csv(data / "cellexplorer-celltype" / "waveforms.csv").dropna(axis=1).to_numpy()
    isi = pd.read_csv(data / "cellexplorer-celltype" / "isi_dist.csv").dropna(axis=1).to_numpy()
    w, t = preproc.preprocess(wf, isi)
    ft = {"wave": torch.from_numpy(w), "time": torch.from_numpy(t)}
    m = len(w)
    lab_of = torch.full((m,), 3, dtype=torch.long)
    ft_tr, ft_te = random_split(list(range(m)), [int(0.1 * m), m - int(0.1 * m)])
    for kind, clip in (("wave", None), ("time", 1.0)):
        om = oms[kind]
        om.exp_avg, om.exp_avg_sq, om.step_count = {}, {}, 0
        _oracle_fit(om, noise2[kind], batches_of(ft[kind], ft_tr, False), batches_of(ft[kind], ft_te, False), 2, lr / 10, clip)
        # (the pipeline does NOT reload a checkpoint after fine-tuning: the embeddings come from the final weights)
    assert {k: v.k for k, v in noise2.items()} == n_forwards, "the oracle walked a different number of forwards than the pipeline"
    embs = {}
    for kind in ("wave", "time"):
        rows = []
        for x, lb in batches_of(ft[kind], ft_tr, False)():
            with torch.no_grad():
                enc = oms[kind].forward((x, lb, None), torch.zeros(x.shape[0], z), training=False)[0]
            rows.append((enc - enc.mean(dim=1)[:, None]) / enc.std(dim=1)[:, None])
        embs[kind] = torch.cat(rows).numpy()
    want = {"waveform": embs["wave"], "isi": embs["time"], "joint": np.concatenate([embs["wave"], embs["time"]], axis=1)}
    for name, ref in want.items():
        df = pd.read_csv(paths[name])
        got = np.stack([np.array(v.strip("[]").split(), dtype=float) for v in df["embeddings"]])
        assert got.shape == ref.shape
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print(f"[pipeline parity] {name}: max rel err {err:.2e}")
        assert err <= 1e-4, (name, err)
