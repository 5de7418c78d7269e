"""CPU oracle for the preprocessing row — TEST INFRASTRUCTURE ONLY.

numpy restatement of what EphysDatasetLabeled.__getitem__ does per item (hippie/dataloading.py:74-96):
float32 cast, log(isi + 1), F.interpolate(size=50|100, mode="linear") with align_corners=False (ATen's
area_pixel_compute_source_index: src = scale*(dst+0.5)-0.5 clamped at 0; upsample_linear1d).
Pinned by tests/golden/datasets_first8.npz (generated from the reference by make_golden_data.py)."""
import numpy as np


def resample_linear(x, L):
    """x: [N, W] float32 -> [N, L] float32."""
    x = np.asarray(x, dtype=np.float32)
    W = x.shape[1]
    scale = np.float32(W) / np.float32(L)
    src = scale * (np.arange(L, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5)
    src = np.maximum(src, np.float32(0))
    x0 = np.minimum(src.astype(np.int64), W - 1)
    x1 = np.minimum(x0 + 1, W - 1)
    w1 = (src - x0.astype(np.float32)).astype(np.float32)
    w0 = (np.float32(1) - w1).astype(np.float32)
    return (w0[None, :] * x[:, x0] + w1[None, :] * x[:, x1]).astype(np.float32)


def preprocess(waveforms, isi_dists):
    """-> (wave [N,1,50], isi [N,1,100]) as the reference's dataset yields them with normalize=False."""
    w = resample_linear(np.asarray(waveforms, dtype=np.float32), 50)
    t = np.log(np.asarray(isi_dists, dtype=np.float32) + np.float32(1)).astype(np.float32)
    t = resample_linear(t, 100)
    return w[:, None, :], t[:, None, :]


def reference_style_batch(rows, L, log1p, idx):
    """One DataLoader batch the way the reference builds it, for the CPU baseline's loader-inclusive figure: per ITEM
    (EphysDatasetLabeled.__getitem__, hippie/dataloading.py:74-101) a float32 tensor, log(x + 1) for the ISI histogram, `F.interpolate(size=L,
    mode="linear")` on a [1, 1, W] view, a [1, L] result — then the default collate's torch.stack over the batch (DataLoader(batch_size=512),
    scripts/train_model_with_multimodal.py:155-166, num_workers=0).  rows: [N, W] float array; idx: the batch's indices.  -> [B, 1, L] float32."""
    import torch
    import torch.nn.functional as F
    items = []
    for i in idx:
        x = torch.as_tensor(rows[int(i)], dtype=torch.float32)
        if log1p:
            x = torch.log(x + 1)
        items.append(F.interpolate(x.view(1, 1, -1), size=L, mode="linear").view(1, L))
    return torch.stack(items, 0)
