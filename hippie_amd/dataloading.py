"""Preprocessing of the reference's datasets on the GPU (hippie/dataloading.py:62-104).

`EphysDatasetLabeled` keeps the reference's constructor and item contract — item i is
`(tensor [1, 50] or [1, 100], label)` — but resamples the WHOLE table once, at construction, with one
HIP launch per modality (HP_OP_RESAMPLE_LINEAR) instead of a Python `F.interpolate` per item."""
from __future__ import annotations

import numpy as np
import torch

from . import program as P
from .program import Ref


def resample_on_device(x: torch.Tensor, L: int, log1p: bool = False) -> torch.Tensor:
    """x: [N, W] float32 CUDA tensor -> [N, L] via libhippie_hip.so."""
    if not x.is_cuda:
        raise P.HipEngineError("resample_on_device needs a CUDA tensor; there is no CPU fallback")
    x = x.contiguous().float()
    N, W = x.shape
    out = torch.empty(N, L, dtype=torch.float32, device=x.device)
    ol = P.OpList()
    # two separate allocations: address them relative to the lower one (space WS = a byte range)
    base = min(x.data_ptr(), out.data_ptr())
    ol.add(P.RESAMPLE_LINEAR, 1 if log1p else 0, [N, W, L], (), [Ref(P.WS, x.data_ptr() - base), Ref(P.WS, out.data_ptr() - base)])
    P.run_single_op(ol.array()[0], [base, 0, 0, 0, 0, 0], torch.cuda.current_stream(x.device).cuda_stream)
    return out


class EphysDatasetLabeled:
    def __init__(self, waveforms, isi_dists, labels, mode, normalize=True, device="cuda"):
        assert mode in ("wave", "time")
        waveforms, isi_dists, labels = np.array(waveforms), np.array(isi_dists), np.array(labels)
        assert len(waveforms) == len(isi_dists)
        assert len(waveforms) == len(labels)
        if normalize:
            # the reference's normalize=True path calls np.min on a torch tensor and raises TypeError
            # (dataloading.py:84); every script passes normalize=False
            raise TypeError("normalize=True is broken in the reference (np.min on a tensor); pass normalize=False")
        self.mode = mode
        self.labels = torch.as_tensor(labels).long().to(device)
        if mode == "wave":
            src = torch.as_tensor(waveforms.astype(np.float32)).to(device)
            self.data = resample_on_device(src, 50)
        else:
            src = torch.as_tensor(isi_dists.astype(np.float32)).to(device)
            self.data = resample_on_device(src, 100, log1p=True)

    def __getitem__(self, idx):
        return self.data[idx].view(1, -1), self.labels[idx]

    def __len__(self):
        return len(self.labels)

    def batches(self, batch_size, indices=None, shuffle=False, generator=None):
        """Batches as a DataLoader over this dataset would yield them: ([B,1,L], labels[B])."""
        n = len(self)
        idx = torch.arange(n) if indices is None else torch.as_tensor(indices)
        if shuffle:
            idx = idx[torch.randperm(len(idx), generator=generator)]
        idx = idx.to(self.data.device)
        for i in range(0, len(idx), batch_size):
            j = idx[i: i + batch_size]
            yield self.data.index_select(0, j).unsqueeze(1), self.labels.index_select(0, j)


class BalancedBatchSampler:
    """Class-balanced oversampling index stream — counterpart of hippie/dataloading.py:107-151, used by the
    supervised stage (scripts/train_model_with_multimodal.py:398, :865).

    Host-side index logic (no device work): classes are keyed in order of first appearance, every class is
    topped up to the size of the largest one with `random.choice` draws from its own (growing) list — the
    same calls on Python's global `random` in the same order as the reference, so a seeded run yields the
    same permutation — and iteration interleaves the classes round-robin.  `len()` = max class size x classes.
    Pass the result to `EphysDatasetLabeled.batches(batch_size, indices=list(sampler))`."""

    def __init__(self, dataset, labels=None):
        import random
        if labels is None:
            raise Exception("You should pass the tensor of labels to the constructor as second argument")
        self.labels = labels
        self.dataset = {}
        for idx in range(len(dataset)):
            lab = self._get_label(dataset, idx)
            self.dataset.setdefault(lab, []).append(idx)
        self.balanced_max = max((len(v) for v in self.dataset.values()), default=0)
        for pool in self.dataset.values():
            while len(pool) < self.balanced_max:
                pool.append(random.choice(pool))
        self.keys = list(self.dataset)
        self.currentkey = 0
        self.indices = [-1] * len(self.keys)

    def _get_label(self, dataset, idx, labels=None):
        return self.labels[idx].item()

    def __iter__(self):
        # the cursor state is kept on the object, as in the reference (:130-135)
        while self.keys and self.indices[self.currentkey] < self.balanced_max - 1:
            self.indices[self.currentkey] += 1
            yield self.dataset[self.keys[self.currentkey]][self.indices[self.currentkey]]
            self.currentkey = (self.currentkey + 1) % len(self.keys)
        self.indices = [-1] * len(self.keys)

    def __len__(self):
        return self.balanced_max * len(self.keys)
