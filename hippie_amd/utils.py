"""get_embeddings of the reference (scripts/utils.py:75-101) for GPU modules."""
from __future__ import annotations

import numpy as np
import torch


def get_embeddings(dataloader_wave, dataloader_time, wave_model, time_model):
    """zip the two loaders, forward both modules, keep `enc` (output[0]), row-standardise with the
    unbiased std (torch.std, ddof=1), concatenate wave | time.  Returns three numpy arrays."""
    emb_w, emb_t = [], []
    for (wave, label_wave), (time, label_time) in zip(dataloader_wave, dataloader_time):
        assert (label_wave == label_time).all()
        # only output[0] is used (scripts/utils.py:84-85): modules that offer it run the encoder half alone
        e_wave = (wave_model.embed((wave, label_wave)) if hasattr(wave_model, "embed") else wave_model((wave, label_wave))[0]).clone()
        e_time = (time_model.embed((time, label_time)) if hasattr(time_model, "embed") else time_model((time, label_time))[0]).clone()
        e_wave = (e_wave - e_wave.mean(dim=1)[:, None]) / e_wave.std(dim=1)[:, None]
        e_time = (e_time - e_time.mean(dim=1)[:, None]) / e_time.std(dim=1)[:, None]
        emb_w.append(e_wave)
        emb_t.append(e_time)
    ew = torch.cat(emb_w, dim=0).detach().cpu().numpy()
    et = torch.cat(emb_t, dim=0).detach().cpu().numpy()
    return ew, et, np.concatenate([ew, et], axis=1)


def get_embeddings_multimodal(loader, model):
    """scripts/train_model_with_multimodal.py:22-35: forward every (data1, data2, labels) batch through the multimodal module in eval
    mode, keep `enc` (output[0]) and row-standardise it with numpy's POPULATION std (np.std, ddof=0 — unlike get_embeddings above,
    which follows scripts/utils.py and torch.std).  Returns one numpy array [N, z]."""
    model.eval()
    out = []
    for sample in loader:
        e = model(sample)[0].detach().cpu().numpy()
        e = (e - np.mean(e, axis=1, keepdims=True)) / np.std(e, axis=1, keepdims=True)
        out.extend(e)
    return np.array(out)
