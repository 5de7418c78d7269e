"""get_embeddings of the reference (scripts/utils.py:75-101) for GPU modules."""
from __future__ import annotations

import numpy as np
import torch


def _side_streams(wave_model, time_model, wave, label_wave, time, label_time):
    """Two HIP streams on which the two modules' encoder passes overlap (hippie_amd.streams: measured, cached per device), or None
    when the modules are not this package's GPU modules."""
    try:
        if not (wave.is_cuda and time.is_cuda and hasattr(wave_model, "embed") and hasattr(time_model, "embed")):
            return None
        from .streams import pick_concurrent_streams
        engs = [m.model.engine(int(x.shape[0]), lab.ndim == 2) for m, x, lab in ((wave_model, wave, label_wave), (time_model, time, label_time))]
        return pick_concurrent_streams(engs, wave.device)
    except Exception as ex:        # the sequential path below is always correct — but ~30 % slower (685 k -> 465 k units/s): say so, once
        global _WARNED
        if not _WARNED:
            _WARNED = True
            import warnings
            warnings.warn(f"get_embeddings: could not pick two concurrent streams ({type(ex).__name__}: {ex}); the wave and the time "
                          "encoder passes run one after the other", RuntimeWarning, stacklevel=3)
        return None


_WARNED = False


def get_embeddings(dataloader_wave, dataloader_time, wave_model, time_model):
    """zip the two loaders, forward both modules, keep `enc` (output[0]), row-standardise with the
    unbiased std (torch.std, ddof=1), concatenate wave | time.  Returns three numpy arrays.
    The two modules are independent: on the GPU their encoder passes run side by side on two streams (same numbers; the
    reference runs them one after the other)."""
    emb_w, emb_t = [], []
    streams = None
    first = True
    for (wave, label_wave), (time, label_time) in zip(dataloader_wave, dataloader_time):
        assert (label_wave == label_time).all()
        if first:
            streams, first = _side_streams(wave_model, time_model, wave, label_wave, time, label_time), False
        outs = []
        cur = torch.cuda.current_stream(wave.device) if streams else None
        for k, (model, x, lab) in enumerate(((wave_model, wave, label_wave), (time_model, time, label_time))):
            if streams:
                streams[k].wait_stream(cur)                 # the batch was produced on the caller's stream
                x.record_stream(streams[k]), lab.record_stream(streams[k])
            with (torch.cuda.stream(streams[k]) if streams else _null()):
                # only output[0] is used (scripts/utils.py:84-85): modules that offer it run the encoder half alone
                e = (model.embed((x, lab)) if hasattr(model, "embed") else model((x, lab))[0]).clone()
                e = (e - e.mean(dim=1)[:, None]) / e.std(dim=1)[:, None]
            outs.append(e)
        emb_w.append(outs[0])
        emb_t.append(outs[1])
    if streams:
        cur = torch.cuda.current_stream(emb_w[0].device)
        for s_ in streams:
            cur.wait_stream(s_)
        for t in emb_w + emb_t:
            t.record_stream(cur)                            # allocated on a side stream, consumed (and freed) on the caller's
    ew = torch.cat(emb_w, dim=0).detach().cpu().numpy()
    et = torch.cat(emb_t, dim=0).detach().cpu().numpy()
    return ew, et, np.concatenate([ew, et], axis=1)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


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
