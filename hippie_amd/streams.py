"""Choosing HIP streams that really run side by side.

The wave and the time model step on two HIP streams (hippie/… trains them one after the other, scripts/train_model_with_multimodal.py:208,224;
here their kernels share the GPU).  ROCm multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, and the firmware
spreads those queues over a few pipes: which queue a `torch.cuda.Stream()` lands on depends on how many streams the process (torch,
RCCL, the profiler) created before it.  Measured on MI355X (`profiles/r03_queue_pair_probe.txt`, tools/micro/queue_pair_probe.py) a
pair of streams is one of three kinds: concurrent (4.40 ms per pair-step), the same hardware queue (5.64 ms = back to back), or two
queues that time-slice each other (7.5-8.6 ms — slower than back to back).  Which pair is which is a lottery, so it is measured:
`pick_concurrent_streams` replays every engine's evaluation-forward graph (no state changes: no running statistics, no parameters,
no step counters) on candidate pairs until one overlaps.
"""
from __future__ import annotations

import torch

CANDIDATES = 6
SEGMENT = "fwd_eval"
_CHOSEN = {}          # device index -> (streams, report): a stream keeps its hardware queue for life, so one measurement per process


def _pair_time(engines, streams, device, replays):
    cur = torch.cuda.current_stream(device)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(cur)
    for s in streams:
        s.wait_stream(cur)
    for _ in range(replays):
        for e, s in zip(engines, streams):
            with torch.cuda.stream(s):
                e.run(SEGMENT, True)
    for s in streams:
        cur.wait_stream(s)
    t1.record(cur)
    t1.synchronize()
    return t0.elapsed_time(t1) * 1e3 / replays


def pick_concurrent_streams(engines, device=None, candidates=CANDIDATES, replays=3, accept=0.85, report=None, refresh=False, prioritise_longer=True):
    """One stream per engine, chosen so that the engines' graphs overlap.  engines: hippie_amd.engine.Engine objects whose plans hold
    a "fwd_eval" segment (every model plan does).  One engine: a fresh stream, nothing to measure.  More than two engines: the pair
    is found for the first two and the rest take the remaining candidates in order.

    Cost matters (the reference's own pretraining fit is 31 steps): each graph is first timed alone; candidate pairs are then tried
    in order and the first one at or below `accept` x the sum of the two is taken (concurrent pairs measure 0.75-0.80 of it, same-queue
    pairs 1.0, time-slicing pairs 1.15-1.5) — normally a handful of measurements of a few milliseconds.  If none qualifies the fastest
    pair seen wins.
    prioritise_longer: the engine whose graph takes longer alone gets a HIGH-priority stream (the other a normal one).  Free-running,
    the shorter chain (the waveform model) otherwise finishes its steps first and the longer one runs its last steps alone:
    tools/micro/balance_probe.py, 4.34 -> 4.26 ms per pair-step when the time model's stream has priority.
    report: optional dict, filled with {"chosen": [i, j], "us": t, "serial_us": t1, "alone_us": [..], "high_priority": k, "tried": {"i,j": us}}.
    The result is kept per device for the life of the process (a later fit reuses it); refresh=True measures again — after something
    else (an RCCL communicator) has created streams of its own."""
    device = torch.device(device if device is not None else engines[0].device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if not refresh and key in _CHOSEN and len(_CHOSEN[key][0]) >= len(engines):
        if report is not None:
            report.update(_CHOSEN[key][1], cached=True)
        return _CHOSEN[key][0][:len(engines)]
    with torch.cuda.device(device):
        n = max(candidates, len(engines))
        pool = [torch.cuda.Stream(device=device) for _ in range(n)]
        if len(engines) < 2:
            return pool[:len(engines)]
        pair = engines[:2]
        for e in pair:
            e.capture_segments((SEGMENT,))
        torch.cuda.synchronize(device)
        _pair_time(pair, (pool[0], pool[0]), device, 1)                     # first replay of a graph uploads it
        # a stream can be slow by itself: best of two
        alone = [min(_pair_time([e], (s,), device, replays) for s in pool[:2]) for e in pair]
        serial = alone[0] + alone[1]
        hi = None
        pools = [pool, pool]
        if prioritise_longer:
            hi = 0 if alone[0] > alone[1] else 1
            pools[hi] = [torch.cuda.Stream(device=device, priority=-1) for _ in range(n)]
        tried = {}
        for j in range(n):
            for i in range(n if hi is not None else j):
                # i indexes the first engine's pool, j the second's (one shared pool without priorities: i < j)
                tried[(i, j)] = _pair_time(pair, (pools[0][i], pools[1][j]), device, replays)
                if tried[(i, j)] <= accept * serial:
                    break
            else:
                continue
            break
        (i, j), t = min(tried.items(), key=lambda kv: kv[1])
        rep = dict(chosen=[i, j], us=round(t, 1), serial_us=round(serial, 1), alone_us=[round(v, 1) for v in alone], high_priority=hi,
                   tried={f"{a},{b}": round(v) for (a, b), v in tried.items()})
        if report is not None:
            report.update(rep)
        chosen = [pools[0][i], pools[1][j]]
        _CHOSEN[key] = (chosen + [s for s in pool if s not in chosen], rep)
        return _CHOSEN[key][0][:len(engines)]


_SIDE = {}            # (device index, busy stream handles) -> (stream, report)


def pick_side_stream(busy, device=None, candidates=12, report=None):
    """A stream for work that must run BESIDE the `busy` streams (the gradient all-reduce beside a model's backward pass): one that does
    not share a hardware queue with any of them.  On a shared queue every `wait_event` of the side stream is a barrier in front of the
    busy stream's own kernels — the two-bucket data-parallel step measured 8.3 instead of 4.3 ms per pair-step that way.  Probed by the
    library (hp_pick_side_stream: a 2 ms spin kernel on the candidate, a timed empty kernel on each busy stream); the busy streams must be
    idle; cached per (device, busy streams) for the life of the process."""
    import ctypes
    from . import program as P
    busy = list(busy)
    device = torch.device(device if device is not None else busy[0].device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), tuple(int(s.cuda_stream) for s in busy))
    if key not in _SIDE:
        lib = P.load_library()
        with torch.cuda.device(device):
            torch.cuda.synchronize(device)
            arr = (ctypes.c_void_p * max(1, len(busy)))(*[ctypes.c_void_p(int(s.cuda_stream)) for s in busy])
            out, rep = ctypes.c_void_p(), (ctypes.c_float * 2)()
            if lib.hp_pick_side_stream(arr, len(busy), int(candidates), ctypes.byref(out), rep) != 0:
                raise P.HipEngineError(lib.hp_last_error().decode())
            _SIDE[key] = (torch.cuda.ExternalStream(out.value, device=device), {"worst_delay_us": round(float(rep[0]), 1), "tried": int(rep[1])})
    if report is not None:
        report.update(_SIDE[key][1])
    return _SIDE[key][0]
