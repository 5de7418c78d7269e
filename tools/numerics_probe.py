"""CPU probe (no GPU): where does the engine's distance to the float64 truth come from?

Emulates the ENGINE's float32 arithmetic of the encoder forward in numpy — conv as fp32 fma chains in the MFMA kernel's
accumulation structure (v_mfma_f32_32x32x2_f32 == fma(a1,b1,fma(a0,b0,c)), measured in round 2), BatchNorm statistics in
float64, the BatchNorm apply in a selectable float32 form — and compares every leaky-ReLU pre-activation with the float64
oracle next to torch-float32 (ATen / mkldnn).  The number that predicts sign flips is E|err(pre)| / sigma(pre): a flip
happens where |pre| < |err|, and the density of pre at zero is ~0.4 / sigma.

    python tools/numerics_probe.py [B] [L] [salt]

fma(a, b, c) in float32 is emulated as float32(float64(a) * float64(b) + float64(c)): the product of two float32 is exact
in float64, the sum rounds once to float64 and once to float32 (double rounding differs from a true fma only on exact
ties of the float64 sum, ~2^-29 of the cases).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cvae_oracle as O      # noqa: E402

f32, f64 = np.float32, np.float64


def fma32(a, b, c):
    return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)


def conv_chain(x, w, stride, pad, chains, order="engine", flush=0):
    """x [B, Cin, L] f32, w [Cout, Cin, k] f32 -> [B, Cout, Lout] f32 with the contraction (taps x Cin) split into `chains`
    sequential fp32 fma chains that are summed pairwise at the end.  chains=4, order="engine": the conv kernel's
    structure (per 32-wide K slice: K-half kh = k//16, group g = (k//8)%2 -> accumulator 2*kh+g ... summed (g0+g1) per half,
    then the halves).  chains=0: exact (float64) accumulation, rounded once.  flush=F > 0: every F K-steps (32*F contraction
    indices) each chain's accumulator is added into a running total and restarted from zero (blocked summation)."""
    B, Cin, L = x.shape
    Cout, _, k = w.shape
    Lout = (L + 2 * pad - k) // stride + 1
    xp = np.zeros((B, Cin, L + 2 * pad), f32)
    xp[:, :, pad: pad + L] = x
    cols = []       # (tap, cin) ordered tap-major, as the kernel consumes them
    for t in range(k):
        xs = xp[:, :, t: t + stride * (Lout - 1) + 1: stride]        # [B, Cin, Lout]
        cols.append(xs.transpose(0, 2, 1).reshape(B * Lout, Cin))
    A = np.concatenate(cols, 1)                                       # [M, k*Cin]
    Wm = w.transpose(2, 1, 0).reshape(k * Cin, Cout)                  # [k*Cin, Cout]
    if chains == 0:
        out = (A.astype(f64) @ Wm.astype(f64)).astype(f32)
    else:
        Kt = A.shape[1]
        acc = [np.zeros((A.shape[0], Cout), f32) for _ in range(chains)]
        tot = [np.zeros((A.shape[0], Cout), f32) for _ in range(chains)]
        for kk in range(Kt):
            if flush and kk and kk % (32 * flush) == 0:
                tot = [(t + a).astype(f32) for t, a in zip(tot, acc)]
                acc = [np.zeros((A.shape[0], Cout), f32) for _ in range(chains)]
            if order == "engine" and chains == 4:
                c = 2 * ((kk % 32) // 16) + ((kk % 16) // 8) if Kt >= 32 else kk % 4
            else:
                c = kk % chains
            acc[c] = fma32(A[:, kk: kk + 1], Wm[kk: kk + 1, :], acc[c])
        if flush:
            acc = [(t + a).astype(f32) for t, a in zip(tot, acc)]
        while len(acc) > 1:
            acc = [(acc[i] + acc[i + 1]).astype(f32) for i in range(0, len(acc), 2)]
        out = acc[0]
    return out.reshape(B, Lout, Cout).transpose(0, 2, 1).copy()


def bn_train(x, gamma, beta, form, eps=1e-5):
    """training-mode BatchNorm of x [B, C, L] (f32) with float64 statistics (the engine's), applied in float32 in `form`:
    scale_shift: fma(x, sc, sh), sh = beta - mean*sc rounded to f32          (round 2's kernels)
    sub_mean:    fma(x - mean_f32, sc, beta)                                 
    sub_mean2:   fma((x - mean_hi) - mean_lo, sc, beta), mean = hi + lo      (two-float mean)
    exact:       the float64 value rounded once (lower bound)"""
    xd = x.astype(f64)
    mean = xd.mean((0, 2))
    var = np.maximum((xd * xd).mean((0, 2)) - mean * mean, 0)
    invstd = 1 / np.sqrt(var + f64(f32(eps)))
    sc = gamma.astype(f64) * invstd
    b = beta.astype(f64)
    bc = lambda v: v[None, :, None]
    if form == "exact":
        return ((xd - bc(mean)) * bc(sc) + bc(b)).astype(f32)
    scf = sc.astype(f32)
    if form == "scale_shift":
        return fma32(x, bc(scf), bc((b - mean * sc).astype(f32)))
    if form == "sub_mean":
        return fma32((x - bc(mean.astype(f32))).astype(f32), bc(scf), bc(beta))
    if form == "sub_mean2":
        hi = mean.astype(f32)
        lo = (mean - hi.astype(f64)).astype(f32)
        return fma32(((x - bc(hi)).astype(f32) - bc(lo)).astype(f32), bc(scf), bc(beta))
    raise ValueError(form)


def lrelu(x, s=0.01):
    return np.where(x > 0, x, x * f32(s)).astype(f32)


def encoder_pre_activations(P, x, chains, form, flush=0):
    """{site -> pre-activation [B,C,L] f32} of the encoder forward under the emulated arithmetic."""
    g = lambda k: P[k].detach().numpy().astype(f32)
    pre = {}
    h = conv_chain(x, g("encoder.conv1.weight"), 2, 1, 1 if chains else 0)
    h = bn_train(h, g("encoder.bn1.weight"), g("encoder.bn1.bias"), form)
    pre["encoder.bn1"] = h
    h = lrelu(h)
    for li in (1, 2, 3, 4):
        for bi in (0, 1):
            s = 2 if (bi == 0 and li > 1) else 1
            p = f"encoder.layer{li}.{bi}."
            r1 = conv_chain(h, g(p + "conv1.weight"), s, 1, chains, flush=flush)
            a1 = bn_train(r1, g(p + "bn1.weight"), g(p + "bn1.bias"), form)
            pre[p + "bn1"] = a1
            r2 = conv_chain(lrelu(a1), g(p + "conv2.weight"), 1, 1, chains, flush=flush)
            o = bn_train(r2, g(p + "bn2.weight"), g(p + "bn2.bias"), form)
            if s == 1:
                sc = h
            else:
                rs = conv_chain(h, g(p + "shortcut.0.weight"), s, 0, chains, flush=flush)
                sc = bn_train(rs, g(p + "shortcut.1.weight"), g(p + "shortcut.1.bias"), form)
            o = (o + sc).astype(f32)
            pre[p + "bn2"] = o
            h = lrelu(o)
    return pre


def torch_pre_activations(om, x):
    """the same sites from the oracle (stock ATen ops) in the model's dtype: pre = out > 0 ? out : out / slope"""
    taps = {}
    src = torch.ones(x.shape[0], dtype=torch.int64)
    eps = torch.zeros(x.shape[0], 10, dtype=om.dtype)
    with torch.no_grad():
        om.forward((x.to(om.dtype), src, None), eps, True, taps=taps)
    out = {}
    for k, v in taps.items():
        if k.startswith("encoder.") and (k.endswith("bn1") or k.endswith("bn2")):
            v = v.numpy().astype(f64)
            out[k] = np.where(v > 0, v, v / 0.01)
    return out


def engine_pre_activations(om32, x, L, B):
    """the REAL engine's pre-activations (GPU only), read back from its workspace like tests/helpers.activation_masks"""
    from hippie_amd import planner
    from hippie_amd.engine import Engine
    eng = Engine(planner.ModelCfg(kind="unimodal", z_dim=10, output_size=L), B, planner.TrainCfg(lr=1e-3))
    eng.load_state_dict({k: v.detach() for k, v in om32.state.items()})
    src = torch.ones(B, dtype=torch.int64)
    eng.set_inputs(x.cuda(), src.cuda(), None, torch.zeros(B, 10).cuda())
    eng.forward(True)
    torch.cuda.synchronize()
    rd = lambda ref, n: eng.ws[ref.offset: ref.offset + 4 * n].view(torch.float32).cpu().numpy()
    out = {}
    for site in eng.plan.act_sites:
        key, M, C = site["key"], site["M"], site["C"]
        if not key.startswith("encoder.") or not (key.endswith("bn1") or key.endswith("bn2")):
            continue
        if site["kind"] == "tensor":
            v = rd(site["out"], M * C).reshape(B, M // B, C).astype(f64)
            pre = np.where(v > 0, v, v / 0.01)
        else:
            raw = rd(site["raw"], M * C).reshape(B, M // B, C)
            cf = rd(site["coef"], 2 * C)
            pre = fma32(raw, cf[None, None, :C], cf[None, None, C:]).astype(f64)
        out[key] = pre.transpose(0, 2, 1)
    return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    salt = int(sys.argv[3]) if len(sys.argv) > 3 else 11
    torch.set_num_threads(8)
    x, _, _, _ = O.synth_inputs(B, L, 10, salt=salt)
    om32 = O.OracleModel("unimodal", 10, L, salt=salt, dtype=torch.float32)
    om64 = O.OracleModel("unimodal", 10, L, salt=salt, dtype=torch.float64)
    t32, t64 = torch_pre_activations(om32, x), torch_pre_activations(om64, x)
    variants = [("engine r2: 4 chains, fma(x,sc,sh)", 4, "scale_shift"), ("4 chains, fma(x-mean,sc,beta)", 4, "sub_mean"),
                ("4 chains, two-float mean", 4, "sub_mean2"), ("4 chains, exact BN apply", 4, "exact"),
                ("exact conv, fma(x,sc,sh)", 0, "scale_shift"), ("exact conv, exact BN apply", 0, "exact"),
                ("8 chains, fma(x,sc,sh)", 8, "scale_shift"), ("16 chains, fma(x,sc,sh)", 16, "scale_shift"),
                ("4 chains, flush every K-step", 4, "scale_shift", 1), ("4 chains, flush every 2 K-steps", 4, "scale_shift", 2),
                ("4 chains, flush every 4 K-steps", 4, "scale_shift", 4)]
    xn = x.numpy().astype(f32)
    res = {v[0]: encoder_pre_activations(om32.state, xn, v[1], v[2], *v[3:]) for v in variants}
    if torch.cuda.is_available():
        variants.insert(0, ("REAL engine (MI355X)",))
        res[variants[0][0]] = engine_pre_activations(om32, x, L, B)
    sites = [k for k in t64 if k in res[variants[0][0]]]
    print(f"B={B} L={L} salt={salt}: mean|err(pre)| / sigma(pre) per leaky-ReLU site, x 1e-7   (flips expected ~ 0.8 * this * elements)")
    hdr = f"{'site':26s} {'|mean|/sig':>9s} {'torch32':>8s} " + " ".join(f"{'v' + str(i):>8s}" for i in range(len(variants)))
    print(hdr)
    tot = np.zeros(len(variants) + 1)
    for k in sites:
        ref = t64[k]
        sig = ref.std()
        ch_mean = np.abs(ref.mean((0, 2))) / ref.std((0, 2))
        row = [np.abs(t32[k] - ref).mean() / sig] + [np.abs(res[name][k].astype(f64) - ref).mean() / sig for name, *_ in variants]
        tot += np.array(row) * ref.size
        print(f"{k:26s} {ch_mean.mean():9.2f} " + " ".join(f"{v * 1e7:8.2f}" for v in row))
    print("expected flips (0.8 * sum err/sigma * n):")
    print(f"  torch32: {0.8 * tot[0]:.2f}")
    for i, (name, *_) in enumerate(variants):
        print(f"  v{i} {name}: {0.8 * tot[i + 1]:.2f}")


if __name__ == "__main__":
    main()
