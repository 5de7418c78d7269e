"""CPU oracle for the HIPPIE cVAE hot path — TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  ``hippie_amd`` must never import anything from ``oracle/``.

It restates, as plain functions over a ``{state_dict key -> tensor}`` mapping,
what the reference computes with ``nn.Module`` classes (citations relative to
``/root/reference``):

* ``enc_forward``   — ``ResNet18Enc.forward``   hippie/backbones.py:94-103
* ``enc_block``     — ``BasicBlockEnc.forward`` hippie/backbones.py:36-41
* ``dec_forward``   — ``ResNet18Dec.forward``   hippie/backbones.py:128-141
* ``dec_block``     — ``BasicBlockDec.forward`` hippie/backbones.py:65-70
* ``resize_conv``   — ``ResizeConv1d.forward``  hippie/backbones.py:13-16
* ``cvae_forward``  — ``hippieUnimodalCVAE.forward`` hippie/model.py:64-72
* ``mm_forward``    — ``MultiModalCVAE.forward``     hippie/model.py:421-432
* ``cvae_losses`` / ``mm_losses`` — ``training_step`` hippie/model.py:95-116, 454-482
* ``adamw_step``    — ``torch.optim.AdamW`` as constructed at hippie/model.py:93
* ``clip_grad_norm``— Lightning ``gradient_clip_val`` (torch ``clip_grad_norm_``)

The arithmetic itself lives in PyTorch (third-party; the reference pins no
version, its dockerfile suggests torch 2.3.1; here torch 2.10.0 CPU): the
restatement calls the same stock ATen ops (`conv1d`, `batch_norm`, `linear`,
`interpolate(nearest)`, `leaky_relu`, `mse_loss`) the reference's modules
call.  Parity is PINNED by ``tests/golden/*.npz``: outputs of the reference's
own modules imported in the build container (generator:
``tests/golden/make_golden.py``), which ``tests/test_oracle_golden.py`` checks
this file against.

Works in float32 (the reference's dtype) or float64 (error analysis).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

ENC_PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
SLOPE_BACKBONE = 0.01   # F.leaky_relu default, backbones.py:37
SLOPE_HEADS = 0.2       # nn.LeakyReLU(0.2), model.py:24


# --------------------------------------------------------------------------
# parameter manifest (state_dict order of the reference modules)
# --------------------------------------------------------------------------
def _bn(prefix, c, out):
    out[prefix + ".weight"] = (c,)
    out[prefix + ".bias"] = (c,)
    out[prefix + ".running_mean"] = (c,)
    out[prefix + ".running_var"] = (c,)
    out[prefix + ".num_batches_tracked"] = ()


def enc_manifest(prefix, z_dim, out):
    out[prefix + "conv1.weight"] = (64, 1, 3)
    _bn(prefix + "bn1", 64, out)
    cin = 64
    for li, planes in enumerate(ENC_PLANES, start=1):
        for bi in range(2):
            stride = 2 if (bi == 0 and li > 1) else 1
            p = f"{prefix}layer{li}.{bi}."
            out[p + "conv1.weight"] = (planes, cin, 3)
            _bn(p + "bn1", planes, out)
            out[p + "conv2.weight"] = (planes, planes, 3)
            _bn(p + "bn2", planes, out)
            if stride != 1:
                out[p + "shortcut.0.weight"] = (planes, cin, 1)
                _bn(p + "shortcut.1", planes, out)
            cin = planes
    out[prefix + "linear.weight"] = (2 * z_dim, 512)
    out[prefix + "linear.bias"] = (2 * z_dim,)


def dec_manifest(prefix, z_dim, output_size, out):
    out[prefix + "linear.weight"] = (512, 2 * z_dim)
    out[prefix + "linear.bias"] = (512,)
    cin = 512
    for li, planes in zip((4, 3, 2, 1), (256, 128, 64, 64)):
        layer_stride = 1 if li == 1 else 2
        for bi, stride in enumerate((1, layer_stride)):
            p = f"{prefix}layer{li}.{bi}."
            cout = cin // stride
            out[p + "conv2.weight"] = (cin, cin, 3)
            _bn(p + "bn2", cin, out)
            if stride == 1:
                out[p + "conv1.weight"] = (cout, cin, 3)
                _bn(p + "bn1", cout, out)
            else:
                out[p + "conv1.conv.weight"] = (cout, cin, 3)
                out[p + "conv1.conv.bias"] = (cout,)
                _bn(p + "bn1", cout, out)
                out[p + "shortcut.0.conv.weight"] = (cout, cin, 3)
                out[p + "shortcut.0.conv.bias"] = (cout,)
                _bn(p + "shortcut.1", cout, out)
        cin = planes
    out[prefix + "conv1.conv.weight"] = (1, 64, 3)
    out[prefix + "conv1.conv.bias"] = (1,)
    out[prefix + "linear_out.weight"] = (output_size, 64)
    out[prefix + "linear_out.bias"] = (output_size,)


def unimodal_manifest(z_dim, output_size, class_hidden_dim, num_sources, num_classes):
    """Key -> shape, in the order of hippieUnimodalCVAE.state_dict() (model.py:13-44)."""
    z, h = z_dim, class_hidden_dim
    m = OrderedDict()
    enc_manifest("encoder.", z, m)
    m["encoder_fc.0.weight"] = (2 * z, 2 * z + 2 * h)
    m["encoder_fc.0.bias"] = (2 * z,)
    _bn("encoder_fc.1", 2 * z, m)
    m["encoder_fc.3.weight"] = (z, 2 * z)
    m["encoder_fc.3.bias"] = (z,)
    _bn("encoder_fc.4", z, m)
    m["source_embedding.weight"] = (num_sources, h)
    m["class_embedding.weight"] = (num_classes, h)
    m["z_mean.weight"] = (z, z)
    m["z_mean.bias"] = (z,)
    m["z_log_var.weight"] = (z, z)
    m["z_log_var.bias"] = (z,)
    m["decoder_fc.0.weight"] = (2 * z, z + 2 * h)
    m["decoder_fc.0.bias"] = (2 * z,)
    m["decoder_fc.2.weight"] = (2 * z, 2 * z)
    m["decoder_fc.2.bias"] = (2 * z,)
    _bn("decoder_fc.3", 2 * z, m)
    dec_manifest("decoder.", z, output_size, m)
    return m


def multimodal_manifest(z_dim, output_size_wave, output_size_isi, class_hidden_dim, num_sources, num_classes):
    """Key -> shape, in the order of MultiModalCVAE.state_dict() (model.py:352-395)."""
    z, h = z_dim, class_hidden_dim
    m = OrderedDict()
    enc_manifest("encoder_mod1.", z, m)
    enc_manifest("encoder_mod2.", z, m)
    m["fusion_encoder.0.weight"] = (2 * z, 4 * z + 2 * h)
    m["fusion_encoder.0.bias"] = (2 * z,)
    _bn("fusion_encoder.1", 2 * z, m)
    m["fusion_encoder.3.weight"] = (z, 2 * z)
    m["fusion_encoder.3.bias"] = (z,)
    m["source_embedding.weight"] = (num_sources, h)
    m["class_embedding.weight"] = (num_classes, h)
    m["z_mean.weight"] = (z, z)
    m["z_mean.bias"] = (z,)
    m["z_log_var.weight"] = (z, z)
    m["z_log_var.bias"] = (z,)
    for mod in ("mod1", "mod2"):
        m[f"decoder_fc_{mod}.0.weight"] = (2 * z, z + 2 * h)
        m[f"decoder_fc_{mod}.0.bias"] = (2 * z,)
        m[f"decoder_fc_{mod}.2.weight"] = (2 * z, 2 * z)
        m[f"decoder_fc_{mod}.2.bias"] = (2 * z,)
        _bn(f"decoder_fc_{mod}.3", 2 * z, m)
    dec_manifest("decoder_mod1.", z, output_size_wave, m)
    dec_manifest("decoder_mod2.", z, output_size_isi, m)
    return m


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


# --------------------------------------------------------------------------
# deterministic closed-form fill (shared recipe: tests/golden/make_golden.py
# applies it to the reference modules through load_state_dict)
# --------------------------------------------------------------------------
def _name_hash(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def unit_noise(name, n, salt=0):
    """n reproducible values in [-1, 1): pure integer arithmetic, no RNG state."""
    idx = np.arange(n, dtype=np.uint64)
    h = np.uint64(_name_hash(name) ^ (salt * 0x9E3779B9 & 0xFFFFFFFF))
    v = (idx * np.uint64(2654435761) + h) & np.uint64(0xFFFFFFFF)
    v ^= v >> np.uint64(15)
    v = (v * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    v ^= v >> np.uint64(13)
    v = (v * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    v ^= v >> np.uint64(16)
    return (v.astype(np.float64) / 2147483648.0) - 1.0


def fill_value(key, shape, salt=0):
    """Closed-form value for one state_dict entry (float64 numpy; int64 for counters)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = unit_noise(key, n, salt)
    if key.endswith("num_batches_tracked"):
        return np.zeros((), dtype=np.int64)
    if key.endswith("running_mean"):
        v = 0.05 * u
    elif key.endswith("running_var"):
        v = 1.0 + 0.25 * np.abs(u)
    elif "embedding" in key:
        v = 1.5 * u
    elif len(shape) == 1:
        is_bn = any(s in key for s in (".bn1.", ".bn2.", "shortcut.1.", "_fc.1.", "_fc.3.", "_fc.4.",
                                       "fusion_encoder.1."))
        # 1-D entries: BatchNorm weight/bias, or a Linear/Conv bias
        if is_bn and key.endswith("weight"):
            v = 1.0 + 0.2 * u
        elif is_bn:
            v = 0.1 * u
        else:
            v = 0.1 * u
    else:
        fan_in = int(np.prod(shape[1:]))
        v = u * math.sqrt(3.0 / fan_in)   # variance 1/fan_in: keeps activations O(1) through 40 layers
    return v.reshape(shape)


def fill_state(manifest, dtype=torch.float32, salt=0):
    sd = OrderedDict()
    for k, shp in manifest.items():
        v = fill_value(k, shp, salt)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        else:
            sd[k] = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
    return sd


def synth_inputs(batch, length, z_dim, num_sources=5, num_classes=5, salt=0, dtype=torch.float32, name="x"):
    """Closed-form inputs: data [B,1,L] in roughly [-1,1], source ids 1..S-1, class ids, eps [B,z]."""
    x = unit_noise(name + ".data", batch * length, salt).reshape(batch, 1, length)
    t = np.linspace(0, 1, length)[None, None, :]
    x = 0.6 * np.sin(6.0 * t + 3.0 * x[:, :, :1]) + 0.4 * x
    src = (np.abs(unit_noise(name + ".src", batch, salt)) * (num_sources - 1)).astype(np.int64) % (num_sources - 1) + 1
    cls = (np.abs(unit_noise(name + ".cls", batch, salt)) * num_classes).astype(np.int64) % num_classes
    # eps: sum of 4 uniforms -> bell-shaped, var = 4/3 * ... scaled to ~unit variance
    e = sum(unit_noise(f"{name}.eps{i}", batch * z_dim, salt) for i in range(4)) * math.sqrt(3.0 / 4.0)
    return (torch.from_numpy(x).to(dtype), torch.from_numpy(src), torch.from_numpy(cls),
            torch.from_numpy(e.reshape(batch, z_dim)).to(dtype))


# --------------------------------------------------------------------------
# functional forward
# --------------------------------------------------------------------------
class _MaskedLeakyReLU(torch.autograd.Function):
    """leaky_relu whose branch is chosen by an injected boolean mask instead of sign(x), forward and backward.

    Why: the loss gradient is discontinuous in the sign of every leaky-ReLU input.  Where an activation is
    closer to zero than float32 resolves, two correct float32 implementations can land on different sides,
    and every gradient upstream of that element then differs at the 1e-3..1e-2 level.  Forcing the oracle
    onto the branch the implementation under test actually took removes that ambiguity: the comparison is
    then between two evaluations of the SAME piecewise-linear function, and the 1e-4 bar applies to every
    gradient.  (The forward value changes by |x|*(1-slope) at a flipped element, i.e. by rounding noise.)"""

    @staticmethod
    def forward(ctx, x, mask, slope):
        ctx.save_for_backward(mask)
        ctx.slope = slope
        return torch.where(mask, x, x * slope)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, g, g * ctx.slope), None, None


class Ctx:
    """training flag + optional tap dict collecting named intermediates + optional injected leaky-ReLU masks
    ({site name -> bool tensor of the activation's shape}; sites are the tap names of the activations)."""

    def __init__(self, training=True, taps=None, masks=None):
        self.training = training
        self.taps = taps
        self.masks = masks
        self.sites = []

    def tap(self, name, t):
        if self.taps is not None:
            self.taps[name] = t

    def lrelu(self, name, x, slope):
        """F.leaky_relu at the named site (backbones.py:37,40,66,69,95; nn.LeakyReLU(0.2) in model.py:24,27,37,40)."""
        self.sites.append(name)
        self.tap(name + "#pre", x)              # the pre-activation: tests bound every sign difference by its magnitude
        if self.masks is not None:
            m = self.masks[name]
            assert m.shape == x.shape and m.dtype == torch.bool, (name, tuple(m.shape), tuple(x.shape))
            out = _MaskedLeakyReLU.apply(x, m, slope)
        else:
            out = F.leaky_relu(x, slope)
        self.tap(name, out)
        return out


def batch_norm(P, prefix, x, ctx):
    if ctx.training:
        P[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, P[prefix + ".running_mean"], P[prefix + ".running_var"], P[prefix + ".weight"],
                        P[prefix + ".bias"], ctx.training, BN_MOMENTUM, BN_EPS)


def resize_conv(P, prefix, x, scale):
    x = F.interpolate(x, scale_factor=scale, mode="nearest")
    return F.conv1d(x, P[prefix + ".conv.weight"], P[prefix + ".conv.bias"], stride=1, padding=1)


def enc_block(P, p, x, stride, ctx):
    out = ctx.lrelu(p + "bn1", batch_norm(P, p + "bn1", F.conv1d(x, P[p + "conv1.weight"], None, stride, 1), ctx), SLOPE_BACKBONE)
    out = batch_norm(P, p + "bn2", F.conv1d(out, P[p + "conv2.weight"], None, 1, 1), ctx)
    if stride == 1:
        sc = x
    else:
        sc = batch_norm(P, p + "shortcut.1", F.conv1d(x, P[p + "shortcut.0.weight"], None, stride, 0), ctx)
    out = ctx.lrelu(p + "bn2", out + sc, SLOPE_BACKBONE)
    ctx.tap(p + "out", out)
    return out


def enc_forward(P, prefix, x, ctx):
    x = ctx.lrelu(prefix + "bn1", batch_norm(P, prefix + "bn1", F.conv1d(x, P[prefix + "conv1.weight"], None, 2, 1), ctx), SLOPE_BACKBONE)
    ctx.tap(prefix + "stem", x)
    for li in (1, 2, 3, 4):
        for bi in (0, 1):
            stride = 2 if (bi == 0 and li > 1) else 1
            x = enc_block(P, f"{prefix}layer{li}.{bi}.", x, stride, ctx)
    x = x.mean(dim=2)           # adaptive_avg_pool1d(x, 1) + view, backbones.py:100-101
    return F.linear(x, P[prefix + "linear.weight"], P[prefix + "linear.bias"])


def dec_block(P, p, x, stride, ctx):
    out = ctx.lrelu(p + "bn2", batch_norm(P, p + "bn2", F.conv1d(x, P[p + "conv2.weight"], None, 1, 1), ctx), SLOPE_BACKBONE)
    if stride == 1:
        out = batch_norm(P, p + "bn1", F.conv1d(out, P[p + "conv1.weight"], None, 1, 1), ctx)
        sc = x
    else:
        out = batch_norm(P, p + "bn1", resize_conv(P, p + "conv1", out, stride), ctx)
        sc = batch_norm(P, p + "shortcut.1", resize_conv(P, p + "shortcut.0", x, stride), ctx)
    out = ctx.lrelu(p + "bn1", out + sc, SLOPE_BACKBONE)
    ctx.tap(p + "out", out)
    return out


def dec_forward(P, prefix, z, ctx):
    x = F.linear(z, P[prefix + "linear.weight"], P[prefix + "linear.bias"])
    x = x.unsqueeze(-1).repeat(1, 1, 4)     # F.interpolate(scale_factor=4) nearest, backbones.py:130-131
    for li in (4, 3, 2, 1):
        layer_stride = 1 if li == 1 else 2
        for bi, stride in enumerate((1, layer_stride)):
            x = dec_block(P, f"{prefix}layer{li}.{bi}.", x, stride, ctx)
    x = resize_conv(P, prefix + "conv1", x, 2)
    x = x.reshape(x.shape[0], -1)
    x = F.linear(x, P[prefix + "linear_out.weight"], P[prefix + "linear_out.bias"])
    return x.unsqueeze(1)


def _embeddings(P, source_labels, class_labels):
    semb = P["source_embedding.weight"][source_labels]
    cemb = P["class_embedding.weight"][class_labels] if class_labels is not None else torch.zeros_like(semb)
    return semb, cemb


def _reparam(mu, logvar, eps):
    return mu + eps * torch.exp(0.5 * logvar)


def cvae_forward(P, data, source_labels, class_labels, eps, ctx):
    """hippieUnimodalCVAE.forward with eps injected (model.py:46-72)."""
    semb, cemb = _embeddings(P, source_labels, class_labels)
    h = enc_forward(P, "encoder.", data, ctx)
    ctx.tap("enc_h", h)
    h = torch.cat([h, semb, cemb], dim=1)
    h = ctx.lrelu("encoder_fc.1", batch_norm(P, "encoder_fc.1", F.linear(h, P["encoder_fc.0.weight"], P["encoder_fc.0.bias"]), ctx), SLOPE_HEADS)
    enc = ctx.lrelu("encoder_fc.4", batch_norm(P, "encoder_fc.4", F.linear(h, P["encoder_fc.3.weight"], P["encoder_fc.3.bias"]), ctx), SLOPE_HEADS)
    mu = F.linear(enc, P["z_mean.weight"], P["z_mean.bias"])
    logvar = F.linear(enc, P["z_log_var.weight"], P["z_log_var.bias"])
    z = _reparam(mu, logvar, eps)
    d = torch.cat([z, semb, cemb], dim=1)
    d = ctx.lrelu("decoder_fc.0", F.linear(d, P["decoder_fc.0.weight"], P["decoder_fc.0.bias"]), SLOPE_HEADS)
    d = ctx.lrelu("decoder_fc.3", batch_norm(P, "decoder_fc.3", F.linear(d, P["decoder_fc.2.weight"], P["decoder_fc.2.bias"]), ctx), SLOPE_HEADS)
    ctx.tap("dec_in", d)
    dec = dec_forward(P, "decoder.", d, ctx)
    return enc, mu, logvar, dec


def mm_forward(P, data1, data2, source_labels, class_labels, eps, ctx):
    """MultiModalCVAE.forward with eps injected (model.py:397-432)."""
    semb, cemb = _embeddings(P, source_labels, class_labels)
    h1 = enc_forward(P, "encoder_mod1.", data1, ctx)
    h2 = enc_forward(P, "encoder_mod2.", data2, ctx)
    h = torch.cat([h1, h2, semb, cemb], dim=1)
    h = ctx.lrelu("fusion_encoder.1", batch_norm(P, "fusion_encoder.1", F.linear(h, P["fusion_encoder.0.weight"], P["fusion_encoder.0.bias"]), ctx), SLOPE_HEADS)
    enc = F.linear(h, P["fusion_encoder.3.weight"], P["fusion_encoder.3.bias"])
    mu = F.linear(enc, P["z_mean.weight"], P["z_mean.bias"])
    logvar = F.linear(enc, P["z_log_var.weight"], P["z_log_var.bias"])
    z = _reparam(mu, logvar, eps)
    zc = torch.cat([z, semb, cemb], dim=1)
    recs = []
    for mod in ("mod1", "mod2"):
        d = ctx.lrelu(f"decoder_fc_{mod}.0", F.linear(zc, P[f"decoder_fc_{mod}.0.weight"], P[f"decoder_fc_{mod}.0.bias"]), SLOPE_HEADS)
        d = ctx.lrelu(f"decoder_fc_{mod}.3", batch_norm(P, f"decoder_fc_{mod}.3", F.linear(d, P[f"decoder_fc_{mod}.2.weight"], P[f"decoder_fc_{mod}.2.bias"]), ctx), SLOPE_HEADS)
        recs.append(dec_forward(P, f"decoder_{mod}.", d, ctx))
    return enc, mu, logvar, recs[0], recs[1]


def kl_rows(mu, logvar):
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - torch.exp(logvar), dim=1)


def cvae_losses(data, mu, logvar, dec, beta=1.0):
    """(loss, mse, kl_mean) of training_step (model.py:103-109)."""
    mse = F.mse_loss(data, dec)
    kl = kl_rows(mu, logvar).mean()
    return mse + beta * kl, mse, kl


def mm_losses(data1, data2, mu, logvar, dec1, dec2, beta=1.0, w1=1.0, w2=1.0):
    """(loss, mse1, mse2, kl_mean) of MultiModalCVAETrainModule.training_step (model.py:465-474)."""
    m1 = F.mse_loss(data1, dec1)
    m2 = F.mse_loss(data2, dec2)
    kl = kl_rows(mu, logvar).mean()
    return w1 * m1 + w2 * m2 + beta * kl, m1, m2, kl


# --------------------------------------------------------------------------
# optimiser restatement
# --------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm, eps=1e-6):
    """torch.nn.utils.clip_grad_norm_ (L2): returns total norm, scales grads in place."""
    gs = [g for g in grads if g is not None]
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in gs]))
    coef = torch.clamp(max_norm / (total + eps), max=1.0)
    for g in gs:
        g.mul_(coef)
    return total


def adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr, weight_decay, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.AdamW step (decoupled decay, no amsgrad); entries with grad None are skipped."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        p.mul_(1.0 - lr * weight_decay)
        exp_avg[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
        exp_avg_sq[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (exp_avg_sq[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(exp_avg[k], denom, value=-lr / bc1)


# --------------------------------------------------------------------------
# a complete training step (used for fixtures, parity tests and the CPU baseline)
# --------------------------------------------------------------------------
class OracleModel:
    """Holds a state (params + buffers) and AdamW state; runs steps with autograd.

    kind: "unimodal" | "multimodal".
    """

    def __init__(self, kind, z_dim, output_size, class_hidden_dim=5, num_sources=5, num_classes=5,
                 output_size2=None, dtype=torch.float32, salt=0):
        self.kind = kind
        self.dtype = dtype
        if kind == "unimodal":
            self.manifest = unimodal_manifest(z_dim, output_size, class_hidden_dim, num_sources, num_classes)
        else:
            self.manifest = multimodal_manifest(z_dim, output_size, output_size2, class_hidden_dim, num_sources, num_classes)
        self.state = fill_state(self.manifest, dtype, salt)
        self.param_keys = [k for k in self.manifest if not is_buffer(k)]
        for k in self.param_keys:
            self.state[k].requires_grad_(True)
        self.exp_avg = {}
        self.exp_avg_sq = {}
        self.step_count = 0

    def load(self, sd):
        with torch.no_grad():
            for k, v in sd.items():
                if k in self.state:
                    self.state[k].copy_(v.to(self.state[k].dtype))

    def forward(self, batch, eps, training=True, taps=None, masks=None):
        """masks: {site -> bool tensor}: leaky-ReLU branches forced (see _MaskedLeakyReLU); None = sign(x)."""
        ctx = Ctx(training, taps, masks)
        if self.kind == "unimodal":
            data, src, cls = batch
            return cvae_forward(self.state, data, src, cls, eps, ctx)
        d1, d2, src, cls = batch
        return mm_forward(self.state, d1, d2, src, cls, eps, ctx)

    def losses(self, batch, outs, beta=1.0, w1=1.0, w2=1.0):
        if self.kind == "unimodal":
            return cvae_losses(batch[0], outs[1], outs[2], outs[3], beta)
        return mm_losses(batch[0], batch[1], outs[1], outs[2], outs[3], outs[4], beta, w1, w2)

    def grads(self):
        return {k: self.state[k].grad for k in self.param_keys}

    def train_step(self, batch, eps, lr, weight_decay=0.01, beta=1.0, clip=None, w1=1.0, w2=1.0):
        for k in self.param_keys:
            self.state[k].grad = None
        outs = self.forward(batch, eps, True)
        ls = self.losses(batch, outs, beta, w1, w2)
        ls[0].backward()
        with torch.no_grad():
            g = self.grads()
            norm = None
            if clip is not None:
                norm = clip_grad_norm(list(g.values()), clip)
            for k in self.param_keys:
                if g[k] is not None and k not in self.exp_avg:
                    self.exp_avg[k] = torch.zeros_like(self.state[k])
                    self.exp_avg_sq[k] = torch.zeros_like(self.state[k])
            self.step_count += 1
            adamw_step({k: self.state[k] for k in self.param_keys}, g, self.exp_avg, self.exp_avg_sq,
                       self.step_count, lr, weight_decay)
        return outs, ls, norm
