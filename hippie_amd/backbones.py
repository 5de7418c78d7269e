"""The reference's backbone classes (hippie/backbones.py) as stand-alone, constructible modules on the MI355X engine.

    from hippie_amd.backbones import ResizeConv1d, BasicBlockEnc, BasicBlockDec, ResNet18Enc, ResNet18Dec
    enc = ResNet18Enc(z_dim=10)            # hippie/backbones.py:74    x [B, 1, L]  -> [B, 2z]
    dec = ResNet18Dec(output_size=50, z_dim=10)      # :107         z [B, 2z]    -> [B, 1, output_size]
    blk = BasicBlockEnc(64, stride=2)      # :20                        x [B, 64, L] -> [B, 128, ceil(L/2)]

Same constructor signatures, forward shapes, train / eval behaviour (training-mode BatchNorm1d with running-statistics
updates; eval mode on the running statistics) and state_dict() keys as the reference classes; constructors draw their initial
weights from torch's global CPU generator in the reference constructors' order.  Each module lowers its own forward program
(planner.lower_backbone) per (batch, length) on first use; all arithmetic runs in libhippie_hip.so — the same conv / BatchNorm
kernels the full cVAE uses — and there is no CPU fallback.  Forward only: inside the cVAE the backward pass of these blocks is
part of the model's program (hippie_amd.model); the reference never trains a backbone on its own either.

Inputs and outputs are torch CUDA tensors in the reference's [B, C, L] layout; the engine's channels-last layout stays
behind this boundary (one transpose in, one out: plumbing, not arithmetic).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

from . import planner
from .engine import Engine


# ---- constructor order of the reference classes (also their state_dict order): (key, shape, kind) ------------------------
def _conv(out, key, co, ci, k, bias=False):
    out.append((key + ".weight", (co, ci, k), "weight"))
    if bias:
        out.append((key + ".bias", (co,), "bias:" + key + ".weight"))


def _linear(out, key, n, k):
    out.append((key + ".weight", (n, k), "weight"))
    out.append((key + ".bias", (n,), "bias:" + key + ".weight"))


def _bn(out, key, c):
    out.append((key + ".weight", (c,), "ones"))
    out.append((key + ".bias", (c,), "zeros"))


def order_enc_block(out, p, cin, stride):
    """BasicBlockEnc.__init__ (hippie/backbones.py:20-34)"""
    planes = cin * stride
    _conv(out, p + "conv1", planes, cin, 3)
    _bn(out, p + "bn1", planes)
    _conv(out, p + "conv2", planes, planes, 3)
    _bn(out, p + "bn2", planes)
    if stride != 1:
        _conv(out, p + "shortcut.0", planes, cin, 1)
        _bn(out, p + "shortcut.1", planes)


def order_dec_block(out, p, cin, stride):
    """BasicBlockDec.__init__ (hippie/backbones.py:45-63)"""
    cout = cin // stride
    _conv(out, p + "conv2", cin, cin, 3)
    _bn(out, p + "bn2", cin)
    if stride == 1:
        _conv(out, p + "conv1", cout, cin, 3)
        _bn(out, p + "bn1", cout)
    else:
        _conv(out, p + "conv1.conv", cout, cin, 3, bias=True)
        _bn(out, p + "bn1", cout)
        _conv(out, p + "shortcut.0.conv", cout, cin, 3, bias=True)
        _bn(out, p + "shortcut.1", cout)


def order_encoder(out, pre, z):
    """ResNet18Enc.__init__ (hippie/backbones.py:74-92)"""
    _conv(out, pre + "conv1", 64, 1, 3)
    _bn(out, pre + "bn1", 64)
    cin = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in (0, 1):
            stride = 2 if (bi == 0 and li > 1) else 1
            order_enc_block(out, f"{pre}layer{li}.{bi}.", cin, stride)
            cin = planes
    _linear(out, pre + "linear", 2 * z, 512)


def order_decoder(out, pre, z, output_size):
    """ResNet18Dec.__init__ (hippie/backbones.py:107-126)"""
    _linear(out, pre + "linear", 512, 2 * z)
    cin = 512
    for li, planes in ((4, 256), (3, 128), (2, 64), (1, 64)):
        for bi, stride in enumerate((1, 1 if li == 1 else 2)):
            order_dec_block(out, f"{pre}layer{li}.{bi}.", cin, stride)
        cin = planes
    _conv(out, pre + "conv1.conv", 1, 64, 3, bias=True)
    _linear(out, pre + "linear_out", output_size, 64)


def init_state(order, generator=None):
    """What the reference constructors leave in their parameters (nn.Conv1d / nn.Linear.reset_parameters: kaiming_uniform_(a=sqrt 5),
    bias ~ U(+-1/sqrt(fan_in)); nn.Embedding: N(0, 1); BatchNorm1d: weight 1, bias 0), drawn in `order` from torch's CPU generator."""
    sd = OrderedDict()
    for key, shape, kind in order:
        t = torch.empty(shape, dtype=torch.float32)
        if kind == "weight":
            torch.nn.init.kaiming_uniform_(t, a=math.sqrt(5), generator=generator)
        elif kind.startswith("bias:"):
            fan_in = sd[kind[5:]][0].numel()
            bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
            torch.nn.init.uniform_(t, -bound, bound, generator=generator)
        elif kind == "emb":
            torch.nn.init.normal_(t, generator=generator)
        elif kind == "ones":
            t.fill_(1.0)
        else:
            t.zero_()
        sd[key] = t
    return sd


class _Backbone:
    """Shared machinery: per-(batch, length) engines over one set of parameter arenas, nn.Module-like surface."""

    _kind = None

    def _init(self, order, device=None, **spec):
        self._spec = spec
        self.device = device
        self.training = True
        self.use_graph = True
        self._engines = {}
        self._root = None
        self._pending_sd = init_state(order)         # drawn at construction, like the reference's constructor

    def _cfg(self, length):
        return planner.BackboneCfg(kind=self._kind, length=int(length), **self._spec)

    def _engine(self, batch, length) -> Engine:
        key = (int(batch), int(length))
        eng = self._engines.get(key)
        if eng is None:
            eng = Engine(self._cfg(length), key[0], planner.TrainCfg(), device=self.device, share_params_from=self._root)
            if self._root is None:
                self._root = eng
                eng.load_state_dict(self._pending_sd, strict=False)
                self._pending_sd = None
            self._engines[key] = eng
        return eng

    def _any_engine(self):
        if self._root is None:
            self._engine(2, self._default_length())
        return self._root

    def _default_length(self):
        return 32

    # ---- nn.Module-like surface ---------------------------------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        if self._root is not None and torch.device(device) != self._root.device:
            raise RuntimeError("parameters already live on " + str(self._root.device))
        self.device = device
        return self

    def state_dict(self, prefix=""):
        return self._any_engine().state_dict(prefix)

    def load_state_dict(self, sd, strict=True, prefix=""):
        return self._any_engine().load_state_dict(sd, strict=strict, prefix=prefix)

    def _run(self, eng: Engine, x_channels_last):
        if self.training and self._has_bn and x_channels_last.shape[0] * (x_channels_last.shape[1] if x_channels_last.ndim == 3 else 1) < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        xs = eng.io("x")
        xs.copy_(x_channels_last.reshape(xs.shape), non_blocking=True)
        mode = "train" if self.training else "eval"
        eng.run("fwd_" + mode, self.use_graph)
        if self.training:
            for p in eng.num_batches_tracked:
                eng.num_batches_tracked[p] += 1
        return eng.io("out_" + mode)

    _has_bn = True

    def forward(self, x):
        """x [B, C, L] -> [B, C', L'] (the reference's layout)."""
        if x.ndim != 3 or x.shape[1] != self._in_channels():
            raise ValueError(f"{type(self).__name__}: expected input [B, {self._in_channels()}, L], got {tuple(x.shape)}")
        eng = self._engine(x.shape[0], x.shape[2])
        out = self._run(eng, x.to(eng.device, torch.float32).permute(0, 2, 1))
        return out.permute(0, 2, 1)

    def __call__(self, x):
        return self.forward(x)


class ResizeConv1d(_Backbone):
    """hippie/backbones.py:6-16: F.interpolate(scale_factor, "nearest") then Conv1d(k, stride 1, padding 1, bias=True).  The
    up-sampling is folded into the conv's row gather; kernel_size must be 3 (the reference hard-codes padding=1, so any
    other size changes the length) and scale_factor 1 or 2 (the values the reference uses)."""
    _kind = "ResizeConv1d"
    _has_bn = False

    def __init__(self, in_channels, out_channels, kernel_size, scale_factor, mode="nearest", device=None):
        if kernel_size != 3 or mode != "nearest":
            raise ValueError("ResizeConv1d: only kernel_size=3, mode='nearest' (what ResNet18Dec builds) are lowered")
        self.scale_factor, self.mode = scale_factor, mode
        order = []
        _conv(order, "conv", out_channels, in_channels, 3, bias=True)
        self._init(order, device, in_channels=in_channels, out_channels=out_channels, stride=int(scale_factor))

    def _in_channels(self):
        return self._spec["in_channels"]


class BasicBlockEnc(_Backbone):
    """hippie/backbones.py:19-41."""
    _kind = "BasicBlockEnc"

    def __init__(self, in_planes, stride=1, device=None):
        order = []
        order_enc_block(order, "", in_planes, stride)
        self._init(order, device, in_channels=in_planes, stride=stride)

    def _in_channels(self):
        return self._spec["in_channels"]


class BasicBlockDec(_Backbone):
    """hippie/backbones.py:44-70."""
    _kind = "BasicBlockDec"

    def __init__(self, in_planes, stride=1, device=None):
        order = []
        order_dec_block(order, "", in_planes, stride)
        self._init(order, device, in_channels=in_planes, stride=stride)

    def _in_channels(self):
        return self._spec["in_channels"]


class ResNet18Enc(_Backbone):
    """hippie/backbones.py:73-103: x [B, nc=1, L] -> [B, 2*z_dim] (any length L >= 2)."""
    _kind = "ResNet18Enc"

    def __init__(self, num_blocks=[2, 2, 2, 2], z_dim=10, nc=1, device=None):      # noqa: B006 (the reference's signature)
        if list(num_blocks) != [2, 2, 2, 2] or nc != 1:
            raise ValueError("ResNet18Enc: only num_blocks=[2,2,2,2], nc=1 (what the reference's models build) are lowered")
        self.z_dim, self.in_planes = z_dim, 512
        order = []
        order_encoder(order, "", z_dim)
        self._init(order, device, z_dim=z_dim)

    def _default_length(self):
        return 50

    def forward(self, x):
        if x.ndim != 3 or x.shape[1] != 1:
            raise ValueError(f"ResNet18Enc: expected input [B, 1, L], got {tuple(x.shape)}")
        eng = self._engine(x.shape[0], x.shape[2])
        return self._run(eng, x.to(eng.device, torch.float32))


class ResNet18Dec(_Backbone):
    """hippie/backbones.py:106-141: z [B, 2*z_dim] -> [B, nc=1, output_size]."""
    _kind = "ResNet18Dec"

    def __init__(self, output_size=64, num_blocks=[2, 2, 2, 2], z_dim=10, nc=1, device=None):      # noqa: B006
        if list(num_blocks) != [2, 2, 2, 2] or nc != 1:
            raise ValueError("ResNet18Dec: only num_blocks=[2,2,2,2], nc=1 (what the reference's models build) are lowered")
        self.z_dim, self.output_size = z_dim, output_size
        order = []
        order_decoder(order, "", z_dim, output_size)
        self._init(order, device, z_dim=z_dim, output_size=output_size)

    def _default_length(self):
        return 4

    def forward(self, z):
        if z.ndim != 2 or z.shape[1] != 2 * self.z_dim:
            raise ValueError(f"ResNet18Dec: expected input [B, {2 * self.z_dim}], got {tuple(z.shape)}")
        eng = self._engine(z.shape[0], 4)
        return self._run(eng, z.to(eng.device, torch.float32))


class VAE:
    """hippie/backbones.py:144-153 (unused by the reference's scripts): encoder -> decoder, forward(x) -> (encoded, decoded)."""

    def __init__(self, z_dim, device=None):
        self.encoder = ResNet18Enc(z_dim=z_dim, device=device)
        self.decoder = ResNet18Dec(z_dim=z_dim, device=device)

    def train(self, mode=True):
        self.encoder.train(mode)
        self.decoder.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def forward(self, x):
        encoded = self.encoder(x)
        return encoded, self.decoder(encoded)

    __call__ = forward


def test_decoder():
    """The shape check the reference ships for its decoder (hippie/backbones.py:156-165), on this implementation."""
    dev = torch.device("cuda", torch.cuda.current_device())
    for output_size in (50, 100):
        out = ResNet18Dec(output_size=output_size)(torch.randn(8, 20, device=dev))
        assert tuple(out.shape) == (8, 1, output_size), tuple(out.shape)
