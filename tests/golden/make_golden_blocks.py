"""Per-block golden vectors from the REAL reference backbone classes (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_blocks.py

Imports /root/reference/hippie/backbones.py (read-only, no bytecode written), builds ResizeConv1d / BasicBlockEnc /
BasicBlockDec / ResNet18Enc / ResNet18Dec with their own constructors, fills every parameter and buffer with the closed-form
recipe of oracle/cvae_oracle.fill_value (keys prefixed "m." so that the recipe recognises BatchNorm entries of a stand-alone
block) through load_state_dict, and stores the eval-mode output, the train-mode output and the BatchNorm running statistics
the train-mode forward leaves, for closed-form inputs.  Only these data files travel; the reference never does."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from hippie import backbones as ref          # noqa: E402  (the reference)
from oracle import cvae_oracle as O          # noqa: E402

torch.set_num_threads(4)


def filled(module, salt):
    new = {}
    for k, v in module.state_dict().items():
        val = O.fill_value("m." + k, tuple(v.shape), salt)
        new[k] = torch.from_numpy(np.ascontiguousarray(val)).to(v.dtype)
    module.load_state_dict(new)
    return module


def block_input(name, shape, salt):
    return torch.from_numpy(O.unit_noise("blocks." + name, int(np.prod(shape)), salt).reshape(shape)).float()


CASES = {
    # name: (constructor, kwargs, input shape [B, C, L] or [B, 2z])
    "resize_64_32_L16": ("ResizeConv1d", dict(in_channels=64, out_channels=32, kernel_size=3, scale_factor=2), (8, 64, 16)),
    "resize_128_64_L8": ("ResizeConv1d", dict(in_channels=128, out_channels=64, kernel_size=3, scale_factor=2), (8, 128, 8)),
    "enc_block_64_s1_L25": ("BasicBlockEnc", dict(in_planes=64, stride=1), (8, 64, 25)),
    "enc_block_64_s2_L25": ("BasicBlockEnc", dict(in_planes=64, stride=2), (8, 64, 25)),
    "dec_block_128_s1_L16": ("BasicBlockDec", dict(in_planes=128, stride=1), (8, 128, 16)),
    "dec_block_128_s2_L16": ("BasicBlockDec", dict(in_planes=128, stride=2), (8, 128, 16)),
    "enc_z10_L50": ("ResNet18Enc", dict(z_dim=10), (8, 1, 50)),
    "enc_z10_L100": ("ResNet18Enc", dict(z_dim=10), (8, 1, 100)),
    "enc_z10_L32": ("ResNet18Enc", dict(z_dim=10), (8, 1, 32)),
    "enc_z10_L256": ("ResNet18Enc", dict(z_dim=10), (8, 1, 256)),
    "dec_z10_out50": ("ResNet18Dec", dict(output_size=50, z_dim=10), (8, 20)),
    "dec_z10_out100": ("ResNet18Dec", dict(output_size=100, z_dim=10), (8, 20)),
}


def main():
    out = {}
    for salt, (name, (cls, kw, shape)) in enumerate(CASES.items(), start=50):
        m = filled(getattr(ref, cls)(**kw), salt)
        x = block_input(name, shape, salt)
        with torch.no_grad():
            m.eval()
            y_eval = m(x)
            m.train()
            y_train = m(x)
        out[name + "/eval"] = y_eval.numpy().astype(np.float32)
        out[name + "/train"] = y_train.numpy().astype(np.float32)
        for k, v in m.state_dict().items():
            if "running_" in k:
                out[name + "/after_train/" + k] = v.numpy().astype(np.float32)
        out[name + "/keys"] = np.array(list(m.state_dict().keys()))
        out[name + "/shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
        print(f"{name:24s} {cls:14s} in {tuple(x.shape)} -> out {tuple(y_train.shape)}  |train| max {float(y_train.abs().max()):.3f}")
    # the reference's own shape-only self-test of the decoder (hippie/backbones.py:156-165)
    ref.test_decoder()
    path = os.path.join(HERE, "blocks.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
