"""Fixture (11): the reference's own get_embeddings (scripts/utils.py:75-101) on reference modules.

Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_embeddings.py
scripts/utils.py imports seaborn at module level (absent here, used by the plotting helpers only); like
pytorch_lightning in make_golden.py it gets an empty in-memory stand-in so that the function can be imported.
Models: closed-form weights (oracle.fill_value, salt 0 / 1), eval mode, two batches of 6 rows.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G                      # noqa: E402  (sets up the pytorch_lightning stand-in + reference imports)

sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
spec = importlib.util.spec_from_file_location("ref_scripts_utils", "/root/reference/scripts/utils.py")
ref_utils = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_utils)
O = G.O


def main():
    z, B, nb = 10, 6, 2
    mods = []
    for L, salt in ((50, 0), (100, 1)):
        net = G.filled(G.ref_model.hippieUnimodalCVAE(z_dim=z, output_size=L, class_hidden_dim=5, num_sources=5, num_classes=5), salt)
        mod = G.ref_model.hippieUnimodalEmbeddingModelCVAE(net, learning_rate=1e-3)
        mod.eval()
        mods.append(mod)
    xw, src, _, _ = O.synth_inputs(B * nb, 50, z, salt=7)
    xt = O.synth_inputs(B * nb, 100, z, salt=8)[0]
    lw = [(xw[i * B:(i + 1) * B], src[i * B:(i + 1) * B]) for i in range(nb)]
    lt = [(xt[i * B:(i + 1) * B], src[i * B:(i + 1) * B]) for i in range(nb)]
    with torch.no_grad():
        ew, et, joint = ref_utils.get_embeddings(lw, lt, mods[0], mods[1])
    np.savez_compressed(os.path.join(HERE, "get_embeddings_z10_B6x2.npz"), waveform=ew, isi=et, joint=joint,
                        meta=np.array([z, B, nb, 7, 8]))
    print(ew.shape, et.shape, joint.shape, float(np.abs(ew).max()))


if __name__ == "__main__":
    main()
