"""The LPIPS-VGG restatement against the golden recorded from the reference's own
LPIPS class (scripts/make_goldens.py; trunk = seeded random VGG16, lin = LPIPS v0.1)."""
import os

import numpy as np
import torch

from oracle import lpips as ol


def _lin():
    from gomavatar_amd import lpips as pl            # data file + seeded weights only (no HIP call)
    d = np.load(pl._DATA)
    return [torch.from_numpy(d[f"lin{k}"]) for k in range(5)], pl.seeded_trunk


def test_lin_weights_are_the_lpips_v01_vgg_set():
    lins, _ = _lin()
    assert [t.numel() for t in lins] == [64, 128, 256, 512, 512]
    assert all(float(t.min()) >= 0 for t in lins)     # NetLinLayer weights are non-negative in the released model


def test_restatement_matches_reference_class(golden_dir):
    g = np.load(os.path.join(golden_dir, "lpips_vgg.npz"))
    lins, seeded_trunk = _lin()
    wb = seeded_trunk(int(g["trunk_seed"]))
    val, res = ol.lpips_vgg(torch.from_numpy(g["in0"]), torch.from_numpy(g["in1"]), wb, lins, per_layer=True)
    np.testing.assert_allclose(val.numpy(), g["val"], rtol=2e-5, atol=1e-7)
    # the reference accumulates in place (`val = res[0]; val += res[l]`, lpips.py:117-119): its res[0] IS the total
    np.testing.assert_allclose(g["res0"], g["val"])
    for k in range(1, 5):
        np.testing.assert_allclose(res[k].numpy(), g[f"res{k}"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(res[0].numpy(), g["val"] - sum(g[f"res{k}"] for k in range(1, 5)), rtol=1e-4, atol=1e-6)
    assert val.shape == (2, 1, 1, 1) and float(val.min()) > 0.01     # a non-degenerate case


def test_identical_inputs_give_zero():
    lins, seeded_trunk = _lin()
    x = torch.rand(1, 3, 32, 32) * 2 - 1
    assert float(ol.lpips_vgg(x, x.clone(), seeded_trunk(1), lins)) == 0.0
