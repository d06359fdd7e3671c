"""NIQE (validation metric) vs scores computed by the reference itself on stored 8-bit images
(tests/golden/niqe.npz, `gen_golden.py niqe`).  CPU only."""
import os

import numpy as np
import pytest
import torch

from real_esrgan_pytorch_amd.image_quality_assessment import NIQE

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "niqe.npz"))
MODEL = os.path.join(HERE, "golden", "niqe_model.mat")


@pytest.mark.parametrize("i", [0, 1])
def test_niqe_matches_reference(i):
    x = torch.from_numpy(G[f"img{i}"]).float() / 255.0
    score = NIQE(int(G[f"crop{i}"]), MODEL)(x)
    np.testing.assert_allclose(np.atleast_1d(score.numpy()), G[f"score{i}"], rtol=1e-6)


def test_niqe_needs_a_block():
    with pytest.raises(ValueError):
        NIQE(0, MODEL)(torch.rand(1, 3, 64, 64))
