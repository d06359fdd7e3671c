"""Host data path vs outputs of the reference itself (tests/golden/dataset.npz, made by
`gen_golden.py dataset` in the build container): a full `TrainValidImageDataset.__getitem__` under fixed seeds --
augmentation draws, the three blur kernels in the reference's draw order -- the validation LR synthesis, MATLAB-style
`image_resize` and the YCbCr conversion.  CPU only."""
import os
import random

import numpy as np
import pytest
import torch

import real_esrgan_pytorch_amd as R
from real_esrgan_pytorch_amd import dataset as D
from real_esrgan_pytorch_amd import imgproc as M

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "dataset.npz"))
IMG_DIR = os.path.join(HERE, "golden", "dataset_images")


@pytest.mark.parametrize("seed", range(6))
def test_train_item_matches_reference(seed):
    ds = D.TrainValidImageDataset(IMG_DIR, 16, 4, "Train", R.config.degradation_model_parameters_dict)
    random.seed(seed); np.random.seed(seed)
    item = ds[0]
    assert set(item) == {"hr", "kernel1", "kernel2", "sinc_kernel"}
    # right-angle rotation / flips are pixel permutations: exact
    assert np.array_equal(item["hr"].numpy(), G[f"train{seed}_hr"])
    for k in ("kernel1", "kernel2", "sinc_kernel"):
        assert item[k].dtype == torch.float32 and tuple(item[k].shape) == (21, 21)
        np.testing.assert_allclose(item[k].numpy(), G[f"train{seed}_{k}"], rtol=0, atol=1e-7)


def test_valid_item_matches_reference():
    ds = D.TrainValidImageDataset(IMG_DIR, 16, 4, "Valid", R.config.degradation_model_parameters_dict)
    item = ds[0]
    assert np.array_equal(item["hr"].numpy(), G["valid_hr"])
    np.testing.assert_allclose(item["lr"].numpy(), G["valid_lr"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("i", range(4))
def test_image_resize(i):
    out = M.image_resize(G[f"resize{i}_in"].copy(), float(G[f"resize{i}_scale"]))
    assert out.shape == G[f"resize{i}_out"].shape
    np.testing.assert_allclose(out, G[f"resize{i}_out"], rtol=0, atol=1e-6)
    t = torch.from_numpy(G[f"resize{i}_in"]).permute(2, 0, 1)
    np.testing.assert_allclose(M.image_resize(t, float(G[f"resize{i}_scale"])).permute(1, 2, 0).numpy(), G[f"resize{i}_out"], rtol=0, atol=1e-6)


def test_rgb2ycbcr():
    x = torch.from_numpy(G["ycbcr_in"])
    np.testing.assert_allclose(M.rgb2ycbcr_torch(x, True).numpy(), G["ycbcr_y"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(M.rgb2ycbcr_torch(x, False).numpy(), G["ycbcr_full"], rtol=0, atol=1e-6)


def test_rotation_quirk_and_validation_batches():
    im = np.arange(6 * 6 * 1, dtype=np.float32).reshape(6, 6, 1) + 1
    r = M._rotate_right_angle(im, 90)
    assert (r[0] == 0).all() and r[1:].min() > 0        # even side: centre (w//2, h//2) is half a pixel off -> one black row
    odd = np.arange(5 * 5 * 3, dtype=np.float32).reshape(5, 5, 3)
    for k in (1, 2, 3):
        assert np.array_equal(M._rotate_right_angle(odd, 90 * k), np.rot90(odd, k))
    ds = D.TrainValidImageDataset(IMG_DIR, 16, 4, "Valid", R.config.degradation_model_parameters_dict)
    from torch.utils.data import DataLoader
    dl = DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)
    b = next(iter(dl))
    assert tuple(b["lr"].shape) == (1, 3, 4, 4) and tuple(b["hr"].shape) == (1, 3, 16, 16) and len(dl) == 1
    cp = D.CPUPrefetcher(dl)                     # reference dataset.py:248-268 (the HBM stager, CUDAPrefetcher: tests/test_gpu_train_harness.py)
    assert len(cp) == 1 and cp.next() is not None and cp.next() is None
