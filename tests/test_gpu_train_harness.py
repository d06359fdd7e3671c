"""Training entry point of SURVEY §8f rank 4 on a tiny synthetic dataset: two epochs of
`real_esrgan_pytorch_amd.train_realesrnet.main()` with dataset-driven degradation, NIQE validation under the EMA
weights, the reference's checkpoint dictionary / file names, and a resume."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _smooth_png(path, size, seed):
    from PIL import Image
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    x = F.interpolate(torch.rand(1, 3, size // 8, size // 8, generator=g), size=(size, size), mode="bicubic").clamp(0, 1)
    x = (0.85 * x + 0.15 * torch.rand(1, 3, size, size, generator=g)).clamp(0, 1)
    Image.fromarray((x[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(path)


def test_train_validate_checkpoint_resume(tmp_path, monkeypatch):
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd import config, imgproc
    from real_esrgan_pytorch_amd import train_realesrnet as T
    for sub, n in (("train", 4), ("valid", 2), ("test_hr", 2)):
        os.makedirs(tmp_path / sub)
        for i in range(n):
            _smooth_png(str(tmp_path / sub / f"{i}.png"), 224, hash(sub) % 100 + i)
    os.makedirs(tmp_path / "test_lr")
    from PIL import Image
    for i in range(2):
        hr = imgproc.read_image_rgb(str(tmp_path / "test_hr" / f"{i}.png"))
        lr = np.clip(imgproc.image_resize(hr, 0.25), 0, 1)
        Image.fromarray((lr * 255).round().astype(np.uint8)).save(str(tmp_path / "test_lr" / f"{i}.png"))
    here = os.path.dirname(os.path.abspath(__file__))
    for k, v in dict(train_image_dir=str(tmp_path / "train"), valid_image_dir=str(tmp_path / "valid"),
                     test_lr_image_dir=str(tmp_path / "test_lr"), test_hr_image_dir=str(tmp_path / "test_hr"),
                     image_size=208, batch_size=2, num_workers=0, epochs=1, print_frequency=1, resume="",
                     lr_scheduler_step_size=1, exp_name="harness_test",
                     niqe_model_path=os.path.join(here, "golden", "niqe_model.mat"),
                     device=torch.device("cuda", 0)).items():
        monkeypatch.setattr(config, k, v, raising=False)
    monkeypatch.chdir(tmp_path)
    T.main()
    ck1 = tmp_path / "samples" / "harness_test" / "g_epoch_1.pth.tar"
    assert ck1.exists() and (tmp_path / "results" / "harness_test" / "g_last.pth.tar").exists()
    assert (tmp_path / "results" / "harness_test" / "g_best.pth.tar").exists()          # first NIQE < 100
    ck = torch.load(ck1, weights_only=False)
    assert set(ck) == {"epoch", "best_niqe", "state_dict", "ema_state_dict", "optimizer", "scheduler"}
    assert ck["epoch"] == 1 and len(ck["state_dict"]) == 702 and np.isfinite(ck["best_niqe"])
    assert all(k.startswith("model.") for k in ck["ema_state_dict"])                     # reference key prefix (inference.py:33)
    tags = [json.loads(l)["tag"] for l in open(tmp_path / "samples" / "logs" / "harness_test" / "scalars.jsonl")]
    assert "Train/Loss" in tags and "Valid/NIQE" in tags and "Test/NIQE" in tags
    # resume for one more epoch
    monkeypatch.setattr(config, "resume", str(ck1))
    monkeypatch.setattr(config, "epochs", 2)
    T.main()
    ck2 = torch.load(tmp_path / "samples" / "harness_test" / "g_epoch_2.pth.tar", weights_only=False)
    assert ck2["epoch"] == 2 and ck2["optimizer"]["state"][0]["step"] == 4               # 2 steps per epoch, state carried over
    # the reference's inference loader reads it: strip "model." (inference.py:33)
    g = R.Generator(3, 3, 4).cuda()
    g.load_state_dict({k[len("model."):]: v for k, v in ck2["ema_state_dict"].items()})
