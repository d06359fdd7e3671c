"""CPU: the drop-in boundary.  The C-ABI library loads and exports every symbol include/resr.h
declares; the nn.Module surface (names, state_dict keys, init RNG stream) matches the reference;
the product refuses CPU tensors instead of falling back."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__
    __graft_entry__.build()
    import real_esrgan_pytorch_amd as R
    return R


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built._lib.LIB_PATH)
    declared = []
    for h in ("resr.h", "resr_debug.h"):      # the drop-in contract / the test and measurement aids
        hdr = open(os.path.join(ROOT, "include", h)).read()
        names = sorted(set(re.findall(r"\b(resr_[a-z0-9_]+)\s*\(", hdr)))
        if h == "resr.h":
            assert len(names) >= 18
            assert not [n for n in names if n.startswith(("resr_debug_", "resr_profile_"))], "debug aids belong in resr_debug.h"
        for name in names:
            assert hasattr(lib, name), f"{name} declared in include/{h} but not exported"
        declared += names
    assert set(built._lib.exported_symbols()) <= set(declared)
    assert lib.resr_version() == built._lib.RESR_VERSION == 3


def test_struct_layouts_match_header(built):
    L = built._lib
    assert ctypes.sizeof(L.ConvDesc) == 26 * 4 + 5 * 8 + 4 * 4 + 4 + 4 + 8 + 4 * 8   # (+ the four q / MX offsets of ABI version 3) 20 original fields + 6 chunk strides + 5 hi->lo offsets (RESR_F16X2) + sparse-tap / group fields + x2_pair_chunks slot, padding, mask_lo_offset
    assert ctypes.sizeof(L.WgradDesc) == 16 * 4 + 4 * 8   # 15 fields + padding + 2 hi->lo offsets + 2 chunk strides
    assert ctypes.sizeof(L.PackChunk) == 64
    assert ctypes.sizeof(L.GeneratorDesc) == 12 * 4   # + x2_plan, reserved_ (ABI version 2)


def test_host_planning_calls_need_no_gpu(built):
    L = built._lib
    d = L.GeneratorDesc(16, 256, 256, 3, 3, 4, 23, L.RESR_F16, 1, 0)
    lib = L.lib()
    assert lib.resr_generator_param_count(ctypes.byref(d)) == 16_697_987
    n = lib.resr_generator_pack_table(ctypes.byref(d), 1, None, 0)
    # forward: conv1 1 + 69 dense blocks x (2+3+4+5+6) + 5 tail convs x 2; backward-data: same count
    # (conv4^T 1 + four 64->64 convs x 2 + conv1^T 2 + 69 x 20)
    assert n == 2 * (1 + 69 * 20 + 5 * 2)
    assert lib.resr_generator_workspace_bytes(ctypes.byref(d)) > 30e9
    # RESR_X2_PLAN_MX_INFER: the packed buffer grows by the MX region (one plain-f16-sized block per chunk) behind the f16 blocks, and
    # an inference workspace by the q tensors (2 more bytes per activation element)
    dx = L.GeneratorDesc(16, 256, 256, 3, 3, 4, 23, L.RESR_F16X2, 0, 0, 33, 0)
    dm = L.GeneratorDesc(16, 256, 256, 3, 3, 4, 23, L.RESR_F16X2, 0, 0, 97, 0)
    plain = (lib.resr_generator_packed_bytes(ctypes.byref(dx), 1) - 16384) // 6
    off = lib.resr_generator_mx_offset(ctypes.byref(dm))
    assert off >= plain * 6 + 16384 and off % 256 == 0 and lib.resr_generator_packed_bytes(ctypes.byref(dm), 0) == off + plain * 2 + 16384
    assert lib.resr_generator_mx_offset(ctypes.byref(d)) == 0
    wx, wm = lib.resr_generator_workspace_bytes(ctypes.byref(dx)), lib.resr_generator_workspace_bytes(ctypes.byref(dm))
    assert 1.45 * wx < wm < 1.55 * wx
    bad = L.GeneratorDesc(1, 7, 8, 3, 3, 2, 23, L.RESR_F16, 0, 0)                # 7 not divisible by 2
    assert lib.resr_generator_workspace_bytes(ctypes.byref(bad)) == 0


def test_generator_surface_matches_reference_init(built):
    z = np.load(os.path.join(G, "generator_init_seed0.npz"))
    torch.manual_seed(0)
    g = built.Generator(3, 3, 4)
    sd = g.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]]
    assert [v.numel() for v in sd.values()] == z["numel"].tolist()
    sums = torch.stack([v.double().sum() for v in sd.values()])
    assert torch.allclose(sums, torch.from_numpy(z["total_sum"]), rtol=0, atol=1e-9), "init RNG stream differs"
    heads = torch.stack([v.reshape(-1)[:4] if v.numel() >= 4 else torch.cat([v.reshape(-1), torch.zeros(4 - v.numel())])
                         for v in sd.values()])
    assert torch.equal(heads, torch.from_numpy(z["head"]))


def test_flat_arena_views_and_roundtrip(built):
    torch.manual_seed(1)
    g = built.Generator(3, 3, 2, n_blocks=1)
    before = {k: v.clone() for k, v in g.state_dict().items()}
    flat = g.flat_parameters()
    assert flat.numel() == sum(p.numel() for p in g.parameters())
    off = 0
    for p in g.parameters():
        assert p.data_ptr() == flat.data_ptr() + 4 * off
        off += p.numel()
    for k, v in g.state_dict().items():
        assert torch.equal(v, before[k])
    g.load_state_dict({k: v + 1 for k, v in before.items()})
    assert torch.equal(g.flat_parameters()[:5], before["conv1.weight"].reshape(-1)[:5] + 1)


def test_no_cpu_fallback(built):
    g = built.Generator(3, 3, 4, n_blocks=1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        g(torch.rand(1, 3, 8, 8))
    with pytest.raises(ValueError):
        built.Generator(3, 3, 3)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "real_esrgan-pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


def test_discriminator_surface_matches_reference_init(built):
    z = np.load(os.path.join(G, "discriminator_init_seed0.npz"))
    torch.manual_seed(0)
    d = built.Discriminator()
    sd = d.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in z["shapes"]]
    sums = torch.stack([v.double().sum() for v in sd.values()])
    assert torch.allclose(sums, torch.from_numpy(z["total_sum"]), rtol=0, atol=1e-9), "init RNG stream differs"
    with pytest.raises(RuntimeError, match="no CPU path"):
        d(torch.rand(1, 3, 16, 16))


def test_natural_sort():
    from real_esrgan_pytorch_amd.test import natural_sorted
    assert natural_sorted(["img10.png", "img2.png", "a.png", "img1.png", "10.png", "9.png"]) == \
        ["9.png", "10.png", "a.png", "img1.png", "img2.png", "img10.png"]


def test_wgrad_quad_plan_host_logic():
    """Host side of the f16 weight-gradient launch (csrc/wgrad.hip build_quads, no GPU): every (X chunk, G tile) product of a
    launch lands in exactly one slot of one 2x2 job; a dense block's 26 products need 7 jobs (6 full + one diagonal)."""
    import ctypes as C
    from real_esrgan_pytorch_amd import _lib as L
    lib = L.lib()

    def plan(cins, couts):
        n = len(cins)
        out = (C.c_int32 * (40 * 4))()
        nq = lib.resr_debug_wgrad_plan((C.c_int32 * n)(*cins), (C.c_int32 * n)(*couts), n, out, 40)
        assert nq > 0, L.last_error() if hasattr(L, "last_error") else nq
        return [[out[q * 4 + p] for p in range(4)] for q in range(nq)]

    def products(cins, couts):
        return sum((ci // 32) * (co // 32) for ci, co in zip(cins, couts))

    # reference ResidualDenseBlock (model.py:73-85): convs 64->32, 96->32, 128->32, 160->32, 192->64 on one workspace
    rdb = ([64, 96, 128, 160, 192], [32, 32, 32, 32, 64])
    jobs = plan(*rdb)
    used = sorted(p for j in jobs for p in j if p >= 0)
    assert used == list(range(products(*rdb))) == list(range(26))
    assert len(jobs) == 7 and sum(1 for j in jobs if all(p >= 0 for p in j)) == 6
    assert sorted(sum(p >= 0 for p in j) for j in jobs)[0] == 2              # the diagonal job
    # single convolutions of the generator's head / tail and of the discriminator
    for cins, couts, njobs in (([64], [64], 1), ([32], [64], 1), ([64], [32], 1), ([32], [32], 1), ([1024], [64], 16),
                               ([160], [32], 3)):
        jobs = plan(cins, couts)
        assert len(jobs) == njobs, (cins, couts, jobs)
        used = sorted(p for j in jobs for p in j if p >= 0)
        assert used == list(range(products(cins, couts)))
    # slot p of a job is (X chunk p & 1, G tile p >> 1): the two slots of a row share the G tile, of a column the X chunk
    j = plan([64], [64])[0]          # products numbered conv-major, then G tile, then X chunk: (g0,x0)=0 (g0,x1)=1 (g1,x0)=2 (g1,x1)=3
    assert j == [0, 1, 2, 3]
    # argument errors come back as negative status, not as a crash
    bad = (C.c_int32 * 4)()
    assert lib.resr_debug_wgrad_plan((C.c_int32 * 1)(48), (C.c_int32 * 1)(32), 1, bad, 1) < 0


def test_entry_points_keep_the_reference_function_names(built):
    """The train / test entry points expose the functions a user of the reference's scripts calls or patches
    (train_realesrnet.py, train_realesrgan.py, test.py, inference.py: main / load_dataset / build_model / define_* / train /
    validate, and the script-level meter classes Summary / AverageMeter / ProgressMeter)."""
    import importlib
    import inspect
    want = {
        "train_realesrnet": ["main", "load_dataset", "build_model", "define_loss", "define_optimizer", "define_scheduler",
                             "load_checkpoint", "save_checkpoint", "train", "validate"],
        "train_realesrgan": ["main", "load_dataset", "build_model", "define_loss", "define_optimizer", "define_scheduler",
                             "train", "validate"],
        "test": ["main"],
        "inference": ["main"],
    }
    for mod, names in want.items():
        m = importlib.import_module(f"real_esrgan_pytorch_amd.{mod}")
        for n in names:
            assert hasattr(m, n), f"{mod}.{n} missing"
    g = importlib.import_module("real_esrgan_pytorch_amd.train_realesrgan")
    # argument order of the reference's GAN train() (train_realesrgan.py:282-294)
    assert list(inspect.signature(g.train).parameters)[:12] == [
        "discriminator", "generator", "ema_model", "train_prefetcher", "pixel_criterion", "content_criterion",
        "adversarial_criterion", "d_optimizer", "g_optimizer", "epoch", "scaler", "writer"]
    for mod in ("train_realesrnet", "train_realesrgan"):            # reference train_realesrnet.py:497-564
        m = importlib.import_module(f"real_esrgan_pytorch_amd.{mod}")
        meter = m.AverageMeter("Loss", ":6.3f", m.Summary.AVERAGE)
        meter.update(2.0, 3)
        meter.update(4.0, 1)
        assert (meter.val, meter.sum, meter.count, meter.avg) == (4.0, 10.0, 4, 2.5)
        assert str(meter) == "Loss  4.000 ( 2.500)" and meter.summary() == "Loss 2.50"
        m.ProgressMeter(10, [meter], prefix="Epoch: [1]").display(3)
    d = importlib.import_module("real_esrgan_pytorch_amd.dataset")    # reference dataset.py:27-30
    assert d.__all__ == ["TrainValidImageDataset", "TestImageDataset", "PrefetchGenerator", "PrefetchDataLoader", "CPUPrefetcher", "CUDAPrefetcher"]
    import torch
    from torch.utils.data import TensorDataset
    pl = d.PrefetchDataLoader(2, dataset=TensorDataset(torch.arange(6.0)), batch_size=2)
    assert [b[0].tolist() for b in pl] == [[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]]
    cp = d.CPUPrefetcher(pl)
    assert len(cp) == 3 and cp.next()[0].tolist() == [0.0, 1.0]
    cp.reset()
    assert [cp.next() is not None for _ in range(4)] == [True, True, True, False]
    n = importlib.import_module("real_esrgan_pytorch_amd.train_realesrnet")
    assert list(inspect.signature(n.validate).parameters) == ["model", "ema_model", "data_prefetcher", "epoch", "writer", "niqe_model", "mode"]


def test_precision_defaults_follow_the_reference_call_sites(built):
    """The reference trains / validates under amp.autocast (train_realesrnet.py:383,461) and runs inference.py:52-53 / test.py:79-80
    in plain fp32: `config.precision` (training) defaults to the f16 mode, `config.inference_precision` to the mode inside the 1e-3
    parity tolerance; the entry points read exactly these."""
    import inspect
    from real_esrgan_pytorch_amd import config, inference, test as dirtest
    if "RESR_PRECISION" not in os.environ:
        assert config.precision == "fast"
    if "RESR_INFERENCE_PRECISION" not in os.environ:
        assert config.inference_precision == "exact16"
    assert "config.inference_precision" in inspect.getsource(inference.main)
    assert "config.inference_precision" in inspect.getsource(dirtest.main)
    with pytest.raises(ValueError):
        built.Generator(3, 3, 4, precision="bf16")


def test_committed_pmc_traffic_names_the_bench_lines_kernel():
    """`roofline.traffic` of the bench line is read from the newest profiles/r*_pmc_traffic_b16.json -- but only under the in-situ name of
    the dominant kernel and only while the file's `csrc_sha16` is the hash of the current kernel sources (bench.pmc_traffic).  Round 6's
    first reduction parsed an older template signature and lost the `...,chain` row, and the line said `traffic: null`: whenever the
    newest file IS from the current sources it must carry that row, and the reduction must know every conv3x3_ws_kernel signature in use."""
    import glob
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from pmc_traffic import short
    assert short("_ZN4resr17conv3x3_ws_kernelIDF16_Li1ELi2ELi8ELi16ELi0ELi0ELi1EEEvNS_8ConvArgsE") == "conv3x3_ws_kernel<f16,1,2,8,chain>"
    assert short("_ZN4resr17conv3x3_ws_kernelIDF16_Li1ELi2ELi8ELi33ELi1ELi0ELi1EEEvNS_8ConvArgsE") == "conv3x3_ws_kernel<f16x2,1,2,8,chain>"
    assert short("_ZN4resr17conv3x3_ws_kernelIDF16_Li2ELi4ELi4ELi2ELi2ELi0ELi0EEEvNS_8ConvArgsE") == "conv3x3_ws_kernel<f16x2,2,4,4>"
    assert short("_ZN4resr17conv3x3_ws_kernelIDF16_Li2ELi4ELi4ELi0ELi0ELi0ELi0EEEvNS_8ConvArgsE") == "conv3x3_ws_kernel<f16,2,4,4>"
    assert short("void resr::wgrad_quad_kernel_t<true>(resr::WgradQuadArgs)") == "wgrad_quad_kernel<mx>"
    spec = importlib.util.spec_from_file_location("bench_for_surface_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_traffic_b16.json")), reverse=True)
    assert files
    newest = json.load(open(files[0]))
    if newest.get("csrc_sha16") == bench.kernel_sources_sha16():
        assert bench.kernel_name(24128) in newest, sorted(k for k in newest if "conv3x3" in k)
        traffic, src = bench.pmc_traffic(bench.kernel_name(24128), 16)
        assert traffic and traffic > 1e9 and src == os.path.basename(files[0])
