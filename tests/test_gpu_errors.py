"""-m gpu: the error contract of the C-ABI (SURVEY.md §8b "Errors" / "Ownership"; include/resr.h:8-11): a bad call returns a
negative resr_status and leaves a message in the THREAD-LOCAL resr_last_error(); nothing is launched, nothing throws across the
boundary, and the next good call works.  The Python mirror turns the code into RuntimeError (real_esrgan_pytorch_amd/_lib.py:check)."""
import ctypes as C
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu

ERR_ARG, ERR_WORKSPACE = -1, -3


def last(lib):
    m = lib.resr_last_error()
    return m.decode() if m else ""


def test_bad_arguments_return_codes_and_messages():
    import real_esrgan_pytorch_amd as R
    L = R._lib
    lib = L.lib()
    st = L.stream_ptr()
    u8 = torch.zeros(1, 3, 16, 16, dtype=torch.uint8, device="cuda")
    f32 = torch.zeros(1, 3, 16, 16, device="cuda")
    q = torch.full((1,), 50.0, device="cuda")
    taps = torch.zeros(4, dtype=torch.int32, device="cuda")
    cases = [
        ("filter2d_u8 even kernel", lambda: lib.resr_filter2d_u8(L.ptr(u8), L.ptr(u8), L.ptr(taps), 1, 3, 16, 16, 2, 2, 0, st), ERR_ARG, "odd"),
        ("filter2d_u8 null", lambda: lib.resr_filter2d_u8(None, L.ptr(u8), L.ptr(taps), 1, 3, 16, 16, 1, 1, 0, st), ERR_ARG, "filter2d_u8"),
        ("resize_u8 bad mode", lambda: lib.resr_resize_u8(L.ptr(u8), L.ptr(u8), 1, 3, 16, 16, 8, 8, 7, None, None, None, None, st), ERR_ARG, "resize_u8"),
        ("resize_u8 missing tables", lambda: lib.resr_resize_u8(L.ptr(u8), L.ptr(u8), 1, 3, 16, 16, 8, 8, 1, None, None, None, None, st), ERR_ARG, "tables"),
        ("jpeg_u8 empty batch", lambda: lib.resr_jpeg_u8(L.ptr(u8), L.ptr(u8), L.ptr(q), None, 0, 16, 16, st), ERR_ARG, "jpeg_u8"),
        ("filter2d null kernel", lambda: lib.resr_filter2d(L.ptr(f32), L.ptr(f32), None, 1, 3, 16, 16, 3, 3, 0, st), ERR_ARG, ""),
    ]
    for name, call, code, needle in cases:
        rc = call()
        assert rc == code, (name, rc)
        msg = last(lib)
        assert msg and needle in msg, (name, msg)
    # the Python mirror raises with the library's message
    with pytest.raises(RuntimeError, match="odd"):
        L.check(lib.resr_filter2d_u8(L.ptr(u8), L.ptr(u8), L.ptr(taps), 1, 3, 16, 16, 2, 2, 0, st), "resr_filter2d_u8")
    # a good call right after the bad ones works and nothing was left pending on the stream
    from real_esrgan_pytorch_amd import imgproc
    out = imgproc.jpeg_u8(torch.full((1, 3, 16, 16), 9, dtype=torch.uint8, device="cuda"), 80)
    torch.cuda.synchronize()
    assert int(out.float().mean().round()) == 9


def test_generator_descriptor_and_workspace_errors():
    import real_esrgan_pytorch_amd as R
    L = R._lib
    lib = L.lib()
    bad = L.GeneratorDesc(1, 24, 24, 3, 3, 3, 1, L.RESR_F16, 0, 0)          # upscale 3: not a Generator the reference has
    assert lib.resr_generator_workspace_bytes(C.byref(bad)) == 0
    d = L.GeneratorDesc(1, 24, 24, 3, 3, 4, 1, L.RESR_F16, 0, 0)
    need = lib.resr_generator_workspace_bytes(C.byref(d))
    assert need > 0
    x = torch.zeros(1, 3, 24, 24, device="cuda")
    y = torch.zeros(1, 3, 96, 96, device="cuda")
    params = torch.zeros(int(lib.resr_generator_param_count(C.byref(d))), device="cuda")
    packed = torch.zeros(int(lib.resr_generator_packed_bytes(C.byref(d), 0)), dtype=torch.uint8, device="cuda")
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    st = L.stream_ptr()
    rc = lib.resr_generator_forward(C.byref(d), L.ptr(x), L.ptr(params), L.ptr(packed), L.ptr(ws), need - 1, L.ptr(y), st)
    assert rc == ERR_WORKSPACE and "workspace" in last(lib)
    rc = lib.resr_generator_forward(C.byref(d), None, L.ptr(params), L.ptr(packed), L.ptr(ws), need, L.ptr(y), st)
    assert rc == ERR_ARG and "null" in last(lib)
    rc = lib.resr_generator_forward(C.byref(bad), L.ptr(x), L.ptr(params), L.ptr(packed), L.ptr(ws), need, L.ptr(y), st)
    assert rc == ERR_ARG and "descriptor" in last(lib)
    dd = L.DiscriminatorDesc(1, 20, 24, L.RESR_F16, 0, 0)                     # H not divisible by 8 (model.py:177-203 needs three halvings)
    assert lib.resr_discriminator_workspace_bytes(C.byref(dd)) == 0
    rc = lib.resr_generator_forward(C.byref(d), L.ptr(x), L.ptr(params), L.ptr(packed), L.ptr(ws), need, L.ptr(y), st)   # all-zero weights: y = clamp(0)
    torch.cuda.synchronize()
    assert rc == 0 and float(y.abs().max()) == 0.0


def test_last_error_is_thread_local():
    import real_esrgan_pytorch_amd as R
    L = R._lib
    lib = L.lib()
    u8 = torch.zeros(1, 3, 16, 16, dtype=torch.uint8, device="cuda")
    taps = torch.zeros(4, dtype=torch.int32, device="cuda")
    st = L.stream_ptr()
    seen = {}

    def worker():
        torch.cuda.set_device(0)
        seen["before"] = last(lib)                       # this thread has made no failing call yet
        rc = lib.resr_resize_u8(L.ptr(u8), L.ptr(u8), 1, 3, 16, 16, 8, 8, 9, None, None, None, None, st)
        seen["rc"], seen["after"] = rc, last(lib)

    assert lib.resr_filter2d_u8(L.ptr(u8), L.ptr(u8), L.ptr(taps), 1, 3, 16, 16, 2, 2, 0, st) == ERR_ARG
    mine = last(lib)
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert "filter2d_u8" in mine and last(lib) == mine   # the worker's failure did not overwrite this thread's message
    assert seen["rc"] == ERR_ARG and "resize_u8" in seen["after"] and "filter2d_u8" not in seen["before"]
