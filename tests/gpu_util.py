"""Test plumbing for the -m gpu parity tests: layout conversion and single-conv packing.

The product is only reached through the C-ABI (real_esrgan_pytorch_amd._lib); torch is used here
for device memory and for converting between the oracle's NCHW fp32 and the library's layouts.
"""
import ctypes as C

import torch

import real_esrgan_pytorch_amd as R

L = R._lib


def tdtype(dtype):
    return torch.float16 if dtype == L.RESR_F16 else torch.float32


def to_nhwc(x_nchw, dtype, c_pad=None, stride=None, offset=0):
    """NCHW fp32 (cpu) -> device NHWC tensor [N,H,W,stride] holding the channels at `offset`."""
    n, c, h, w = x_nchw.shape
    c_pad = c_pad or c
    stride = stride or c_pad
    buf = torch.zeros(n, h, w, stride, dtype=tdtype(dtype), device="cuda")
    buf[..., offset:offset + c] = x_nchw.permute(0, 2, 3, 1).to(tdtype(dtype)).cuda()
    return buf


def from_nhwc(buf, c, offset=0):
    return buf[..., offset:offset + c].float().cpu().permute(0, 3, 1, 2).contiguous()


def pack_conv(weight, dtype, transposed=False, scale=1.0):
    """Pack one OIHW fp32 conv weight through resr_pack_weights (forward or backward-data form)."""
    cout, cin = weight.shape[:2]
    m_real, k_real = (cin, cout) if transposed else (cout, cin)
    m_pad = (m_real + 31) // 32 * 32
    k_pad = (k_real + 31) // 32 * 32
    mt = m_pad // 32
    nck = k_pad // 32
    chunks = (L.PackChunk * nck)()
    for ck in range(nck):
        kc = min(32, k_real - ck * 32)
        chunks[ck] = L.PackChunk(0, ck * 9 * mt * 1024, cout, cin, 0, m_real, ck * 32, kc, mt,
                                 1 if transposed else 0, scale, 0, None)
    table = torch.frombuffer(bytearray(bytes(chunks)), dtype=torch.uint8).cuda()
    arena = weight.reshape(-1).float().cuda()
    es = {L.RESR_F16: 2, L.RESR_F32: 4, L.RESR_F16X2: 6}[dtype]   # exact16: three f16 blocks per chunk
    packed = torch.zeros(nck * 9 * mt * 1024 * es + 16384, dtype=torch.uint8, device="cuda")
    L.check(L.lib().resr_pack_weights(L.ptr(table), nck, L.ptr(arena), L.ptr(packed), dtype, L.stream_ptr()),
            "resr_pack_weights")
    return packed


def quant(x, dtype):
    """Round a cpu fp32 tensor to the storage type the kernel will see."""
    return x.to(tdtype(dtype)).float()


def sptr(t, elem_offset=0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + elem_offset * t.element_size())
