"""MI355X-native networks behind the reference's `model.py` surface.

Same class names, constructor arguments, forward semantics and state_dict keys as the reference
(`Generator`, `ResidualDenseBlock`, `ResidualResidualDenseBlock`, `EMA`; reference model.py:22-27),
but `forward`/`backward` are one C-ABI call each into libresr_hip.so (include/resr.h), which
enqueues the hand-written gfx950 kernels.  There is no PyTorch/CPU fallback: CPU tensors raise.

Parameters stay OIHW fp32 `nn.Parameter`s (optimiser / checkpoint surface) but are *views into
one flat arena* in reference `named_parameters()` order, so weight packing, EMA and the
data-parallel all-reduce are single passes over contiguous HBM.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _lib

__all__ = ["EMA", "ResidualDenseBlock", "ResidualResidualDenseBlock", "Generator"]


def _precision_to_dtype(precision: str) -> int:
    if precision == "fast":
        return _lib.RESR_F16
    if precision == "strict":
        return _lib.RESR_F32
    if precision == "exact16":
        return _lib.RESR_F16X2
    raise ValueError("precision must be 'fast' (f16 MFMA, fp32 accumulate), 'exact16' (split-operand f16 MFMA, "
                     f"fp32-class results) or 'strict' (f32 MFMA), got {precision!r}")


class ResidualDenseBlock(nn.Module):
    """Parameter container for one dense block (reference model.py:64-106).

    The convs are `nn.Conv2d` objects so that construction consumes the RNG exactly like the
    reference (same weights under the same seed) and the state_dict keys match; the arithmetic
    runs inside `Generator.forward`."""

    def __init__(self, channels: int, growth_channels: int) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(channels + growth_channels * 0, growth_channels, (3, 3), (1, 1), (1, 1))
        self.conv2 = nn.Conv2d(channels + growth_channels * 1, growth_channels, (3, 3), (1, 1), (1, 1))
        self.conv3 = nn.Conv2d(channels + growth_channels * 2, growth_channels, (3, 3), (1, 1), (1, 1))
        self.conv4 = nn.Conv2d(channels + growth_channels * 3, growth_channels, (3, 3), (1, 1), (1, 1))
        self.conv5 = nn.Conv2d(channels + growth_channels * 4, channels, (3, 3), (1, 1), (1, 1))
        self.leaky_relu = nn.LeakyReLU(0.2, True)
        self.identity = nn.Identity()
        for m in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):  # model.py:100-106
            nn.init.kaiming_normal_(m.weight)
            m.weight.data *= 0.1
            nn.init.constant_(m.bias, 0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """model.py:87-98 on the conv kernel (forward only: training runs through `Generator`)."""
        return _dense_blocks_forward([self], x, rrdb=False)


class ResidualResidualDenseBlock(nn.Module):
    """Parameter container for one RRDB (reference model.py:109-132)."""

    def __init__(self, channels: int, growth_channels: int) -> None:
        super().__init__()
        self.rdb1 = ResidualDenseBlock(channels, growth_channels)
        self.rdb2 = ResidualDenseBlock(channels, growth_channels)
        self.rdb3 = ResidualDenseBlock(channels, growth_channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """model.py:123-132 on the conv kernel (forward only: training runs through `Generator`)."""
        return _dense_blocks_forward([self.rdb1, self.rdb2, self.rdb3], x, rrdb=True)


def _pack_one(conv: nn.Conv2d, dtype: int) -> torch.Tensor:
    """Packed forward weights of ONE conv (resr_pack_weights on a one-conv table)."""
    w = conv.weight.detach().float().contiguous()
    cout, cin = w.shape[:2]
    mt, nck = (cout + 31) // 32, (cin + 31) // 32
    chunks = (_lib.PackChunk * nck)()
    for ck in range(nck):
        chunks[ck] = _lib.PackChunk(0, ck * 9 * mt * 1024, cout, cin, 0, cout, ck * 32, min(32, cin - ck * 32), mt, 0, 1.0, 0, None)
    table = torch.frombuffer(bytearray(bytes(chunks)), dtype=torch.uint8).to(w.device)
    es = {_lib.RESR_F16: 2, _lib.RESR_F32: 4, _lib.RESR_F16X2: 6}[dtype]
    packed = torch.zeros(nck * 9 * mt * 1024 * es + 16384, dtype=torch.uint8, device=w.device)
    _lib.check(_lib.lib().resr_pack_weights(_lib.ptr(table), nck, _lib.ptr(w.reshape(-1)), _lib.ptr(packed), dtype,
                                            _lib.stream_ptr(w)), "resr_pack_weights")
    return packed


def _dense_blocks_forward(rdbs, x: torch.Tensor, rrdb: bool, precision: Optional[str] = None) -> torch.Tensor:
    """Standalone forward of one dense block / one RRDB (the reference exports both, model.py:22-27): five (fifteen)
    `resr_conv3x3` passes over interleaved [N,H,W,192] workspaces -- conv_k reads the channel prefix, writes its own
    32-channel slice, conv5 carries the `*0.2 + x` epilogue (and the RRDB's second residual).  Forward only."""
    _lib.require_cuda(x, "ResidualDenseBlock.forward")
    params = [p for r in rdbs for p in r.parameters()]
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        raise RuntimeError("standalone dense blocks run forward-only on the MI355X path: wrap the call in torch.no_grad() "
                           "(training differentiates through Generator, where the blocks are fused)")
    dtype = _precision_to_dtype(precision or os.environ.get("RESR_PRECISION", "fast"))
    L, lib = _lib, _lib.lib()
    n, c, h, w = x.shape
    if c != 64:
        raise RuntimeError("dense blocks take 64 channels")
    x2 = dtype == L.RESR_F16X2
    T = torch.float32 if dtype == L.RESR_F32 else torch.float16
    pairs, px = (2 if x2 else 1), n * h * w
    lo = px * 192 if x2 else 0
    st = L.stream_ptr(x)
    xc = x.detach().float().contiguous()
    tmp = torch.empty((pairs, n, h, w, 64), dtype=T, device=x.device)
    L.check(lib.resr_nchw_to_nhwc(L.ptr(xc), L.ptr(tmp), n, 64, h, w, 1, 64, dtype, None, st), "resr_nchw_to_nhwc")
    bufs = [torch.zeros((pairs, n, h, w, 192), dtype=T, device=x.device) for _ in range(len(rdbs) + 1)]
    bufs[0][..., :64] = tmp
    es = 4 if dtype == L.RESR_F32 else 2
    for r, rdb in enumerate(rdbs):
        cur, nxt = bufs[r], bufs[r + 1]
        for k in range(1, 6):
            conv = getattr(rdb, f"conv{k}")
            cin, cout = 64 + 32 * (k - 1), (32 if k < 5 else 64)
            d = L.ConvDesc(n, h, w, cin, cin, 192, 0, cout, cout, 192, 192, 192, 0, dtype, L.CONV_LRELU if k < 5 else 0,
                           0.2, 1.0, 0.2, 1.0, 0.2)
            d.in0_lo_offset = d.out_lo_offset = d.res0_lo_offset = d.res1_lo_offset = lo
            bias = conv.bias.detach().float().contiguous()
            if k < 5:
                out = C.c_void_p(cur.data_ptr() + cin * es)
                res0 = res1 = None
            else:
                out, res0 = L.ptr(nxt), L.ptr(cur)
                res1 = L.ptr(bufs[0]) if (rrdb and r == len(rdbs) - 1) else None
            L.check(lib.resr_conv3x3(C.byref(d), L.ptr(cur), None, L.ptr(_pack_one(conv, dtype)), L.ptr(bias), res0, res1, None,
                                     out, None, st), "resr_conv3x3")
    y = torch.empty((n, 64, h, w), dtype=torch.float32, device=x.device)
    L.check(lib.resr_nhwc_to_nchw(L.ptr(bufs[-1]), L.ptr(y), n, 64, h, w, 1, 192, dtype, st), "resr_nhwc_to_nchw")
    return y


class _Workspace:
    """One activation workspace; `busy` while an autograd graph that saved into it is alive.  `owner` counts the
    training-mode forwards that took it: only the graph that still owns it may release it (a stale token of an earlier
    graph, collected late, must not free a workspace a newer graph saved its activations in)."""

    def __init__(self, nbytes: int, device, zero_head: int = 0) -> None:
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if zero_head:      # the chain state of the dense-block launches: zero once (include/resr.h)
            self.buf[:zero_head].zero_()
        self.busy = False
        self.owner = 0

    def acquire(self) -> int:
        self.owner += 1
        self.busy = True
        return self.owner

    def release(self, owner: int) -> None:
        if owner == self.owner:
            self.busy = False


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "Generator", training: bool, x: torch.Tensor, *params: torch.Tensor):
        y, desc, ws = module._run_forward(x, training)
        ctx.module, ctx.desc, ctx.ws = module, desc, ws
        ctx.x_needs_grad = x.requires_grad
        ctx.n_params = len(params)
        ctx.owner = 0
        if training:
            ctx.owner = ws.acquire()
            ctx._token = _WsToken(ws, ctx.owner, module)   # frees the workspace when the graph is dropped without backward
        return y

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        module: Generator = ctx.module
        # Another live graph of this module (two forwards before either backward: GAN-style losses, or both in one
        # .backward()) may run its backward before autograd has consumed the views handed out here -- and that backward
        # overwrites the arena.  Only the last live graph may hand out arena views; the others hand out copies.
        private = module._live_graphs > 1
        grads, gx = module._run_backward(ctx.desc, ctx.ws, gy.contiguous().float(), ctx.x_needs_grad, private)
        ctx._token.finish()
        if len(grads) != ctx.n_params:             # flat_parameter() mode: the one alias was the graph's only parameter input
            grads = [None] * ctx.n_params
        return (None, None, gx) + tuple(grads)


class _WsToken:
    """Lifetime of one training-mode graph: counts it among the module's live graphs and gives the workspace back
    when its backward has run or when the graph is dropped without one."""

    def __init__(self, ws: _Workspace, owner: int, module: "Generator") -> None:
        self.ws, self.owner, self.module, self.open = ws, owner, module, True
        module._live_graphs += 1

    def finish(self) -> None:
        if self.open:
            self.open = False
            self.module._live_graphs -= 1
            self.ws.release(self.owner)

    def __del__(self) -> None:
        self.finish()


class Generator(nn.Module):
    """RRDBNet generator (reference model.py:206-275) on hand-written gfx950 kernels.

    Args mirror the reference: Generator(in_channels, out_channels, upscale_factor) with
    upscale_factor in {1, 2, 4}.  Extra keyword `precision`: "fast" = f16 operands on
    v_mfma_f32_32x32x16_f16 with fp32 accumulation (the reference's CUDA-autocast numerics class),
    "exact16" = the same matrix pipe with split operands (activations and weights as hi + lo f16 pairs, three
    MFMAs per product, fp32 accumulate: meets the 1e-3 parity tolerance vs the fp32 CPU path at about a third
    of fast mode's throughput), "strict" = f32 operands on v_mfma_f32_32x32x2_f32 (bit-for-bit fp32 FMA chains).
    Default from $RESR_PRECISION, else "fast".
    `x2_plan` (exact16 only; bit set of _lib.X2_PLAN_*, default from $RESR_X2_PLAN, else 763 = bits 0, 1, 3, 4, 5, 6, 7, 9; 59 = round 5's default without the MX stages): which tensors of the dense
    blocks are single f16 instead of hi/lo pairs -- bit 0: the growth planes o1..o4 of an INFERENCE forward (50 instead of 60
    stages per block; forward ~1e-6 at the reference's init scale, gate 2e-4), bit 1: the growth-plane gradients of the backward
    pass are READ as single f16 (two stages / two tap-products on their chunks; the bias sums still take hi + lo; worst gradient
    tensor 3-5e-4 vs float64, gate 1e-3), bit 2 (opt-in, with bit 1): they are stored single as well (~4 % faster, worst bias tensor
    6.7e-4), bit 3: the weight products read the growth planes (the X chunks of conv2..conv5 behind the residual stream) as their
    hi tensor (54 instead of 68 tap-products per block; the worst tensor does not move), bit 4 (with bit 3): conv5's products of the
    growth planes also take g_y's hi tensor alone (46 per block), bit 5 (with bit 0): the growth chunks of an inference forward meet
    the f16 weights W0 alone -- one stage each, 40 per block (forward 2.3e-6 at the init scale, 2.9e-5 at dense weights x 4), bit 6 (with bits 0 + 5): the PAIR
    chunks of an inference forward take one f16 stage + one MX stage -- both 2^-12-weighted correction products as nine
    v_mfma_scale_f32_32x32x64_f8f6f4 per output row on unscaled bf8 operands ([bf8(x_hi) | bf8(x_lo)] records written by the producing
    epilogues, [bf8(W1) | bf8(W2)] blocks from the packer): 30 stage-equivalents per block, forward ~1e-4 (gate 2e-4), bit 7: the dense
    blocks' backward-data passes read EVERY gradient chunk as a pair on one f16 + one MX stage (40 stage-equivalents per block instead of
    50; the growth-plane gradients enter with both halves again), bit 9 (with bits 7 and 3): the dense blocks' weight gradients take both
    2^-12-weighted tap-products of every stream chunk as ONE MX job (8-bit transpose reads of the q records, K = 32 pixels twice) --
    conv1..conv4 get their (x_hi, g_lo) term back at no cost: worst gradient tensor 5-7e-5 against the all-pairs plan instead of 2-3e-4;
    bit 8 (opt-in): exact16's forward in front of fast mode's backward pass; bit 10 (opt-in, with bit 9): the 4x-resolution tail (conv3, conv4, upsampling2)
    on MX stages / MX jobs too -- + 0.85 % on the step, every gradient tensor still within 6-8e-5 of the all-pairs plan (median 2.6e-5 -> 4.3e-5).  x2_plan=0 = pairs everywhere: forward 1.8e-6, every gradient tensor 5.8e-6.
    The backward pass of the 16-bit modes (exact16, fast) does not depend on the caller's loss scale: an incoming gradient whose largest element is below 2^6 is
    lifted by a power of two inside the native pass and the results are handed back unscaled (bit-identical gradients at loss scale
    1 and 2^20; csrc/generator.hip, $RESR_X2_GRAD_PRESCALE_LOG2 / RESR_X2_NO_GRAD_PRESCALE=1).
    forward(x[N,C,H,W] float in [0,1]) -> [N,out,H*s,W*s] clamped to [0,1]; differentiable.
    """

    N_BLOCKS = 23

    def __init__(self, in_channels: int, out_channels: int, upscale_factor: int,
                 precision: Optional[str] = None, n_blocks: Optional[int] = None, x2_plan: Optional[int] = None) -> None:
        super().__init__()
        if upscale_factor not in (1, 2, 4):
            raise ValueError("upscale_factor must be 1, 2 or 4")
        self.in_channels, self.out_channels, self.upscale_factor = in_channels, out_channels, upscale_factor
        self.precision = precision or os.environ.get("RESR_PRECISION", "fast")
        self._dtype = _precision_to_dtype(self.precision)
        self.x2_plan = int(os.environ.get("RESR_X2_PLAN", "763")) if x2_plan is None else int(x2_plan)
        if not 0 <= self.x2_plan <= 2047:
            raise ValueError(f"x2_plan must be a bit set of X2_PLAN_GROWTH_F16_INFER (1) | X2_PLAN_GROWTH_GRAD_F16 (2) | "
                             f"X2_PLAN_GROWTH_GRAD_STORE_F16 (4) | X2_PLAN_GROWTH_ACT_F16_WGRAD (8) | X2_PLAN_GROWTH_ACT_G_HI_WGRAD (16) | X2_PLAN_GROWTH_W16_INFER (32) | "
                             f"X2_PLAN_MX_INFER (64) | X2_PLAN_MX_BWD (128) | X2_PLAN_F16_BACKWARD (256) | X2_PLAN_MX_WGRAD (512) | X2_PLAN_MX_TAIL (1024), got {self.x2_plan}")
        if (self.x2_plan & 128) and (self.x2_plan & 4):
            raise ValueError(f"x2_plan={self.x2_plan}: MX_BWD (128) reads the growth-plane gradients as pairs; GROWTH_GRAD_STORE_F16 (4) stores them single")
        # a bit that only refines another one means nothing without it: refuse instead of silently ignoring it
        for bit, needs, name in ((4, 2, "GROWTH_GRAD_STORE_F16 (4) refines GROWTH_GRAD_F16 (2)"), (16, 8, "GROWTH_ACT_G_HI_WGRAD (16) refines GROWTH_ACT_F16_WGRAD (8)"),
                                 (32, 1, "GROWTH_W16_INFER (32) refines GROWTH_F16_INFER (1)"), (64, 33, "MX_INFER (64) rides on GROWTH_F16_INFER (1) + GROWTH_W16_INFER (32)"),
                                 (512, 128 + 8, "MX_WGRAD (512) rides on MX_BWD (128: the gradient planes' q tensors) + GROWTH_ACT_F16_WGRAD (8: the stream chunks are the pair chunks)"),
                                 (1024, 512 + 128 + 8, "MX_TAIL (1024) extends MX_WGRAD (512) to the 4x-resolution tail")):
            if (self.x2_plan & bit) and (self.x2_plan & needs) != needs:
                raise ValueError(f"x2_plan={self.x2_plan}: {name}")
        self.n_blocks = n_blocks or self.N_BLOCKS
        if upscale_factor == 2:
            conv_in, downscale_factor = in_channels * 4, 2
        elif upscale_factor == 1:
            conv_in, downscale_factor = in_channels * 16, 4
        else:
            conv_in, downscale_factor = in_channels, 1
        # same construction order as the reference so the RNG stream (hence the init) is identical
        self.downsampling = nn.PixelUnshuffle(downscale_factor)
        self.conv1 = nn.Conv2d(conv_in, 64, (3, 3), (1, 1), (1, 1))
        self.trunk = nn.Sequential(*[ResidualResidualDenseBlock(64, 32) for _ in range(self.n_blocks)])
        self.conv2 = nn.Conv2d(64, 64, (3, 3), (1, 1), (1, 1))
        self.upsampling1 = nn.Sequential(nn.Conv2d(64, 64, (3, 3), (1, 1), (1, 1)), nn.LeakyReLU(0.2, True))
        self.upsampling2 = nn.Sequential(nn.Conv2d(64, 64, (3, 3), (1, 1), (1, 1)), nn.LeakyReLU(0.2, True))
        self.conv3 = nn.Sequential(nn.Conv2d(64, 64, (3, 3), (1, 1), (1, 1)), nn.LeakyReLU(0.2, True))
        self.conv4 = nn.Conv2d(64, out_channels, (3, 3), (1, 1), (1, 1))

        self._flat: Optional[torch.Tensor] = None        # fp32 parameter arena
        self._flat_grad: Optional[torch.Tensor] = None   # fp32 gradient arena (written by backward)
        self._packed: Optional[torch.Tensor] = None
        self._packed_f16: Optional[torch.Tensor] = None   # x2_plan bit 8: a RESR_F16 packing of the same table for the f16 backward pass
        self._table_dev: Dict[int, tuple] = {}
        self._workspaces: Dict[tuple, List[_Workspace]] = {}
        self.grad_hook = None   # callable(flat_grad) run after backward wrote the arena (data-parallel all-reduce)
        # callable(flat_grad, ranges, events): data-parallel exchange OVERLAPPED with the backward pass -- events[i] fires on
        # the backward stream as soon as arena range ranges[i] is final (resr_generator_backward's grad_ready_events)
        self.grad_ready_hook = None
        self._events: List[torch.cuda.Event] = []
        self._live_graphs = 0   # training-mode forwards whose backward has not run yet
        self.__dict__["_flat_param"] = None   # see flat_parameter(); kept out of nn.Module's parameter registry

    # ---- flat arena ---------------------------------------------------------------------------
    def _ordered_params(self) -> List[nn.Parameter]:
        return [p for _, p in self.named_parameters()]

    def _flatten(self) -> None:
        params = self._ordered_params()
        dev = params[0].device
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).float())
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self._flat = flat
        self._flat_grad = None
        self._packed = None
        self._table_dev.clear()
        self._workspaces.clear()

    def _arena_ok(self) -> bool:
        if self._flat is None:
            return False
        params = self._ordered_params()
        base = self._flat.data_ptr()
        off = 0
        for p in params:
            if p.data_ptr() != base + off * 4 or p.dtype != torch.float32:
                return False
            off += p.numel()
        return off == self._flat.numel()

    def flat_parameters(self) -> torch.Tensor:
        """The fp32 arena all parameters are views of (reference named_parameters order)."""
        if not self._arena_ok():
            self._flatten()
        return self._flat

    def flat_grad(self) -> Optional[torch.Tensor]:
        return self._flat_grad

    def grad_ranges(self) -> List[tuple]:
        """Element ranges of the gradient arena in the order the backward pass finishes them (the order of
        resr_generator_backward's grad_ready_events): the tail convs (end of the arena), RRDB n-1 ... RRDB 0, conv1.
        Consecutive entries are adjacent, descending."""
        starts, off = {}, 0
        for name, p in self.named_parameters():
            starts.setdefault(name.split(".")[1] if name.startswith("trunk.") else name.split(".")[0], off)
            off += p.numel()
        total = off
        tail = starts["conv2"]
        blocks = [starts[str(b)] for b in range(self.n_blocks)] + [tail]
        ranges = [(tail, total)]
        for b in range(self.n_blocks - 1, -1, -1):
            ranges.append((blocks[b], blocks[b + 1]))
        ranges.append((0, blocks[0]))
        return ranges

    def flat_parameter(self) -> nn.Parameter:
        """One leaf Parameter that aliases the whole fp32 arena, for an optimizer that should see a single tensor:
        `optim.Adam([g.flat_parameter()], ..., fused=True)` is one elementwise launch (and one GradScaler unscale /
        inf-check) instead of multi-tensor launches over 702 views.  After this call backward hands the gradient arena
        to this Parameter's `.grad` (overwritten, not accumulated) and returns no per-tensor gradients.  The per-tensor
        Parameters stay valid views of the same storage; `state_dict()` is unchanged, the optimizer's own state_dict
        then holds one tensor (not loadable into the reference's per-tensor Adam state)."""
        flat = self.flat_parameters()
        fp = self.__dict__["_flat_param"]
        if fp is None:
            fp = nn.Parameter(flat, requires_grad=True)
            self.__dict__["_flat_param"] = fp
        elif fp.data_ptr() != flat.data_ptr():
            fp.data = flat
        return fp

    # ---- C-ABI plumbing -------------------------------------------------------------------------
    def _desc(self, x: torch.Tensor, training: bool) -> _lib.GeneratorDesc:
        n, c, h, w = x.shape
        if c != self.in_channels:
            raise RuntimeError(f"Generator: expected {self.in_channels} input channels, got {c}")
        return _lib.GeneratorDesc(n, h, w, self.in_channels, self.out_channels, self.upscale_factor,
                                  self.n_blocks, self._dtype, 1 if training else 0,
                                  int(os.environ.get("RESR_WGRAD_SPLITS", "0")),
                                  self.x2_plan if self._dtype == _lib.RESR_F16X2 else 0, 0)

    def _pack(self, desc: _lib.GeneratorDesc, backward: bool) -> None:
        L = _lib.lib()
        flat = self.flat_parameters()
        key = 1 if backward else 0
        if key not in self._table_dev:
            n = L.resr_generator_pack_table(C.byref(desc), key, None, 0)
            if n <= 0:
                _lib.check(int(n) if n < 0 else -1, "resr_generator_pack_table")
            host = (_lib.PackChunk * n)()
            n2 = L.resr_generator_pack_table(C.byref(desc), key, C.cast(host, C.c_void_p), n)
            assert n2 == n
            raw = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(flat.device)
            self._table_dev[key] = (raw, int(n))
        nbytes = L.resr_generator_packed_bytes(C.byref(desc), 1)
        if self._packed is None or self._packed.numel() < nbytes or self._packed.device != flat.device:
            self._packed = torch.zeros(nbytes, dtype=torch.uint8, device=flat.device)
        raw, n = self._table_dev[key]
        _lib.check(L.resr_pack_weights(_lib.ptr(raw), n, _lib.ptr(flat), _lib.ptr(self._packed), self._dtype,
                                       _lib.stream_ptr(flat)), "resr_pack_weights")
        if self._dtype == _lib.RESR_F16X2 and backward and (desc.x2_plan & _lib.X2_PLAN_F16_BACKWARD):
            fd = _lib.GeneratorDesc(desc.n, desc.h, desc.w, desc.in_channels, desc.out_channels, desc.upscale, desc.n_blocks, _lib.RESR_F16, 1, 0, 0, 0)
            nb16 = L.resr_generator_packed_bytes(C.byref(fd), 1)
            if self._packed_f16 is None or self._packed_f16.numel() < nb16 or self._packed_f16.device != flat.device:
                self._packed_f16 = torch.zeros(nb16, dtype=torch.uint8, device=flat.device)
            _lib.check(L.resr_pack_weights(_lib.ptr(raw), n, _lib.ptr(flat), _lib.ptr(self._packed_f16), _lib.RESR_F16, _lib.stream_ptr(flat)),
                       "resr_pack_weights (f16 backward)")
        if self._dtype == _lib.RESR_F16X2 and (((desc.x2_plan & _lib.X2_PLAN_MX_INFER) and not desc.training) or
                                               ((desc.x2_plan & _lib.X2_PLAN_MX_BWD) and desc.training)):
            # the MX blocks of the same table ([bf8(W1) | bf8(W2)] per tap and row), behind the f16 blocks of the packed buffer
            mx_off = int(L.resr_generator_mx_offset(C.byref(desc)))
            _lib.check(L.resr_pack_weights_mx(_lib.ptr(raw), n, _lib.ptr(flat), C.c_void_p(self._packed.data_ptr() + mx_off),
                                              _lib.stream_ptr(flat)), "resr_pack_weights_mx")

    def _workspace(self, desc: _lib.GeneratorDesc, device) -> _Workspace:
        L = _lib.lib()
        key = (desc.n, desc.h, desc.w, desc.training, desc.dtype, desc.wgrad_splits, desc.x2_plan & (_lib.X2_PLAN_MX_INFER | _lib.X2_PLAN_MX_BWD | _lib.X2_PLAN_MX_WGRAD | _lib.X2_PLAN_MX_TAIL))   # (the MX plans carry q tensors)
        pool = self._workspaces.setdefault(key, [])
        for ws in pool:
            if not ws.busy:
                return ws
        nbytes = L.resr_generator_workspace_bytes(C.byref(desc))
        if nbytes == 0:
            raise RuntimeError("resr_generator_workspace_bytes: unsupported shape")
        ws = _Workspace(nbytes, device, int(L.resr_generator_chain_state_bytes(C.byref(desc))))
        pool.append(ws)
        return ws

    def _run_forward(self, x: torch.Tensor, training: bool):
        _lib.require_cuda(x, "Generator.forward")
        L = _lib.lib()
        flat = self.flat_parameters()
        _lib.require_cuda(flat, "Generator parameters")
        xc = x.detach().float().contiguous()  # also normalises channels_last strides (inference.py:28,49)
        desc = self._desc(xc, training)
        self._pack(desc, backward=training)
        ws = self._workspace(desc, xc.device)
        s = self.upscale_factor
        y = torch.empty((desc.n, self.out_channels, desc.h * s, desc.w * s), dtype=torch.float32, device=xc.device)
        _lib.check(L.resr_generator_forward(C.byref(desc), _lib.ptr(xc), _lib.ptr(flat), _lib.ptr(self._packed),
                                            _lib.ptr(ws.buf), ws.buf.numel(), _lib.ptr(y), _lib.stream_ptr(xc)),
                   "resr_generator_forward")
        return y, desc, ws

    def _run_backward(self, desc, ws: _Workspace, gy: torch.Tensor, need_gx: bool, private: bool = False):
        L = _lib.lib()
        flat = self.flat_parameters()
        if self._flat_grad is None or self._flat_grad.device != flat.device:
            self._flat_grad = torch.zeros_like(flat)
        gx = None
        if need_gx:
            gx = torch.empty((desc.n, self.in_channels, desc.h, desc.w), dtype=torch.float32, device=gy.device)
        # The native backward OVERWRITES the gradient arena.  If gradients of an earlier backward are still held (no
        # zero_grad in between: gradient accumulation, or two losses through one generator), they may alias the arena
        # (autograd keeps the views it was handed), so they are parked and put back, and this call's gradients are
        # handed out as a separate tensor for autograd to add.
        # (flat_parameter() mode keeps its documented overwrite semantics: the alias is not a module parameter, so
        # `model.zero_grad()` never clears its .grad, and every backward simply replaces the arena.)
        fp = self.__dict__["_flat_param"]
        accumulate = fp is None and any(p.grad is not None for p in self._ordered_params())
        prev = self._flat_grad.clone() if accumulate else None
        ev_arr, n_ev = None, 0
        if self.grad_ready_hook is not None:
            n_ev = self.n_blocks + 2
            while len(self._events) < n_ev:               # torch creates the HIP event at the first record()
                e = torch.cuda.Event()
                e.record(torch.cuda.current_stream(gy.device))
                self._events.append(e)
            ev_arr = (C.c_void_p * n_ev)(*[e.cuda_event for e in self._events[:n_ev]])
        f16bwd = self._dtype == _lib.RESR_F16X2 and (desc.x2_plan & _lib.X2_PLAN_F16_BACKWARD)
        _lib.check(L.resr_generator_backward(C.byref(desc), _lib.ptr(gy), _lib.ptr(flat), _lib.ptr(self._packed_f16 if f16bwd else self._packed),
                                             _lib.ptr(ws.buf), ws.buf.numel(), _lib.ptr(self._flat_grad),
                                             _lib.ptr(gx), _lib.stream_ptr(gy), ev_arr, n_ev),
                   "resr_generator_backward")
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(self._flat_grad, self.grad_ranges(), self._events[:n_ev])
        elif self.grad_hook is not None:
            self.grad_hook(self._flat_grad)
        if fp is not None:                       # flat_parameter() mode: the arena is the gradient of the alias
            if fp.data_ptr() != flat.data_ptr():
                fp.data = flat
            fp.grad = self._flat_grad
            return [None] * len(self._ordered_params()), gx
        src = self._flat_grad
        if prev is not None:
            src = self._flat_grad.clone()        # this call's gradients, for autograd to accumulate
            self._flat_grad.copy_(prev)          # the earlier ones back where .grad may be looking
        elif private:
            src = self._flat_grad.clone()        # another live graph will overwrite the arena before autograd reads these
        grads, off = [], 0
        for p in self._ordered_params():
            n = p.numel()
            grads.append(src[off:off + n].view(p.shape) if p.requires_grad else None)
            off += n
        return grads, gx

    # ---- module surface ---------------------------------------------------------------------------
    def _forward_impl(self, x: torch.Tensor) -> torch.Tensor:
        flat = self.flat_parameters()
        fp = self.__dict__["_flat_param"]
        if fp is not None and fp.data_ptr() != flat.data_ptr():
            fp.data = flat                       # the arena was rebuilt (.to(), new tensors loaded): keep the alias on it
        # flat_parameter() mode: the alias stands for all 702 tensors in the autograd graph (its .grad is set by hand in backward),
        # so a backward pass does not walk 702 AccumulateGrad nodes that would each receive None
        params = [fp] if fp is not None else self._ordered_params()
        # grad mode is off inside autograd.Function.forward, so decide here whether to keep activations
        training = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _GeneratorFn.apply(self, training, x, *params)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._forward_impl(x)


class EMA(nn.Module):
    """Exponential moving average of a model's trainable parameters (reference model.py:30-61).

    Same surface (`register/update/apply_shadow/restore`, `.shadow`, `.backup` dicts).  When the
    wrapped model exposes `flat_parameters()` the update is one fused launch over the flat arena
    (resr_ema_update) that reproduces the reference's rounding: (1-d)*p and d*shadow rounded
    separately, then added."""

    def __init__(self, model: nn.Module, weight_decay: float) -> None:
        super().__init__()
        self.model = model
        self.weight_decay = weight_decay
        self.shadow: Dict[str, torch.Tensor] = {}
        self.backup: Dict[str, torch.Tensor] = {}
        self._flat_shadow: Optional[torch.Tensor] = None

    def _flat_ok(self) -> bool:
        return (hasattr(self.model, "flat_parameters")
                and all(p.requires_grad for p in self.model.parameters()))

    def register(self) -> None:
        if self._flat_ok():
            flat = self.model.flat_parameters()
            _lib.require_cuda(flat, "EMA.register")
            self._flat_shadow = flat.detach().clone()
            off = 0
            self.shadow = {}
            for name, p in self.model.named_parameters():
                n = p.numel()
                self.shadow[name] = self._flat_shadow[off:off + n].view(p.shape)
                off += n
            return
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                _lib.require_cuda(param, "EMA.register")
                self.shadow[name] = param.data.clone()

    def update(self) -> None:
        if self._flat_shadow is not None and self._flat_ok():
            flat = self.model.flat_parameters()
            _lib.check(_lib.lib().resr_ema_update(_lib.ptr(self._flat_shadow), _lib.ptr(flat), flat.numel(),
                                                  float(self.weight_decay), _lib.stream_ptr(flat)), "resr_ema_update")
            return
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                assert name in self.shadow
                s = self.shadow[name]
                _lib.check(_lib.lib().resr_ema_update(_lib.ptr(s), _lib.ptr(param.data.contiguous()), s.numel(),
                                                      float(self.weight_decay), _lib.stream_ptr(s)), "resr_ema_update")

    def apply_shadow(self) -> None:
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                assert name in self.shadow
                self.backup[name] = param.data
                param.data = self.shadow[name]

    def restore(self) -> None:
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                assert name in self.backup
                param.data = self.backup[name]
