"""Spatial tiling of large inputs with a hipGraph-captured tile loop (BASELINE config 5: x2 model on a
3840x2160 LR frame).  Net-new relative to the reference, which runs whole images (inference.py:53).

The frame is cut into a grid of equal tiles; every tile is computed on a fixed-size window (tile + 2*halo, shifted
inwards at the frame border so that all windows have the same shape) and only the tile's own pixels are stitched into
the result.  Per pixel the arithmetic is that of `Generator` on the window, so with halo >= the network's
receptive-field radius the stitched image equals whole-image inference bit for bit (tested with a short trunk); with
the 23-block trunk (radius ~350 LR px) the halo is a quality / speed choice.

Tile choice (`tile=None`): the FEWEST equal tiles whose window still fits the conv kernels' 32-bit addressing (2^24
output pixels per tensor) and that compute the fewest pixels: fewer, larger windows waste less on halos and on the inward
shift -- a 3840x2160 frame of the x2 model is three full-height columns (windows 2160x1344: 1.05x the frame's pixels;
fixed 1024-pixel tiles computed 1.71x).

Precision: the tiler runs the `Generator` it is given, in that model's arithmetic.  The reference has no tiler; its whole-image
inference call sites run fp32 (inference.py:52-53), so a caller that wants the reference's numbers builds the model the way
`inference.py` does -- `Generator(..., precision=config.inference_precision)`, i.e. "exact16" -- and one that wants frames per
second (BASELINE config 5: "fp16-class" tiled inference) builds it with precision="fast".

hipGraph: the WHOLE frame loop -- for every tile: window gather, the ~350 launches of the generator, the stitch copy --
is captured once per (frame shape, parameter arena) into one graph over static frame-in / frame-out buffers and replayed
per frame: one host call per frame instead of ~355 per tile.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

from .model import Generator

_MAX_OUT_PIXELS = 1 << 24        # conv3x3_ws.hip: per-tensor pixel limit of the 32-bit lane offsets
# Halo (LR pixels) of the tiles `super_resolve` cuts when a frame exceeds that limit.  The 23-block trunk's receptive field is
# ~350 LR pixels, but what a far pixel contributes decays fast (every dense block and every RRDB adds its branch times 0.2):
# measured seam error against a whole-image pass on a 512^2 frame, 23 blocks (tools/tile_halo_error.py) --
# the default keeps it below the entry points' uint8 output step at the reference's init scale AND at dense weights x 4.
DEFAULT_HALO = int(__import__("os").environ.get("RESR_TILE_HALO", "64"))


def fits_whole(model: Generator, n: int, H: int, W: int) -> bool:
    """Can `model` take an [n, c, H, W] batch as ONE launch sequence?  The producer/consumer conv kernels address a tensor with
    24 x 24-bit offsets: at most 2^24 pixels per tensor, i.e. of the HR-resolution tail (csrc/conv3x3_ws.hip,
    conv3x3_ws_supported; csrc/generator.hip additionally wants n * h * w * 512 < 2^31 elements per plane stack)."""
    s = model.upscale_factor
    r = {4: 1, 2: 2, 1: 4}[s]
    return n * H * s * W * s <= _MAX_OUT_PIXELS and n * (H // r) * (W // r) * 512 <= 0x7fffffff


@torch.no_grad()
def super_resolve(model: Generator, x: torch.Tensor, halo: Optional[int] = None) -> torch.Tensor:
    """`model(x)` for a frame of ANY size (the reference's whole-image call sites, inference.py:52-53 and test.py:79-80, run
    whatever memory allows): one pass when the frame fits the kernels' per-tensor limit, else equal tiles of the fewest pixels
    with `halo` LR pixels of context (TiledGenerator, no graph: one frame per call) stitched into the full image."""
    n, _, H, W = x.shape
    if fits_whole(model, n, H, W):
        return model(x)
    return TiledGenerator(model, tile=None, halo=DEFAULT_HALO if halo is None else halo, use_graph=False)(x.contiguous())


class TiledGenerator:
    def __init__(self, model: Generator, tile=None, halo: int = 32, use_graph: bool = True) -> None:
        """tile: None (automatic, see module docstring), an int (square tiles of that edge at most) or (tile_h, tile_w)."""
        self.model, self.tile, self.halo, self.use_graph = model, tile, halo, use_graph
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._key = None
        self._frame_in: Optional[torch.Tensor] = None
        self._frame_out: Optional[torch.Tensor] = None
        self._held = None

    # ---- geometry -------------------------------------------------------------------------------------------
    def _grid(self, n: int, H: int, W: int) -> Tuple[int, int]:
        """(rows, cols) of the tile grid."""
        if self.tile is not None:
            th, tw = (self.tile, self.tile) if isinstance(self.tile, int) else self.tile
            return max(1, math.ceil(H / th)), max(1, math.ceil(W / tw))
        s = self.model.upscale_factor
        r = {4: 1, 2: 2, 1: 4}[s]                       # windows are rounded up to the pixel-unshuffle factor (plan()): price them that way
        halo = math.ceil(self.halo / r) * r
        best = None
        for ny in range(1, 65):
            for nx in range(1, 65):
                wh = min(H, math.ceil(math.ceil(H / ny) / r) * r + 2 * halo)
                ww = min(W, math.ceil(math.ceil(W / nx) / r) * r + 2 * halo)
                if n * wh * s * ww * s > _MAX_OUT_PIXELS:
                    continue
                cost = (ny * nx * wh * ww, ny * nx)
                if best is None or cost < best[0]:
                    best = (cost, ny, nx)
        if best is None:
            raise RuntimeError("TiledGenerator: no tile grid fits the frame")
        return best[1], best[2]

    def plan(self, n: int, H: int, W: int):
        """[(y0, y1, x0, x1, wy, wx)] tile bounds and window origins, plus the common window size."""
        r = {4: 1, 2: 2, 1: 4}[self.model.upscale_factor]          # pixel-unshuffle factor: windows stay multiples of it
        ny, nx = self._grid(n, H, W)
        th, tw = math.ceil(H / ny), math.ceil(W / nx)
        th, tw = math.ceil(th / r) * r, math.ceil(tw / r) * r
        halo = math.ceil(self.halo / r) * r
        win_h, win_w = min(H, th + 2 * halo), min(W, tw + 2 * halo)
        tiles: List[tuple] = []
        for y0 in range(0, H, th):
            y1 = min(H, y0 + th)
            wy = min(max(y0 - halo, 0), H - win_h)
            for x0 in range(0, W, tw):
                x1 = min(W, x0 + tw)
                wx = min(max(x0 - halo, 0), W - win_w)
                tiles.append((y0, y1, x0, x1, wy, wx))
        return tiles, win_h, win_w

    # ---- execution ------------------------------------------------------------------------------------------
    def _run_tiles(self, x: torch.Tensor, out: torch.Tensor, tiles, win_h: int, win_w: int) -> None:
        s = self.model.upscale_factor
        for (y0, y1, x0, x1, wy, wx) in tiles:
            sr = self.model(x[:, :, wy:wy + win_h, wx:wx + win_w])      # the generator makes its own contiguous copy
            out[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = sr[:, :, (y0 - wy) * s:(y1 - wy) * s, (x0 - wx) * s:(x1 - wx) * s]

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns a NEW tensor per call (or fills `out`): in graph mode the replay writes a static buffer, which the next
        frame overwrites -- `[tg(f) for f in frames]` must not alias it.  `out=` saves the copy's allocation in a frame loop."""
        n, c, H, W = x.shape
        s = self.model.upscale_factor
        tiles, win_h, win_w = self.plan(n, H, W)
        if not self.use_graph:
            if out is None:
                out = torch.empty((n, self.model.out_channels, H * s, W * s), dtype=torch.float32, device=x.device)
            self._run_tiles(x.float(), out, tiles, win_h, win_w)
            return out
        # The captured launches bake in the addresses of the parameter arena, the packed weights and the workspace:
        # re-capture when the model re-flattened its parameters (EMA.apply_shadow / restore, .to(), loading into new
        # tensors), and hold references to everything the graph reads so the allocator cannot recycle it meanwhile.
        key = (tuple(x.shape), str(x.device), self.model.flat_parameters().data_ptr())
        if self._graph is None or self._key != key:
            self._graph = None
            self._frame_in = x.float().clone()
            self._frame_out = torch.empty((n, self.model.out_channels, H * s, W * s), dtype=torch.float32, device=x.device)
            for _ in range(2):                          # warm-up: one-time init + workspace allocation outside capture
                self.model(self._frame_in[:, :, :win_h, :win_w])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._run_tiles(self._frame_in, self._frame_out, tiles, win_h, win_w)
            self._graph, self._key = g, key
            self._held = (self.model.flat_parameters(), self.model._packed, dict(self.model._workspaces))
        self._frame_in.copy_(x)
        self._graph.replay()
        if out is None:
            return self._frame_out.clone()
        out.copy_(self._frame_out)
        return out
