"""Host logic of the tiler (real_esrgan-pytorch_amd/tiling.py): grid choice and window geometry, no GPU needed."""
import real_esrgan_pytorch_amd as R
from real_esrgan_pytorch_amd.tiling import TiledGenerator


def _cover(tiles, H, W):
    seen = [[0] * W for _ in range(H)]
    for (y0, y1, x0, x1, wy, wx) in tiles:
        for y in range(y0, y1):
            for x in range(x0, x1):
                seen[y][x] += 1
    return all(v == 1 for row in seen for v in row)


def test_config5_frame_is_a_few_large_windows():
    g = R.Generator(3, 3, 2, n_blocks=1)
    tg = TiledGenerator(g, halo=32)
    tiles, wh, ww = tg.plan(1, 2160, 3840)
    assert len(tiles) <= 4                                                  # 1 x 3 columns of 2160 x 1344 windows
    assert len(tiles) * wh * ww / (2160 * 3840) < 1.10                      # pixels computed per frame pixel (1024-px tiles: 1.71)
    assert wh * 2 * ww * 2 <= 1 << 24                                       # the conv kernels' per-tensor pixel limit
    for (y0, y1, x0, x1, wy, wx) in tiles:
        assert 0 <= wy <= y0 and y1 <= wy + wh <= 2160 and 0 <= wx <= x0 and x1 <= wx + ww <= 3840
        assert wy % 2 == 0 and wx % 2 == 0 and wh % 2 == 0 and ww % 2 == 0   # pixel-unshuffle alignment of the x2 model
        # every interior edge of a tile has its full halo
        assert (y0 == 0 or y0 - wy >= 32) and (y1 == 2160 or wy + wh - y1 >= 32)
        assert (x0 == 0 or x0 - wx >= 32) and (x1 == 3840 or wx + ww - x1 >= 32)


def test_tiles_partition_the_frame():
    g4 = R.Generator(3, 3, 4, n_blocks=1)
    for tile, (H, W) in ((None, (37, 53)), (16, (37, 53)), ((12, 20), (37, 53)), (None, (2048, 2048))):
        tiles, wh, ww = TiledGenerator(g4, tile=tile, halo=4).plan(1, H, W)
        assert wh <= H and ww <= W
        if H * W < 10000:
            assert _cover(tiles, H, W)
        else:
            assert sum((y1 - y0) * (x1 - x0) for (y0, y1, x0, x1, _, _) in tiles) == H * W
