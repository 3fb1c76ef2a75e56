"""CPU: the tile order of csrc/gemm_f16.hip's 8-phase kernels (vlfm_gemm_f16_tile_order evaluates the kernels' own tile_of() on the
host).  The kernels trust it blindly -- a tile listed twice is computed twice and one is never written -- so: a bijection for every
shape and grouping, half n-tiles last within each XCD, and the persistent kernel's walk (workgroup w takes list positions w, w + grid,
...) covers every tile exactly once while a workgroup stays on one XCD."""
import ctypes

import numpy as np
import pytest

SHAPES = [(65792, 6144), (65792, 4224), (65792, 1408), (8224, 6144), (257, 6144), (7000, 2568), (5000, 4224), (300, 264), (130, 8),
          (256, 256), (2049, 640), (100000, 1408), (33 * 256, 129), (256 * 9, 256 * 3 + 128), (1, 8)]


def _order(m, n, gm):
    from vlfm_amd import _lib

    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.vlfm_gemm_f16_tile_order.restype = ctypes.c_int
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    out = np.zeros((tiles, 2), np.int32)
    got = L.vlfm_gemm_f16_tile_order(m, n, gm, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), tiles)
    assert got == tiles
    return out


@pytest.mark.parametrize("gm", [0, 1, 2, 4, 8, 32])
@pytest.mark.parametrize("shape", SHAPES)
def test_tile_order_is_a_bijection(shape, gm):
    m, n = shape
    o = _order(m, n, gm)
    tm_n, tn_n = (m + 255) // 256, (n + 255) // 256
    assert o[:, 0].min() >= 0 and o[:, 0].max() < tm_n and o[:, 1].min() >= 0 and o[:, 1].max() < tn_n
    keys = o[:, 0].astype(np.int64) * tn_n + o[:, 1]
    assert len(np.unique(keys)) == tm_n * tn_n


@pytest.mark.parametrize("shape", [(65792, 4224), (65792, 1408), (7000, 2568), (256 * 9, 256 * 3 + 128)])
def test_half_tiles_come_last_within_each_xcd(shape):
    m, n = shape
    o = _order(m, n, 0)
    last = (n + 255) // 256 - 1
    assert n - last * 256 <= 128                     # these shapes end in a half-empty n-tile
    for xcd in range(8):
        mine = o[xcd::8, 1] == last                  # list positions xcd, xcd + 8, ...: the XCD's share in the order it runs
        first_half = int(np.argmax(mine)) if mine.any() else len(mine)
        assert mine[first_half:].all(), (xcd, mine)


@pytest.mark.parametrize("grid", [256, 304, 32, 8])
@pytest.mark.parametrize("shape", [(65792, 6144), (65792, 4224), (7000, 2568)])
def test_persistent_walk_covers_every_tile_once_and_keeps_its_xcd(shape, grid):
    m, n = shape
    o = _order(m, n, 0)
    tiles = len(o)
    g = min(grid, tiles)
    seen = np.zeros(tiles, np.int32)
    for w in range(g):
        pos = np.arange(w, tiles, g)
        seen[pos] += 1
        if g % 8 == 0:
            assert ((pos & 7) == (w & 7)).all()
    assert (seen == 1).all()


def test_tile_order_rejects_bad_arguments():
    from vlfm_amd import _lib

    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.vlfm_gemm_f16_tile_order.restype = ctypes.c_int
    out = (ctypes.c_int * 4)()
    assert L.vlfm_gemm_f16_tile_order(0, 8, 0, out, 2) < 0
    assert L.vlfm_gemm_f16_tile_order(1024, 1024, 0, out, 2) < 0      # 16 tiles do not fit 2 pairs
    assert L.vlfm_gemm_f16_tile_order(256, 256, 0, None, 2) < 0


@pytest.mark.parametrize("grid", [256, 304, 32, 8])
@pytest.mark.parametrize("shape", SHAPES)
def test_work_items_cover_every_tile_once_whole_or_as_two_halves(shape, grid):
    from vlfm_amd import _lib

    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.vlfm_gemm_f16_work_items.restype = ctypes.c_int
    m, n = shape
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    out = np.full((2 * tiles, 2), -7, np.int32)
    items = L.vlfm_gemm_f16_work_items(m, n, grid, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), 2 * tiles)
    assert tiles <= items <= tiles + grid // 2
    it = out[:items]
    whole = np.zeros(tiles, np.int32)
    halves = np.zeros((tiles, 2), np.int32)
    for pos, h in it:
        assert 0 <= pos < tiles and h in (-1, 0, 1)
        if h < 0:
            whole[pos] += 1
        else:
            halves[pos, h] += 1
    split = halves.sum(axis=1) > 0
    assert (whole[~split] == 1).all() and (whole[split] == 0).all() and (halves[split] == 1).all()
    g = min(grid, tiles)
    left = tiles % g if tiles > g else 0
    if left and 2 * left <= g:
        assert split.sum() == left and split[tiles - left:].all()      # exactly the ragged round, and it becomes one full-width round
        assert items == tiles + left
    else:
        assert not split.any()
    per_wg = np.bincount(np.arange(items) % g, minlength=g)
    assert per_wg.max() - per_wg.min() <= 1


def test_the_vit_fc1_shapes_end_in_split_rounds():
    """fc1 at 256 / 128 / 64 / 32 images: 24.1 / 12.1 / 6.1 / 3.1 rounds of 256 workgroups -- the 24 leftover tiles become 48 half items."""
    from vlfm_amd import _lib

    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.vlfm_gemm_f16_work_items.restype = ctypes.c_int
    for images in (256, 128, 64, 32):
        m = images * 257
        tiles = ((m + 255) // 256) * 24
        out = np.zeros((2 * tiles, 2), np.int32)
        assert L.vlfm_gemm_f16_work_items(m, 6144, 256, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), 2 * tiles) == tiles + tiles % 256
        assert tiles % 256 == 24
