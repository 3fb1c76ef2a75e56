"""CPU: the two facts the workgroup-parallel border follower (vlfm_amd/csrc/border_parallel.h) rests on, checked with a plain
NumPy model of its state machine against the restated cv2.findContours (oracle/cvport.c):

  1. the successor map f(p, s_back) = (p', s_back') -- p' the first set neighbour of p counter-clockwise after s_back,
     s_back' the direction from p' back to p -- is INJECTIVE on the valid states of any bitmap (so every state lies on a cycle);
  2. the outer border OpenCV traces from a component's raster-first pixel is exactly the cycle of f through
     state0 = (i0, direction of the first neighbour found by the clockwise search from west), and CHAIN_APPROX_SIMPLE keeps a
     state iff its outgoing direction differs from s_back ^ 4.
"""
import numpy as np
import pytest

DX = [1, 1, 0, -1, -1, -1, 0, 1]      # chain codes: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE (y grows downwards)
DY = [0, -1, -1, -1, 0, 1, 1, 1]


def _nbr8(img, x, y):
    h, w = img.shape
    m = 0
    for s in range(8):
        xx, yy = x + DX[s], y + DY[s]
        if 0 <= xx < w and 0 <= yy < h and img[yy, xx]:
            m |= 1 << s
    return m


def _succ(img, x, y, s_back):
    nb = _nbr8(img, x, y)
    for k in range(1, 9):
        s = (s_back + k) & 7
        if nb >> s & 1:
            return x + DX[s], y + DY[s], (s + 4) & 7, s
    raise AssertionError("a valid state has a set neighbour")


def _valid_states(img):
    ys, xs = np.nonzero(img)
    for x, y in zip(xs.tolist(), ys.tolist()):
        nb = _nbr8(img, x, y)
        for s in range(8):
            if nb >> s & 1:
                yield x, y, s


def _images():
    rng = np.random.default_rng(4)
    out = [(rng.uniform(size=(24, 31)) < d).astype(np.uint8) for d in (0.15, 0.4, 0.6, 0.85)]
    blob = np.zeros((30, 40), np.uint8)
    blob[3:27, 4:36] = 1
    blob[10:20, 12:28] = 0
    blob[13:17, 16:24] = 1          # nested component
    blob[5, 4:20] = 0               # a slit
    blob[28, 10] = 1                # isolated pixel
    blob[0, 0:5] = 1                # touches the image border
    out.append(blob)
    chk = (np.add.outer(np.arange(20), np.arange(26)) % 2 == 0).astype(np.uint8)
    out.append(chk)
    return out


@pytest.mark.parametrize("k", range(6))
def test_successor_map_is_injective(k):
    img = _images()[k]
    seen = {}
    for st in _valid_states(img):
        nx, ny, nsb, _ = _succ(img, *st)
        assert img[ny, nx] and (_nbr8(img, nx, ny) >> nsb & 1)       # a valid state again
        assert (nx, ny, nsb) not in seen, (st, seen.get((nx, ny, nsb)))
        seen[(nx, ny, nsb)] = st


@pytest.mark.parametrize("method", [1, 2])
@pytest.mark.parametrize("k", range(6))
def test_cycle_through_state0_is_the_opencv_chain(k, method):
    from oracle import cv

    img = _images()[k]
    want, _ = cv.findContours(img, cv.RETR_EXTERNAL, method)
    for contour in want:
        pts = contour.reshape(-1, 2)
        # the border's start pixel is its raster-first one: smallest y, then smallest x
        order = np.lexsort((pts[:, 0], pts[:, 1]))
        x0, y0 = (int(v) for v in pts[order[0]])
        nb = _nbr8(img, x0, y0)
        assert not (nb >> 4 & 1)                                      # west neighbour clear
        if nb == 0:                                                   # isolated pixel: one point, no state
            assert len(pts) == 1
            continue
        s0 = next(s for s in (3, 2, 1, 0, 7, 6, 5) if nb >> s & 1)    # clockwise search from west: s_end - 1, s_end - 2, ...
        chain, state = [], (x0, y0, s0)
        while True:
            x, y, sb = state
            nx, ny, nsb, s_out = _succ(img, x, y, sb)
            if method == 1 or s_out != (sb ^ 4):
                chain.append((x, y))
            state = (nx, ny, nsb)
            if state == (x0, y0, s0):
                break
            assert len(chain) <= 8 * img.size
        assert np.array_equal(np.array(chain), pts), (k, method, len(chain), len(pts))
