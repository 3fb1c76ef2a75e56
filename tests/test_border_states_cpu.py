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


# ----------------------------------------------------------------------------------------------------------------------
# Round 4: the facts the LDS-table form (wg_build_rank_lds / wg_emit_border_lds) adds on top of the two above.
#   3. a traced border (outer or hole) only visits BORDER pixels (set, with a clear 8-neighbour), so states entered from an
#      interior pixel can be left out, and so can states whose successor pixel is not a border pixel (v2_alloc_mask);
#   4. with those ids, a state is either on a genuine cycle or its chain ends in a fixed point ("dead"): nothing leads into a
#      genuine cycle from outside, and every state of every traced border has an id and is alive;
#   5. the in-place rounds (A = (jump, dist), B = (largest id in the window, distance to its first occurrence), a reader takes
#      A before B, a writer publishes B before A) end, under ANY update order within a round, with B = (cycle's largest id,
#      distance to it) for every live state, and stop one round after that is first true;
#   6. the closed form of the 8-connected line iterator used by raster_edge_shared.
def _border_mask(img):
    h, w = img.shape
    pad = np.pad(img, 1)
    full = np.ones_like(img)
    for s in range(8):
        full &= pad[1 + DY[s]:1 + DY[s] + h, 1 + DX[s]:1 + DX[s] + w]
    return (img == 1) & (full == 0)


def _alloc_states(img):
    """ids in the kernel's order (raster order of the border pixels, then direction) and the successor id of every state (its own
    id for a fixed point)."""
    border = _border_mask(img).astype(np.uint8)
    ids, order = {}, []
    ys, xs = np.nonzero(border)
    for x, y in zip(xs.tolist(), ys.tolist()):
        bnb = _nbr8(border, x, y)
        for d in range(8):
            if bnb >> d & 1:
                nx, ny, _, s = _succ(img, x, y, d)
                if border[ny, nx]:
                    ids[(x, y, d)] = len(order)
                    order.append((x, y, d))
    nxt = []
    for (x, y, d) in order:
        nx, ny, nsb, _ = _succ(img, x, y, d)
        nxt.append(ids.get((nx, ny, nsb), ids[(x, y, d)]))
    return ids, order, np.array(nxt, np.int64), border


@pytest.mark.parametrize("k", range(6))
def test_traced_borders_visit_border_pixels_and_their_states_have_ids(k):
    from oracle import cv

    img = _images()[k]
    ids, order, nxt, border = _alloc_states(img)
    # fixed points absorb their chains; everything else is a permutation of itself (no state outside a cycle leads into one)
    n = len(order)
    alive = np.ones(n, bool)
    alive[nxt == np.arange(n)] = False
    for _ in range(n):
        dead_next = ~alive[nxt] & alive
        if not dead_next.any():
            break
        alive[dead_next] = False
    live = np.flatnonzero(alive)
    assert len(set(nxt[live].tolist())) == len(live) and set(nxt[live].tolist()) == set(live.tolist())
    for mode in (cv.RETR_EXTERNAL, cv.RETR_LIST):
        contours, _ = cv.findContours(img, mode, 1)
        for c in contours:
            pts = c.reshape(-1, 2)
            assert all(border[y, x] for x, y in pts.tolist())
    # every state of every traced outer border has an id and is alive
    for contour in cv.findContours(img, cv.RETR_EXTERNAL, 1)[0]:
        pts = contour.reshape(-1, 2)
        o = np.lexsort((pts[:, 0], pts[:, 1]))
        x0, y0 = (int(v) for v in pts[o[0]])
        nb = _nbr8(img, x0, y0)
        if nb == 0:
            continue
        s0 = next(s for s in (3, 2, 1, 0, 7, 6, 5) if nb >> s & 1)
        state = (x0, y0, s0)
        while True:
            assert state in ids and alive[ids[state]], state
            nx, ny, nsb, _ = _succ(img, *state)
            state = (nx, ny, nsb)
            if state == (x0, y0, s0):
                break


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("k", range(6))
def test_in_place_ranking_under_any_update_order(k, seed):
    img = _images()[k]
    _, order, nxt, _ = _alloc_states(img)
    n = len(order)
    if n == 0:
        return
    DEAD = 0xFFFF
    rng = np.random.default_rng(100 * k + seed)
    jump, dist = nxt.copy(), np.ones(n, np.int64)
    m = np.where(nxt == np.arange(n), DEAD, np.arange(n))
    dm = np.zeros(n, np.int64)
    # ground truth by walking
    want_m, want_dm = np.full(n, DEAD), np.zeros(n, np.int64)
    for i in range(n):
        seen, j, steps = [], i, 0
        while j not in seen and nxt[j] != j:
            seen.append(j)
            j = nxt[j]
        if nxt[j] == j:
            continue                      # the chain ends in a fixed point: dead
        cyc = seen[seen.index(j):]
        assert i in cyc                   # (no live state outside a cycle: fact 4)
        h = max(cyc)
        want_m[i], j, steps = h, i, 0
        while j != h:
            j, steps = nxt[j], steps + 1
        want_dm[i] = steps
    rounds, first_done = 0, None
    while True:
        rounds += 1
        assert rounds <= int(np.ceil(np.log2(max(n, 2)))) + 2
        m_start = m.copy()
        open_ = False
        for i in rng.permutation(n):      # any order: a state sees its target's words as they are NOW (possibly already updated)
            j = jump[i]
            aj_jump, aj_dist = jump[j], dist[j]          # A before B
            mj, dmj = m[j], dm[j]
            if mj != m_start[i]:
                open_ = True
            if mj > m[i]:
                m[i], dm[i] = mj, (dist[i] + dmj) & 0xFFFF   # B first ...
            jump[i], dist[i] = aj_jump, (dist[i] + aj_dist) & 0xFFFF   # ... then A
        good = bool(np.array_equal(m, want_m) and np.array_equal(dm[want_m != DEAD], want_dm[want_m != DEAD]))
        if good and first_done is None:
            first_done = rounds
        if not open_:
            break
    assert np.array_equal(m, want_m)
    live = want_m != DEAD
    assert np.array_equal(dm[live], want_dm[live])
    assert first_done is not None and rounds <= first_done + 1      # the stop test fires one round after the tables are final


def test_closed_form_of_the_8_connected_line_iterator():
    for dx in range(0, 70):
        for dy in range(0, dx + 1):
            err, m = dx - 2 * dy, 0
            for i in range(dx + 1):
                num = 2 * dy * i - dx
                assert m == (0 if num <= 0 else (num + 2 * dx - 1) // (2 * dx)), (dx, dy, i)
                step = err < 0
                err += -2 * dy + (2 * dx if step else 0)
                m += step
