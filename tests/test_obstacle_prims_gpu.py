"""-m gpu: bit-packed primitives of the obstacle-map kernels against the oracle's OpenCV restatement."""
import numpy as np
import pytest
import torch

from vlfm_amd import _lib

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pack(img_u8: torch.Tensor) -> torch.Tensor:
    planes, rows, cols = img_u8.shape
    out = torch.zeros((planes, rows, (cols + 31) // 32), dtype=torch.int32, device=img_u8.device)
    _lib.check(_lib.lib().vlfm_bits_pack(img_u8.data_ptr(), out.data_ptr(), planes, rows, cols, _stream()))
    return out


def _unpack(bits: torch.Tensor, cols: int) -> torch.Tensor:
    planes, rows, _ = bits.shape
    out = torch.empty((planes, rows, cols), dtype=torch.uint8, device=bits.device)
    _lib.check(_lib.lib().vlfm_bits_unpack(bits.data_ptr(), out.data_ptr(), planes, rows, cols, _stream()))
    return out


@pytest.mark.parametrize("shape", [(50, 70), (64, 64), (100, 1000), (33, 31)])
def test_pack_unpack_and_dilate(gpu_device, shape):
    from oracle import cv

    rng = np.random.default_rng(0)
    img = (rng.uniform(size=(3,) + shape) < 0.03).astype(np.uint8)
    d = torch.from_numpy(img).to(gpu_device)
    bits = _pack(d)
    assert torch.equal(_unpack(bits, shape[1]), d)
    for (kw, kh) in [(3, 3), (5, 5), (7, 7), (1, 3), (9, 1)]:
        out = torch.zeros_like(bits)
        _lib.check(_lib.lib().vlfm_bits_dilate(bits.data_ptr(), out.data_ptr(), 3, shape[0], shape[1], kw, kh, _stream()))
        got = _unpack(out, shape[1]).cpu().numpy()
        for p in range(3):
            assert np.array_equal(got[p], cv.dilate(img[p], np.ones((kh, kw), np.uint8)))


def _gpu_contours(img: np.ndarray, method: int, device):
    from oracle import cv  # noqa: F401

    planes, rows, cols = img.shape
    bits = _pack(torch.from_numpy(img).to(device))
    stride = (cols + 31) // 32
    scratch = torch.zeros((2, planes, rows, stride), dtype=torch.int32, device=device)
    cap_p, cap_c = 1 << 16, 4096
    pts = torch.zeros((planes, cap_p, 2), dtype=torch.int32, device=device)
    starts = torch.zeros((planes, cap_c), dtype=torch.int32, device=device)
    lens = torch.zeros((planes, cap_c), dtype=torch.int32, device=device)
    counts = torch.zeros((planes, 3), dtype=torch.int32, device=device)
    _lib.check(_lib.lib().vlfm_find_contours_external(bits.data_ptr(), planes, rows, cols, method, scratch.data_ptr(),
                                                     pts.data_ptr(), cap_p, starts.data_ptr(), lens.data_ptr(), cap_c,
                                                     counts.data_ptr(), _stream()))
    pts, starts, lens, counts = pts.cpu().numpy(), starts.cpu().numpy(), lens.cpu().numpy(), counts.cpu().numpy()
    out = []
    for p in range(planes):
        assert counts[p, 2] == 0
        cs = [pts[p, starts[p, k]:starts[p, k] + lens[p, k]] for k in range(counts[p, 0])]
        out.append(cs[::-1])  # OpenCV order = reverse discovery order
    return out


@pytest.mark.parametrize("density", [0.02, 0.3, 0.55, 0.8])
@pytest.mark.parametrize("method", [1, 2])
def test_find_contours_external_bit_exact(gpu_device, density, method):
    from oracle import cv

    rng = np.random.default_rng(int(density * 100) + method)
    img = (rng.uniform(size=(4, 60, 90)) < density).astype(np.uint8)
    img[1] = 0
    img[1, 10:40, 20:70] = 1
    img[1, 15:30, 30:50] = 0
    img[1, 20:25, 35:45] = 1          # component nested in a hole: not external
    img[2, 0, :] = 1                  # touches every image border
    img[2, :, 0] = 1
    img[2, -1, :] = 1
    img[2, :, -1] = 1
    got = _gpu_contours(img, method, gpu_device)
    for p in range(4):
        want, _ = cv.findContours(img[p], cv.RETR_EXTERNAL, method)
        assert len(got[p]) == len(want), (p, len(got[p]), len(want))
        for g, w in zip(got[p], want):
            assert np.array_equal(g, w.reshape(-1, 2))


def _walk_paths(reset: bool = True):
    """(borders from the ranked tables, planes with the per-pixel tables in global memory, borders walked by one lane although the
    ranked tables existed, planes with every table in LDS)."""
    import ctypes

    out = np.zeros(4, np.int64)
    _lib.check(_lib.lib().vlfm_walk_path_counters(ctypes.c_void_p(out.ctypes.data), int(reset)))
    return out


def _gpu_contours_wg(img: np.ndarray, method: int, device, cap_p: int = 1 << 14, cap_c: int = 4096, global_tables: bool = False):
    """The workgroup-parallel border follower (csrc/border_parallel.h) through vlfm_find_contours_external_wg: tables in LDS and
    every border ranked at once (default), or tables in global memory and one ranking per border (``global_tables``)."""
    method = method | (0x100 if global_tables else 0)
    planes, rows, cols = img.shape
    bits = _pack(torch.from_numpy(img).to(device))
    nbytes = _lib.lib().vlfm_find_contours_wg_scratch_bytes(planes, rows, cols, cap_p)
    assert nbytes > 0
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    pts = torch.zeros((planes, cap_p, 2), dtype=torch.int32, device=device)
    starts = torch.zeros((planes, cap_c), dtype=torch.int32, device=device)
    lens = torch.zeros((planes, cap_c), dtype=torch.int32, device=device)
    counts = torch.zeros((planes, 3), dtype=torch.int32, device=device)
    _lib.check(_lib.lib().vlfm_find_contours_external_wg(bits.data_ptr(), planes, rows, cols, method, scratch.data_ptr(), nbytes,
                                                        pts.data_ptr(), cap_p, starts.data_ptr(), lens.data_ptr(), cap_c,
                                                        counts.data_ptr(), _stream()))
    pts, starts, lens, counts = pts.cpu().numpy(), starts.cpu().numpy(), lens.cpu().numpy(), counts.cpu().numpy()
    out = []
    for p in range(planes):
        assert counts[p, 2] == 0
        cs = [pts[p, starts[p, k]:starts[p, k] + lens[p, k]] for k in range(counts[p, 0])]
        out.append(cs[::-1])
    return out


def _follower_states(img: np.ndarray):
    """(border pixels, states with an id) of the LDS follower for one plane: a state = (border pixel, direction of a border-pixel
    neighbour) from which the counter-clockwise search reaches a border pixel (csrc/border_parallel.h: v2_alloc_mask)."""
    from scipy import ndimage

    dx, dy = [1, 1, 0, -1, -1, -1, 0, 1], [0, -1, -1, -1, 0, 1, 1, 1]
    I = np.pad(img.astype(np.uint8), 2)
    B = (I == 1) & (ndimage.minimum_filter(I, 3) == 0)
    n = 0
    for y, x in zip(*np.nonzero(B)):
        nb = [I[y + dy[d], x + dx[d]] for d in range(8)]
        bn = [B[y + dy[d], x + dx[d]] for d in range(8)]
        for d in range(8):
            if bn[d]:
                s = next((d + k) & 7 for k in range(1, 9) if nb[(d + k) & 7])
                n += bool(bn[s])
    return int(B.sum()), n


def _lds_tables_fit(plane: np.ndarray, cap_p: int = 1 << 15):
    """Which form of the follower takes this plane (csrc/border_parallel.h: wg_build_rank_lds): "lds" = every table in the LDS behind
    the three padded window planes (158 KB in all), "gpix" = the per-pixel tables in global memory, None = the global-table form."""
    rows, cols = plane.shape
    plane_words = (rows + 2) * (((cols + 31) // 32 + 2) | 1)
    nb, n = _follower_states(plane)
    arena = 158 * 1024 - 12 * plane_words - (4 if plane_words & 1 else 0)
    rr = (2 * (rows + 1) + 7) & ~7
    pix = (4 * nb + 2 * (nb + 1) + 7) & ~7
    w32b = 8 * ((n + 31) // 32 + 1)

    def states_fit(left):
        return n < 65535 and (8 * n + w32b <= left or 4 * n + 4 * max(0, n - 2 * plane_words) + w32b <= left)

    if nb <= 65535 and rr + pix + 5 * nb <= arena and states_fit(arena - rr - pix):
        return "lds"
    if nb + 1 <= min(65535, 2 * cap_p) and rr <= arena and states_fit(arena - rr):
        return "gpix"
    return None


def _structured_planes():
    """Bitmaps that stress the state machine of the border follower: long ragged outlines with holes (the explored-area
    shape), one-pixel spurs and diagonals (states that are entered and left through the same neighbour), checkerboards
    (every pixel a border pixel with up to 4 diagonal neighbours), nested components, thick solids (interior successors)."""
    from scipy import ndimage

    rng = np.random.default_rng(11)
    H, W = 240, 330
    planes = []
    blobs = ndimage.binary_dilation(rng.uniform(size=(H, W)) < 0.004, iterations=7)
    holes = ndimage.binary_dilation(rng.uniform(size=(H, W)) < 0.002, iterations=3)
    planes.append(blobs & ~holes)
    cave = ndimage.binary_opening(rng.uniform(size=(H, W)) < 0.62, iterations=1)      # one sprawling component with many holes
    planes.append(cave)
    chk = (np.add.outer(np.arange(H), np.arange(W)) % 2 == 0)
    chk[:, 110:] = False
    chk[120:, :] = False
    chk[60:120, 200:300] = True                                                         # a solid block beside the checkerboard
    chk[80:100, 230:260] = False
    chk[88:92, 240:250] = True
    planes.append(chk)
    lines = np.zeros((H, W), bool)
    for k in range(0, 200, 7):
        lines[k, 5:300 - k] = True                                                      # one-pixel horizontal spurs
        lines[np.arange(5, 200), np.minimum(np.arange(5, 200) + k, W - 1)] = True       # diagonals crossing them
    lines[10:230, 310] = True
    planes.append(lines)
    solid = np.zeros((H, W), bool)
    solid[3:237, 3:327] = True                                                          # one big solid: long straight outline
    solid[100:140, 100:230] = False
    solid[110:130, 120:200] = True
    planes.append(solid)
    return np.stack(planes).astype(np.uint8)


@pytest.mark.parametrize("method", [1, 2])
def test_parallel_border_follower_equals_findcontours(gpu_device, method):
    """Workgroup-parallel Suzuki-Abe (successor tables + list ranking) against cv2.findContours(RETR_EXTERNAL) restated in
    oracle/cvport.c AND against the one-lane walk, on random noise of four densities and on structured bitmaps."""
    from oracle import cv

    for global_tables in (False, True):
        _walk_paths()
        n_borders = 0
        n_fit = {"lds": 0, "gpix": 0, None: 0}
        for density in (0.02, 0.3, 0.55, 0.8):
            rng = np.random.default_rng(int(density * 100) + method)
            img = (rng.uniform(size=(3, 60, 90)) < density).astype(np.uint8)
            got = _gpu_contours_wg(img, method, gpu_device, global_tables=global_tables)
            serial = _gpu_contours(img, method, gpu_device)
            for p in range(3):
                want, _ = cv.findContours(img[p], cv.RETR_EXTERNAL, method)
                assert len(got[p]) == len(want) == len(serial[p]), (density, p, len(got[p]), len(want))
                fit = _lds_tables_fit(img[p], 1 << 14)
                n_fit[fit] += 1
                n_borders += len(want) if fit else 0
                for g, w, q in zip(got[p], want, serial[p]):
                    assert np.array_equal(g, w.reshape(-1, 2)) and np.array_equal(g, q)
        full = _structured_planes()
        longest = 0
        # the full planes (most of them too intricate for the LDS tables: the global-table form or one lane) and three crops that fit
        for y0, y1, x0, x1 in ((0, 240, 0, 330), (0, 120, 0, 165), (60, 180, 100, 265), (120, 240, 165, 330)):
            img = np.ascontiguousarray(full[:, y0:y1, x0:x1])
            got = _gpu_contours_wg(img, method, gpu_device, cap_p=1 << 15, global_tables=global_tables)
            for p in range(img.shape[0]):
                want, _ = cv.findContours(img[p], cv.RETR_EXTERNAL, method)
                assert len(got[p]) == len(want), (global_tables, y0, x0, p, len(got[p]), len(want))
                fit = _lds_tables_fit(img[p])
                n_fit[fit] += 1
                n_borders += len(want) if fit else 0
                for k, (g, w) in enumerate(zip(got[p], want)):
                    assert np.array_equal(g, w.reshape(-1, 2)), (global_tables, y0, x0, p, k, len(g), len(w))
                    longest = max(longest, len(g))
        assert longest > (500 if method == 1 else 100)   # the ranking path ran (borders closing within 16 steps are walked by one lane)
        from_lds, planes_gpix, one_lane, planes_lds = _walk_paths()
        if global_tables:
            assert from_lds == 0 and planes_lds == 0 and planes_gpix == 0
        else:
            # every plane whose tables fit had them in LDS (all of them, or all but the per-pixel ones), and every border of those
            # came out of them: no start state was dead or missing
            assert n_fit["lds"] >= 20, n_fit
            assert planes_lds == n_fit["lds"] and planes_gpix == n_fit["gpix"] and one_lane == 0 and from_lds == n_borders, (
                from_lds, planes_gpix, one_lane, planes_lds, n_fit, n_borders)


def test_lds_border_tables_on_explored_area_shapes(gpu_device):
    """The shapes the map kernels trace -- a mid-episode explored area (one sprawling component, ragged outline, pillars as
    holes, specks around it) and obstacle blobs in a fog window -- at window sizes from the fog kernel's 205 x 205 to a
    20 m x 20 m world: LDS tables whenever they fit behind the window planes (predicted here from the state count), every border
    of such a plane from them, same chains as cv2.findContours either way."""
    from oracle import cv
    from scipy import ndimage

    rng = np.random.default_rng(23)
    fitted = 0
    forms = []
    # (rows, cols, smoothing of the outline, pillar / speck / obstacle-seed densities); the last one is a window whose three planes
    # leave 31 KB of LDS: per-pixel tables in global memory
    for rows, cols, sigma, pillars, specks, seeds in ((205, 205, 9, 4e-4, 5e-4, 2e-3), (300, 420, 14, 4e-4, 5e-4, 2e-3),
                                                    (402, 430, 18, 4e-4, 5e-4, 2e-3), (556, 454, 32, 1e-4, 1e-4, 4e-4)):
        planes = []
        field = ndimage.gaussian_filter(rng.uniform(size=(rows, cols)), sigma)
        area = field > np.quantile(field, 0.45)
        area &= ~ndimage.binary_dilation(rng.uniform(size=(rows, cols)) < pillars, iterations=3)
        area |= rng.uniform(size=(rows, cols)) < specks
        planes.append(area)
        planes.append(ndimage.binary_dilation(rng.uniform(size=(rows, cols)) < seeds, iterations=2))   # obstacle blobs
        img = np.stack(planes).astype(np.uint8)
        fits = [_lds_tables_fit(img[p]) for p in range(2)]
        fitted += sum(f is not None for f in fits)
        forms += fits
        for method in (1, 2):
            _walk_paths()
            got = _gpu_contours_wg(img, method, gpu_device, cap_p=1 << 15)
            n_lds = 0
            for p in range(2):
                want, _ = cv.findContours(img[p], cv.RETR_EXTERNAL, method)
                assert len(got[p]) == len(want), (rows, cols, method, p, len(got[p]), len(want))
                n_lds += len(want) if fits[p] else 0
                for g, w in zip(got[p], want):
                    assert np.array_equal(g, w.reshape(-1, 2))
            from_lds, planes_gpix, one_lane, planes_lds = _walk_paths()
            assert planes_lds == fits.count("lds") and planes_gpix == fits.count("gpix") and one_lane == 0 and from_lds == n_lds, (
                rows, cols, method, fits, from_lds, planes_gpix, one_lane, planes_lds, n_lds)
    assert fitted >= 7 and forms.count("gpix") >= 2, forms


def test_parallel_border_follower_falls_back_when_the_tables_do_not_fit(gpu_device):
    """More border pixels than the follower's tables hold (19 680 in 6 560 three-pixel dashes against a capacity of 16 384) but
    fewer chain points than the output holds: every border is walked by one lane instead, same result."""
    from oracle import cv

    img = np.zeros((1, 240, 330), np.uint8)
    for y in range(0, 160, 2):
        for x0 in range(0, 328, 4):
            img[0, y, x0:x0 + 3] = 1
    got = _gpu_contours_wg(img, 2, gpu_device, cap_p=1 << 14, cap_c=8192)[0]
    want, _ = cv.findContours(img[0], cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    assert len(got) == len(want) == 80 * 82
    for g, w in zip(got, want):
        assert np.array_equal(g, w.reshape(-1, 2))


def test_find_contours_map_sized(gpu_device):
    from oracle import cv
    from scipy import ndimage

    rng = np.random.default_rng(5)
    blobs = ndimage.binary_dilation(rng.uniform(size=(1000, 1000)) < 0.0005, iterations=9).astype(np.uint8)
    got = _gpu_contours(blobs[None], 2, gpu_device)[0]
    want, _ = cv.findContours(blobs, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    assert len(got) == len(want) and len(want) > 20
    for g, w in zip(got, want):
        assert np.array_equal(g, w.reshape(-1, 2))


def test_hoisted_reciprocal_division_equals_the_ieee_division(gpu_device):
    """csrc/depth_ingest.hip: (u - W/2) z / fx with y = RN(1/fx) hoisted, q0 = RN(a y), r = fma(-fx, q0, a), q = fma(r, y, q0)
    -- Markstein's division step -- must give the bits of the IEEE division for the numerators the scatter produces
    (integer pixel offset x an f32 depth widened to f64) and for arbitrary doubles, for several focal lengths."""
    from vlfm_amd import _lib

    rng = np.random.default_rng(12)
    n = 1 << 22
    z = rng.uniform(0.05, 12.0, n).astype(np.float32).astype(np.float64)
    u = rng.integers(-2048, 2048, n).astype(np.float64)
    sets = [u * z, rng.standard_normal(n) * 1e3, np.ldexp(rng.uniform(1, 2, n), rng.integers(-40, 40, n)),
            np.concatenate([[0.0, -0.0, 1.0, -1.0], np.nextafter(388.19, 1e9) * np.arange(1, n - 3)])]
    L = _lib.lib()
    for fx in (388.1926244788403, 320.0, 776.3852489576806, 1.0 / 3.0, 554.2562584220407, 0.1):
        for k, a in enumerate(sets):
            d_a = torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
            bad = torch.zeros(1, dtype=torch.int32, device=gpu_device)
            _lib.check(L.vlfm_selftest_div_exact(d_a.data_ptr(), d_a.numel(), float(fx), bad.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), "selftest_div_exact")
            assert int(bad.item()) == 0, (fx, k, int(bad.item()))


@pytest.mark.gpu
def test_lds_follower_on_tall_and_degenerate_planes(gpu_device):
    """Planes taller than the workgroup (the start search walks the rows in blocks of 1024 lanes, the row-rank table has more than
    1024 entries), a plane of isolated pixels only (no state at all), an empty plane and a full one: same chains as
    cv2.findContours, both methods."""
    from oracle import cv

    rng = np.random.default_rng(31)
    tall = np.zeros((2, 1500, 40), np.uint8)
    tall[0] = rng.uniform(size=(1500, 40)) < 0.45
    tall[1, 5:1495, 3:37] = 1
    tall[1, 700:720, 10:30] = 0
    tall[1, 1100:1103, 0:40] = 0          # splits the solid in two components, the second one starting below row 1024
    specks = np.zeros((3, 64, 96), np.uint8)
    specks[0, ::3, ::5] = 1               # isolated pixels only
    specks[2] = 1                         # full plane
    for img in (tall, specks):
        for method in (1, 2):
            _walk_paths()
            got = _gpu_contours_wg(img, method, gpu_device, cap_p=1 << 15, cap_c=8192)
            for p in range(img.shape[0]):
                want, _ = cv.findContours(img[p], cv.RETR_EXTERNAL, method)
                assert len(got[p]) == len(want), (img.shape, method, p, len(got[p]), len(want))
                for g, w in zip(got[p], want):
                    assert np.array_equal(g, w.reshape(-1, 2))
            assert _walk_paths()[2] == 0      # no border needed the one-lane walk
