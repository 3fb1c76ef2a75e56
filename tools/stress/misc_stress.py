"""More seeds / sizes for the DBSCAN, erosion + cloud extraction and NMS parity tests (bodies of the committed tests)."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util, numpy as np, torch
def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", name + ".py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
om, dt = load("test_object_map_gpu"), load("test_detect_gpu")
dev = torch.device("cuda:0")
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
rng = np.random.default_rng(a)
for seed in range(a, b):
    n = int(rng.choice([1, 2, 63, 64, 65, 99, 100, 101, 257, 700, 1500, 3000, 4096]))
    for what, fn in (("dbscan", lambda: om.test_dbscan_matches_sequential_oracle(dev, n, seed)),
                     ("nms", lambda: dt.test_nms_matches_oracle(dev, max(1, n // 2)))):
        try:
            if what == "nms":
                torch.manual_seed(seed); np.random.seed(seed)
            fn()
        except AssertionError as e:
            bad += 1
            tb = traceback.extract_tb(e.__traceback__)[-1]
            print(what, "seed", seed, "n", n, "FAILED line", tb.lineno, str(e)[:200].replace("\n", " "))
        except Exception as e:
            bad += 1
            print(what, "seed", seed, "n", n, "RAISED", type(e).__name__, str(e)[:200])
print(f"dbscan + nms seeds {a}..{b - 1}: {bad} failed")
# erosion + cloud order on random masks
from oracle.ref_object_map import extract_object_cloud
from vlfm_amd.mapping.object_point_cloud_map import ObjectPointCloudMap
from vlfm_amd.synthetic import depth_frame, camera_intrinsics, MIN_DEPTH, MAX_DEPTH
fx, fy, _ = camera_intrinsics(640)
bad2 = 0
for seed in range(a, b):
    r = np.random.default_rng(10_000 + seed)
    depth = depth_frame(r, 480, 640, holes=bool(seed % 2))
    m = np.zeros((480, 640), np.uint8)
    for _ in range(int(r.integers(1, 6))):
        y0, x0 = int(r.integers(0, 470)), int(r.integers(0, 630)); h, w = int(r.integers(1, 200)), int(r.integers(1, 260))
        m[y0:y0 + h, x0:x0 + w] = 1
    if seed % 3 == 0:
        m &= (r.uniform(size=m.shape) < 0.97).astype(np.uint8)       # pinholes: erosion eats around them
    it = int(r.integers(0, 7))
    for use_db in (False, True):
        o = ObjectPointCloudMap(erosion_size=it, device=dev); o.use_dbscan = use_db
        np.random.seed(seed); got = o._extract_object_cloud(depth, m, MIN_DEPTH, MAX_DEPTH, fx, fy)
        np.random.seed(seed); want = extract_object_cloud(depth, m, it, MIN_DEPTH, MAX_DEPTH, fx, fy, use_dbscan=use_db)
        if not (got.shape == want.shape and np.array_equal(got, want)):
            bad2 += 1; print("cloud seed", seed, "erosion", it, "dbscan", use_db, got.shape, want.shape)
print(f"erosion + cloud extraction seeds {a}..{b - 1}: {bad2} failed")
