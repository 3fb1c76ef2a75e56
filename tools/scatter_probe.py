"""Where does the scatter pass spend its time?  Same frames, height band moved so that nothing / everything is in band."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vlfm_amd import _lib
from vlfm_amd.mapping.obstacle_map import ObstacleMapBatch
from vlfm_amd.synthetic import MIN_DEPTH, MAX_DEPTH, camera_intrinsics, depth_frame, pose_to_tf
dev = torch.device("cuda:0"); E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
fx, fy, fov = camera_intrinsics(640)
rng = np.random.default_rng(0)
depth = torch.from_numpy(np.stack([depth_frame(rng) for _ in range(E)])).to(dev)
tf = np.stack([pose_to_tf(0.1 * e % 3, 0.0, 0.3 * e) for e in range(E)])
for name, lo, hi, thr in (("default band", 0.61, 0.88, 100000), ("nothing in band", 10.0, 11.0, 100000),
                          ("wide band -5..5", -5.0, 5.0, 100000), ("default, single pass (thresh -1)", 0.61, 0.88, -1)):
    om = ObstacleMapBatch(E, lo, hi, 0.18, 1.5, hole_area_thresh=thr, device=dev)
    for _ in range(3): om.ingest(depth, tf, MIN_DEPTH, MAX_DEPTH, fx, fy, want_colmax=True); om.colmax_keys.zero_()
    torch.cuda.synchronize()
    _lib.lib().vlfm_profile_enable(1)
    for _ in range(10): om.ingest(depth, tf, MIN_DEPTH, MAX_DEPTH, fx, fy, want_colmax=True); om.colmax_keys.zero_()
    torch.cuda.synchronize()
    r = {k: round(_lib.profile_read(k)[0] * 1e3, 1) for k in ("depth_ingest_kernel", "depth_scatter_kernel", "depth_ingest_scatter_kernel") if _lib.profile_read(k)[1]}
    _lib.lib().vlfm_profile_enable(0)
    print(f"{name:36s} {r}  set bits: {int(om._unpack(om.obstacle_bits).sum())}")
