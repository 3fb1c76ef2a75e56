"""Times the combined depth pass (csrc/depth_ingest.hip: column maxima + obstacle scatter) and the streaming-only pass (column
maxima alone: the floor) on mid-episode frames of the rooms world, and prints a checksum of the obstacle planes.
    python tools/ingest_probe.py                     -> 256 x 480x640, 16 x 720x1280, 8 x 480x640 (one process each)
    python tools/ingest_probe.py <tag> [E H W]       -> one geometry"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def one(variant: int, E: int = 256, H: int = 480, W: int = 640, steps: int = 30) -> None:
    import numpy as np
    import torch

    from vlfm_amd import _lib
    from vlfm_amd.harness import RoomsRenderer
    from vlfm_amd.mapping import ObstacleMapBatch
    from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, camera_intrinsics

    dev = torch.device("cuda:0")
    fx, fy, fov = camera_intrinsics(W)
    rr = RoomsRenderer(list(range(E)), 500, H, W, dev)
    ob = ObstacleMapBatch(E, min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5, device=dev)
    frames = [rr.render(150 + t) for t in range(4)]
    for t in range(3):
        ob.ingest(frames[t], rr.tf_table[150 + t], MIN_DEPTH, MAX_DEPTH, fx, fy, want_colmax=True)
        ob.colmax_keys.zero_()
    torch.cuda.synchronize()
    _lib.lib().vlfm_profile_enable(1)
    for t in range(steps):
        k = t % 4
        ob.ingest(frames[k], rr.tf_table[150 + k], MIN_DEPTH, MAX_DEPTH, fx, fy, want_colmax=True)
        ob.colmax_keys.zero_()
    torch.cuda.synchronize()
    ms, n = _lib.profile_read("depth_ingest_scatter_kernel")
    for t in range(10):   # the same frames through the streaming-only form (column maxima, no obstacle plane): the floor
        ob.ingest(frames[t % 4], rr.tf_table[150 + t % 4], MIN_DEPTH, MAX_DEPTH, fx, fy, want_colmax=True, update_obstacles=False)
        ob.colmax_keys.zero_()
    torch.cuda.synchronize()
    ms0, n0 = _lib.profile_read("depth_ingest_kernel")
    print(f"streaming only (column maxima): {ms0 * 1e3:8.1f} us over {n0} launches = {E * H * W * 4 / (ms0 * 1e-3) / 1e12:.2f} TB/s",
          flush=True)
    bits = ob.obstacle_bits.cpu().numpy()
    print(f"variant {variant}: {ms * 1e3:8.1f} us over {n} launches ({E} x {H}x{W}: "
          f"{E * H * W * 4 / (ms * 1e-3) / 1e12:.2f} TB/s of depth), obstacle bits {int(np.unpackbits(bits.view(np.uint8)).sum())}, "
          f"checksum {int(bits.astype(np.int64).sum()) & 0xFFFFFFFF:08x}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(int(sys.argv[1]), *(int(a) for a in sys.argv[2:]))
    else:
        for shape in (["256", "480", "640"], ["16", "720", "1280"], ["8", "480", "640"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "0"] + shape, check=False)
