"""The networks of the configs[2] full step one at a time at E envs (synchronised before and after: wall = host + GPU of that part).
python tools/full_step_parts_probe.py [E]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from vlfm_amd.harness import BatchedEpisodes
from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
from vlfm_amd.vlm.blip2itm import BLIP2ITM
from vlfm_amd.vlm.sam import MobileSAM
from vlfm_amd.vlm.yolov7 import YOLOv7
E = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
blip2 = BLIP2ITM(device=dev, allow_random_init=True)
det = YOLOv7(device=dev, allow_random_init=True)
sam = MobileSAM(device=dev, allow_random_init=True)
pn = WrappedPointNavResNetPolicy(None, device=dev, n_envs=E, discrete_actions=True)
import bench
from vlfm_amd.harness import ScriptedSightings
sim = BatchedEpisodes(E, device=dev, blip2=blip2, detector=det, sam=sam, select_frontiers=True, pointnav=pn, object_maps=True,
                      sightings=ScriptedSightings(**bench.SIGHTING_SCRIPT), scripted_masks=True)
sim.warm_up_segmenter()
sim.fast_forward(40)
for _ in range(3): sim.step()
rgb = sim.rgb_pool[0]
depth = sim.rooms.frame(50) if sim.rooms is not None else sim.depth_pool[0]
sel = list(range(0, E, 4))
box = torch.tensor([[[0.3 * sim.W, 0.3 * sim.H, 0.7 * sim.W, 0.8 * sim.H]]] * len(sel))
def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
parts = {
    "BLIP-2 cosine_batch": lambda: blip2.cosine_batch(rgb, sim.prompts),
    "detector predict_batch": lambda: det.predict_batch(rgb),
    f"MobileSAM segment_bboxes ({len(sel)} frames)": lambda: sam.segment_bboxes(rgb[sel], box),
    "PointNav act_on_depth": lambda: pn.act_on_depth(depth[..., None] if depth.dim() == 3 else depth, torch.rand(E, 2, device=dev), torch.ones(E, 1, dtype=torch.bool, device=dev)),
    "whole step": lambda: sim.step(),
}
for name, fn in parts.items():
    try:
        print(f"{name:45s} {timed(fn):7.2f} ms")
    except Exception as exc:  # noqa: BLE001
        print(f"{name:45s} failed: {type(exc).__name__}: {exc}")
