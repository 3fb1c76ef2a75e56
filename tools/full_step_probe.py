"""configs[2] full step (BLIP-2 + detector + MobileSAM + maps + frontier selection + PointNav) at E envs: wall time per step and,
under `rocprofv3 --kernel-trace --stats`, where the GPU time goes.  python tools/full_step_probe.py [E] [yolo|gdino]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from vlfm_amd.harness import BatchedEpisodes
from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
from vlfm_amd.vlm.blip2itm import BLIP2ITM
from vlfm_amd.vlm.sam import MobileSAM
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "yolo"
dev = torch.device("cuda:0")
blip2 = BLIP2ITM(device=dev, allow_random_init=True)
if which == "gdino":
    from vlfm_amd.vlm.grounding_dino import GroundingDINO
    det = GroundingDINO(device=dev, allow_random_init=True)
else:
    from vlfm_amd.vlm.yolov7 import YOLOv7
    det = YOLOv7(device=dev, allow_random_init=True)
sim = BatchedEpisodes(E, device=dev, blip2=blip2, detector=det, sam=MobileSAM(device=dev, allow_random_init=True),
                      select_frontiers=True,
                      pointnav=WrappedPointNavResNetPolicy(None, device=dev, n_envs=E, discrete_actions=True))
sim.fast_forward(40)
for _ in range(3): sim.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 8
for _ in range(N): sim.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
print(f"E={E} {which}: {dt * 1e3:.1f} ms per step = {E / dt:.0f} env-steps/s")
