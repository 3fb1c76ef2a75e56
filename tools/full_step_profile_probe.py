"""Host profile (cProfile, cumulative) of the configs[2] full step at E environments: where the step's wall time goes on the host side
(syncs show up as the time of the call that waits).   python tools/full_step_profile_probe.py [E] [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
import bench
from vlfm_amd.harness import BatchedEpisodes, ScriptedSightings
from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
from vlfm_amd.vlm.blip2itm import BLIP2ITM
from vlfm_amd.vlm.sam import MobileSAM
from vlfm_amd.vlm.yolov7 import YOLOv7
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
sim = BatchedEpisodes(E, device=dev, blip2=BLIP2ITM(device=dev, allow_random_init=True), detector=YOLOv7(device=dev, allow_random_init=True),
                      sam=MobileSAM(device=dev, allow_random_init=True), select_frontiers=True, object_maps=True,
                      sightings=ScriptedSightings(**bench.SIGHTING_SCRIPT), scripted_masks=True,
                      pointnav=WrappedPointNavResNetPolicy(None, device=dev, n_envs=E, discrete_actions=True))
sim.fast_forward(40)
sim.warm_up_segmenter()
sim.prepare(N + 4)
for _ in range(4): sim.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
prof = cProfile.Profile(); prof.enable()
for _ in range(N): sim.step()
torch.cuda.synchronize()
prof.disable()
print(f"E={E}: {(time.perf_counter() - t0) / N * 1e3:.1f} ms per step; object stats {sim.object_stats}")
pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
