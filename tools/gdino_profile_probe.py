"""GroundingDINO.predict_batch alone at B frames, steady state: wall ms per call, and (under rocprofv3 --kernel-trace) the window
of the last `reps` calls printed so that tools/rocprof_tail.py can cut the steady state out of the trace.
    python tools/gdino_profile_probe.py [B] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from vlfm_amd.vlm.grounding_dino import GroundingDINO
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
g = GroundingDINO(device=dev, allow_random_init=True)
cap = "chair . bed . potted plant . toilet . tv . couch ."
img = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, device=dev)
for _ in range(2):
    g.predict_batch(img, [cap])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    g.predict_batch(img, [cap])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"GroundingDINO.predict_batch B={B}: {dt / reps * 1e3:.1f} ms per call; steady-state window = last {dt * 1e3:.0f} ms")
with open(os.environ.get("GDINO_WINDOW_FILE", "/tmp/gdino_window_ms"), "w") as f:
    f.write(str(int(dt * 1e3)))
