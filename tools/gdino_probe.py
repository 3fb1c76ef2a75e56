"""GroundingDINO (HF geometry, random-init) predict_batch timing at batch B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)  # idle OpenMP spinners exhaust the container's CPU quota
from vlfm_amd.vlm.grounding_dino import GroundingDINO
dev = torch.device("cuda:0")
g = GroundingDINO(device=dev, allow_random_init=True)
cap = "chair . bed . potted plant . toilet . tv . couch ."
for B in (int(a) for a in (sys.argv[1:] or ["8"])):
    img = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, device=dev)
    for _ in range(2): g.predict_batch(img, [cap])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): g.predict_batch(img, [cap])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    print(f"B={B}: {dt*1e3:.1f} ms per batch, {dt/B*1e3:.2f} ms per image")
