"""E in {1, 8}: eager vs graphed BLIP-2 in the harness."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlfm_amd.harness import BatchedEpisodes
from vlfm_amd.vlm.blip2itm import BLIP2ITM
dev = torch.device("cuda:0"); torch.set_num_threads(1)
blip = BLIP2ITM(device=dev, allow_random_init=True)
for E in (1, 8, 16):
    for graph in (False, True):
        sim = BatchedEpisodes(E, device=dev, blip2=blip, graph_blip2=graph)
        for _ in range(4): sim.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): sim.step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        a = sim.last_cosines.clone()
        print(f"E={E} graph={graph}: {dt*1e3:.2f} ms/step  {E/dt:.1f} env-steps/s  cos[0]={float(a[0]):.6f}")
        del sim
