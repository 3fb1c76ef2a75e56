"""Per-step wall times of the map-only harness (diagnostic for host-side stalls)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlfm_amd.harness import BatchedEpisodes
E = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sim = BatchedEpisodes(E, device=torch.device("cuda:0"), use_blip2=False, overlap=(os.environ.get("NO_OVERLAP") is None))
for _ in range(3): sim.step()
torch.cuda.synchronize()
ts = []
for _ in range(steps):
    t0 = time.perf_counter(); sim.step(); ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print(f"E={E} env={ {k: os.environ[k] for k in ('HSA_ENABLE_INTERRUPT','HSA_ENABLE_SDMA','NO_OVERLAP') if k in os.environ} }")
print("  ms/step:", " ".join(f"{t:.1f}" for t in ts))
