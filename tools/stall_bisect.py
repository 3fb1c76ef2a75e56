"""Which part of the map step provokes the ~100 ms host stalls?  mode in {ingest, obst, obst_read, update, full}"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vlfm_amd.harness import BatchedEpisodes
from vlfm_amd.synthetic import MIN_DEPTH, MAX_DEPTH
E = int(sys.argv[1]); mode = sys.argv[2]; steps = 24
sim = BatchedEpisodes(E, device=torch.device("cuda:0"), use_blip2=False, overlap=False)
for _ in range(3): sim.step()
torch.cuda.synchronize()
ts = []
cos = torch.full((E, 1), 0.3, device="cuda:0", dtype=torch.float64)
for i in range(steps):
    t0 = time.perf_counter()
    k = sim.t % 4
    depth = sim.current_depth(sim.E); tf = sim.tf_table[sim.t % 500]
    if mode in ("ingest", "obst", "obst_read", "full", "nav") or mode.startswith("sub"):
        colmax = sim.obstacles.ingest(depth, tf, MIN_DEPTH, MAX_DEPTH, sim.fx, sim.fy, want_colmax=True)
    if mode in ("obst", "obst_read", "full"):
        sim.obstacles.update_after_ingest(tf, MAX_DEPTH, sim.fov)
    if mode == "nav":
        sim.obstacles.update_after_ingest(tf, MAX_DEPTH, sim.fov, explore=False)
    if mode.startswith("sub"):   # subN: only the first N envs of the E resident ones go through the explore pipeline
        n = int(mode[3:])
        sim.obstacles.update_after_ingest(tf[:n], MAX_DEPTH, sim.fov, env_ids=list(range(n)))
    if mode in ("obst_read", "full"):
        sim.obstacles.frontier_list()
    if mode in ("update",):
        colmax = sim.values.column_max(depth)
    if mode in ("update", "full"):
        sim.values.update(cos, None, tf, MIN_DEPTH, MAX_DEPTH, sim.fov, colmax=colmax)
    elif mode != "update":
        sim.obstacles.colmax_keys.zero_()
    torch.cuda.synchronize()
    sim.t += 1
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"E={E} mode={mode}: " + " ".join(f"{t:.1f}" for t in ts))
