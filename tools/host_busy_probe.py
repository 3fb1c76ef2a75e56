"""Where does a rank's host CPU go inside one batched step?  Wraps the step's host-visible calls with (wall, thread CPU)
timers and prints both per call, plus the busiest threads of the process over the timed steps.

    python tools/host_busy_probe.py [envs] [steps] [prof | profN]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from bench import thread_cpu_seconds  # noqa: E402
from vlfm_amd import _lib  # noqa: E402
from vlfm_amd.mapping import obstacle_map as om  # noqa: E402
from vlfm_amd.mapping import value_map as vm  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
device = torch.device("cuda:0")
torch.cuda.set_device(device)
_lib.host_wait_blocking(device)
torch.set_num_threads(1)
from vlfm_amd.harness import BatchedEpisodes  # noqa: E402

acc = {}


def timed(name, fn):
    def run(*a, **k):
        w0, c0 = time.perf_counter(), time.thread_time()
        try:
            return fn(*a, **k)
        finally:
            w, c = acc.get(name, (0.0, 0.0))
            acc[name] = (w + time.perf_counter() - w0, c + time.thread_time() - c0)
    return run


sim = BatchedEpisodes(E, device=device)
PROF = [int(a[4:] or 1) for a in sys.argv if a.startswith("prof")]
if PROF:
    _lib.lib().vlfm_profile_enable(PROF[0])   # prof = every launch, profN = every N-th launch of each kernel
sim.fast_forward(150)
sim.prepare(3 + STEPS)
for _ in range(3):
    sim.step()
torch.cuda.synchronize()
vm.wait_stream = timed("wait_stream (value_map)", vm.wait_stream)
om.wait_stream = timed("wait_stream (obstacle_map)", om.wait_stream)
vm.UploadRing.upload = timed("UploadRing.upload", vm.UploadRing.upload)
sim.blip2.cosine_batch = timed("blip2.cosine_batch (enqueue)", sim.blip2.cosine_batch)
sim.obstacles.ingest = timed("obstacles.ingest", sim.obstacles.ingest)
sim.obstacles.update_after_ingest = timed("obstacles.update_after_ingest", sim.obstacles.update_after_ingest)
sim.obstacles.frontier_list = timed("obstacles.frontier_list", sim.obstacles.frontier_list)
sim.values.update = timed("values.update", sim.values.update)
sim.values.waypoint_values = timed("values.waypoint_values", sim.values.waypoint_values)
t0, c0, th0 = time.perf_counter(), time.thread_time(), thread_cpu_seconds()
for _ in range(STEPS):
    sim.step()
torch.cuda.synchronize()
wall, cpu, th1 = time.perf_counter() - t0, time.thread_time() - c0, thread_cpu_seconds()
print(f"{E} envs, {STEPS} steps: {wall / STEPS * 1e3:.1f} ms/step, main thread {cpu / wall:.2f} cores")
for name, (w, c) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:34s} wall {w / STEPS * 1e3:8.2f} ms/step   thread CPU {c / STEPS * 1e3:8.2f} ms/step")
busy = sorted(((th1[t][1] - th0.get(t, (None, 0.0))[1]) / wall for t in th1), reverse=True)[:4]
print("  busiest threads:", ", ".join(f"{b:.2f}" for b in busy), "cores")
