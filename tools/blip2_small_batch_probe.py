"""BLIP-2 ITC forward (cosine_batch) at a small batch: ms per forward.  Under rocprofv3 --kernel-trace, tools/rocprof_tail.py on the
database gives the steady-state kernel table (profiles/r06_blip2_batch1_kernels.txt).
    python tools/blip2_small_batch_probe.py [envs] [forwards]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlfm_amd.vlm import ops
from vlfm_amd.vlm.blip2itm import BLIP2ITM
print("tuned library GEMMs:", bool(ops.use_tuned_gemms()))      # (vlfm_amd/tunableop_results.csv, as bench.py does)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
blip = BLIP2ITM(device=dev, allow_random_init=True)
rgb = torch.randint(0, 255, (E, 480, 640, 3), dtype=torch.uint8, device=dev)
with torch.inference_mode():
    for _ in range(5):
        c = blip.cosine_batch(rgb, ["Seems like there is a chair ahead."])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        c = blip.cosine_batch(rgb, ["Seems like there is a chair ahead."])
    torch.cuda.synchronize()
print(f"E={E}: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per BLIP-2 ITC forward; cos[0] = {float(c[0]):.5f}")
