"""Pick the library GEMM solutions of the BLIP-2 forward with PyTorch's TunableOp (hipBLASLt / rocBLAS candidates timed per shape)
and write them to vlfm_amd/tunableop_results.csv; vlfm_amd/vlm/ops.py:use_tuned_gemms() loads that file (no tuning at run time).
    python tools/tune_gemms.py 256 64        # batch sizes to tune for
The file is keyed by shape, dtype and the validators TunableOp records (ROCm, hipBLASLt, rocBLAS, arch): another image re-tunes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.cuda.tunable as tunable

out = os.path.join(ROOT, "gpurun_out", "tunableop_results.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(out)
tunable.set_max_tuning_duration(int(os.environ.get("VLFM_TUNE_MS", "12")))
tunable.set_max_tuning_iterations(20)
from vlfm_amd.vlm.blip2itm import BLIP2ITM

dev = torch.device("cuda:0")
m = BLIP2ITM(device=dev, allow_random_init=True)
g = torch.Generator(device="cpu").manual_seed(0)
for B in [int(a) for a in sys.argv[1:]] or [256]:
    rgb = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, generator=g).to(dev)
    t0 = time.perf_counter()
    with torch.inference_mode():
        for _ in range(2):
            m.cosine_batch(rgb, ["Seems like there is a bed ahead."])
    torch.cuda.synchronize()
    print(f"B={B}: tuned in {time.perf_counter() - t0:.1f} s, {len(tunable.get_results())} entries", flush=True)
# TunableOp writes its table to `out` when the process exits (this torch has no explicit write call): show what is known so far
for row in tunable.get_results():
    print(",".join(str(v) for v in row))
print(f"-> {out} at exit; copy it to vlfm_amd/tunableop_results.csv")
