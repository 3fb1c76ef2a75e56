"""Q-Former query branch at batch E: exact-split K/V projection (f16 MFMA GEMMs) vs plain f32 GEMMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlfm_amd.vlm.blip2itm import BLIP2ITM
E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
m = BLIP2ITM(device=dev, allow_random_init=True).model
tok = torch.randn(E, 257, 1408, device=dev, dtype=torch.float16)
out = {}
for mode in (False, True, False, True):
    m.split_kv = mode
    with torch.inference_mode():
        for _ in range(2): y = m.query_features(tok)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): y = m.query_features(tok)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    out[mode] = y
    print(f"E={E} split_kv={mode}: {dt*1e3:.2f} ms per Q-Former query branch")
print("max |split - plain| =", float((out[True] - out[False]).abs().max()), " mean |y| =", float(out[False].abs().mean()))
