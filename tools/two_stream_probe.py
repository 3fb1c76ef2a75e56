"""ViT-g + Q-Former at batch E: one stream vs K streams of E/K images each (compute-bound GEMMs of one part overlapping
the memory-bound LayerNorm / GELU / attention kernels of another?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlfm_amd.vlm.blip2itm import BLIP2ITM
E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
m = BLIP2ITM(device=dev, allow_random_init=True).model
pat = torch.randn(E, 256, 588, device=dev, dtype=torch.float16)
def fwd(x):
    return m.query_features(m.vision_tokens(x))
def run(K, n=4, sizes=None):
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    parts = list(pat.chunk(K)) if sizes is None else list(pat.split(sizes))
    def once():
        cur = torch.cuda.current_stream(dev)
        outs = []
        for s, p in zip(streams, parts):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(fwd(p))
        for s in streams:
            cur.wait_stream(s)
        return outs
    with torch.inference_mode():
        for _ in range(2): once()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): once()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
with torch.inference_mode():
    for _ in range(2): fwd(pat)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): fwd(pat)
    torch.cuda.synchronize(); base = (time.perf_counter() - t0) / 4 * 1e3
print(f"E={E}: one stream {base:.2f} ms")
for K in (2, 4):
    print(f"E={E}: {K} streams x {E // K} images: {run(K):.2f} ms")
# NOTE: three concurrent forwards (e.g. sizes [48, 40, 40]) hung this probe on the GPU box (stream-K GEMMs of three
# streams waiting for workgroups that cannot become resident?) -- do not extend the list below without a short timeout.
