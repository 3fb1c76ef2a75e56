"""ViT-g forward at batch E: plain block path vs the deferred-bias fast path (ms per forward, output difference)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlfm_amd.vlm.blip2itm import BLIP2ITM
E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
blip = BLIP2ITM(device=dev, allow_random_init=True)
m = blip.model
with torch.no_grad():
    for n, p in m.named_parameters():
        if p.dim() == 1 and "norm" not in n.lower():
            p.copy_(torch.randn_like(p) * 0.1)
for blk in m.blocks:
    blk.pack_heads()   # the packed copies (library-attention path) follow the edited biases
m.weights_changed()
pat = torch.randn(E, 256, 588, device=dev, dtype=torch.float16)
outs = {}
for mode in (False, True, False, True):
    m.deferred_bias = mode
    with torch.inference_mode():
        for _ in range(2): y = m.vision_tokens(pat)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): y = m.vision_tokens(pat)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    outs[mode] = y.float()
    print(f"E={E} deferred_bias={mode}: {dt*1e3:.2f} ms per ViT forward")
d = (outs[True] - outs[False]).abs()
print("max |fast - plain| =", float(d.max()), " mean =", float(d.mean()), " mean |y| =", float(outs[False].abs().mean()))
