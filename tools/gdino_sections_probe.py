"""Where the GroundingDINO forward spends GPU time, by module class (events around every forward of the listed classes; nested
classes are reported separately, parents include their children), eager, B frames.
    python tools/gdino_sections_probe.py [B] [fast 0|1] [precision]"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from vlfm_amd.vlm.grounding_dino import GroundingDINO
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fast = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
prec = sys.argv[3] if len(sys.argv) > 3 else "split"
dev = torch.device("cuda:0")
g = GroundingDINO(device=dev, allow_random_init=True, fast=fast, gemm_precision=prec, graph=False)
WATCH = ("SwinLayer", "SwinPatchMerging", "SwinEmbeddings", "GroundingDinoConvEncoder", "GroundingDinoFusionLayer", "GroundingDinoTextEnhancerLayer",
         "GroundingDinoDeformableLayer", "GroundingDinoMultiscaleDeformableAttention", "GroundingDinoDecoderLayer", "GroundingDinoEncoder",
         "GroundingDinoDecoder", "GroundingDinoModel", "MultiScaleDeformableAttention", "BertModel")
spans = collections.defaultdict(list)
def hook(mod, name):
    plain = mod.forward
    def fwd(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = plain(*a, **k); e.record(); spans[name].append((s, e)); return out
    mod.forward = fwd
for m in g.model.modules():
    if type(m).__name__ in WATCH:
        hook(m, type(m).__name__)
cap = "chair . bed . potted plant . toilet . tv . couch ."
img = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, device=dev)
for _ in range(2): g.predict_batch(img, [cap])
spans.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n): g.predict_batch(img, [cap])
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n * 1e3
print(f"B={B} fast={fast} precision={prec}: {wall:.1f} ms wall per predict_batch (eager)")
for name, ev in sorted(spans.items(), key=lambda kv: -sum(s.elapsed_time(e) for s, e in kv[1])):
    tot = sum(s.elapsed_time(e) for s, e in ev) / n
    print(f"  {name:46s} {tot:8.2f} ms  ({len(ev) // n} calls)")
