"""Can HF's GroundingDINO forward be captured in a HIP graph?  Yes (with the small host constants memoised and
transformers' per-call shape check off), but it gains only 2-9 %: the detector is bound by its f32 kernels (4.8-5.8 ms per
image), not by the host.  Kept as a probe; the product path stays eager."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from vlfm_amd.vlm import det_ops
from vlfm_amd.vlm.grounding_dino import GroundingDINO, preprocess_caption
class HostConstantCache:
    """HF's GroundingDINO forward builds a few small device tensors from Python lists on every call
    (``torch.tensor(SPECIAL_TOKENS, device=...)``, ``torch.as_tensor(spatial_shapes_list, device=...)``): host-to-device
    copies, which a HIP-graph capture does not allow.  Inside this context ``torch.tensor`` / ``torch.as_tensor`` of a
    Python list / tuple / number onto a GPU are memoised by (value, dtype, device): run the forward once eagerly inside the
    context (records the constants), then capture inside it (every constant is a cache hit, nothing crosses PCIe).  The
    values are compile-time constants of the model for a fixed input geometry, so sharing one tensor per value is safe as
    long as nobody writes into it -- HF's forward does not."""

    def __init__(self) -> None:
        self.cache = {}
        self._saved = None

    def _wrap(self, fn):
        def cached(data, *args, **kw):
            dev = kw.get("device", None)
            if isinstance(data, (list, tuple, int, float, bool)) and dev is not None and torch.device(dev).type == "cuda":
                key = (fn.__name__, repr(data), str(kw.get("dtype", None)), str(torch.device(dev)), repr(args))
                if key not in self.cache:
                    self.cache[key] = fn(data, *args, **kw)
                return self.cache[key]
            return fn(data, *args, **kw)
        return cached

    def __enter__(self):
        self._saved = (torch.tensor, torch.as_tensor)
        torch.tensor, torch.as_tensor = self._wrap(self._saved[0]), self._wrap(self._saved[1])
        return self

    def __exit__(self, *exc):
        torch.tensor, torch.as_tensor = self._saved
        return False


dev = torch.device("cuda:0")
g = GroundingDINO(device=dev, allow_random_init=True)
cap = "chair . bed . potted plant . toilet . tv . couch ."
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
img = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, device=dev)
ids = torch.tensor([g.tokenizer(preprocess_caption(cap))] * B, device=dev)
mask = torch.ones_like(ids)
tt = torch.zeros_like(ids)
pix = det_ops.to_tensor_normalize(img)
def fwd():
    return g.model(pixel_values=pix, input_ids=ids, attention_mask=mask, token_type_ids=tt)
with torch.inference_mode():
    for _ in range(3): out = fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): out = fwd()
    torch.cuda.synchronize(); print(f"eager B={B}: {(time.perf_counter()-t0)/4*1e3:.1f} ms")
    ref_logits, ref_boxes = out.logits.clone(), out.pred_boxes.clone()
    out2 = fwd()
    fin = torch.isfinite(ref_logits)
    print("eager vs eager: max|dbox|", float((out2.pred_boxes - ref_boxes).abs().max()), "max|dlogits| (finite)",
          float((out2.logits[fin] - ref_logits[fin]).abs().max()), "same inf pattern", bool((torch.isfinite(out2.logits) == fin).all()))
    try:
        consts = HostConstantCache()
        consts.__enter__()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): fwd()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gout = fwd()
        torch.cuda.synchronize()
        for _ in range(2): graph.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): graph.replay()
        torch.cuda.synchronize(); print(f"graph B={B}: {(time.perf_counter()-t0)/4*1e3:.1f} ms; max|dlogits| (finite) {float((gout.logits[fin]-ref_logits[fin]).abs().max()):.2e} max|dbox| {float((gout.pred_boxes-ref_boxes).abs().max()):.2e} same inf pattern {bool((torch.isfinite(gout.logits) == fin).all())}")
    except Exception:
        traceback.print_exc()
