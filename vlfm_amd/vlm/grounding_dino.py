"""vlfm.vlm.grounding_dino, MI355X in-process (reference: /root/reference/vlfm/vlm/grounding_dino.py:22-85).

``GroundingDINO.predict(image, caption)`` keeps the reference's pipeline: to_tensor + ImageNet normalise at NATIVE resolution
(no resize, :52-54) -> model -> ``predict()`` post-processing of the un-vendored groundingdino package [ext] (caption
lower-cased and forced to end with "."; per query the sigmoid maximum over text tokens must exceed ``box_threshold``;
phrase = the tokens whose probability exceeds ``text_threshold``) -> ``ObjectDetections`` (cxcywh -> xyxy) -> keep only
phrases that exactly match a class of the caption (:70-72).

Network: HF ``GroundingDinoForObjectDetection`` at the Swin-T / BERT-base geometry of ``groundingdino_swint_ogc`` (172 M
parameters) on PyTorch-ROCm; its deformable attention runs through the pure-PyTorch sampling path (the reference's CUDA
extension has no ROCm build).  Pretrained weights are not available offline: ``model_dir`` (or
``GROUNDING_DINO_MODEL_DIR``) loads an HF checkpoint + tokenizer, otherwise weights are random-init and a deterministic
word-level tokenizer stands in (special ids 101/102/1012 kept so that the model's phrase-block attention masks form)."""
from __future__ import annotations

import os
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import det_ops
from .detections import ObjectDetections

CLASSES = "chair . person . dog ."  # grounding_dino.py:19


class WordTokenizer:
    """Stand-in for BertTokenizer when no vocabulary is on disk: words hash into the vocabulary, "." -> 1012,
    [CLS]/[SEP] = 101/102; ``decode`` maps ids back to the words seen so far."""

    def __init__(self, vocab_size: int = 30522, max_len: int = 256) -> None:
        self.vocab_size, self.max_len = vocab_size, max_len
        self._words: Dict[int, str] = {101: "[CLS]", 102: "[SEP]", 1012: "."}

    def _id(self, w: str) -> int:
        if w == ".":
            return 1012
        h = 2166136261
        for ch in w.encode():
            h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
        i = 2000 + h % (self.vocab_size - 2000)
        self._words[i] = w
        return i

    def __call__(self, text: str) -> List[int]:
        toks = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", text.lower())
        return ([101] + [self._id(t) for t in toks][: self.max_len - 2] + [102])

    def decode(self, ids: Sequence[int]) -> str:
        return " ".join(self._words.get(int(i), "[UNK]") for i in ids)


def preprocess_caption(caption: str) -> str:
    """groundingdino.util.inference.preprocess_caption [ext]."""
    result = caption.lower().strip()
    return result if result.endswith(".") else result + "."


def _bert_tokenizer(tokenizer_dir: Optional[str]):
    """The groundingdino package builds ``bert-base-uncased``'s tokenizer from the hub [ext]; offline it has to be on disk:
    ``tokenizer_dir`` / ``GROUNDING_DINO_TOKENIZER`` (a directory holding ``vocab.txt``, or the file), else the local
    transformers cache.  There is no stand-in vocabulary for real weights."""
    from transformers import BertTokenizer

    path = tokenizer_dir or os.environ.get("GROUNDING_DINO_TOKENIZER")
    if path:
        vocab = path if os.path.isfile(path) else os.path.join(path, "vocab.txt")
        if not os.path.isfile(vocab):
            raise FileNotFoundError(f"no vocab.txt under {path!r}")
        return BertTokenizer(vocab_file=vocab, do_lower_case=True)
    msg = ("GroundingDINO weights need the bert-base-uncased vocabulary: pass tokenizer_dir (or set GROUNDING_DINO_TOKENIZER) "
           "to a directory with its vocab.txt")
    try:
        tok = BertTokenizer.from_pretrained("bert-base-uncased", local_files_only=True)
    except Exception as exc:  # noqa: BLE001
        raise ValueError(msg) from exc
    if len(tok) < 30522:   # transformers can hand back an EMPTY tokenizer (special tokens only) when nothing is cached
        raise ValueError(msg)
    return tok


class GroundingDINO:
    """grounding_dino.py:22-74.  The reference hands ``config_path`` / ``weights_path`` (GroundingDINO_SwinT_OGC.py,
    groundingdino_swint_ogc.pth) to the un-vendored groundingdino package.  Here the network is transformers'
    ``GroundingDinoForObjectDetection`` and BOTH forms of the weights load:

    * the reference's own files: ``weights_path`` = the ``.pth``, ``config_path`` = the ``.py`` hyper-parameter file (parsed,
      never executed; omitted = the Swin-T OGC geometry).  The state dict goes through ``gdino_weights``' key map, strictly
      (every tensor of the file placed, every parameter of the graph fed, shapes equal);
    * ``model_dir`` (or ``GROUNDING_DINO_MODEL_DIR``; ``weights_path`` may also name such a directory): a checkpoint in
      transformers' format, e.g. ``IDEA-Research/grounding-dino-tiny`` = the converted groundingdino_swint_ogc.

    A path that is given must load; random weights need ``allow_random_init=True``."""

    def __init__(self, config_path: Optional[str] = None, weights_path: Optional[str] = None, caption: str = CLASSES,
                 box_threshold: float = 0.35, text_threshold: float = 0.25, device=None, model_dir: Optional[str] = None,
                 hf_config=None, seed: int = 0, allow_random_init: bool = False, tokenizer_dir: Optional[str] = None,
                 fast: Optional[bool] = None, gemm_precision: Optional[str] = None, graph: Optional[bool] = None) -> None:
        from transformers import GroundingDinoConfig, GroundingDinoForObjectDetection

        from ..mapping.base_map import require_gpu

        self.device = require_gpu(device)
        self.caption, self.box_threshold, self.text_threshold = caption, box_threshold, text_threshold
        model_dir = model_dir or os.environ.get("GROUNDING_DINO_MODEL_DIR")
        original = None
        if not model_dir and weights_path:
            if os.path.isdir(weights_path):
                model_dir = weights_path
            elif not os.path.exists(weights_path):
                raise FileNotFoundError(f"GroundingDINO weights {weights_path!r} not found")
            else:
                original = weights_path
        if original:
            from . import gdino_weights

            if config_path and not os.path.isfile(config_path):
                raise FileNotFoundError(f"GroundingDINO config {config_path!r} not found")
            cfg = hf_config or (gdino_weights.config_from_groundingdino_py(config_path) if config_path
                                else GroundingDinoConfig())
            ckpt = torch.load(original, map_location="cpu", weights_only=True)
            sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
            self.model = GroundingDinoForObjectDetection(cfg)
            feed = gdino_weights.convert_groundingdino_state_dict(sd, self.model.state_dict())
            res = self.model.load_state_dict(feed, strict=False)
            if res.unexpected_keys or not set(res.missing_keys) <= set(gdino_weights._NOT_FED):   # (not an assert: python -O)
                raise KeyError(f"GroundingDINO checkpoint {original!r} does not fill the graph: unexpected "
                               f"{list(res.unexpected_keys)[:4]}, missing {list(res.missing_keys)[:4]}")
            tok = _bert_tokenizer(tokenizer_dir)
            self.tokenizer = lambda t: tok(t)["input_ids"]
            self.decode = tok.decode
            self.weights = f"groundingdino checkpoint:{original}"
        elif model_dir:
            from transformers import AutoTokenizer

            self.model = GroundingDinoForObjectDetection.from_pretrained(model_dir)
            tok = AutoTokenizer.from_pretrained(model_dir)
            self.tokenizer = lambda t: tok(t)["input_ids"]
            self.decode = tok.decode
            self.weights = f"pretrained:{model_dir}"
        elif allow_random_init or hf_config is not None:
            torch.manual_seed(seed)
            cfg = hf_config or GroundingDinoConfig()
            self.model = GroundingDinoForObjectDetection(cfg)
            wt = WordTokenizer(cfg.text_config.vocab_size, cfg.max_text_len)
            self.tokenizer, self.decode = wt, wt.decode
            self.weights = "random-init"
        else:
            raise ValueError("GroundingDINO needs weights_path (groundingdino_swint_ogc.pth) or model_dir / "
                             "GROUNDING_DINO_MODEL_DIR (a transformers-format checkpoint); pass allow_random_init=True for a "
                             "randomly initialised network (benchmarks only)")
        self.description = f"GroundingDINO (HF Swin-T + BERT-base geometry, 172 M parameters) {self.weights}"
        self.model.eval().to(self.device)
        # transformers validates the spatial shapes of EVERY deformable-attention call with a device read-back (`.item()`):
        # 12 host syncs per forward, and illegal inside a HIP-graph capture.  The shapes come from this wrapper and are
        # consistent by construction; transformers' own switch turns the check off.
        os.environ.setdefault("TRANSFORMERS_DISABLE_TORCH_CHECK", "1")
        self.hip_deform_attn = det_ops.patch_hf_deformable_attention(self.model)  # HIP MsDeformAttn
        self.gemm_convs = det_ops.patch_convs_as_gemm(self.model)   # patch-embed / 1x1 convolutions as GEMMs
        det_ops.cache_text_branch(self.model)                       # the caption is constant over an episode
        # fused forwards on csrc/gemm_f32.hip + csrc/sam_ops.hip (vlm/gdino_fast.py).  VLFM_GDINO_FAST=0 keeps transformers' own
        # forward; gemm_precision "split" (default) = f32-grade GEMMs from f16 operand pairs with the exact library GEMMs as the
        # automatic fallback when an operand leaves f16's range, "library" = hipBLASLt f32 throughout; graph = replay the forward of
        # a recurring input signature from a HIP graph
        from . import gdino_fast

        self.fast = (os.environ.get("VLFM_GDINO_FAST", "1") != "0") if fast is None else bool(fast)
        self.gemm_precision = gemm_precision or os.environ.get("VLFM_GDINO_GEMM", "split")
        self.fast_summary = gdino_fast.accelerate(self.model, self.gemm_precision) if self.fast else None
        self.use_graph = self.fast and ((os.environ.get("VLFM_GDINO_GRAPH", "1") != "0") if graph is None else bool(graph))
        self._graphed = gdino_fast.GraphedForward(self.model) if self.use_graph else None
        self.overflow_fallbacks = 0
        self._phrase_cache: Dict = {}
        if self.fast:
            self.description += (f"; fused MI355X forward (vlm/gdino_fast.py), f32 GEMMs: {self.gemm_precision}"
                                 + (", HIP-graph replay" if self.use_graph else ""))

    @torch.inference_mode()
    def predict_batch(self, images_u8: torch.Tensor, captions: Sequence[str]) -> List[ObjectDetections]:
        """images_u8 [B,H,W,3] u8 RGB on device; one caption per image (or a single shared caption)."""
        B = images_u8.shape[0]
        raw = list(captions) if len(captions) == B else list(captions) * B
        caps = [preprocess_caption(c) for c in raw]
        ids = [self.tokenizer(c) for c in caps]
        L = max(len(i) for i in ids)
        input_ids = torch.zeros((B, L), dtype=torch.long)
        mask = torch.zeros((B, L), dtype=torch.long)
        for b, i in enumerate(ids):
            input_ids[b, : len(i)] = torch.tensor(i)
            mask[b, : len(i)] = 1
        pix = det_ops.to_tensor_normalize(images_u8)
        self.model.vlfm_text_key = tuple(caps)   # the BERT branch is memoised per caption batch (det_ops.cache_text_branch)
        logits, pred_boxes = self._forward(pix, input_ids.to(self.device), mask.to(self.device), tuple(caps))
        best, boxes, bits = self._postprocess_device(logits, pred_boxes, ids)
        if self.fast and self.model.vlfm_fast.precision == "split":
            from . import ops

            flag = ops.gemm_f32_overflow_flag(self.device, "gdino")
            if int(flag.item()):     # (the copies above already synchronised) an operand left f16's range: the split results are void
                flag.zero_()
                self.overflow_fallbacks += 1
                self.model.vlfm_fast.precision = "library"
                try:
                    logits, pred_boxes = self._forward(pix, input_ids.to(self.device), mask.to(self.device), tuple(caps))
                    best, boxes, bits = self._postprocess_device(logits, pred_boxes, ids)
                finally:
                    # a WEIGHT outside f16's range poisons its cached planes for good: stay on the library from then on
                    self.model.vlfm_fast.precision = "library" if ops.split_weights_bad(self.device, "gdino") else "split"
        return [self._detections(best[b], boxes[b], bits[b], ids[b], raw[b]) for b in range(B)]

    def _postprocess_device(self, logits: torch.Tensor, pred_boxes: torch.Tensor, ids):
        """The reductions of groundingdino.util.inference.predict [ext] where the logits are: per query the best token probability
        and the SET of caption tokens above ``text_threshold`` as a bit pattern (bit k = token k; [CLS], [SEP] and everything behind
        excluded, get_phrases_from_posmap [ext]).  1.3 MB cross to the host for 64 frames instead of the 59 MB probability tensor."""
        B = logits.shape[0]
        L = max(len(i) for i in ids)
        if L > 62:
            raise NotImplementedError("captions beyond 62 tokens: the token-set bit pattern no longer fits an int64")
        probs = logits.sigmoid().float()                             # [B, nq, max_text_len]
        best = probs.amax(dim=2)
        pos = probs[:, :, :L] > self.text_threshold
        valid = torch.zeros((B, L), dtype=torch.bool)
        for b, i in enumerate(ids):
            valid[b, 1:len(i) - 1] = True
        pos &= valid.to(self.device)[:, None, :]
        bits = (pos.to(torch.int64) << torch.arange(L, device=self.device)).sum(dim=2)
        return best.cpu().numpy(), pred_boxes.float().cpu().numpy(), bits.cpu().numpy()

    def _forward(self, pix: torch.Tensor, input_ids: torch.Tensor, mask: torch.Tensor, key):
        tt = torch.zeros_like(input_ids)
        if self._graphed is not None:
            try:
                return self._graphed(key, pix, input_ids, mask, tt)
            except Exception as exc:  # noqa: BLE001 -- a capture that the runtime refuses must not take the detector down
                import warnings

                warnings.warn(f"GroundingDINO: HIP-graph capture failed ({type(exc).__name__}: {exc}); running eagerly from now on")
                self._graphed, self.use_graph = None, False
                torch.cuda.synchronize(self.device)
        out = self.model(pixel_values=pix, input_ids=input_ids, attention_mask=mask, token_type_ids=tt)
        return out.logits, out.pred_boxes

    def _detections(self, best: np.ndarray, boxes: np.ndarray, bits: np.ndarray, ids: Sequence[int], raw_caption: str) -> ObjectDetections:
        """groundingdino.util.inference.predict's post-processing [ext] + grounding_dino.py:70-72 for one image: queries whose best
        token probability exceeds box_threshold; each one's phrase = the caption tokens above text_threshold (``bits``, from
        ``_postprocess_device``).  One decode per DISTINCT token set and caption, memoised (queries share a handful of phrases, and an
        episode keeps its caption)."""
        keep = best > self.box_threshold
        box, conf, pat = boxes[keep], best[keep], bits[keep]
        classes = raw_caption[: -len(" .")].split(" . ")   # grounding_dino.py:70-72, literally (the caller's caption ends with " ." -- App. C9)
        key = tuple(int(i) for i in ids)
        table = self._phrase_cache.setdefault(key, {})
        if len(table) > 65536:
            table.clear()
        idv = np.asarray(ids)
        uniq, inv = (np.unique(pat, return_inverse=True) if len(pat) else (pat, np.zeros(0, np.int64)))
        phr_u = []
        for p_ in uniq.tolist():
            ph = table.get(p_)
            if ph is None:
                toks = idv[[k for k in range(len(ids)) if (p_ >> k) & 1]].tolist()
                ph = table[p_] = self.decode(toks).replace(".", "").strip()
            phr_u.append(ph)
        # filter_by_class (detections.py:73-80) on the distinct phrases: with hundreds of queries above the box threshold (an untrained
        # network: all 900) the per-query Python loop was the cost of the call
        ok_u = np.array([ph in classes for ph in phr_u], bool) if len(phr_u) else np.zeros(0, bool)
        sel = ok_u[inv.reshape(-1)] if len(pat) else np.zeros(0, bool)
        phrases = [phr_u[i] for i in inv.reshape(-1)[sel].tolist()]
        det = ObjectDetections(torch.from_numpy(np.ascontiguousarray(box[sel])),
                               torch.from_numpy(np.ascontiguousarray(conf[sel])) if sel.any() else torch.zeros(0), phrases,
                               image_source=None)
        det.filter_by_class(classes)       # (every phrase left is a class: kept for the reference's call sequence)
        return det

    def predict(self, image: np.ndarray, caption: Optional[str] = None) -> ObjectDetections:
        caption_to_use = self.caption if caption is None else caption
        img = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)[None]
        det = self.predict_batch(img, [caption_to_use])[0]
        det.image_source = image
        return det


class GroundingDINOClient:
    """grounding_dino.py:77-85; ``port`` accepted and ignored (the model lives in this process).  ``emulate_jpeg=True``
    reproduces the reference's quality-90 JPEG transport (server_wrapper.py:57-68) for A/B checks."""

    _shared: Dict[str, GroundingDINO] = {}

    def __init__(self, port: int = 12181, device=None, emulate_jpeg: bool = False, **model_kwargs) -> None:
        key = str(device)
        if key not in GroundingDINOClient._shared:
            GroundingDINOClient._shared[key] = GroundingDINO(device=device, **model_kwargs)
        self._model = GroundingDINOClient._shared[key]
        self._emulate_jpeg = emulate_jpeg
        self.url = f"inprocess://gdino (port {port} ignored)"

    def predict(self, image_numpy: np.ndarray, caption: Optional[str] = "") -> ObjectDetections:
        seen = image_numpy
        if self._emulate_jpeg:
            from .transport import jpeg_roundtrip

            seen = jpeg_roundtrip(image_numpy)
        det = self._model.predict(seen, caption=caption)
        return ObjectDetections.from_json(det.to_json(), image_source=image_numpy)
