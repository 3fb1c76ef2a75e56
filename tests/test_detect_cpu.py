"""CPU: host logic of the detector wrappers -- ObjectDetections (drop-in container), yolov7 post-processing with a plain
NMS plugged in, caption handling -- and the stand-in networks' output contracts at tiny sizes."""
import numpy as np
import torch

from vlfm_amd.vlm import det_ops
from vlfm_amd.vlm.detections import ObjectDetections, box_convert


def test_object_detections_container():
    boxes = torch.tensor([[0.5, 0.5, 0.2, 0.4], [0.25, 0.3, 0.1, 0.1], [0.8, 0.7, 0.3, 0.2]])
    d = ObjectDetections(boxes, torch.tensor([0.9, 0.4, 0.8]), ["chair", "bed", "chair"], image_source=None)
    assert torch.allclose(d.boxes[0], torch.tensor([0.4, 0.3, 0.6, 0.7]))            # cxcywh -> xyxy (detections.py:30-33)
    assert d.num_detections == 3
    d.filter_by_class(["chair", "tv"])
    assert d.phrases == ["chair", "chair"] and d.boxes.shape == (2, 4)
    d.filter_by_conf(0.8)                                                               # >= (detections.py:70)
    assert d.num_detections == 2
    d.filter_by_conf(0.85)
    assert d.phrases == ["chair"] and torch.allclose(d.logits, torch.tensor([0.9]))
    j = d.to_json()
    r = ObjectDetections.from_json(j)
    assert r.phrases == d.phrases and torch.allclose(r.boxes, d.boxes) and "chair (0.90)" in repr(r)
    d.filter_by_class([])
    assert d.num_detections == 0 and repr(d) == "No detections"
    assert torch.equal(box_convert(boxes, "cxcywh", "cxcywh"), boxes)


def _torch_nms(b, s, thr, max_keep=None):
    from oracle.ref_detect import nms

    k = torch.from_numpy(nms(b.numpy(), s.numpy(), thr))
    return k if max_keep is None else k[:max_keep]


def test_yolo_postprocessing_host_logic():
    g = torch.Generator().manual_seed(0)
    pred = torch.zeros(2, 50, 85)
    pred[..., :2] = torch.rand(2, 50, 2, generator=g) * torch.tensor([640.0, 448.0])
    pred[..., 2:4] = torch.rand(2, 50, 2, generator=g) * 150 + 10
    pred[..., 4] = torch.rand(2, 50, generator=g)
    pred[..., 5:] = torch.rand(2, 50, 80, generator=g)
    out = det_ops.non_max_suppression(pred, 0.25, 0.45, nms_fn=_torch_nms)
    assert len(out) == 2
    for o in out:
        assert o.shape[1] == 6 and (o[:, 4] > 0.25).all() and o.shape[0] <= 300
        assert (o[:, 5] == o[:, 5].round()).all() and (o[:-1, 4] >= o[1:, 4]).all()   # kept in score order
    only = det_ops.non_max_suppression(pred, 0.25, 0.45, classes=[3, 7], nms_fn=_torch_nms)
    assert all(set(o[:, 5].tolist()) <= {3.0, 7.0} for o in only)
    # scale_coords reproduces the reference's quirk: the image was resized NON-uniformly (480 -> 448) but boxes are mapped
    # back with the letterbox formula (gain = min, pad on x)
    c = det_ops.scale_coords((448, 640), torch.tensor([[100.0, 100.0, 300.0, 400.0]]), (480, 640, 3))
    gain = 448 / 480
    pad = (640 - 640 * gain) / 2
    assert torch.allclose(c, torch.tensor([[(100 - pad) / gain, 100 / gain, (300 - pad) / gain, 400 / gain]]))


def test_resize_area_oracle_known_answers():
    from oracle.ref_detect import resize_area_u8

    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    assert np.array_equal(resize_area_u8(img, 6, 4), img)                              # identity
    half = resize_area_u8(img, 3, 2)                                                   # integer factor = box mean
    want = img.reshape(2, 2, 3, 2, 3).astype(np.float32).mean(axis=(1, 3))
    assert np.array_equal(half, np.rint(want).astype(np.uint8))
    const = np.full((480, 640, 3), 77, np.uint8)
    assert (resize_area_u8(const, 640, 448) == 77).all()                               # weights sum to 1


def test_stand_in_networks_contracts():
    from vlfm_amd.vlm.sam import TinyViT

    with torch.inference_mode():   # (the detector's graph has its own file: tests/test_yolov7_cpu.py)
        e = TinyViT().eval()(torch.rand(1, 3, 128, 128))[0]
        assert e.shape == (1, 256, 8, 8)
    from vlfm_amd.vlm.grounding_dino import WordTokenizer, preprocess_caption

    tok = WordTokenizer()
    ids = tok(preprocess_caption("Chair . potted plant"))
    assert ids[0] == 101 and ids[-1] == 102 and ids.count(1012) == 2
    assert tok.decode(ids[1:2]) == "chair" and tok.decode(ids[3:5]) == "potted plant"


def test_object_detections_match_reference_fixture():
    """tests/golden/detections.npz was produced by the reference's own vlfm/vlm/detections.py."""
    from golden_util import load

    g = load("detections")
    names = ["chair", "bed", "potted plant", "tv"]
    d = ObjectDetections(torch.from_numpy(g["in_boxes"]), torch.from_numpy(g["in_logits"]),
                         [names[i] for i in g["in_phrase_idx"]], image_source=None)
    assert np.array_equal(d.boxes.numpy(), g["boxes_xyxy"])
    d.filter_by_class(["chair", "tv", "sofa"])
    assert np.array_equal(d.boxes.numpy(), g["after_class_boxes"]) and d.phrases == list(g["after_class_phrases"])
    d.filter_by_conf(0.8)
    j = d.to_json()
    assert np.array_equal(np.array(j["boxes"], np.float64).reshape(-1, 4), g["after_conf_boxes"])
    assert np.array_equal(np.array(j["logits"]), g["after_conf_logits"]) and j["phrases"] == list(g["after_conf_phrases"])
    assert d.num_detections == int(g["num"])
    assert np.array_equal(ObjectDetections.from_json(j).boxes.numpy(), g["roundtrip_boxes"])


def test_gemm_shaped_convolutions_match_conv2d():
    """det_ops.patch_convs_as_gemm: a kernel == stride (patch embedding) or 1x1 convolution evaluated as one GEMM gives
    nn.Conv2d's result (up to the summation order), and leaves every other convolution alone."""
    import torch
    import torch.nn as nn

    from vlfm_amd.vlm.det_ops import patch_convs_as_gemm

    torch.manual_seed(0)
    net = nn.ModuleDict({"patch": nn.Conv2d(3, 24, 4, 4), "point": nn.Conv2d(24, 16, 1), "keep3": nn.Conv2d(16, 8, 3, 2, 1),
                         "grouped": nn.Conv2d(16, 16, 1, groups=2), "ragged": nn.Conv2d(3, 5, (4, 2), (4, 2), bias=False)})
    x = torch.randn(2, 3, 30, 41)                      # not a multiple of the patch: the remainder is dropped, as conv does
    want = {k: None for k in net}
    with torch.no_grad():
        p = net["patch"](x)
        want = dict(patch=p, point=net["point"](p), keep3=net["keep3"](net["point"](p)), grouped=net["grouped"](net["point"](p)),
                    ragged=net["ragged"](x))
        assert patch_convs_as_gemm(net) == 3           # patch, point, ragged; the 3x3 and the grouped one are left alone
        p = net["patch"](x)
        got = dict(patch=p, point=net["point"](p), keep3=net["keep3"](net["point"](p)), grouped=net["grouped"](net["point"](p)),
                   ragged=net["ragged"](x))
    for k in want:
        assert got[k].shape == want[k].shape, k
        assert torch.allclose(got[k], want[k], atol=1e-5, rtol=1e-5), k


def test_text_branch_cache_returns_the_same_features():
    import torch
    from transformers import GroundingDinoConfig, GroundingDinoForObjectDetection

    from vlfm_amd.vlm.det_ops import cache_text_branch

    tiny = GroundingDinoConfig(num_queries=10, d_model=32, encoder_layers=1, decoder_layers=2, encoder_ffn_dim=64,
                               decoder_ffn_dim=64, encoder_attention_heads=2, decoder_attention_heads=2,
                               backbone_config={"model_type": "swin", "embed_dim": 16, "depths": [1, 1, 1, 1],
                                                "num_heads": [1, 2, 2, 2], "window_size": 4,
                                                "out_features": ["stage2", "stage3", "stage4"]},
                               text_config={"model_type": "bert", "hidden_size": 32, "num_hidden_layers": 1,
                                            "num_attention_heads": 2, "intermediate_size": 64, "vocab_size": 3000,
                                            "max_position_embeddings": 64})
    torch.manual_seed(0)
    model = GroundingDinoForObjectDetection(tiny).eval()
    calls = []
    plain = model.model.text_backbone.forward
    model.model.text_backbone.forward = lambda *a, **k: (calls.append(1), plain(*a, **k))[1]
    cache_text_branch(model)
    pix = torch.randn(1, 3, 96, 128)
    ids = torch.tensor([[101, 2100, 1012, 2200, 1012, 102]])
    with torch.inference_mode():
        model.vlfm_text_key = ("chair . bed .",)
        a = model(pixel_values=pix, input_ids=ids, attention_mask=torch.ones_like(ids))
        b = model(pixel_values=pix, input_ids=ids, attention_mask=torch.ones_like(ids))
        other = torch.tensor([[101, 2300, 1012, 102, 0, 0]])
        model.vlfm_text_key = ("tv .",)
        model(pixel_values=pix, input_ids=other, attention_mask=(other != 0).long())
        model.vlfm_text_key = None                     # no announced caption: the backbone runs as usual
        model(pixel_values=pix, input_ids=other, attention_mask=(other != 0).long())
    assert len(calls) == 3                             # second identical caption: no BERT forward
    assert torch.equal(a.logits, b.logits) and torch.equal(a.pred_boxes, b.pred_boxes)


def test_fold_batchnorm_is_the_same_function():
    """det_ops.fold_batchnorm_ (conv + eval-mode BatchNorm [+ SiLU / GELU module] -> conv + ops.BiasAct) on the two patterns it
    serves: the YOLOv7-class conv + bn + SiLU triple and TinyViT's Conv2d_BN, with non-trivial running statistics; the BiasAct
    module's framework path (CPU) must reproduce bias + activation."""
    import torch.nn as nn

    from vlfm_amd.vlm import ops
    from vlfm_amd.vlm.sam import _ConvBN, _MBConv

    g = torch.Generator().manual_seed(0)

    def randomise(m):
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.2)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
                mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
                mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.2)

    tri = nn.Sequential(nn.Conv2d(5, 7, 3, 1, 1, bias=False), nn.BatchNorm2d(7), nn.SiLU(inplace=True)).eval()
    mb = _MBConv(8).eval()
    cb = _ConvBN(4, 6, 3, 2, 1).eval()
    for net, x in ((tri, torch.randn(2, 5, 9, 11, generator=g)), (mb, torch.randn(2, 8, 12, 12, generator=g)),
                   (cb, torch.randn(1, 4, 10, 10, generator=g))):
        randomise(net)
        with torch.inference_mode():
            want = net(x.clone())
            n = det_ops.fold_batchnorm_(net)
            got = net(x.clone())
        assert n >= 1 and not any(isinstance(m, nn.BatchNorm2d) for m in net.modules())
        assert torch.allclose(got, want, atol=2e-5, rtol=1e-5), float((got - want).abs().max())
    assert isinstance(tri[1], ops.BiasAct) and tri[1].act == "silu" and isinstance(tri[2], nn.Identity) and tri[0].bias is None
    assert isinstance(cb.bn, ops.BiasAct) and cb.bn.act is None
    assert det_ops.fold_batchnorm_(tri) == 0          # idempotent


def test_grounding_dino_postprocessing_equals_the_per_query_loop():
    """GroundingDINO._postprocess_device + _detections (reductions where the logits are, token sets as bit patterns, one decode per
    distinct token set) against the upstream per-query loop
    (groundingdino.util.inference.predict [ext] + grounding_dino.py:70-72) on random probabilities: few, many and no queries
    above the box threshold."""
    from vlfm_amd.vlm.grounding_dino import GroundingDINO, WordTokenizer, preprocess_caption

    class Stub(GroundingDINO):
        def __init__(self):
            wt = WordTokenizer(30522, 256)
            self.tokenizer, self.decode = wt, wt.decode
            self.box_threshold, self.text_threshold = 0.35, 0.25
            self.device, self._phrase_cache = torch.device("cpu"), {}

    g = Stub()
    raw = "chair . bed . potted plant . toilet . tv . couch ."
    ids = g.tokenizer(preprocess_caption(raw))

    def loop(probs, boxes):
        probs, boxes = torch.from_numpy(probs), torch.from_numpy(boxes)
        keep = probs.max(dim=1)[0] > g.box_threshold
        logit, box = probs[keep], boxes[keep]
        phrases = []
        for row in logit:
            pos = row[: len(ids)] > g.text_threshold
            pos[0] = False
            pos[len(ids) - 1:] = False
            phrases.append(g.decode([ids[k] for k in torch.nonzero(pos).flatten().tolist()]).replace(".", "").strip())
        det = ObjectDetections(box, logit.max(dim=1)[0] if len(logit) else torch.zeros(0), phrases, image_source=None)
        det.filter_by_class(raw[: -len(" .")].split(" . "))
        return det

    for seed, power, nq in ((0, 3, 300), (1, 1, 120), (2, 8, 300), (3, 60, 200)):
        rng = np.random.default_rng(seed)
        want = np.clip(rng.uniform(size=(nq, 256)) ** power, 1e-6, 1 - 1e-6)
        logits = torch.from_numpy(np.log(want / (1 - want)).astype(np.float32))
        probs = logits.sigmoid().float().numpy()          # what the device path thresholds: sigmoid of the f32 logits
        boxes = rng.uniform(size=(nq, 4)).astype(np.float32)
        best, bx, bits = g._postprocess_device(logits[None], torch.from_numpy(boxes)[None], [ids])
        a, b = loop(probs, boxes), g._detections(best[0], bx[0], bits[0], ids, raw)
        assert a.phrases == b.phrases and torch.equal(a.boxes, b.boxes) and torch.equal(a.logits, b.logits)
