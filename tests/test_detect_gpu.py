"""-m gpu: detector-side HIP kernels against their CPU restatements (oracle/ref_detect.py), the real Pillow, plain PyTorch,
and the in-process YOLOv7 / GroundingDINO / MobileSAM wrappers (random-init weights: contracts + GPU-vs-CPU self-parity)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(480, 640), (720, 1280), (448, 640)])
def test_resize_area_matches_oracle_bit_exact(gpu_device, shape):
    from oracle.ref_detect import resize_area_u8
    from vlfm_amd.vlm import det_ops

    rng = np.random.default_rng(3)
    imgs = rng.integers(0, 256, size=(2, shape[0], shape[1], 3), dtype=np.uint8)
    got = det_ops.resize_area(torch.from_numpy(imgs).to(gpu_device), 448, 640, torch.float32).cpu()
    for i in range(2):
        want = torch.from_numpy(resize_area_u8(imgs[i], 640, 448)).permute(2, 0, 1).float() / 255.0
        assert torch.equal(got[i], want)
    half = det_ops.resize_area(torch.from_numpy(imgs).to(gpu_device), 448, 640, torch.float16).cpu()
    u8 = torch.from_numpy(np.stack([resize_area_u8(im, 640, 448) for im in imgs])).permute(0, 3, 1, 2)
    assert torch.equal(half, u8.half() / 255.0)                     # yolov7.py:81-82: img.half(); img /= 255.0


def test_to_tensor_normalize_bit_exact(gpu_device):
    from vlfm_amd.vlm import det_ops

    rng = np.random.default_rng(4)
    imgs = rng.integers(0, 256, size=(2, 120, 160, 3), dtype=np.uint8)
    got = det_ops.to_tensor_normalize(torch.from_numpy(imgs).to(gpu_device)).cpu()
    t = torch.from_numpy(imgs).permute(0, 3, 1, 2).float().div(255)                    # to_tensor
    mean = torch.tensor(det_ops.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(det_ops.IMAGENET_STD).view(1, 3, 1, 1)
    assert torch.equal(got, (t - mean) / std)                                          # sub_(mean).div_(std)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 700, 3000])
def test_nms_matches_oracle(gpu_device, n):
    from oracle.ref_detect import nms as ref_nms
    from vlfm_amd.vlm import det_ops

    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 600, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 200, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1)
    boxes[n // 2] = boxes[0]                                        # exact duplicates and ties in score
    scores = rng.permutation(n).astype(np.float32) / n              # distinct scores: the order is unambiguous
    for thr in (0.45, 0.1, 0.9):
        got = det_ops.nms(torch.from_numpy(boxes).to(gpu_device), torch.from_numpy(scores).to(gpu_device), thr).cpu().numpy()
        want = ref_nms(boxes, scores, thr)
        assert np.array_equal(got, want), (n, thr, len(got), len(want))
    capped = det_ops.nms(torch.from_numpy(boxes).to(gpu_device), torch.from_numpy(scores).to(gpu_device), 0.45, 5).cpu().numpy()
    assert np.array_equal(capped, ref_nms(boxes, scores, 0.45)[:5])
    assert det_ops.nms(torch.zeros((0, 4), device=gpu_device), torch.zeros(0, device=gpu_device), 0.5).numel() == 0


def test_preprocess_sam_matches_pil(gpu_device):
    from PIL import Image

    from vlfm_amd.vlm import ops

    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, size=(2, 480, 640, 3), dtype=np.uint8)
    got, (oh, ow) = ops.preprocess_sam(torch.from_numpy(imgs).to(gpu_device))
    assert (oh, ow) == (768, 1024) and got.shape == (2, 3, 1024, 1024)
    mean = torch.tensor(ops.SAM_MEAN).view(3, 1, 1)
    std = torch.tensor(ops.SAM_STD).view(3, 1, 1)
    for i in range(2):
        pil = np.asarray(Image.fromarray(imgs[i]).resize((ow, oh), Image.BILINEAR))   # ResizeLongestSide.apply_image
        want = (torch.from_numpy(pil.copy()).permute(2, 0, 1).float() - mean) / std     # Sam.preprocess
        assert torch.equal(got[i, :, :oh, :ow].cpu(), want)
        assert (got[i, :, oh:, :] == 0).all()                                           # zero padding after normalisation


def test_yolov7_pipeline(gpu_device):
    from vlfm_amd.vlm import det_ops
    from vlfm_amd.vlm.coco_classes import COCO_CLASSES
    from vlfm_amd.vlm.yolov7 import YOLOv7, YOLOv7Client

    model = YOLOv7(device=gpu_device, width_multiple=0.2, allow_random_init=True)   # the E6E topology in miniature
    with torch.no_grad():   # make the random net emit confident boxes
        for d in model.model.model[-1].m:
            d.bias.fill_(0.0)
            d.bias.view(3, 85)[:, 4] = 1.5
    rng = np.random.default_rng(6)
    imgs = rng.integers(0, 256, size=(2, 480, 640, 3), dtype=np.uint8)
    dets = model.predict_batch(torch.from_numpy(imgs).to(gpu_device))
    assert len(dets) == 2
    for d in dets:
        assert d.boxes.shape[1] == 4 and d.num_detections == len(d.logits) <= 300
        assert (d.logits > 0.25).all() and all(p in COCO_CLASSES for p in d.phrases)
        assert (d.boxes >= 0).all() and (d.boxes[:, [0, 2]] <= 1.0 + 1e-6).all()
    # the single-image entry point is the batched one at B = 1
    one = model.predict(imgs[1])
    assert one.phrases == dets[1].phrases and torch.allclose(one.boxes, dets[1].boxes)
    # post-processing parity: same raw predictions -> HIP NMS == reference NMS
    from oracle.ref_detect import nms as ref_nms

    img = det_ops.resize_area(torch.from_numpy(imgs).to(gpu_device), 448, 640, torch.float16)
    with torch.inference_mode():
        pred = model.model(img).float()
    a = det_ops.non_max_suppression(pred, 0.25, 0.45)
    b = det_ops.non_max_suppression(pred.cpu(), 0.25, 0.45,
                                    nms_fn=lambda bx, s, t, m: torch.from_numpy(ref_nms(bx.numpy(), s.numpy(), t))[:m])
    for x, y in zip(a, b):
        assert torch.equal(x.cpu(), y)
    c = YOLOv7Client(port=12184, device=gpu_device, width_multiple=0.2, allow_random_init=True)
    assert isinstance(c.predict(imgs[0]).to_json()["phrases"], list)


def test_grounding_dino_wrapper(gpu_device):
    from transformers import GroundingDinoConfig

    from vlfm_amd.vlm.grounding_dino import GroundingDINO

    tiny = GroundingDinoConfig(num_queries=30, d_model=32, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=64,
                               decoder_ffn_dim=64, encoder_attention_heads=2, decoder_attention_heads=2,
                               backbone_config={"model_type": "swin", "embed_dim": 16, "depths": [1, 1, 1, 1],
                                                "num_heads": [1, 2, 2, 2], "window_size": 4,
                                                "out_features": ["stage2", "stage3", "stage4"]},
                               text_config={"model_type": "bert", "hidden_size": 32, "num_hidden_layers": 1,
                                            "num_attention_heads": 2, "intermediate_size": 64, "vocab_size": 30522,
                                            "max_position_embeddings": 64})
    gd = GroundingDINO(device=gpu_device, hf_config=tiny, box_threshold=0.0, text_threshold=0.0)
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(96, 128, 3), dtype=np.uint8)
    det = gd.predict(img, caption="chair . potted plant .")
    # thresholds 0 -> every query survives with the phrase "chair potted plant", which is not an exact class -> filtered
    assert det.num_detections == 0
    gd.text_threshold = 2.0   # no token passes -> empty phrases -> filtered as well
    assert gd.predict(img, caption="chair .").num_detections == 0
    # raw model outputs: GPU == CPU fp32 of the same weights (the deformable attention runs in plain PyTorch on both)
    from vlfm_amd.vlm import det_ops

    pix = det_ops.to_tensor_normalize(torch.from_numpy(img).to(gpu_device)[None])
    ids = torch.tensor([gd.tokenizer("chair .")])
    kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), token_type_ids=torch.zeros_like(ids))
    with torch.inference_mode():
        g = gd.model(pixel_values=pix, **{k: v.to(gpu_device) for k, v in kw.items()})
        cpu_model = gd.model.to("cpu")
        c = cpu_model(pixel_values=pix.cpu(), **kw)
    assert torch.allclose(g.pred_boxes.cpu(), c.pred_boxes, atol=2e-3)
    assert torch.allclose(g.logits.cpu().sigmoid(), c.logits.sigmoid(), atol=2e-3)


def test_mobile_sam_wrapper(gpu_device):
    from vlfm_amd.vlm.sam import MobileSAM, MobileSAMClient

    sam = MobileSAM(device=gpu_device, allow_random_init=True)
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    mask = sam.segment_bbox(img, [100, 120, 400, 380])
    assert mask.shape == (480, 640) and mask.dtype == np.bool_          # full-frame mask (sam.py:49-57 quirk)
    imgs = torch.from_numpy(np.stack([img, img[::-1].copy()])).to(gpu_device)
    boxes = torch.tensor([[[100, 120, 400, 380], [10, 10, 200, 200]], [[300, 50, 630, 470], [0, 0, 639, 479]]], dtype=torch.float32)
    masks = sam.segment_bboxes(imgs, boxes)
    assert masks.shape == (2, 2, 480, 640) and masks.dtype == torch.bool
    assert np.array_equal(masks[0, 0].cpu().numpy(), mask)              # batched == single
    c = MobileSAMClient(port=12183, device=gpu_device, allow_random_init=True)
    assert c.segment_bbox(img, [100, 120, 400, 380]).shape == (480, 640)


def test_ms_deform_attn_matches_hf_pytorch_path(gpu_device):
    """HIP multi-scale deformable attention vs the HF module's own grid_sample formulation (fp32, 1e-5)."""
    from transformers.models.grounding_dino.modeling_grounding_dino import MultiScaleDeformableAttention

    from vlfm_amd.vlm import det_ops

    g = torch.Generator().manual_seed(0)
    shapes = [(60, 80), (30, 40), (15, 20), (8, 10)]
    B, Q, heads, D, L, P = 2, 50, 8, 32, 4, 4
    S = sum(h * w for h, w in shapes)
    value = torch.randn(B, S, heads, D, generator=g)
    loc = torch.rand(B, Q, heads, L, P, 2, generator=g) * 1.3 - 0.15       # some samples fall outside: zero padding
    w = torch.softmax(torch.randn(B, Q, heads, L * P, generator=g), -1).view(B, Q, heads, L, P)
    start = torch.tensor([0] + list(np.cumsum([h * w_ for h, w_ in shapes])[:-1]))
    want = MultiScaleDeformableAttention()(value, torch.tensor(shapes), shapes, start, loc, w, 64)
    got = det_ops.ms_deform_attn(value.to(gpu_device), shapes, start.to(gpu_device), loc.to(gpu_device), w.to(gpu_device)).cpu()
    assert got.shape == want.shape == (B, Q, heads * D)
    assert torch.allclose(got, want, atol=1e-5, rtol=1e-5), float((got - want).abs().max())


def test_yolov7_weight_routes_and_no_silent_fallback(gpu_device, tmp_path):
    """The weights routes of YOLOv7 (yolov7.py:35-38 in the reference hands `yolov7-e6e.pt` to attempt_load): a TorchScript
    export whose output is the inference tensor [B, N, 85] (exercised with a traced miniature: same detections as the eager
    module), a checkpoint / state dict of the FULL yolov7-e6e graph (synthetic tensors under the checkpoint's own keys: loaded
    strictly, then the same detections as the network they came from); a missing path, a file that is neither, a state dict
    of another network, or no weights at all must raise."""
    from vlfm_amd.vlm.yolov7 import YOLOv7
    from vlfm_amd.vlm.yolov7_e6e import YoloV7E6E

    def confident(net):
        with torch.no_grad():   # make the random net emit confident boxes
            for m in net.model[-1].m:
                m.bias.fill_(0.0)
                m.bias.view(3, 85)[:, 4] = 1.5
        return net

    torch.manual_seed(3)
    with torch.device(gpu_device):
        net = confident(YoloV7E6E(width_multiple=0.2).init_random(3).eval())
    example = torch.rand(1, 3, 448, 640, device=gpu_device)
    path = str(tmp_path / "mini.torchscript.pt")
    torch.jit.trace(net, example).save(path)
    loaded = YOLOv7(weights=path, device=gpu_device, half_precision=False)
    assert loaded.weights == f"torchscript:{path}"
    eager = YOLOv7(device=gpu_device, width_multiple=0.2, allow_random_init=True, half_precision=False)
    eager.model = net
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    a, b = loaded.predict(img), eager.predict(img)
    assert a.num_detections == b.num_detections and a.num_detections > 0
    assert torch.allclose(a.boxes, b.boxes, atol=1e-5) and a.phrases == b.phrases
    assert "conv GFLOPs" in eager.description and "M parameters" in eager.description and eager.gflops > 0
    # ---- the full graph through the checkpoint route
    with torch.device(gpu_device):
        full = confident(YoloV7E6E().init_random(5).eval())
    full.half().float()     # f16-representable weights: the file stores halves (like the real one), the comparison stays exact
    ck = str(tmp_path / "yolov7-e6e.pt")
    torch.save({"model": {k: v.half() if v.is_floating_point() else v for k, v in full.state_dict().items()}, "ema": None}, ck)
    from_ck = YOLOv7(weights=ck, device=gpu_device)
    assert from_ck.weights.startswith("yolov7-e6e checkpoint (unfused)") and "151.8 M parameters" in from_ck.description
    ref = YOLOv7(device=gpu_device, allow_random_init=True)
    ref.model = full.fuse_().half()      # what YOLOv7.__init__ does with a loaded graph: fold in f32, then halve
    a, b = from_ck.predict(img), ref.predict(img)
    assert a.num_detections == b.num_detections > 0 and a.phrases == b.phrases and torch.equal(a.boxes, b.boxes)
    with pytest.raises(FileNotFoundError):
        YOLOv7(weights=str(tmp_path / "nope.pt"), device=gpu_device)
    bogus = tmp_path / "pickled.pt"
    torch.save({"model": "neither"}, str(bogus))
    with pytest.raises(ValueError, match="neither a TorchScript module"):
        YOLOv7(weights=str(bogus), device=gpu_device)
    other = tmp_path / "other_net.pt"
    torch.save(net.state_dict(), str(other))       # a state dict, but of a different network (the miniature's shapes)
    with pytest.raises(ValueError, match="shape"):
        YOLOv7(weights=str(other), device=gpu_device)
    with pytest.raises(ValueError, match="allow_random_init"):
        YOLOv7(device=gpu_device)


def test_detectors_and_segmenter_refuse_unusable_weights(gpu_device, tmp_path):
    from vlfm_amd.vlm.grounding_dino import GroundingDINO
    from vlfm_amd.vlm.sam import MobileSAM

    with pytest.raises(ValueError, match="allow_random_init"):
        GroundingDINO(device=gpu_device)
    pth = tmp_path / "groundingdino_swint_ogc.pth"
    torch.save({"model": {}}, str(pth))
    with pytest.raises(KeyError, match="does not have"):         # an empty / foreign state dict is not silently accepted
        GroundingDINO(weights_path=str(pth), device=gpu_device)
    with pytest.raises(FileNotFoundError):
        GroundingDINO(config_path=str(tmp_path / "GroundingDINO_SwinT_OGC.py"), weights_path=str(pth), device=gpu_device)
    with pytest.raises(FileNotFoundError):
        GroundingDINO(weights_path=str(tmp_path / "missing.pth"), device=gpu_device)
    with pytest.raises(ValueError, match="allow_random_init"):
        MobileSAM(device=gpu_device)
    with pytest.raises(FileNotFoundError):
        MobileSAM(sam_checkpoint=str(tmp_path / "mobile_sam.pt"), device=gpu_device)


def test_grounding_dino_loads_the_references_own_checkpoint_format(gpu_device, tmp_path):
    """grounding_dino.py:18-19,33: ``weights_path`` = a groundingdino ``.pth`` ({"model": state dict under the ORIGINAL names,
    q | k | v fused}).  A synthetic file in that layout (a miniature network's weights re-packed) goes through the constructor;
    the detections equal those of the network the tensors came from.  Real weights need the BERT vocabulary on disk."""
    from transformers import GroundingDinoConfig, GroundingDinoForObjectDetection

    from vlfm_amd.vlm import gdino_weights as gw
    from vlfm_amd.vlm.grounding_dino import GroundingDINO

    tiny = GroundingDinoConfig(num_queries=30, d_model=32, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=64,
                               decoder_ffn_dim=64, encoder_attention_heads=2, decoder_attention_heads=2,
                               backbone_config={"model_type": "swin", "embed_dim": 16, "depths": [1, 1, 1, 1],
                                                "num_heads": [1, 2, 2, 2], "window_size": 4,
                                                "out_features": ["stage2", "stage3", "stage4"]},
                               text_config={"model_type": "bert", "hidden_size": 32, "num_hidden_layers": 1,
                                            "num_attention_heads": 2, "intermediate_size": 64, "vocab_size": 30522,
                                            "max_position_embeddings": 64})
    torch.manual_seed(4)
    src = GroundingDinoForObjectDetection(tiny).eval()
    hf, spec = src.state_dict(), gw.original_state_dict_spec(tiny)
    orig = {}
    for k, t in hf.items():
        s = gw.source_of(k)
        if s is None:
            continue
        if s[1] is None:
            orig[s[0]] = t.clone()
        else:
            n = spec[s[0]][0] // 3
            orig.setdefault(s[0], torch.zeros(spec[s[0]]))[s[1] * n:(s[1] + 1) * n] = t
    for k, shape in spec.items():
        orig.setdefault(k, torch.zeros(shape, dtype=torch.long if k.endswith(("position_ids", "position_index")) else torch.float32))
    pth = str(tmp_path / "groundingdino_swint_ogc.pth")
    torch.save({"model": {"module." + k: v for k, v in orig.items()}}, pth)
    vocab = tmp_path / "vocab.txt"
    words = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + \
            [f"w{i}" for i in range(908)] + ["."] + ["chair", "bed", "plant", "potted", "tv", "couch", "toilet"]
    vocab.write_text("\n".join(words) + "\n")
    with pytest.raises(ValueError, match="vocabulary"):
        GroundingDINO(weights_path=pth, hf_config=tiny, device=gpu_device)       # no vocabulary on disk, no stand-in
    gd = GroundingDINO(weights_path=pth, hf_config=tiny, device=gpu_device, tokenizer_dir=str(tmp_path),
                       box_threshold=0.0, text_threshold=0.0)
    assert gd.weights.startswith("groundingdino checkpoint:")
    ref = GroundingDINO(hf_config=tiny, device=gpu_device, box_threshold=0.0, text_threshold=0.0)
    ref.model.load_state_dict(hf)
    ref.tokenizer, ref.decode = gd.tokenizer, gd.decode
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    a, b = gd.predict(img, caption="chair . bed ."), ref.predict(img, caption="chair . bed .")
    assert a.phrases == b.phrases and torch.equal(a.boxes, b.boxes) and torch.equal(a.logits, b.logits)


def test_mobile_sam_loads_a_checkpoint_and_matches_the_oracle(gpu_device, tmp_path):
    """sam.py:35-38: ``sam_checkpoint`` must load.  A synthetic mobile_sam.pt (the file's keys and shapes) goes through the
    constructor; the full segment_bbox pipeline on the GPU (HIP preprocessing, TinyViT, box prompt, decoder, two bilinear
    resizes, threshold) is compared with the oracle's network fed the same preprocessed pixels."""
    from oracle import ref_mobile_sam as ref
    from vlfm_amd.vlm import ops
    from vlfm_amd.vlm.sam import MobileSAM

    sd = ref.synthetic_checkpoint(11)
    path = str(tmp_path / "mobile_sam.pt")
    torch.save(sd, path)
    sam = MobileSAM(sam_checkpoint=path, device=gpu_device)
    assert "mobile_sam.pt" in sam.weights
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    box = [192, 144, 448, 384]
    got = sam.segment_bbox(img, box)
    pix, (oh, ow) = ops.preprocess_sam(torch.from_numpy(img).to(gpu_device)[None])
    boxes_1024 = torch.tensor([box], dtype=torch.float32) * torch.tensor([ow / 640, oh / 480, ow / 640, oh / 480])
    low = ref.predict_low_res(sd, pix.float().cpu(), boxes_1024)[None]
    m = torch.nn.functional.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)[..., :oh, :ow]
    want = (torch.nn.functional.interpolate(m, (480, 640), mode="bilinear", align_corners=False) > 0)[0, 0].numpy()
    assert got.shape == want.shape == (480, 640)
    assert (got != want).mean() <= 2e-3, (got != want).mean()    # fp32 on both sides; only logits within ~1e-4 of 0 may flip
    assert 0.02 < want.mean() < 0.98


@pytest.mark.parametrize("case", [(2, 16, 32, 32, 1), (2, 16, 32, 32, 2), (1, 8, 64, 128, 1), (3, 5, 24, 40, 2), (1, 4, 12, 1024, 1)])
def test_depthwise_conv3x3_kernel(gpu_device, case):
    """vlfm_dwconv3x3_f32 (TinyViT's depthwise convolutions) against F.conv2d evaluated in float64: borders, both strides,
    with and without bias and the fused exact GELU."""
    import torch.nn.functional as F

    from vlfm_amd.vlm import ops

    n, c, h, w, stride = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(c, generator=g)
    for bias in (b, None):
        for gelu in (False, True):
            want = F.conv2d(x.double(), wt.double(), None if bias is None else bias.double(), stride, 1, 1, c)
            if gelu:
                want = F.gelu(want)
            got = ops.depthwise_conv3x3(x.to(gpu_device), wt.to(gpu_device), None if bias is None else bias.to(gpu_device),
                                        stride, gelu).cpu()
            assert got.shape == want.shape
            assert torch.allclose(got.double(), want, atol=2e-6, rtol=2e-6), float((got.double() - want).abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("act", [None, "gelu", "silu"])
def test_bias_act_kernel(gpu_device, dtype, act):
    """vlfm_bias_act_nchw (the folded BatchNorm's bias + the activation in one in-place pass) against the framework ops in
    float64: aligned planes, planes whose size is not a multiple of the vector width, a single-pixel plane."""
    import torch.nn.functional as F

    from vlfm_amd.vlm import ops

    g = torch.Generator().manual_seed(3)
    for shape in [(2, 5, 16, 24), (3, 7, 5, 7), (1, 3, 1, 1), (1, 2, 33, 130)]:
        x = (torch.randn(shape, generator=g) * 2).to(dtype)
        b = torch.randn(shape[1], generator=g).to(dtype)
        want = x.double() + b.double().view(1, -1, 1, 1)
        want = F.gelu(want) if act == "gelu" else F.silu(want) if act == "silu" else want
        y = x.to(gpu_device).clone()
        pure = ops.bias_act(y, b.to(gpu_device), act)             # default: the input is left alone (framework ops)
        assert pure.data_ptr() != y.data_ptr() and torch.equal(y.cpu(), x)
        out = ops.bias_act(y, b.to(gpu_device), act, inplace=True)
        assert out.data_ptr() == y.data_ptr()          # in place: the HIP kernel
        tol = 2e-6 if dtype == torch.float32 else 2e-3
        assert torch.allclose(out.cpu().double(), want, atol=tol, rtol=tol), (shape, float((out.cpu().double() - want).abs().max()))
        assert torch.allclose(pure.cpu().double(), want, atol=tol, rtol=tol)
