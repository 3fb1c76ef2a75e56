"""-m gpu: the fused GroundingDINO forward (vlfm_amd/vlm/gdino_fast.py) against transformers' own forward of the SAME weights at the
Swin-T / BERT-base geometry the reference loads (vlfm/vlm/grounding_dino.py:18-33), and its pieces against the modules they replace.
The network runs in fp32 in the reference; the bar for the whole graph is the existing GPU-vs-CPU bar of test_detect_gpu.py (2e-3 on
boxes and token probabilities), for the Swin blocks 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _swin_layer(dim, heads, shift, device):
    from transformers import SwinConfig
    from transformers.models.swin.modeling_swin import SwinLayer

    torch.manual_seed(dim + shift)
    cfg = SwinConfig(embed_dim=dim, window_size=7)
    layer = SwinLayer(cfg, dim=dim, input_resolution=(30, 40), num_heads=heads, shift_size=shift).eval().to(device)
    with torch.no_grad():
        layer.attention.relative_position_bias.relative_position_bias_table.normal_(0, 0.5)
    return layer


@pytest.mark.parametrize("dim,heads,shift,hw", [(96, 3, 0, (30, 40)), (96, 3, 3, (30, 40)), (192, 6, 3, (28, 35)), (384, 12, 3, (15, 20)),
                                                 (768, 24, 0, (8, 10))])
@pytest.mark.parametrize("precision", ["library", "split"])
def test_fused_swin_layer_equals_the_module(gpu_device, dim, heads, shift, hw, precision):
    import copy

    from vlfm_amd.vlm import gdino_fast

    layer = _swin_layer(dim, heads, shift, gpu_device)
    fused = copy.deepcopy(layer)
    st = gdino_fast._State()
    st.precision = precision
    assert gdino_fast.patch_swin_layers(torch.nn.ModuleList([fused]), st) == 1
    H, W = hw
    x = torch.randn(3, H * W, dim, device=gpu_device) * 2.0
    with torch.inference_mode():
        want = layer(x.clone(), (H, W))[0]
        got = fused.forward(x.clone(), (H, W))[0]
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-4 * scale, (float((got - want).abs().max()), scale)


def test_fused_forward_equals_the_hf_forward_at_full_geometry(gpu_device):
    from vlfm_amd.vlm import det_ops, ops
    from vlfm_amd.vlm.grounding_dino import GroundingDINO

    caption = "chair . bed . potted plant . toilet . tv . couch ."
    plain = GroundingDINO(device=gpu_device, allow_random_init=True, seed=3, fast=False)
    fused = GroundingDINO(device=gpu_device, allow_random_init=True, seed=3, fast=True, graph=True)
    fused.model.load_state_dict(plain.model.state_dict())
    rng = np.random.default_rng(11)
    imgs = torch.from_numpy(rng.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8)).to(gpu_device)
    pix = det_ops.to_tensor_normalize(imgs)
    ids = torch.tensor([plain.tokenizer(caption.lower())] * 2, device=gpu_device)
    kw = dict(pixel_values=pix, input_ids=ids, attention_mask=torch.ones_like(ids), token_type_ids=torch.zeros_like(ids))
    with torch.inference_mode():
        want = plain.model(**kw)
        res = {}
        for prec in ("library", "split"):
            fused.model.vlfm_fast.precision = prec
            ops.gemm_f32_overflow_flag(gpu_device, "gdino").zero_()
            out = fused.model(**kw)
            torch.cuda.synchronize()
            assert int(ops.gemm_f32_overflow_flag(gpu_device, "gdino").item()) == 0
            res[prec] = (out.logits.clone(), out.pred_boxes.clone())
            # everything in front of the two-stage query selection: strict
            for name in ("encoder_last_hidden_state_vision", "encoder_last_hidden_state_text", "enc_outputs_class",
                         "enc_outputs_coord_logits"):
                a, b = getattr(out, name), getattr(want, name)
                ok = torch.isfinite(b)
                assert bool((torch.isfinite(a) == ok).all()), (prec, name)
                err, scale = float((a[ok] - b[ok]).abs().max()), float(b[ok].abs().max())
                assert err <= 1e-3 * max(scale, 1.0), (prec, name, err, scale)
            # behind it: the decoder starts from the 900 best of 6380 proposals; with random weights the scores at the cut are a
            # few 1e-6 apart, so an f32-roundoff-level difference can swap a proposal in or out, and THAT query then differs
            # completely.  Queries fed the same proposal must agree to the usual bar; nearly all must be such queries.
            fin = torch.isfinite(want.logits)
            assert bool((torch.isfinite(out.logits) == fin).all())
            dq = (out.pred_boxes - want.pred_boxes).abs().amax(dim=2)                                     # [B, nq]
            dpq = torch.where(fin, (out.logits.sigmoid() - want.logits.sigmoid()).abs(), torch.zeros_like(want.logits)).amax(dim=2)
            same = (dq <= 2e-3) & (dpq <= 2e-3)
            assert float(same.float().mean()) >= 0.97, (prec, float(same.float().mean()), float(dq.max()), float(dpq.max()))
        # the split form against the library form of the SAME fused graph: f32-grade (same-proposal queries)
        d = (res["split"][1] - res["library"][1]).abs().amax(dim=2)
        assert float((d <= 5e-4).float().mean()) >= 0.97, float(d.max())
    # the product path: predict_batch (HIP-graph replay) == the same forward run eagerly, twice in a row
    fused.model.vlfm_fast.precision = "split"
    fused.box_threshold = fused.text_threshold = 0.0
    a = fused.predict_batch(imgs, [caption])
    b = fused.predict_batch(imgs, [caption])
    assert fused.use_graph, "HIP-graph capture of the fused forward failed"
    assert fused.overflow_fallbacks == 0
    for x, y in zip(a, b):
        assert torch.equal(x.boxes, y.boxes) and torch.equal(x.logits, y.logits)


def test_overflowing_operands_fall_back_to_the_exact_gemms(gpu_device):
    """An activation beyond f16's range must not come back as garbage: the split GEMMs flag it and predict_batch repeats the batch
    on the library's f32 GEMMs."""
    from vlfm_amd.vlm.grounding_dino import GroundingDINO

    gd = GroundingDINO(device=gpu_device, allow_random_init=True, seed=4, fast=True, graph=False)
    with torch.no_grad():   # blow up one backbone MLP so that its hidden activation exceeds 65504
        gd.model.model.backbone.conv_encoder.model.swin.encoder.layers[0].blocks[0].mlp.fc1.weight.mul_(1.0e5)
    img = torch.randint(0, 256, (1, 480, 640, 3), dtype=torch.uint8, device=gpu_device)
    out = gd.predict_batch(img, ["chair ."])
    assert gd.overflow_fallbacks == 1 and len(out) == 1


@pytest.mark.parametrize("coords,queries", [(2, 6380), (4, 900)])
def test_fused_deformable_attention_equals_the_module(gpu_device, coords, queries):
    """GroundingDinoMultiscaleDeformableAttention [ext] (value / offsets / logits Linears, softmax, sampling locations, bilinear
    sampling, output projection) against the patched forward with the fused sampling kernel, encoder (reference points) and decoder
    (reference boxes) forms."""
    import copy

    from transformers import GroundingDinoConfig
    from transformers.models.grounding_dino.modeling_grounding_dino import GroundingDinoMultiscaleDeformableAttention

    from vlfm_amd.vlm import det_ops, gdino_fast

    torch.manual_seed(coords)
    cfg = GroundingDinoConfig()
    mod = GroundingDinoMultiscaleDeformableAttention(cfg, num_heads=8, n_points=4).eval().to(gpu_device)
    with torch.no_grad():
        mod.sampling_offsets.weight.normal_(0, 0.05)
        mod.attention_weights.weight.normal_(0, 0.2)
    shapes = [(60, 80), (30, 40), (15, 20), (8, 10)]
    S = sum(h * w for h, w in shapes)
    B = 2
    enc = torch.randn(B, S, 256, device=gpu_device)
    hid = torch.randn(B, queries, 256, device=gpu_device)
    pos = torch.randn(B, queries, 256, device=gpu_device) * 0.1
    ref = torch.rand(B, queries, 4, coords, device=gpu_device) * (0.9 if coords == 2 else 0.5) + 0.05
    spatial = torch.tensor(shapes, device=gpu_device)
    lsi = torch.tensor([0, 4800, 6000, 6300], device=gpu_device)
    resid = torch.randn(B, queries, 256, device=gpu_device)
    plain = copy.deepcopy(mod)
    det_ops.patch_hf_deformable_attention(plain)
    st = gdino_fast._State()
    wrap = torch.nn.ModuleList([mod])
    det_ops.patch_hf_deformable_attention(wrap)
    kw = dict(hidden_states=hid, attention_mask=None, encoder_hidden_states=enc, position_embeddings=pos, reference_points=ref,
              spatial_shapes=spatial, spatial_shapes_list=shapes, level_start_index=lsi)
    with torch.inference_mode():
        want = plain(**kw)[0] + resid
        for prec in ("library", "split"):
            st.precision = prec
            for fused in (False, True):
                st.fused_sampling = fused
                m2 = copy.deepcopy(mod)
                det_ops.patch_hf_deformable_attention(m2)
                assert gdino_fast.patch_deformable(torch.nn.ModuleList([m2]), st) == 1
                got = m2.forward(residual=resid, **kw)[0]
                err = float((got - want).abs().max())
                assert err <= 2e-4 * float(want.abs().max()), (prec, fused, err)


def test_reassociated_fusion_layer_equals_the_module(gpu_device):
    """GroundingDinoFusionLayer [ext] against its re-associated form (text pushed through the vision-side weights) at the encoder's real
    sizes: 6380 vision tokens, a 15-token caption with padding in one batch entry."""
    import copy

    from transformers import GroundingDinoConfig
    from transformers.models.grounding_dino.modeling_grounding_dino import GroundingDinoFusionLayer

    from vlfm_amd.vlm import gdino_fast

    torch.manual_seed(21)
    layer = GroundingDinoFusionLayer(GroundingDinoConfig()).eval().to(gpu_device)
    with torch.no_grad():
        layer.vision_param.fill_(0.3)
        layer.text_param.fill_(0.2)
        for p in layer.attn.parameters():
            p.normal_(0, 0.05)
    fused = copy.deepcopy(layer)
    st = gdino_fast._State()
    assert gdino_fast.patch_fusion(torch.nn.ModuleList([fused]), st) == 1
    B, Lv, Lt = 3, 6380, 15
    v = torch.randn(B, Lv, 256, device=gpu_device)
    t = torch.randn(B, Lt, 256, device=gpu_device)
    mt = torch.zeros(B, Lt, dtype=torch.bool, device=gpu_device)
    mt[1, 11:] = True
    mv = torch.zeros(B, Lv, dtype=torch.bool, device=gpu_device)
    with torch.inference_mode():
        (v0, pv0), (t0, pt0) = layer(v, t, attention_mask_vision=mv, attention_mask_text=mt)
        for prec in ("library", "split"):
            st.precision = prec
            (v1, pv1), (t1, pt1) = fused.forward(v, t, attention_mask_vision=mv, attention_mask_text=mt)
            errs = {"vision": float((v0 - v1).abs().max()), "text": float((t0 - t1).abs().max()),
                    "p_vision": float((pv0 - pv1).abs().max()), "p_text": float((pt0 - pt1).abs().max())}
            assert errs["vision"] <= 2e-5 * float(v0.abs().max()) and errs["text"] <= 2e-5 * float(t0.abs().max()), (prec, errs)
            assert errs["p_vision"] <= 1e-5 and errs["p_text"] <= 1e-5, (prec, errs)
