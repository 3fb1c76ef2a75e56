"""CPU checks of the BLIP-2 ITC path: (1) our lean ITC graph == Hugging Face's Blip2ForImageTextRetrieval on the same
(random) weights -- the installed HF port is the only executable statement of the LAVIS graph in this environment
(SURVEY.md 3.4); (2) the Pillow-restated bicubic taps reproduce the real PIL.Image.resize bit for bit."""
import numpy as np
import pytest
import torch

from vlfm_amd.vlm.blip2itm import Blip2ITCConfig, Blip2ITCModel, HashTokenizer, blip_caption


def _hf_model(cfg: Blip2ITCConfig):
    from transformers import Blip2Config, Blip2ForImageTextRetrieval, Blip2QFormerConfig, Blip2VisionConfig

    v = Blip2VisionConfig(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_mlp, num_hidden_layers=cfg.v_layers,
                          num_attention_heads=cfg.v_heads, image_size=cfg.image_size, patch_size=cfg.patch_size,
                          layer_norm_eps=cfg.v_ln_eps, qkv_bias=True)
    q = Blip2QFormerConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.q_hidden, num_hidden_layers=cfg.q_layers,
                           num_attention_heads=cfg.q_heads, intermediate_size=cfg.q_mlp,
                           max_position_embeddings=cfg.max_position_embeddings, layer_norm_eps=cfg.q_ln_eps,
                           cross_attention_frequency=cfg.cross_attention_frequency, encoder_hidden_size=cfg.v_hidden,
                           use_qformer_text_input=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    c = Blip2Config(vision_config=v.to_dict(), qformer_config=q.to_dict(), text_config={"model_type": "opt"},
                    num_query_tokens=cfg.num_query_tokens, image_text_hidden_size=cfg.proj_dim)
    c.image_token_index = None
    m = Blip2ForImageTextRetrieval(c).eval().float()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.08)
        for mod in m.modules():
            if isinstance(mod, torch.nn.LayerNorm):
                mod.weight.add_(1.0)
    return m


def test_itc_graph_matches_hf_port():
    cfg = Blip2ITCConfig.tiny()
    hf = _hf_model(cfg)
    ours = Blip2ITCModel(cfg).eval()
    ours.load_hf_state_dict(hf.state_dict())
    g = torch.Generator().manual_seed(2)
    pix = torch.randn(3, 3, cfg.image_size, cfg.image_size, generator=g)
    ids = torch.randint(0, cfg.vocab_size, (1, 9), generator=g)
    with torch.inference_mode():
        want = hf(pixel_values=pix, input_ids=ids.expand(3, -1), attention_mask=torch.ones(3, 9, dtype=torch.long),
                  use_image_text_matching_head=False).logits_per_image
        # HF scores every image against every text; the diagonal block is image i vs its own (identical) text
        want = want[:, 0] if want.dim() == 2 else want
        q = ours.query_features(ours.vision_tokens(pix))
        got = ours.itc_reference_head(q, ours.text_feature(ids))
    assert got.shape == (3,)
    assert torch.allclose(got, want.reshape(-1)[:3] if want.numel() == 3 else want.flatten()[:3], atol=2e-5, rtol=0), (got, want)
    assert got.abs().max() <= 1.0 + 1e-6


def test_text_branch_padding_mask_and_caption_processor():
    cfg = Blip2ITCConfig.tiny()
    m = Blip2ITCModel(cfg).init_random(3).eval()
    ids = torch.tensor([[5, 6, 7, 8]])
    padded = torch.tensor([[5, 6, 7, 8, 0, 0]])
    mask = torch.tensor([[1, 1, 1, 1, 0, 0]])
    with torch.inference_mode():
        a = m.text_feature(ids)
        b = m.text_feature(padded, mask)
    assert torch.allclose(a, b, atol=1e-6)
    assert blip_caption('Seems like there is a  "Potted Plant" ahead.') == "seems like there is a potted plant ahead"
    tok = HashTokenizer(30523, 32)
    t1, t2 = tok("seems like there is a chair ahead"), tok("seems like there is a chair ahead")
    assert t1 == t2 and t1[0] == 101 and t1[-1] == 102 and len(t1) == 9
    assert len(tok(" ".join(["w"] * 100))) == 32


def test_flagship_config_matches_blip2_vitg():
    c = Blip2ITCConfig()
    m = Blip2ITCModel.__new__(Blip2ITCModel)  # geometry only: count parameters analytically (no 1.2 GB alloc)
    vit = c.v_layers * (4 * c.v_hidden * c.v_hidden + 2 * c.v_hidden * c.v_mlp)
    assert (c.image_size // c.patch_size) ** 2 + 1 == 257 and c.v_layers == 39 and c.v_hidden == 1408
    assert 0.95e9 < vit < 1.05e9  # ViT-g/14 minus the last block ~ 0.99 B weights in the blocks
    # forward flops per frame ~ 2 * params * tokens + attention
    flops = 2 * vit * 257 + c.v_layers * 4 * 257 * 257 * c.v_hidden
    assert 0.5e12 < flops < 0.56e12  # the 0.52 TFLOP/frame of SURVEY.md 7


@pytest.mark.parametrize("shape", [(480, 640), (720, 1280), (97, 131)])
def test_resample_taps_reproduce_real_pil(shape):
    from PIL import Image

    from vlfm_amd.vlm.ops import resample_coeffs

    rng = np.random.default_rng(0)
    H, W = shape
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((224, 224), Image.BICUBIC))
    hb, hk, _ = resample_coeffs(W, 224)
    vb, vk, _ = resample_coeffs(H, 224)
    # horizontal pass (u8 round trip), then vertical pass -- Pillow's ImagingResample order
    tmp = np.zeros((H, 224, 3), np.uint8)
    for xx in range(224):
        x0, cnt = hb[xx]
        acc = (img[:, x0:x0 + cnt, :].astype(np.int64) * hk[xx, :cnt, None]).sum(axis=1) + (1 << 21)
        tmp[:, xx, :] = np.clip(acc >> 22, 0, 255)
    out = np.zeros((224, 224, 3), np.uint8)
    for yy in range(224):
        y0, cnt = vb[yy]
        acc = (tmp[y0:y0 + cnt].astype(np.int64) * vk[yy, :cnt, None, None]).sum(axis=0) + (1 << 21)
        out[yy] = np.clip(acc >> 22, 0, 255)
    assert np.array_equal(out, want)


def test_head_padding_is_exact():
    """_VitBlock.pack_heads zero-pads the 88-wide heads to 96 inside the weights: outputs must not change."""
    cfg = Blip2ITCConfig(image_size=28, patch_size=14, v_hidden=88 * 2, v_layers=2, v_heads=2, v_mlp=64, q_hidden=24,
                         q_layers=2, q_heads=4, q_mlp=48, vocab_size=97, max_position_embeddings=40,
                         num_query_tokens=5, proj_dim=8)
    m = Blip2ITCModel(cfg).init_random(1).eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() > 1:
                p.mul_(8)
            else:
                p.add_(torch.randn_like(p) * 0.1)
    x = torch.randn(2, 3, 28, 28)
    with torch.inference_mode():
        want = m.vision_tokens(x)
        for blk in m.blocks:
            blk.pack_heads()
        assert m.blocks[0]._packed is not None and m.blocks[0]._packed[3] == 96
        got = m.vision_tokens(x)
    assert torch.allclose(got, want, atol=1e-6, rtol=0)


# ------------------------------------------------------------------------------------------------ the reference's artefacts
def _lavis_spec(cfg: Blip2ITCConfig, with_vit_in_file: bool):
    """Names -> shapes of LAVIS' artefacts for this geometry, from LAVIS' own module definitions [ext] (Blip2Qformer in
    lavis/models/blip2_models/blip2_qformer.py + Qformer.py, VisionTransformer in lavis/models/eva_vit.py):
    ``blip2_pretrained.pth['model']`` and ``eva_vit_g.pth`` (the full 40-block EVA tower with its classifier norm)."""
    H, I, Hq, Iq = cfg.v_hidden, cfg.v_mlp, cfg.q_hidden, cfg.q_mlp
    n_tok = (cfg.image_size // cfg.patch_size) ** 2 + 1
    vit = {"cls_token": (1, 1, H), "pos_embed": (1, n_tok, H), "patch_embed.proj.weight": (H, 3, cfg.patch_size, cfg.patch_size),
           "patch_embed.proj.bias": (H,), "norm.weight": (H,), "norm.bias": (H,)}
    for i in range(cfg.v_layers + 1):          # LAVIS builds depth - 1 blocks and loads non-strictly: the file has one more
        o = f"blocks.{i}."
        vit.update({o + "norm1.weight": (H,), o + "norm1.bias": (H,), o + "attn.q_bias": (H,), o + "attn.v_bias": (H,),
                    o + "attn.qkv.weight": (3 * H, H), o + "attn.proj.weight": (H, H), o + "attn.proj.bias": (H,),
                    o + "norm2.weight": (H,), o + "norm2.bias": (H,), o + "mlp.fc1.weight": (I, H), o + "mlp.fc1.bias": (I,),
                    o + "mlp.fc2.weight": (H, I), o + "mlp.fc2.bias": (H,)})
    q = {"query_tokens": (1, cfg.num_query_tokens, Hq), "ln_vision.weight": (H,), "ln_vision.bias": (H,),
         "Qformer.bert.embeddings.position_ids": (1, cfg.max_position_embeddings),
         "Qformer.bert.embeddings.word_embeddings.weight": (cfg.vocab_size, Hq),
         "Qformer.bert.embeddings.position_embeddings.weight": (cfg.max_position_embeddings, Hq),
         "Qformer.bert.embeddings.LayerNorm.weight": (Hq,), "Qformer.bert.embeddings.LayerNorm.bias": (Hq,),
         "vision_proj.weight": (cfg.proj_dim, Hq), "vision_proj.bias": (cfg.proj_dim,), "text_proj.weight": (cfg.proj_dim, Hq),
         "text_proj.bias": (cfg.proj_dim,), "itm_head.weight": (2, Hq), "itm_head.bias": (2,), "temp": (),
         "Qformer.cls.predictions.bias": (cfg.vocab_size,), "Qformer.cls.predictions.transform.dense.weight": (Hq, Hq),
         "Qformer.cls.predictions.transform.dense.bias": (Hq,), "Qformer.cls.predictions.transform.LayerNorm.weight": (Hq,),
         "Qformer.cls.predictions.transform.LayerNorm.bias": (Hq,), "Qformer.cls.predictions.decoder.weight": (cfg.vocab_size, Hq),
         "Qformer.cls.predictions.decoder.bias": (cfg.vocab_size,)}
    for i in range(cfg.q_layers):
        o = f"Qformer.bert.encoder.layer.{i}."
        atts = [("attention", Hq)] + ([("crossattention", H)] if i % cfg.cross_attention_frequency == 0 else [])
        for att, kv in atts:
            q.update({o + f"{att}.self.query.weight": (Hq, Hq), o + f"{att}.self.query.bias": (Hq,),
                      o + f"{att}.self.key.weight": (Hq, kv), o + f"{att}.self.key.bias": (Hq,),
                      o + f"{att}.self.value.weight": (Hq, kv), o + f"{att}.self.value.bias": (Hq,),
                      o + f"{att}.output.dense.weight": (Hq, Hq), o + f"{att}.output.dense.bias": (Hq,),
                      o + f"{att}.output.LayerNorm.weight": (Hq,), o + f"{att}.output.LayerNorm.bias": (Hq,)})
        for a, b in (("intermediate", "output"), ("intermediate_query", "output_query")):
            q.update({o + f"{a}.dense.weight": (Iq, Hq), o + f"{a}.dense.bias": (Iq,), o + f"{b}.dense.weight": (Hq, Iq),
                      o + f"{b}.dense.bias": (Hq,), o + f"{b}.LayerNorm.weight": (Hq,), o + f"{b}.LayerNorm.bias": (Hq,)})
    if with_vit_in_file:
        q.update({"visual_encoder." + k: v for k, v in vit.items() if not k.startswith(("norm.", f"blocks.{cfg.v_layers}."))})
        return q, None
    return q, vit


@pytest.mark.parametrize("vit_in_file", [False, True])
def test_lavis_artefacts_load_strictly_and_give_the_hf_ports_scores(vit_in_file):
    """blip2itm.py:29-34 loads LAVIS' blip2_pretrained.pth (+ eva_vit_g.pth).  Synthetic files with LAVIS' names and shapes:
    every tensor placed or on the explicit not-used list, every parameter fed; and the SAME tensors fed to transformers'
    Blip2ForImageTextRetrieval under ITS names give the same ITC score -- the two loaders agree on where each tensor goes
    (incl. EVA's [q_bias, 0, v_bias])."""
    cfg = Blip2ITCConfig.tiny()
    g = torch.Generator().manual_seed(11)
    q_spec, vit_spec = _lavis_spec(cfg, vit_in_file)

    def rnd(spec):
        out = {}
        for k, shape in spec.items():
            t = torch.randn(shape, generator=g) * 0.08
            out[k] = t + 1.0 if (("norm" in k.lower() or "ln_vision" in k) and k.endswith("weight")) else t
        return out

    sd = rnd(q_spec)
    vit = rnd(vit_spec) if vit_spec is not None else None
    ours = Blip2ITCModel(cfg).eval()
    ours.load_lavis_state_dict({"module." + k: v for k, v in sd.items()}, vit)
    # the same tensors under transformers' names
    V = (lambda k: vit[k]) if vit is not None else (lambda k: sd["visual_encoder." + k])
    hf = _hf_model(cfg)
    t = {"vision_model.embeddings.class_embedding": V("cls_token"), "vision_model.embeddings.position_embedding": V("pos_embed"),
         "vision_model.embeddings.patch_embedding.weight": V("patch_embed.proj.weight"),
         "vision_model.embeddings.patch_embedding.bias": V("patch_embed.proj.bias"),
         "vision_model.post_layernorm.weight": sd["ln_vision.weight"], "vision_model.post_layernorm.bias": sd["ln_vision.bias"],
         "query_tokens": sd["query_tokens"], "embeddings.word_embeddings.weight": sd["Qformer.bert.embeddings.word_embeddings.weight"],
         "embeddings.position_embeddings.weight": sd["Qformer.bert.embeddings.position_embeddings.weight"],
         "qformer.layernorm.weight": sd["Qformer.bert.embeddings.LayerNorm.weight"],
         "qformer.layernorm.bias": sd["Qformer.bert.embeddings.LayerNorm.bias"],
         "vision_projection.weight": sd["vision_proj.weight"], "vision_projection.bias": sd["vision_proj.bias"],
         "text_projection.weight": sd["text_proj.weight"], "text_projection.bias": sd["text_proj.bias"]}
    for i in range(cfg.v_layers):
        o, d = f"blocks.{i}.", f"vision_model.encoder.layers.{i}."
        for a, b in (("norm1", "layer_norm1"), ("norm2", "layer_norm2"), ("attn.proj", "self_attn.projection"),
                     ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2")):
            t[d + b + ".weight"], t[d + b + ".bias"] = V(o + a + ".weight"), V(o + a + ".bias")
        t[d + "self_attn.qkv.weight"] = V(o + "attn.qkv.weight")
        t[d + "self_attn.qkv.bias"] = torch.cat([V(o + "attn.q_bias"), torch.zeros(cfg.v_hidden), V(o + "attn.v_bias")])
    for k, v in sd.items():
        if k.startswith("Qformer.bert.encoder."):
            t["qformer." + k[len("Qformer.bert."):].replace(".self.", ".attention.")] = v
    missing = [k for k in hf.state_dict() if k not in t and not k.startswith(("itm_head", "temp"))]
    res = hf.load_state_dict(t, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys[:5]
    assert all(k.startswith(("itm_head", "temp")) or "position_ids" in k for k in res.missing_keys), res.missing_keys[:5]
    pix = torch.randn(2, 3, cfg.image_size, cfg.image_size, generator=g)
    ids = torch.randint(0, cfg.vocab_size, (1, 7), generator=g)
    with torch.inference_mode():
        want = hf(pixel_values=pix, input_ids=ids.expand(2, -1), attention_mask=torch.ones(2, 7, dtype=torch.long),
                  use_image_text_matching_head=False).logits_per_image
        got = ours.itc_reference_head(ours.query_features(ours.vision_tokens(pix)), ours.text_feature(ids))
    assert torch.allclose(got, want[:, 0] if want.dim() == 2 else want.flatten()[:2], atol=2e-5, rtol=0), (got, want)
    # strictness
    extra = dict(sd); extra["Qformer.bert.encoder.layer.0.crossattention.self.gate"] = torch.zeros(1)
    with pytest.raises(KeyError, match="no place"):
        Blip2ITCModel(cfg).load_lavis_state_dict(extra, vit)
    lacking = dict(sd); lacking.pop("vision_proj.bias")
    with pytest.raises(KeyError, match="vision_proj.bias"):
        Blip2ITCModel(cfg).load_lavis_state_dict(lacking, vit)
    if not vit_in_file:
        with pytest.raises(KeyError, match="eva_vit_g"):
            Blip2ITCModel(cfg).load_lavis_state_dict(sd, None)           # the frozen tower is not in blip2_pretrained.pth
        both = dict(sd); both["visual_encoder.cls_token"] = vit["cls_token"]
        with pytest.raises(KeyError, match="twice"):
            Blip2ITCModel(cfg).load_lavis_state_dict(both, vit)


def test_tuned_gemm_table_is_well_formed_and_optional():
    """vlfm_amd/tunableop_results.csv (tools/tune_gemms.py): TunableOp's validators + one solution per library-GEMM shape of the
    BLIP-2 forward at the benchmark's batch sizes; loading it is optional (no GPU here: use_tuned_gemms() says no and does not raise)."""
    import os

    import vlfm_amd
    from vlfm_amd.vlm import ops

    path = os.path.join(os.path.dirname(vlfm_amd.__file__), "tunableop_results.csv")
    rows = [line.strip().split(",") for line in open(path) if line.strip()]
    validators = {r[1] for r in rows if r[0] == "Validator"}
    assert {"PT_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"} <= validators
    shapes = {r[1] for r in rows if r[0] != "Validator"}
    tokens = 256 * 257                                       # the headline: 256 images x 257 tokens
    for n, k in ((4224, 1408), (1408, 1408), (1408, 6144)):  # qkv, projection, fc2 of ViT-g
        assert f"tn_{n}_{tokens}_{k}_ld_{k}_{k}_{n}" in shapes
    assert all(len(r) == 4 and float(r[3]) > 0 for r in rows if r[0] != "Validator")
    if not torch.cuda.is_available():
        assert ops.use_tuned_gemms() is False
