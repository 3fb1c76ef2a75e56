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
