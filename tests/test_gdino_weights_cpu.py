"""The reference's own GroundingDINO artefacts (/root/reference/vlfm/vlm/grounding_dino.py:18-19,33:
GroundingDINO_SwinT_OGC.py + groundingdino_swint_ogc.pth, read by the un-vendored groundingdino package [ext]) onto
transformers' GroundingDinoForObjectDetection: the key map is strict both ways on a synthetic state dict with the original's
names and shapes (written down from the original module definitions, independently of the map), fused q|k|v tensors are split
in storage order, and the hyper-parameter file is parsed without being executed."""
import pytest
import torch

from vlfm_amd.vlm import gdino_weights as gw

OGC = '''
batch_size = 1
modelname = "groundingdino"
backbone = "swin_T_224_1k"
position_embedding = "sine"
pe_temperatureH = 20
pe_temperatureW = 20
return_interm_indices = [1, 2, 3]
backbone_freeze_keywords = None
enc_layers = 6
dec_layers = 6
pre_norm = False
dim_feedforward = 2048
hidden_dim = 256
dropout = 0.0
nheads = 8
num_queries = 900
query_dim = 4
num_patterns = 0
num_feature_levels = 4
enc_n_points = 4
dec_n_points = 4
two_stage_type = "standard"
two_stage_bbox_embed_share = False
two_stage_class_embed_share = False
transformer_activation = "relu"
dec_pred_bbox_embed_share = True
dn_box_noise_scale = 1.0
dn_label_noise_ratio = 0.5
dn_label_coef = 1.0
dn_bbox_coef = 1.0
embed_init_tgt = True
dn_labelbook_size = 2000
max_text_len = 256
text_encoder_type = "bert-base-uncased"
use_text_enhancer = True
use_fusion_layer = True
use_checkpoint = True
use_transformer_ckpt = True
use_text_cross_attention = True
text_dropout = 0.0
fusion_dropout = 0.0
fusion_droppath = 0.1
sub_sentence_present = True
'''   # groundingdino/config/GroundingDINO_SwinT_OGC.py [ext], as published


def _tiny():
    from transformers import GroundingDinoConfig

    return GroundingDinoConfig(num_queries=10, d_model=32, encoder_layers=2, decoder_layers=2, encoder_ffn_dim=64,
                               decoder_ffn_dim=64, encoder_attention_heads=2, decoder_attention_heads=2,
                               backbone_config={"model_type": "swin", "embed_dim": 16, "depths": [1, 1, 2, 1],
                                                "num_heads": [1, 2, 2, 2], "window_size": 4,
                                                "out_features": ["stage2", "stage3", "stage4"]},
                               text_config={"model_type": "bert", "hidden_size": 32, "num_hidden_layers": 2,
                                            "num_attention_heads": 2, "intermediate_size": 64, "vocab_size": 300,
                                            "max_position_embeddings": 64})


def test_the_swin_t_ogc_layout_converts_strictly_both_ways():
    """Full-size geometry on meta tensors: the 939 tensors of the original layout (174.3 M elements incl. the shared box
    heads listed under every name, BERT's pooler and the index buffers) fill the 1046 parameters of the target but the two
    of ``swin.layernorm``; nothing is left over."""
    from transformers import GroundingDinoConfig, GroundingDinoForObjectDetection

    cfg = GroundingDinoConfig()
    with torch.device("meta"):
        hf = GroundingDinoForObjectDetection(cfg).state_dict()
    spec = gw.original_state_dict_spec(cfg)
    assert "transformer.encoder.fusion_layers.5.attn.values_l_proj.weight" in spec and "bert.pooler.dense.weight" in spec
    assert spec["backbone.0.layers.2.blocks.5.attn.qkv.weight"] == (1152, 384)
    assert spec["transformer.decoder.layers.0.ca_text.in_proj_weight"] == (768, 256) and spec["input_proj.3.0.weight"] == (256, 768, 3, 3)
    orig = {k: torch.empty(s, device="meta") for k, s in spec.items()}
    out = gw.convert_groundingdino_state_dict({"module." + k: v for k, v in orig.items()}, hf)
    assert set(hf) - set(out) == set(gw._NOT_FED)
    assert all(tuple(out[k].shape) == tuple(hf[k].shape) for k in out)


def test_round_trip_on_a_miniature_keeps_every_tensor_and_the_function():
    """A miniature target's own weights, re-packed into the original's layout (q | k | v concatenated, original names), come
    back through the converter bit for bit, and the converted network computes what the source network computes."""
    from transformers import GroundingDinoForObjectDetection

    cfg = _tiny()
    torch.manual_seed(0)
    src = GroundingDinoForObjectDetection(cfg).eval()
    hf = src.state_dict()
    spec = gw.original_state_dict_spec(cfg)
    orig = {}
    for k, t in hf.items():
        s = gw.source_of(k)
        if s is None:
            continue
        okey, part = s
        if part is None:
            orig[okey] = t.clone()
        else:
            orig.setdefault(okey, torch.zeros(spec[okey]))
            n = spec[okey][0] // 3
            orig[okey][part * n:(part + 1) * n] = t
    for k, shape in spec.items():          # what only the original has
        if k not in orig:
            assert gw._ignorable(k), k
            orig[k] = torch.zeros(shape)
    assert {k: tuple(v.shape) for k, v in orig.items()} == spec
    feed = gw.convert_groundingdino_state_dict(orig, hf)
    assert all(torch.equal(feed[k], hf[k]) for k in feed)
    dst = GroundingDinoForObjectDetection(cfg).eval()
    res = dst.load_state_dict(feed, strict=False)
    assert not res.unexpected_keys and set(res.missing_keys) == set(gw._NOT_FED)
    dst.load_state_dict({k: hf[k] for k in gw._NOT_FED}, strict=False)   # (unused by the detector; equalised for the comparison)
    pix = torch.randn(1, 3, 96, 128)
    ids = torch.tensor([[101, 210, 112, 220, 112, 102]])
    with torch.inference_mode():
        a = src(pixel_values=pix, input_ids=ids, attention_mask=torch.ones_like(ids))
        b = dst(pixel_values=pix, input_ids=ids, attention_mask=torch.ones_like(ids))
    assert torch.equal(a.logits, b.logits) and torch.equal(a.pred_boxes, b.pred_boxes)
    # strictness
    extra = dict(orig); extra["transformer.decoder.layers.0.ca_image.weight"] = torch.zeros(1)
    with pytest.raises(KeyError, match="no place"):
        gw.convert_groundingdino_state_dict(extra, hf)
    missing = dict(orig); missing.pop("feat_map.weight")
    with pytest.raises(KeyError, match="feat_map.weight"):
        gw.convert_groundingdino_state_dict(missing, hf)
    wrong = dict(orig); wrong["transformer.level_embed"] = torch.zeros(3, 32)
    with pytest.raises(ValueError, match="level_embed"):
        gw.convert_groundingdino_state_dict(wrong, hf)


def test_the_reference_config_file_is_parsed_not_executed(tmp_path):
    from transformers import GroundingDinoConfig

    p = tmp_path / "GroundingDINO_SwinT_OGC.py"
    p.write_text(OGC)
    got, want = gw.config_from_groundingdino_py(str(p)).to_dict(), GroundingDinoConfig().to_dict()
    differ = {k for k in set(got) | set(want) if got.get(k) != want.get(k)}
    assert differ <= {"dropout"}, differ           # (training-only; the published file says 0.0)
    assert got["backbone_config"]["depths"] == [2, 2, 6, 2] and got["num_queries"] == 900 and got["max_text_len"] == 256
    (tmp_path / "code.py").write_text(OGC + "\nimport os\nnum_queries = os.cpu_count()\n")
    with pytest.raises(ValueError, match="literal"):
        gw.config_from_groundingdino_py(str(tmp_path / "code.py"))
    (tmp_path / "other.py").write_text(OGC.replace("swin_T_224_1k", "resnet50"))
    with pytest.raises(ValueError, match="backbone"):
        gw.config_from_groundingdino_py(str(tmp_path / "other.py"))
    (tmp_path / "onestage.py").write_text(OGC.replace('two_stage_type = "standard"', 'two_stage_type = "no"'))
    with pytest.raises(ValueError, match="two_stage_type"):
        gw.config_from_groundingdino_py(str(tmp_path / "onestage.py"))
    b = gw.config_from_groundingdino_py(_write(tmp_path, OGC.replace("swin_T_224_1k", "swin_B_384_22k")))
    assert b.backbone_config.embed_dim == 128 and list(b.backbone_config.depths) == [2, 2, 18, 2]


def _write(tmp_path, text):
    p = tmp_path / "cfg_b.py"
    p.write_text(text)
    return str(p)
