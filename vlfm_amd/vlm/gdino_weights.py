"""The reference's own GroundingDINO artefacts onto the graph this package runs (reference:
/root/reference/vlfm/vlm/grounding_dino.py:18-19,33 -- ``GroundingDINO_SwinT_OGC.py`` + ``data/groundingdino_swint_ogc.pth``
handed to the un-vendored groundingdino package's ``load_model`` [ext]).

The network here is transformers' ``GroundingDinoForObjectDetection`` (same architecture, other parameter names).  This file
is the key map between the two, written against the module definitions of groundingdino (models/GroundingDINO/
groundingdino.py, transformer.py, fuse_modules.py, backbone/swin_transformer.py [ext]) and the parameter names of the
installed transformers:

* `convert_groundingdino_state_dict` -- strict both ways: every tensor of the ``.pth`` is consumed (or is on the short,
  explicit list of tensors the target does not have: BERT's pooler and position-id buffer, Swin's relative-position index
  buffers), every parameter of the target is fed (except ``swin.layernorm``, which transformers' SwinModel owns but the
  detector's feature maps never pass through), all shapes equal.  Fused projections of the original (Swin ``qkv``,
  ``nn.MultiheadAttention.in_proj_*``) are split in their storage order q | k | v.
* `config_from_groundingdino_py` -- reads the reference's hyper-parameter file (a Python file of literal assignments) WITHOUT
  executing it and returns the matching ``GroundingDinoConfig``; anything the transformers graph cannot express raises.
* `original_state_dict_spec` -- names and shapes of the original checkpoint, derived from the original module definitions
  (independently of the key map: tests feed a synthetic state dict built from it through the converter).
"""
from __future__ import annotations

import ast
import re
from typing import Dict, List, Optional, Tuple

import torch

Source = Tuple[str, Optional[int]]   # (key in the original checkpoint, which third of a fused q|k|v tensor or None)

_SWIN = "model.backbone.conv_encoder.model.swin."
_NOT_FED = ("model.backbone.conv_encoder.model.swin.layernorm.weight", "model.backbone.conv_encoder.model.swin.layernorm.bias")


def _ignorable(orig_key: str) -> bool:
    return (orig_key.endswith("attn.relative_position_index") or orig_key.endswith("attn_mask")
            or orig_key == "bert.embeddings.position_ids" or orig_key.startswith("bert.pooler."))


def source_of(hf_key: str) -> Optional[Source]:
    """Where a parameter of transformers' GroundingDinoForObjectDetection comes from in groundingdino's checkpoint."""
    k = hf_key
    qkv = {"q_proj": 0, "k_proj": 1, "v_proj": 2, "query": 0, "key": 1, "value": 2}
    if k in _NOT_FED:
        return None
    if k.startswith(_SWIN):
        r = k[len(_SWIN):]
        m = re.fullmatch(r"embeddings\.patch_embeddings\.projection\.(weight|bias)", r)
        if m:
            return f"backbone.0.patch_embed.proj.{m.group(1)}", None
        m = re.fullmatch(r"embeddings\.norm\.(weight|bias)", r)
        if m:
            return f"backbone.0.patch_embed.norm.{m.group(1)}", None
        m = re.fullmatch(r"encoder\.layers\.(\d+)\.blocks\.(\d+)\.(.+)\.(weight|bias|relative_position_bias_table)", r)
        if m:
            i, j, what, leaf = m.groups()
            o = f"backbone.0.layers.{i}.blocks.{j}."
            if what in ("attention.q_proj", "attention.k_proj", "attention.v_proj"):
                return o + f"attn.qkv.{leaf}", qkv[what.split(".")[1]]
            if what == "attention.relative_position_bias":
                return o + "attn.relative_position_bias_table", None
            table = {"attention.o_proj": "attn.proj", "layernorm_before": "norm1", "layernorm_after": "norm2",
                     "mlp.fc1": "mlp.fc1", "mlp.fc2": "mlp.fc2"}
            return o + f"{table[what]}.{leaf}", None
        m = re.fullmatch(r"encoder\.layers\.(\d+)\.downsample\.(reduction|norm)\.(weight|bias)", r)
        if m:
            return f"backbone.0.layers.{m.group(1)}.downsample.{m.group(2)}.{m.group(3)}", None
        raise KeyError(f"no rule for {hf_key}")
    m = re.fullmatch(r"model\.backbone\.conv_encoder\.model\.hidden_states_norms\.stage(\d)\.(weight|bias)", k)
    if m:   # out_indices (1, 2, 3) of the original = stages 2, 3, 4
        return f"backbone.0.norm{int(m.group(1)) - 1}.{m.group(2)}", None
    m = re.fullmatch(r"model\.input_proj_vision\.(\d+)\.(\d)\.(weight|bias)", k)
    if m:
        return f"input_proj.{m.group(1)}.{m.group(2)}.{m.group(3)}", None
    if k.startswith("model.text_backbone."):
        return "bert." + k[len("model.text_backbone."):], None
    m = re.fullmatch(r"model\.text_projection\.(weight|bias)", k)
    if m:
        return f"feat_map.{m.group(1)}", None
    simple = {"model.level_embed": "transformer.level_embed",
              "model.query_position_embeddings.weight": "transformer.tgt_embed.weight"}
    if k in simple:
        return simple[k], None
    m = re.fullmatch(r"model\.(enc_output|enc_output_norm)\.(weight|bias)", k)
    if m:
        return f"transformer.{m.group(1)}.{m.group(2)}", None
    m = re.fullmatch(r"model\.encoder_output_bbox_embed\.layers\.(\d)\.(weight|bias)", k)
    if m:
        return f"transformer.enc_out_bbox_embed.layers.{m.group(1)}.{m.group(2)}", None
    m = re.fullmatch(r"model\.encoder\.layers\.(\d+)\.(deformable_layer|text_enhancer_layer|fusion_layer)\.(.+)", k)
    if m:
        i, part, r = m.groups()
        if part == "deformable_layer":
            o = f"transformer.encoder.layers.{i}."
            table = {"self_attn_layer_norm": "norm1", "fc1": "linear1", "fc2": "linear2", "final_layer_norm": "norm2"}
            head, leaf = r.rsplit(".", 1)
            return o + (f"{table[head]}.{leaf}" if head in table else r), None     # self_attn.* keep their names
        if part == "text_enhancer_layer":
            o = f"transformer.encoder.text_layers.{i}."
            head, leaf = r.rsplit(".", 1)
            if head in ("self_attn.query", "self_attn.key", "self_attn.value"):
                return o + f"self_attn.in_proj_{leaf}", qkv[head.split(".")[1]]
            table = {"self_attn.out_proj": "self_attn.out_proj", "fc1": "linear1", "fc2": "linear2",
                     "layer_norm_before": "norm1", "layer_norm_after": "norm2"}
            return o + f"{table[head]}.{leaf}", None
        o = f"transformer.encoder.fusion_layers.{i}."
        if r in ("vision_param", "text_param"):
            return o + {"vision_param": "gamma_v", "text_param": "gamma_l"}[r], None
        head, leaf = r.rsplit(".", 1)
        table = {"layer_norm_vision": "layer_norm_v", "layer_norm_text": "layer_norm_l", "attn.vision_proj": "attn.v_proj",
                 "attn.text_proj": "attn.l_proj", "attn.values_vision_proj": "attn.values_v_proj",
                 "attn.values_text_proj": "attn.values_l_proj", "attn.out_vision_proj": "attn.out_v_proj",
                 "attn.out_text_proj": "attn.out_l_proj"}
        return o + f"{table[head]}.{leaf}", None
    m = re.fullmatch(r"model\.decoder\.layers\.(\d+)\.(.+)\.(weight|bias)", k)
    if m:
        i, head, leaf = m.groups()
        o = f"transformer.decoder.layers.{i}."
        for mine, theirs in (("self_attn", "self_attn"), ("encoder_attn_text", "ca_text")):
            if head in (f"{mine}.query", f"{mine}.key", f"{mine}.value"):
                return o + f"{theirs}.in_proj_{leaf}", qkv[head.split(".")[1]]
            if head == f"{mine}.out_proj":
                return o + f"{theirs}.out_proj.{leaf}", None
        if head.startswith("encoder_attn."):
            return o + "cross_attn." + head[len("encoder_attn."):] + f".{leaf}", None
        table = {"self_attn_layer_norm": "norm2", "encoder_attn_text_layer_norm": "catext_norm",
                 "encoder_attn_layer_norm": "norm1", "fc1": "linear1", "fc2": "linear2", "final_layer_norm": "norm3"}
        return o + f"{table[head]}.{leaf}", None
    m = re.fullmatch(r"model\.decoder\.layer_norm\.(weight|bias)", k)
    if m:
        return f"transformer.decoder.norm.{m.group(1)}", None
    m = re.fullmatch(r"model\.decoder\.reference_points_head\.layers\.(\d)\.(weight|bias)", k)
    if m:
        return f"transformer.decoder.ref_point_head.layers.{m.group(1)}.{m.group(2)}", None
    m = re.fullmatch(r"model\.decoder\.bbox_embed\.(\d+)\.layers\.(\d)\.(weight|bias)", k)
    if m:
        return f"transformer.decoder.bbox_embed.{m.group(1)}.layers.{m.group(2)}.{m.group(3)}", None
    m = re.fullmatch(r"bbox_embed\.(\d+)\.layers\.(\d)\.(weight|bias)", k)
    if m:
        return k, None
    raise KeyError(f"no rule for {hf_key}")


def convert_groundingdino_state_dict(orig: Dict[str, torch.Tensor], hf_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``orig``: the ``model`` entry of groundingdino_swint_ogc.pth (a leading ``module.`` is dropped, as groundingdino's
    clean_state_dict does [ext]); ``hf_state``: ``state_dict()`` of the target (names and shapes).  Returns the tensors
    under the target's names; raises on any tensor that is missing, left over or of the wrong shape."""
    orig = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in orig.items()}
    out: Dict[str, torch.Tensor] = {}
    used = set()
    for k, t in hf_state.items():
        src = source_of(k)
        if src is None:
            continue
        okey, part = src
        if okey not in orig:
            raise KeyError(f"{k} needs {okey}, which the checkpoint does not have")
        v = orig[okey]
        used.add(okey)
        if part is not None:
            if v.shape[0] % 3:
                raise ValueError(f"{okey}: fused q|k|v tensor with {v.shape[0]} rows")
            n = v.shape[0] // 3
            v = v[part * n:(part + 1) * n]
        if tuple(v.shape) != tuple(t.shape):
            raise ValueError(f"{k}: {tuple(t.shape)} in the graph, {tuple(v.shape)} from {okey}")
        out[k] = v
    left = sorted(k for k in orig if k not in used and not _ignorable(k))
    if left:
        raise KeyError(f"{len(left)} tensors of the checkpoint have no place in the graph, e.g. {left[:4]}")
    return out


# ------------------------------------------------------------------------------------------------ the reference's config file
def _literal_assignments(path: str) -> Dict[str, object]:
    tree = ast.parse(open(path).read(), filename=path)
    out: Dict[str, object] = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            try:
                out[node.targets[0].id] = ast.literal_eval(node.value)
            except ValueError as exc:
                raise ValueError(f"{path}: `{node.targets[0].id}` is not a literal") from exc
        elif not isinstance(node, (ast.Expr, ast.Import, ast.ImportFrom)):
            raise ValueError(f"{path}: only literal assignments are read from a GroundingDINO config file")
    return out


def config_from_groundingdino_py(path: str):
    """GroundingDINO_SwinT_OGC.py (groundingdino/config [ext]) -> GroundingDinoConfig.  The file is parsed, never executed."""
    from transformers import GroundingDinoConfig

    c = _literal_assignments(path)
    swin = {"swin_T_224_1k": dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7),
            "swin_B_384_22k": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=12)}
    if c.get("backbone") not in swin:
        raise ValueError(f"backbone {c.get('backbone')!r}: only the Swin backbones of the published checkpoints are mapped")
    need = {"two_stage_type": "standard", "position_embedding": "sine", "use_text_enhancer": True, "use_fusion_layer": True,
            "use_text_cross_attention": True, "dec_pred_bbox_embed_share": True, "embed_init_tgt": True,
            "text_encoder_type": "bert-base-uncased", "transformer_activation": "relu", "pre_norm": False,
            "return_interm_indices": [1, 2, 3], "num_patterns": 0, "two_stage_bbox_embed_share": False,
            "two_stage_class_embed_share": False}
    for k, v in need.items():
        if k in c and c[k] != v:
            raise ValueError(f"{path}: {k} = {c[k]!r}; transformers' GroundingDino graph is built for {v!r}")
    if c.get("enc_n_points", 4) != c.get("dec_n_points", 4):
        raise ValueError("different numbers of sampling points in encoder and decoder are not expressible")
    hidden = int(c.get("hidden_dim", 256))
    kw = dict(backbone_config=dict(model_type="swin", out_features=["stage2", "stage3", "stage4"], **swin[c["backbone"]]),
              text_config=dict(model_type="bert"),
              num_queries=int(c.get("num_queries", 900)), d_model=hidden,
              encoder_layers=int(c.get("enc_layers", 6)), decoder_layers=int(c.get("dec_layers", 6)),
              encoder_attention_heads=int(c.get("nheads", 8)), decoder_attention_heads=int(c.get("nheads", 8)),
              encoder_ffn_dim=int(c.get("dim_feedforward", 2048)), decoder_ffn_dim=int(c.get("dim_feedforward", 2048)),
              num_feature_levels=int(c.get("num_feature_levels", 4)),
              encoder_n_points=int(c.get("enc_n_points", 4)), decoder_n_points=int(c.get("dec_n_points", 4)),
              max_text_len=int(c.get("max_text_len", 256)), fusion_droppath=float(c.get("fusion_droppath", 0.1)),
              fusion_dropout=float(c.get("fusion_dropout", 0.0)), dropout=float(c.get("dropout", 0.0)),
              positional_embedding_temperature=int(c.get("pe_temperatureH", 20)), two_stage=True,
              query_dim=int(c.get("query_dim", 4)))
    if c.get("pe_temperatureH", 20) != c.get("pe_temperatureW", 20):
        raise ValueError("pe_temperatureH != pe_temperatureW is not expressible")
    return GroundingDinoConfig(**kw)


# ------------------------------------------------------------------------------------------------ the original's key layout
def original_state_dict_spec(cfg) -> Dict[str, Tuple[int, ...]]:
    """Names -> shapes of groundingdino's ``model.state_dict()`` for the architecture ``cfg`` describes, written down from the
    ORIGINAL module definitions (not from the key map above): SwinTransformer (patch_embed, layers[i].blocks[j] with norm1 /
    attn.{qkv, proj, relative_position_bias_table, relative_position_index} / norm2 / mlp.{fc1, fc2}, layers[i].downsample,
    norm1-3), input_proj (conv + GroupNorm per level), BertModel, feat_map, Transformer (level_embed, tgt_embed, enc_output,
    encoder.{layers, text_layers, fusion_layers}, decoder.{layers, norm, ref_point_head, bbox_embed}, enc_out_bbox_embed),
    bbox_embed."""
    s: Dict[str, Tuple[int, ...]] = {}
    b = cfg.backbone_config
    E, depths, heads, ws = b.embed_dim, list(b.depths), list(b.num_heads), b.window_size
    d, ffn, L = cfg.d_model, cfg.encoder_ffn_dim, cfg.num_feature_levels

    def lin(name, o, i, bias=True):
        s[name + ".weight"] = (o, i)
        if bias:
            s[name + ".bias"] = (o,)

    def norm(name, n):
        s[name + ".weight"], s[name + ".bias"] = (n,), (n,)

    s["backbone.0.patch_embed.proj.weight"], s["backbone.0.patch_embed.proj.bias"] = (E, 3, b.patch_size, b.patch_size), (E,)
    norm("backbone.0.patch_embed.norm", E)
    for i, (n_blocks, h) in enumerate(zip(depths, heads)):
        C = E * 2 ** i
        for j in range(n_blocks):
            o = f"backbone.0.layers.{i}.blocks.{j}."
            norm(o + "norm1", C)
            s[o + "attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, h)
            s[o + "attn.relative_position_index"] = (ws * ws, ws * ws)
            lin(o + "attn.qkv", 3 * C, C)
            lin(o + "attn.proj", C, C)
            norm(o + "norm2", C)
            lin(o + "mlp.fc1", int(C * b.mlp_ratio), C)
            lin(o + "mlp.fc2", C, int(C * b.mlp_ratio))
        if i < len(depths) - 1:
            lin(f"backbone.0.layers.{i}.downsample.reduction", 2 * C, 4 * C, bias=False)
            norm(f"backbone.0.layers.{i}.downsample.norm", 4 * C)
    for k in (1, 2, 3):
        norm(f"backbone.0.norm{k}", E * 2 ** k)
    feats = [E * 2, E * 4, E * 8]
    for lvl in range(L):
        cin, ksz = (feats[lvl], 1) if lvl < 3 else (feats[-1] if lvl == 3 else d, 3)
        s[f"input_proj.{lvl}.0.weight"], s[f"input_proj.{lvl}.0.bias"] = (d, cin, ksz, ksz), (d,)
        norm(f"input_proj.{lvl}.1", d)
    t = cfg.text_config
    H, I = t.hidden_size, t.intermediate_size
    s["bert.embeddings.position_ids"] = (1, t.max_position_embeddings)
    s["bert.embeddings.word_embeddings.weight"] = (t.vocab_size, H)
    s["bert.embeddings.position_embeddings.weight"] = (t.max_position_embeddings, H)
    s["bert.embeddings.token_type_embeddings.weight"] = (t.type_vocab_size, H)
    norm("bert.embeddings.LayerNorm", H)
    for i in range(t.num_hidden_layers):
        o = f"bert.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            lin(o + f"attention.self.{n}", H, H)
        lin(o + "attention.output.dense", H, H)
        norm(o + "attention.output.LayerNorm", H)
        lin(o + "intermediate.dense", I, H)
        lin(o + "output.dense", H, I)
        norm(o + "output.LayerNorm", H)
    lin("bert.pooler.dense", H, H)
    lin("feat_map", d, H)
    s["transformer.level_embed"] = (L, d)
    s["transformer.tgt_embed.weight"] = (cfg.num_queries, d)
    lin("transformer.enc_output", d, d)
    norm("transformer.enc_output_norm", d)
    nh_e, np_e = cfg.encoder_attention_heads, cfg.encoder_n_points

    def deform(o, heads_, points):
        lin(o + "sampling_offsets", heads_ * L * points * 2, d)
        lin(o + "attention_weights", heads_ * L * points, d)
        lin(o + "value_proj", d, d)
        lin(o + "output_proj", d, d)

    def mha(o, dim):
        s[o + "in_proj_weight"], s[o + "in_proj_bias"] = (3 * dim, dim), (3 * dim,)
        lin(o + "out_proj", dim, dim)

    for i in range(cfg.encoder_layers):
        o = f"transformer.encoder.layers.{i}."
        deform(o + "self_attn.", nh_e, np_e)
        norm(o + "norm1", d); lin(o + "linear1", ffn, d); lin(o + "linear2", d, ffn); norm(o + "norm2", d)
        o = f"transformer.encoder.text_layers.{i}."
        mha(o + "self_attn.", d)
        lin(o + "linear1", ffn // 2, d); lin(o + "linear2", d, ffn // 2); norm(o + "norm1", d); norm(o + "norm2", d)
        o = f"transformer.encoder.fusion_layers.{i}."
        norm(o + "layer_norm_v", d); norm(o + "layer_norm_l", d)
        e = ffn // 2   # BiAttentionBlock(embed_dim = dim_feedforward // 2)
        for n in ("v_proj", "l_proj", "values_v_proj", "values_l_proj"):
            lin(o + "attn." + n, e, d)
        lin(o + "attn.out_v_proj", d, e); lin(o + "attn.out_l_proj", d, e)
        s[o + "gamma_v"], s[o + "gamma_l"] = (d,), (d,)
    for i in range(cfg.decoder_layers):
        o = f"transformer.decoder.layers.{i}."
        deform(o + "cross_attn.", cfg.decoder_attention_heads, cfg.decoder_n_points)
        norm(o + "norm1", d)
        mha(o + "ca_text.", d); norm(o + "catext_norm", d)
        mha(o + "self_attn.", d); norm(o + "norm2", d)
        lin(o + "linear1", cfg.decoder_ffn_dim, d); lin(o + "linear2", d, cfg.decoder_ffn_dim); norm(o + "norm3", d)
    norm("transformer.decoder.norm", d)
    lin("transformer.decoder.ref_point_head.layers.0", d, cfg.query_dim // 2 * d)
    lin("transformer.decoder.ref_point_head.layers.1", d, d)

    def mlp3(o):
        lin(o + "layers.0", d, d); lin(o + "layers.1", d, d); lin(o + "layers.2", 4, d)

    for i in range(cfg.decoder_layers):   # one MLP object shared by reference: the state dict lists it under every name
        mlp3(f"bbox_embed.{i}.")
        mlp3(f"transformer.decoder.bbox_embed.{i}.")
    mlp3("transformer.enc_out_bbox_embed.")
    return s
