"""oracle/ref_mobile_sam.py -- TEST INFRASTRUCTURE ONLY (parity oracle).

CPU restatement of what /root/reference/vlfm/vlm/sam.py:40-57 runs through the un-vendored ``mobile_sam`` package
(pyproject.toml:26, git HEAD): ``SamPredictor.set_image`` + ``predict(box=..., multimask_output=False)`` of
``sam_model_registry["vit_t"]`` = TinyViT-5M image encoder + segment-anything's prompt encoder and two-way mask decoder.
PARITY UNPINNED against the real package (not installable here, weights not obtainable); it restates the published
architecture, and it is written in the package's OWN layout -- token tensors [B, L, C] through the encoder, explicit
softmax attention, functional code reading the checkpoint's own key names (``image_encoder.*``, ``prompt_encoder.*``,
``mask_decoder.*``) -- so that it checks BOTH the product's network (vlfm_amd/vlm/sam.py: NCHW encoder modules + the
``transformers`` SAM decoder) AND the checkpoint key map that feeds it.  Nothing under vlfm_amd/ may import this.

``expected_checkpoint_shapes()`` is the key/shape table of ``mobile_sam.pt`` derived from the architecture numbers
(dims 64/128/160/320, depths 2/2/6/2, heads 2/4/5/10, windows 7/7/14/7, 1000-way head, 256-d neck, SAM decoder)."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

DIMS, DEPTHS, HEADS, WINDOWS = (64, 128, 160, 320), (2, 2, 6, 2), (2, 4, 5, 10), (7, 7, 14, 7)


# ------------------------------------------------------------------------------------------------ key / shape table
def expected_checkpoint_shapes() -> Dict[str, Tuple[int, ...]]:
    t: Dict[str, Tuple[int, ...]] = {}

    def conv_bn(prefix, cin, cout, ks, groups=1):
        t[prefix + ".c.weight"] = (cout, cin // groups, ks, ks)
        for n in ("weight", "bias", "running_mean", "running_var"):
            t[prefix + ".bn." + n] = (cout,)
        t[prefix + ".bn.num_batches_tracked"] = ()

    def linear(prefix, cin, cout):
        t[prefix + ".weight"], t[prefix + ".bias"] = (cout, cin), (cout,)

    def norm(prefix, c):
        t[prefix + ".weight"], t[prefix + ".bias"] = (c,), (c,)

    e = "image_encoder."
    conv_bn(e + "patch_embed.seq.0", 3, 32, 3)
    conv_bn(e + "patch_embed.seq.2", 32, 64, 3)
    for b in range(DEPTHS[0]):
        p = f"{e}layers.0.blocks.{b}"
        conv_bn(p + ".conv1", 64, 256, 1)
        conv_bn(p + ".conv2", 256, 256, 3, groups=256)
        conv_bn(p + ".conv3", 256, 64, 1)
    for i in range(3):
        p = f"{e}layers.{i}.downsample"
        conv_bn(p + ".conv1", DIMS[i], DIMS[i + 1], 1)
        conv_bn(p + ".conv2", DIMS[i + 1], DIMS[i + 1], 3, groups=DIMS[i + 1])
        conv_bn(p + ".conv3", DIMS[i + 1], DIMS[i + 1], 1)
    for i in range(1, 4):
        d, w = DIMS[i], WINDOWS[i]
        for b in range(DEPTHS[i]):
            p = f"{e}layers.{i}.blocks.{b}"
            norm(p + ".attn.norm", d)
            linear(p + ".attn.qkv", d, 3 * d)
            linear(p + ".attn.proj", d, d)
            t[p + ".attn.attention_biases"] = (HEADS[i], w * w)   # distinct (|dy|, |dx|) offsets in a w x w window
            norm(p + ".mlp.norm", d)
            linear(p + ".mlp.fc1", d, 4 * d)
            linear(p + ".mlp.fc2", 4 * d, d)
            conv_bn(p + ".local_conv", d, d, 3, groups=d)
    norm(e + "norm_head", 320)
    linear(e + "head", 320, 1000)
    t[e + "neck.0.weight"] = (256, 320, 1, 1)
    norm(e + "neck.1", 256)
    t[e + "neck.2.weight"] = (256, 256, 3, 3)
    norm(e + "neck.3", 256)
    p = "prompt_encoder."
    t[p + "pe_layer.positional_encoding_gaussian_matrix"] = (2, 128)
    for i in range(4):
        t[f"{p}point_embeddings.{i}.weight"] = (1, 256)
    t[p + "not_a_point_embed.weight"] = (1, 256)
    t[p + "no_mask_embed.weight"] = (1, 256)
    t[p + "mask_downscaling.0.weight"], t[p + "mask_downscaling.0.bias"] = (4, 1, 2, 2), (4,)
    norm(p + "mask_downscaling.1", 4)
    t[p + "mask_downscaling.3.weight"], t[p + "mask_downscaling.3.bias"] = (16, 4, 2, 2), (16,)
    norm(p + "mask_downscaling.4", 16)
    t[p + "mask_downscaling.6.weight"], t[p + "mask_downscaling.6.bias"] = (256, 16, 1, 1), (256,)
    m = "mask_decoder."

    def attn(prefix, internal):
        for n in ("q_proj", "k_proj", "v_proj"):
            linear(f"{prefix}.{n}", 256, internal)
        linear(prefix + ".out_proj", internal, 256)

    for layer in range(2):
        q = f"{m}transformer.layers.{layer}"
        attn(q + ".self_attn", 256)
        attn(q + ".cross_attn_token_to_image", 128)
        attn(q + ".cross_attn_image_to_token", 128)
        linear(q + ".mlp.lin1", 256, 2048)
        linear(q + ".mlp.lin2", 2048, 256)
        for k in range(1, 5):
            norm(f"{q}.norm{k}", 256)
    attn(m + "transformer.final_attn_token_to_image", 128)
    norm(m + "transformer.norm_final_attn", 256)
    t[m + "iou_token.weight"], t[m + "mask_tokens.weight"] = (1, 256), (4, 256)
    t[m + "output_upscaling.0.weight"], t[m + "output_upscaling.0.bias"] = (256, 64, 2, 2), (64,)
    norm(m + "output_upscaling.1", 64)
    t[m + "output_upscaling.3.weight"], t[m + "output_upscaling.3.bias"] = (64, 32, 2, 2), (32,)
    for i in range(4):
        q = f"{m}output_hypernetworks_mlps.{i}.layers"
        linear(q + ".0", 256, 256)
        linear(q + ".1", 256, 256)
        linear(q + ".2", 256, 32)
    q = m + "iou_prediction_head.layers"
    linear(q + ".0", 256, 256)
    linear(q + ".1", 256, 256)
    linear(q + ".2", 256, 4)
    return t


def synthetic_checkpoint(seed: int = 0) -> Dict[str, torch.Tensor]:
    """A state dict with mobile_sam.pt's keys and shapes, filled with seeded values at initialisation-like scales (so that
    the forward pass stays well conditioned); every tensor is distinct."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in expected_checkpoint_shapes().items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(7, dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(shape, generator=g) * 0.05
        elif k.endswith(".bn.weight") or ("norm" in k and k.endswith("weight")) or k.endswith("neck.1.weight") or \
                k.endswith("neck.3.weight") or k.endswith("output_upscaling.1.weight") or k.endswith("mask_downscaling.1.weight") \
                or k.endswith("mask_downscaling.4.weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif k.endswith("positional_encoding_gaussian_matrix"):
            sd[k] = torch.randn(shape, generator=g)
        elif k.endswith("bias") or k.endswith("attention_biases"):
            sd[k] = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = max(1, int(torch.tensor(shape[1:]).prod().item())) if len(shape) > 1 else 1
            sd[k] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return sd


# ------------------------------------------------------------------------------------------------ TinyViT (token layout)
def _conv_bn(sd, p, x, stride=1, pad=0, groups=1):
    x = F.conv2d(x, sd[p + ".c.weight"], None, stride, pad, 1, groups)
    return F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"], sd[p + ".bn.bias"],
                        False, 0.0, 1e-5)


def _patch_merging(sd, p, x, res, out_dim):
    """tokens [B, L, C] (or a map [B, C, H, W] from the conv stage) -> tokens [B, L', out_dim]."""
    if x.dim() == 3:
        b = x.shape[0]
        x = x.view(b, res[0], res[1], -1).permute(0, 3, 1, 2)
    x = F.gelu(_conv_bn(sd, p + ".conv1", x))
    stride = 1 if out_dim in (320, 448, 576) else 2
    x = F.gelu(_conv_bn(sd, p + ".conv2", x, stride, 1, out_dim))
    x = _conv_bn(sd, p + ".conv3", x)
    return x.flatten(2).transpose(1, 2)


def _bias_index(window: int) -> torch.Tensor:
    pts = [(i, j) for i in range(window) for j in range(window)]
    offsets: Dict[Tuple[int, int], int] = {}
    idxs = []
    for p1 in pts:
        for p2 in pts:
            off = (abs(p1[0] - p2[0]), abs(p1[1] - p2[1]))
            if off not in offsets:
                offsets[off] = len(offsets)
            idxs.append(offsets[off])
    return torch.tensor(idxs).view(len(pts), len(pts))


def _window_attention(sd, p, x, heads, window):
    b, n, c = x.shape
    kd = c // heads
    x = F.layer_norm(x, (c,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]).view(b, n, heads, -1)
    q, k, v = qkv.split([kd, kd, kd], dim=3)
    q, k, v = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)
    attn = (q @ k.transpose(-2, -1)) * kd ** -0.5 + sd[p + ".attention_biases"][:, _bias_index(window)]
    x = (attn.softmax(dim=-1) @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def _tinyvit_block(sd, p, x, res, heads, window):
    h, w = res
    b, l, c = x.shape
    res_x = x
    if h == window and w == window:
        x = _window_attention(sd, p + ".attn", x, heads, window)
    else:
        x = x.view(b, h, w, c)
        pad_b, pad_r = (window - h % window) % window, (window - w % window) % window
        if pad_b or pad_r:
            x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        ph, pw = h + pad_b, w + pad_r
        nh, nw = ph // window, pw // window
        x = x.view(b, nh, window, nw, window, c).transpose(2, 3).reshape(b * nh * nw, window * window, c)
        x = _window_attention(sd, p + ".attn", x, heads, window)
        x = x.view(b, nh, nw, window, window, c).transpose(2, 3).reshape(b, ph, pw, c)
        if pad_b or pad_r:
            x = x[:, :h, :w].contiguous()
        x = x.view(b, l, c)
    x = res_x + x
    x = x.transpose(1, 2).reshape(b, c, h, w)
    x = _conv_bn(sd, p + ".local_conv", x, 1, 1, c)
    x = x.view(b, c, l).transpose(1, 2)
    y = F.layer_norm(x, (c,), sd[p + ".mlp.norm.weight"], sd[p + ".mlp.norm.bias"], 1e-5)
    y = F.linear(F.gelu(F.linear(y, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])), sd[p + ".mlp.fc2.weight"],
                 sd[p + ".mlp.fc2.bias"])
    return x + y


def _layer_norm_2d(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def image_encoder(sd, x):
    """[B, 3, 1024, 1024] normalised, padded image -> [B, 256, 64, 64] embedding."""
    e = "image_encoder."
    x = _conv_bn(sd, e + "patch_embed.seq.0", x, 2, 1)
    x = _conv_bn(sd, e + "patch_embed.seq.2", F.gelu(x), 2, 1)                      # 256 x 256, 64 channels
    for b in range(DEPTHS[0]):                                                       # MBConv stage
        p = f"{e}layers.0.blocks.{b}"
        y = F.gelu(_conv_bn(sd, p + ".conv1", x))
        y = F.gelu(_conv_bn(sd, p + ".conv2", y, 1, 1, 256))
        x = F.gelu(_conv_bn(sd, p + ".conv3", y) + x)
    res = (256, 256)
    x = _patch_merging(sd, e + "layers.0.downsample", x, res, DIMS[1])
    res = (128, 128)
    for i in range(1, 4):
        for b in range(DEPTHS[i]):
            x = _tinyvit_block(sd, f"{e}layers.{i}.blocks.{b}", x, res, HEADS[i], WINDOWS[i])
        if i < 3:
            x = _patch_merging(sd, f"{e}layers.{i}.downsample", x, res, DIMS[i + 1])
            if DIMS[i + 1] not in (320, 448, 576):
                res = (res[0] // 2, res[1] // 2)
    b, _, c = x.shape
    x = x.view(b, 64, 64, c).permute(0, 3, 1, 2)
    x = F.conv2d(x, sd[e + "neck.0.weight"])
    x = _layer_norm_2d(x, sd[e + "neck.1.weight"], sd[e + "neck.1.bias"])
    x = F.conv2d(x, sd[e + "neck.2.weight"], padding=1)
    return _layer_norm_2d(x, sd[e + "neck.3.weight"], sd[e + "neck.3.bias"])


# ------------------------------------------------------------------------------------------------ prompt encoder
def _pe_encoding(sd, coords):
    coords = 2 * coords - 1
    coords = coords @ sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    coords = 2 * math.pi * coords
    return torch.cat([torch.sin(coords), torch.cos(coords)], dim=-1)


def dense_pe(sd, size=(64, 64)):
    h, w = size
    grid = torch.ones((h, w), dtype=torch.float32)
    y = (grid.cumsum(dim=0) - 0.5) / h
    x = (grid.cumsum(dim=1) - 0.5) / w
    return _pe_encoding(sd, torch.stack([x, y], dim=-1)).permute(2, 0, 1).unsqueeze(0)


def embed_boxes(sd, boxes, input_size=(1024, 1024)):
    """boxes [B, 4] xyxy in the 1024-frame -> sparse prompt embeddings [B, 2, 256]."""
    coords = (boxes + 0.5).reshape(-1, 2, 2).clone()
    coords[:, :, 0] = coords[:, :, 0] / input_size[1]
    coords[:, :, 1] = coords[:, :, 1] / input_size[0]
    corner = _pe_encoding(sd, coords.to(torch.float32))
    corner[:, 0, :] += sd["prompt_encoder.point_embeddings.2.weight"][0]
    corner[:, 1, :] += sd["prompt_encoder.point_embeddings.3.weight"][0]
    return corner


# ------------------------------------------------------------------------------------------------ mask decoder
def _attention(sd, p, q, k, v, heads=8):
    q = F.linear(q, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"])
    k = F.linear(k, sd[p + ".k_proj.weight"], sd[p + ".k_proj.bias"])
    v = F.linear(v, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"])

    def split(t):
        b, n, c = t.shape
        return t.reshape(b, n, heads, c // heads).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    attn = torch.softmax(q @ k.permute(0, 1, 3, 2) / math.sqrt(q.shape[-1]), dim=-1)
    out = (attn @ v).transpose(1, 2)
    out = out.reshape(out.shape[0], out.shape[1], -1)
    return F.linear(out, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def two_way_transformer(sd, image_embedding, image_pe, point_embedding):
    m = "mask_decoder.transformer"
    keys = image_embedding.flatten(2).permute(0, 2, 1)
    key_pe = image_pe.flatten(2).permute(0, 2, 1)
    queries, query_pe = point_embedding, point_embedding
    for i in range(2):
        p = f"{m}.layers.{i}"
        if i == 0:
            queries = _attention(sd, p + ".self_attn", queries, queries, queries)
        else:
            q = queries + query_pe
            queries = queries + _attention(sd, p + ".self_attn", q, q, queries)
        queries = _ln(sd, p + ".norm1", queries)
        q, k = queries + query_pe, keys + key_pe
        queries = _ln(sd, p + ".norm2", queries + _attention(sd, p + ".cross_attn_token_to_image", q, k, keys))
        mlp = F.linear(F.relu(F.linear(queries, sd[p + ".mlp.lin1.weight"], sd[p + ".mlp.lin1.bias"])),
                       sd[p + ".mlp.lin2.weight"], sd[p + ".mlp.lin2.bias"])
        queries = _ln(sd, p + ".norm3", queries + mlp)
        q, k = queries + query_pe, keys + key_pe
        keys = _ln(sd, p + ".norm4", keys + _attention(sd, p + ".cross_attn_image_to_token", k, q, queries))
    q, k = queries + query_pe, keys + key_pe
    queries = _ln(sd, m + ".norm_final_attn", queries + _attention(sd, m + ".final_attn_token_to_image", q, k, keys))
    return queries, keys


def _mlp3(sd, p, x):
    x = F.relu(F.linear(x, sd[p + ".layers.0.weight"], sd[p + ".layers.0.bias"]))
    x = F.relu(F.linear(x, sd[p + ".layers.1.weight"], sd[p + ".layers.1.bias"]))
    return F.linear(x, sd[p + ".layers.2.weight"], sd[p + ".layers.2.bias"])


def mask_decoder(sd, image_embeddings, sparse):
    """image_embeddings [1, 256, 64, 64], sparse [K, 2, 256] (K boxes) -> (low-res masks [K, 4, 256, 256], iou [K, 4])."""
    m = "mask_decoder."
    k = sparse.shape[0]
    out_tokens = torch.cat([sd[m + "iou_token.weight"], sd[m + "mask_tokens.weight"]], dim=0)
    tokens = torch.cat([out_tokens.unsqueeze(0).expand(k, -1, -1), sparse], dim=1)
    dense = sd["prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(k, -1, 64, 64)
    src = torch.repeat_interleave(image_embeddings, k, dim=0) + dense
    pos = torch.repeat_interleave(dense_pe(sd), k, dim=0)
    b, c, h, w = src.shape
    hs, src = two_way_transformer(sd, src, pos, tokens)
    iou_token_out, mask_tokens_out = hs[:, 0, :], hs[:, 1:5, :]
    src = src.transpose(1, 2).view(b, c, h, w)
    up = F.conv_transpose2d(src, sd[m + "output_upscaling.0.weight"], sd[m + "output_upscaling.0.bias"], stride=2)
    up = F.gelu(_layer_norm_2d(up, sd[m + "output_upscaling.1.weight"], sd[m + "output_upscaling.1.bias"]))
    up = F.gelu(F.conv_transpose2d(up, sd[m + "output_upscaling.3.weight"], sd[m + "output_upscaling.3.bias"], stride=2))
    hyper = torch.stack([_mlp3(sd, f"{m}output_hypernetworks_mlps.{i}", mask_tokens_out[:, i, :]) for i in range(4)], dim=1)
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)
    return masks, _mlp3(sd, m + "iou_prediction_head", iou_token_out)


def predict_low_res(sd, pixel_values, boxes_1024):
    """SamPredictor.predict(box=..., multimask_output=False) up to the low-resolution logits: pixel_values [1,3,1024,1024]
    (already resized / normalised / padded), boxes [K,4] xyxy in the resized frame -> [K, 256, 256] logits of mask 0."""
    with torch.no_grad():
        emb = image_encoder(sd, pixel_values)
        masks, _ = mask_decoder(sd, emb, embed_boxes(sd, boxes_1024))
        return masks[:, 0]
