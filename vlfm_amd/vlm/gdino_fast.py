"""GroundingDINO's forward on MI355X kernels (reference: vlfm/vlm/grounding_dino.py:38-74 runs the network in fp32 through the
un-vendored groundingdino package; here the graph is transformers' ``GroundingDinoForObjectDetection`` -- same parameters, loaded
by vlm/gdino_weights.py -- and this module replaces the forward of its heavy sub-modules, leaving their parameters where the
checkpoint loaders put them).

Where a 64-frame forward went before (profiles/r04_gdino_b64_before.txt: 291 ms of kernels, 368 ms wall): f32 GEMMs 147 ms
(hipBLASLt at 107-147 TFLOP/s: already 0.7-0.9 of the f32 matrix peak -- there is no TF32-like mode on gfx950), MsDeformAttn
39 ms, ~2 150 launches of elementwise / LayerNorm / permute kernels 100 ms, and 77 ms in which the GPU waits for the host.

What this module does about it:
  * every eligible ``nn.Linear`` runs on csrc/gemm_f32.hip's SPLIT form (f32 in / f32 out / f32 accumulate, operands as two f16:
    f32-grade results at 1.5-2x hipBLASLt's f32 rate), with bias / ReLU / exact GELU / residual in the epilogue where the graph
    allows; an operand outside f16's range raises a sticky device flag and ``GroundingDINO.predict_batch`` repeats the batch on
    the library's exact f32 GEMMs;
  * Swin blocks on NHWC rows: LayerNorm + pad + cyclic shift + window partition in one kernel, one fused q|k|v GEMM in the
    attention kernel's per-head layout, window attention with the relative-position bias and the shifted-window mask
    (csrc/sam_ops.hip), output projection, window reverse + roll back + crop + residual add in one in-place kernel, LayerNorm
    rows, fc1 + GELU, fc2 + residual: 8 launches per block instead of ~45, no permute / roll / pad / contiguous copies;
  * encoder deformable layers: value / offsets+weights (one GEMM) / output projection + residual, row LayerNorms, fc1 + ReLU,
    fc2 + residual; the padding mask fill is skipped when the batch has no padding (all frames share one size);
  * fusion layers, re-associated: the caption has ~15 tokens, so the text keys / values are pushed through the vision-side
    projection weights once per image and the 6380 vision tokens only meet [256 x 60] and [60 x 256] matrices -- the three
    256 <-> 1024 GEMMs over all vision tokens (a quarter of the network's matrix work), their head transposes and the 15-column
    batched GEMMs disappear (patch_fusion).
``accelerate(model)`` installs everything and returns a summary; ``model.vlfm_gemm_precision`` switches the GEMM form."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class _State:
    """Per-model switches read by the patched forwards."""

    def __init__(self) -> None:
        self.precision = "split"        # "split" | "library" (F.linear = hipBLASLt exact f32)
        self.no_padding = True          # every frame of a batch has the same size: the encoder's padding mask is empty
        self.fused_sampling = True      # deformable attention: softmax + sampling locations inside the sampling kernel


MIN_ROWS_OWN_GEMM = 1024


def _linear(st: _State, x: torch.Tensor, lin_w: torch.Tensor, lin_b, act=None, residual=None, out=None) -> torch.Tensor:
    """act(x W^T + b) + residual on the matrix cores when the shape allows and the model runs in a fused precision; else torch."""
    # (the caption's ~15 tokens per frame: a 128 x 128-tile kernel walks K alone on one or two CUs -- 25-60 us per Linear at 8 frames
    # against ~10 us for the library's 16 x 16 tiles; tools/gdino_gemm_shapes_probe.py)
    if st.precision != "library" and ops.linear_f32_supported(x, lin_w) and x.numel() // x.shape[-1] >= MIN_ROWS_OWN_GEMM:
        return ops.linear_f32(x, lin_w, lin_b, act=act, residual=residual, precision=st.precision, out=out, owner="gdino")
    y = F.linear(x, lin_w, lin_b)
    if act == "relu":
        y = torch.relu_(y)
    elif act == "gelu":
        y = F.gelu(y)
    if residual is not None:
        y = y.add_(residual.reshape(y.shape))
    if out is not None:
        out.copy_(y.reshape(out.shape))
        return out.view(y.shape)
    return y


def _ln(x: torch.Tensor, ln: nn.LayerNorm) -> torch.Tensor:
    """LayerNorm over the last dimension through the row kernel (f32 GPU tensors, C % 4 == 0, C <= 1024); else torch."""
    C = x.shape[-1]
    if x.is_cuda and x.dtype == torch.float32 and C % 4 == 0 and C <= 1024 and x.is_contiguous() and not torch.is_grad_enabled():
        return ops.layernorm_rows(x.view(-1, 1, 1, C), ln.weight, ln.bias, ln.eps).view(x.shape)
    return ln(x)


def patch_linears(model: nn.Module, st: _State) -> int:
    n = 0
    for mod in model.modules():
        if isinstance(mod, nn.Linear) and not getattr(mod, "_vlfm_fast", False):
            def forward(x, _m=mod):
                return _linear(st, x, _m.weight, _m.bias)

            mod.forward = forward
            mod._vlfm_fast = True
            n += 1
    return n


# ------------------------------------------------------------------------------------------------ Swin
class _Cache:
    """Derived tensors of a module (fused weights, transposed biases), rebuilt when a source parameter changes."""

    def __init__(self) -> None:
        self.key, self.val = None, None

    def get(self, params, build):
        key = tuple((p.data_ptr(), ops.tensor_version(p), str(p.device)) for p in params)
        if key != self.key:
            self.key, self.val = key, build()
        return self.val


def _swin_qkv(att) -> tuple:
    """The q / k / v Linears as ONE [heads * 96, C] weight in the attention kernel's layout (per head: q | k | v, 32 rows each)."""
    heads, hd = att.num_attention_heads, att.head_dim
    ws = [l.weight.detach().view(heads, hd, -1) for l in (att.q_proj, att.k_proj, att.v_proj)]
    w = torch.stack(ws, dim=1).reshape(heads * 3 * hd, -1).contiguous()
    b = None
    if att.q_proj.bias is not None:
        b = torch.stack([l.bias.detach().view(heads, hd) for l in (att.q_proj, att.k_proj, att.v_proj)], dim=1).reshape(-1).contiguous()
    return w, b


def patch_swin_layers(backbone: nn.Module, st: _State) -> int:
    n = 0
    for stage in backbone.modules():     # blocks after the first of a stage own their input (it is the previous block's output)
        blocks = getattr(stage, "blocks", None)
        if type(stage).__name__.endswith("SwinStage") and blocks is not None:
            for i, blk in enumerate(blocks):
                blk.vlfm_stage_entry = i == 0
    for layer in backbone.modules():
        if type(layer).__name__ != "SwinLayer" or layer.attention.head_dim != 32:
            continue
        plain = layer.forward
        qkv_cache, bias_cache, mask_cache = _Cache(), _Cache(), {}

        def forward(hidden_states, input_dimensions, always_partition=False, _l=layer, _plain=plain, _qkv=qkv_cache,
                    _bias=bias_cache, _masks=mask_cache, **kw):
            x = hidden_states
            C = x.shape[-1]
            if not (x.is_cuda and x.dtype == torch.float32 and C % 4 == 0 and not torch.is_grad_enabled()
                    and not kw.get("output_attentions")):
                return _plain(hidden_states, input_dimensions, always_partition=always_partition, **kw)
            if not always_partition:
                _l.set_shift_and_window_size(input_dimensions)
            H, W = input_dimensions
            B = x.shape[0]
            ws, shift = int(_l.window_size), int(_l.shift_size)
            att = _l.attention
            heads = att.num_attention_heads
            x = x.reshape(B, H, W, C)
            if not x.is_contiguous():
                x = x.contiguous()
            elif getattr(_l, "vlfm_stage_entry", True):
                # the block updates its rows in place: the first block of a stage must not do that to the CALLER's tensor (the
                # embeddings / the previous stage's output, which `output_hidden_states` and `out_features` hand out)
                x = x.clone()
            nwy, nwx = -(-H // ws), -(-W // ws)
            wins = ops.layernorm_rows(x, _l.layernorm_before.weight, _l.layernorm_before.bias, _l.layernorm_before.eps,
                                      window=ws, shift=shift, pad_zero=True)                       # [B nW, ws^2, C]
            wq, bq = _qkv.get([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight] +
                              ([att.q_proj.bias, att.k_proj.bias, att.v_proj.bias] if att.q_proj.bias is not None else []),
                              lambda: _swin_qkv(att))
            qkv = _linear(st, wins, wq, bq)                                                       # [B nW, ws^2, heads * 96]
            rpb = att.relative_position_bias
            bias_t = _bias.get([rpb.relative_position_bias_table],
                               lambda: rpb()[0].detach().float().transpose(1, 2).contiguous())    # [heads, j, i]
            mask = None
            if shift > 0:
                mk = (nwy * ws, nwx * ws, ws, shift, str(x.device))
                if mk not in _masks:
                    _masks[mk] = _l.get_attn_mask(nwy * ws, nwx * ws, dtype=torch.float32, device=x.device).contiguous()
                mask = _masks[mk]
            a = ops.window_attention(qkv.view(B * nwy * nwx, ws * ws, heads * 96), bias_t, heads, att.scaling, mask_t=mask)
            o = _linear(st, a, att.o_proj.weight, att.o_proj.bias)
            ops.window_reverse_add_(x, o.reshape(B * nwy * nwx, ws * ws, C).contiguous(), ws, shift)   # x += attention (in place)
            y = ops.layernorm_rows(x, _l.layernorm_after.weight, _l.layernorm_after.bias, _l.layernorm_after.eps)
            hid = _linear(st, y.view(B * H * W, C), _l.mlp.fc1.weight, _l.mlp.fc1.bias, act="gelu")
            x2 = x.view(B * H * W, C)
            _linear(st, hid, _l.mlp.fc2.weight, _l.mlp.fc2.bias, residual=x2, out=x2)                  # x += mlp (in place)
            return x.view(B, H * W, C), None

        layer.forward = forward
        n += 1
    return n


# ------------------------------------------------------------------------------------------------ deformable attention + encoder layer
def patch_deformable(model: nn.Module, st: _State) -> int:
    n = 0
    for mod in model.modules():
        if type(mod).__name__ != "GroundingDinoMultiscaleDeformableAttention":
            continue
        ow_cache = _Cache()

        def forward(hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                    position_embeddings=None, reference_points=None, spatial_shapes=None, spatial_shapes_list=None,
                    level_start_index=None, output_attentions=False, residual=None, _m=mod, _ow=ow_cache):
            """GroundingDinoMultiscaleDeformableAttention.forward [ext] with: value projection, ONE GEMM for sampling offsets +
            attention logits, the output projection taking the caller's residual in its epilogue."""
            q = hidden_states if position_embeddings is None else hidden_states + position_embeddings
            B, Q, _ = q.shape
            S = encoder_hidden_states.shape[1]
            value = _linear(st, encoder_hidden_states, _m.value_proj.weight, _m.value_proj.bias)
            if attention_mask is not None and not st.no_padding:
                value = value.masked_fill(~attention_mask[..., None], float(0))
            value = value.view(B, S, _m.n_heads, _m.d_model // _m.n_heads)
            n_off = _m.sampling_offsets.out_features
            w, b = _ow.get([_m.sampling_offsets.weight, _m.sampling_offsets.bias, _m.attention_weights.weight, _m.attention_weights.bias],
                           lambda: (torch.cat([_m.sampling_offsets.weight.detach(), _m.attention_weights.weight.detach()], 0).contiguous(),
                                    torch.cat([_m.sampling_offsets.bias.detach(), _m.attention_weights.bias.detach()], 0).contiguous()))
            ow = _linear(st, q, w, b)
            if (st.fused_sampling and _m.n_heads == 8 and _m.d_model == 256 and value.is_cuda and value.dtype == torch.float32
                    and reference_points.shape[-1] in (2, 4) and reference_points.dtype == torch.float32):
                # softmax + sampling locations + bilinear sampling in one kernel (csrc/detect_ops.hip)
                from . import det_ops

                out = det_ops.ms_deform_attn_fused(value, spatial_shapes_list, level_start_index, ow.view(B, Q, -1), reference_points,
                                                   _m.n_levels, _m.n_points)
                return _linear(st, out, _m.output_proj.weight, _m.output_proj.bias, residual=residual), None
            sampling_offsets = ow[..., :n_off].reshape(B, Q, _m.n_heads, _m.n_levels, _m.n_points, 2)
            aw = ow[..., n_off:].reshape(B, Q, _m.n_heads, _m.n_levels * _m.n_points)
            aw = F.softmax(aw, -1).view(B, Q, _m.n_heads, _m.n_levels, _m.n_points)
            if reference_points.shape[-1] == 2:
                norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
                loc = reference_points[:, :, None, :, None, :] + sampling_offsets / norm[None, None, None, :, None, :]
            elif reference_points.shape[-1] == 4:
                loc = (reference_points[:, :, None, :, None, :2]
                       + sampling_offsets / _m.n_points * reference_points[:, :, None, :, None, 2:] * 0.5)
            else:
                raise ValueError(f"Last dim of reference_points must be 2 or 4, but got {reference_points.shape[-1]}")
            out = _m.attn(value, spatial_shapes, spatial_shapes_list, level_start_index, loc, aw, _m.im2col_step)
            out = _linear(st, out, _m.output_proj.weight, _m.output_proj.bias, residual=residual)
            return out, aw

        mod.forward = forward
        n += 1
    for mod in model.modules():
        if type(mod).__name__ != "GroundingDinoDeformableLayer":
            continue

        def layer_forward(hidden_states, attention_mask, position_embeddings=None, reference_points=None, spatial_shapes=None,
                          spatial_shapes_list=None, level_start_index=None, output_attentions=False, _m=mod):
            """GroundingDinoDeformableLayer.forward [ext] (eval mode: the dropouts are identities)."""
            x, aw = _m.self_attn(hidden_states=hidden_states, attention_mask=attention_mask, encoder_hidden_states=hidden_states,
                                 encoder_attention_mask=attention_mask, position_embeddings=position_embeddings,
                                 reference_points=reference_points, spatial_shapes=spatial_shapes,
                                 spatial_shapes_list=spatial_shapes_list, level_start_index=level_start_index,
                                 output_attentions=output_attentions, residual=hidden_states)
            x = _ln(x, _m.self_attn_layer_norm)
            h = _linear(st, x, _m.fc1.weight, _m.fc1.bias, act="relu")     # config.activation_function == "relu" (checked at install)
            x = _linear(st, h, _m.fc2.weight, _m.fc2.bias, residual=x)
            return _ln(x, _m.final_layer_norm), aw

        mod.forward = layer_forward
        n += 1
    return n


# ------------------------------------------------------------------------------------------------ fusion layer
def patch_fusion(model: nn.Module, st: _State) -> int:
    """GroundingDinoFusionLayer + GroundingDinoBiMultiHeadAttention [ext], eval mode, RE-ASSOCIATED.  The module projects every one of
    the B x 6380 vision tokens to 1024-d queries and 1024-d values and its 1024-d attention output back to 256-d (three GEMMs of
    0.21 TFLOP per layer at 64 frames: a quarter of the network's matrix work, plus head-transposing copies of 1.7 GB each and batched
    GEMMs with 15 columns), although the other side of the attention is the CAPTION -- L_t ~ 15 tokens.  With x_v the normalised vision
    token, k_t / u_t the text keys / values of head h:
        score[v, h, t] = (W_q[h] x_v + b_q[h]) . k_t[h]       =  x_v . (W_q[h]^T k_t[h])  +  b_q[h] . k_t[h]
        out_v[v]       = W_o concat_h(sum_t p[v, h, t] u_t[h]) =  sum_{h, t} p[v, h, t] (W_o[:, h] u_t[h])
        out_t[t, h]    = sum_v p'[t, h, v] (W_v[h] x_v + b_v[h]) =  W_v[h] (sum_v p'[t, h, v] x_v)  +  b_v[h]
    i.e. the text side is pushed through the vision-side weights once per image (4 x L_t vectors of 256), and the vision tokens only
    meet [256 x 4 L_t] and [4 L_t x 256] matrices: the same function in exact arithmetic, f32 rounding in a different order (checked
    against the module in tests/test_gdino_fast_gpu.py).  The module's global-maximum subtraction and +-50000 clamps are shifts that a
    softmax cancels (they guard fp16 overflow, which an f32 graph does not have) and are dropped."""
    n = 0
    for mod in model.modules():
        if type(mod).__name__ != "GroundingDinoFusionLayer":
            continue

        def forward(vision_features, text_features, attention_mask_vision=None, attention_mask_text=None, _m=mod):
            a = _m.attn
            v = _ln(vision_features, _m.layer_norm_vision)                          # [B, Lv, 256]
            t = _m.layer_norm_text(text_features)                                   # [B, Lt, 256]
            B, Lv, Cv = v.shape
            Lt = t.shape[1]
            Hh, hd, E = a.num_heads, a.head_dim, a.embed_dim
            k = F.linear(t, a.text_proj.weight, a.text_proj.bias).view(B, Lt, Hh, hd)             # text keys
            u = F.linear(t, a.values_text_proj.weight, a.values_text_proj.bias).view(B, Lt, Hh, hd)   # text values
            Wq = (a.vision_proj.weight * a.scale).view(Hh, hd, Cv)
            bq = (a.vision_proj.bias * a.scale).view(Hh, hd)
            A = torch.einsum("hdc,bthd->bhtc", Wq, k).reshape(B, Hh * Lt, Cv)                      # [B, 4 Lt, 256]
            s0 = torch.einsum("hd,bthd->bht", bq, k).reshape(B, 1, Hh * Lt)
            S = torch.bmm(v, A.transpose(1, 2)).add_(s0).view(B, Lv, Hh, Lt)                        # scores[v, h, t]
            # text <- vision: softmax over the vision tokens (per head and text token).  Evaluated on the TRANSPOSED scores
            # [B, 4 Lt, Lv] (a second 60-row GEMM; s0 is constant along the vision tokens and cancels): the softmax then runs along
            # the contiguous dimension -- along dim 1 of S it was torch's strided "spatial" softmax kernel, 409 us per layer at 8
            # frames, 9 % of the forward's GPU time -- and its result is already the left operand of the next product
            St = torch.bmm(A, v.transpose(1, 2))                                                    # [B, 4 Lt, Lv]
            if attention_mask_vision is not None and not st.no_padding:
                St = St.masked_fill(attention_mask_vision[:, None, :], float("-inf"))
            Pt = torch.softmax(St, dim=-1)                                                          # [B, 4 Lt, Lv]
            Y = torch.bmm(Pt, v).view(B, Hh, Lt, Cv)                                                # sum_v p' x_v
            Wv = a.values_vision_proj.weight.view(Hh, hd, Cv)
            ot = (torch.einsum("hdc,bhtc->bthd", Wv, Y) + a.values_vision_proj.bias.view(1, 1, Hh, hd)).reshape(B, Lt, E)
            t_out = t + _m.text_param * F.linear(ot, a.out_text_proj.weight, a.out_text_proj.bias)
            # vision <- text: softmax over the text tokens
            Sv = S
            if attention_mask_text is not None:
                Sv = S.masked_fill(attention_mask_text[:, None, None, :], float("-inf"))
            P = torch.softmax(Sv, dim=-1)
            Wo = (a.out_vision_proj.weight * _m.vision_param[:, None]).view(Cv, Hh, hd)            # layer scale folded in
            Cm = torch.einsum("chd,bthd->bhtc", Wo, u) + (a.out_vision_proj.bias * _m.vision_param / Hh).view(1, 1, 1, Cv)
            v_out = torch.baddbmm(v, P.view(B, Lv, Hh * Lt), Cm.reshape(B, Hh * Lt, Cv))          # every head's p sums to 1: bias / heads each
            return (v_out, P.permute(0, 2, 1, 3).reshape(B * Hh, Lv, Lt)), (t_out, Pt.view(B * Hh, Lt, Lv))

        mod.forward = forward
        n += 1
    return n


# ------------------------------------------------------------------------------------------------ decoder / text self- and cross-attention
def patch_mha(model: nn.Module, st: _State) -> int:
    """GroundingDinoMultiheadAttention.forward [ext] (the decoder's 900-query self-attention and text cross-attention, the text
    enhancer) without materialising the [B x heads, Lq, Lk] score and probability tensors (1.7 GB each for the decoder's self-attention
    at 64 frames): projections on the shared GEMM path, softmax(q k^T / sqrt(d) + mask) v through scaled_dot_product_attention.  The
    attention probabilities the module can return are dropped by every caller unless ``output_attentions`` is requested from the
    model; this path returns None for them."""
    n = 0
    for mod in model.modules():
        if type(mod).__name__ != "GroundingDinoMultiheadAttention":
            continue
        plain = mod.forward

        def forward(queries, keys, values, attention_mask=None, output_attentions=False, _m=mod, _plain=plain):
            if torch.is_grad_enabled() or not queries.is_cuda:
                return _plain(queries, keys, values, attention_mask, output_attentions)
            B = queries.shape[0]
            H, hd = _m.num_attention_heads, _m.attention_head_size
            q = _linear(st, queries, _m.query.weight, _m.query.bias).view(B, -1, H, hd).transpose(1, 2)
            k = _linear(st, keys, _m.key.weight, _m.key.bias).view(B, -1, H, hd).transpose(1, 2)
            v = _linear(st, values, _m.value.weight, _m.value.bias).view(B, -1, H, hd).transpose(1, 2)
            mask = attention_mask
            if mask is not None and mask.dtype != torch.bool:
                mask = mask.to(q.dtype)
            ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
            ctx = ctx.transpose(1, 2).reshape(B, -1, _m.all_head_size)
            out = _linear(st, ctx, _m.out_proj.weight, _m.out_proj.bias)
            return (out, None) if output_attentions else (out,)

        mod.forward = forward
        n += 1
    return n


def accelerate(model: nn.Module, precision: str = "split") -> Dict[str, int]:
    """Install the fused forwards on a transformers GroundingDinoForObjectDetection in eval mode.  ``model.vlfm_fast`` holds the
    switches (``precision``: "split" / "library"; ``no_padding``)."""
    st = _State()
    st.precision = precision
    cfg = model.config
    assert cfg.activation_function == "relu", "the fused encoder FFN assumes ReLU (GroundingDINO_SwinT_OGC)"
    assert not model.training, "fused forwards are inference-only: call model.eval() first"
    model.vlfm_fast = st

    def refuse_padding(_mod, args, kwargs):
        """The fused encoder / fusion forwards skip the padding masks (``no_padding``): every frame of a batch this repo feeds has the
        same size.  An explicit ``pixel_mask`` with holes would silently give other results than the unpatched model: refuse it."""
        pm = kwargs.get("pixel_mask", None)
        if st.no_padding and pm is not None and not bool(pm.all()):
            raise NotImplementedError("accelerated GroundingDINO forward: padded batches (pixel_mask with zeros) are not supported; "
                                      "set model.vlfm_fast.no_padding = False to take the masked paths")
        return None

    model.register_forward_pre_hook(refuse_padding, with_kwargs=True)
    out = {"swin_layers": patch_swin_layers(model.model.backbone, st), "deformable": patch_deformable(model, st),
           "fusion_layers": patch_fusion(model, st), "attention": patch_mha(model, st), "linears": patch_linears(model, st)}
    return out


# ------------------------------------------------------------------------------------------------ HIP-graph replay of the forward
class HostConstantCache:
    """transformers' GroundingDINO forward builds a few small device tensors from Python lists on every call
    (``torch.tensor(SPECIAL_TOKENS, device=...)``, ``torch.as_tensor(spatial_shapes_list, device=...)``): host-to-device copies,
    which a HIP-graph capture does not allow.  Inside this context ``torch.tensor`` / ``torch.as_tensor`` of a Python list / tuple /
    number onto a GPU are memoised by (value, dtype, device): run the forward once eagerly inside the context (records the
    constants), then capture inside it (every constant is a cache hit, nothing crosses PCIe).  The values are compile-time
    constants of the model for a fixed input geometry; sharing one tensor per value is safe as long as nobody writes into it --
    the forward does not."""

    def __init__(self) -> None:
        self.cache: Dict = {}
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


class GraphedForward:
    """The whole detector forward of one input signature (batch, frame size, caption batch, GEMM precision) as ONE HIP graph: at 64
    frames the eager forward issues ~1 200 launches even after the fusions above and the GPU waits for Python between them.  The
    inputs are copied into the graph's static buffers, the outputs (logits, boxes) are the graph's static tensors: read them
    before the next replay of the same signature."""

    def __init__(self, model: nn.Module, max_graphs: int = 4) -> None:
        self.model, self.max_graphs = model, max_graphs
        self.graphs: Dict = {}
        self.consts = HostConstantCache()

    def __call__(self, key, pixel_values, input_ids, attention_mask, token_type_ids):
        key = (key, tuple(pixel_values.shape), tuple(input_ids.shape), self.model.vlfm_fast.precision)
        g = self.graphs.get(key)
        if g is None:
            if len(self.graphs) >= self.max_graphs:
                self.graphs.pop(next(iter(self.graphs)))
            static = [t.clone() for t in (pixel_values, input_ids, attention_mask, token_type_ids)]

            def run():
                out = self.model(pixel_values=static[0], input_ids=static[1], attention_mask=static[2], token_type_ids=static[3])
                return out.logits, out.pred_boxes

            with self.consts:
                side = torch.cuda.Stream(device=pixel_values.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):      # fills every cache (split weights, fused weights, masks, host constants) before the capture
                        run()
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    outs = run()
            g = self.graphs[key] = (graph, static, outs)
        graph, static, outs = g
        for dst, src in zip(static, (pixel_values, input_ids, attention_mask, token_type_ids)):
            dst.copy_(src, non_blocking=True)
        graph.replay()
        return outs
