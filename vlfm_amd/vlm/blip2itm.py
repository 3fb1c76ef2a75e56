"""vlfm.vlm.blip2itm, in-process and batched (reference: /root/reference/vlfm/vlm/blip2itm.py).

The reference runs LAVIS ``blip2_image_text_matching``/``pretrain`` behind a Flask server and calls
``model({"image","text_input"}, match_head="itc")`` one frame at a time (blip2itm.py:37-54); the client ships a
JPEG over HTTP (blip2itm.py:57-64, server_wrapper.py:71-164).  Here the same ITC graph lives in the
policy process and is batched across environments:

    u8 HWC frames --[HIP: PIL-exact bicubic 224x224 + CLIP normalise]--> f16 CHW
      --> EVA ViT-g/14 (39 blocks, PyTorch-ROCm GEMMs + SDPA, fp16) --> LayerNorm
      --> Q-Former query branch (12 BERT layers, 32 queries, cross-attention every 2nd layer, fp32)
      --[HIP: vision_proj 768->256, L2 normalise, dot with the cached text feature, max over 32 queries]--> cosine

The text branch runs once per distinct prompt and is cached (the prompt only changes with the target object,
itm_policy.py:197).  Parameter names follow Hugging Face ``Blip2ForImageTextRetrieval`` so a
``Salesforce/blip2-itm-vit-g`` state dict loads directly; offline (no weights, no tokenizer vocabulary) the model is
randomly initialised and prompts are hashed into token ids -- throughput-faithful, not semantically meaningful.
"""
from __future__ import annotations

import hashlib
import math
import os
import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class Blip2ITCConfig:
    # vision tower (EVA-CLIP ViT-g/14 as shipped with BLIP-2: 39 of 40 blocks)
    image_size: int = 224
    patch_size: int = 14
    v_hidden: int = 1408
    v_layers: int = 39
    v_heads: int = 16
    v_mlp: int = 6144
    v_ln_eps: float = 1e-6
    # Q-Former (BERT-base geometry)
    q_hidden: int = 768
    q_layers: int = 12
    q_heads: int = 12
    q_mlp: int = 3072
    q_ln_eps: float = 1e-12
    cross_attention_frequency: int = 2
    vocab_size: int = 30523
    max_position_embeddings: int = 512
    num_query_tokens: int = 32
    proj_dim: int = 256
    max_txt_len: int = 32

    @staticmethod
    def tiny() -> "Blip2ITCConfig":
        """Small geometry for CPU parity tests."""
        return Blip2ITCConfig(image_size=28, patch_size=14, v_hidden=32, v_layers=2, v_heads=4, v_mlp=64,
                              q_hidden=24, q_layers=4, q_heads=4, q_mlp=48, vocab_size=97,
                              max_position_embeddings=40, num_query_tokens=5, proj_dim=8)


# measured at 256 images (tools/gemm_f16_probe.py, profiles/r05_gemm_probe_*.txt): fc1 + GELU 1.08 ms here against 1.24 ms for
# hipBLASLt + a GELU pass.  As plain GEMMs the kernel ties the library: projection 283 against 290-296 us alone but 0.7 % SLOWER
# inside the network (profiles/r05_full_step_ab.txt: 1 682 against 1 694 env-steps/s), qkv 685 against 670 us, fc2 1 021 against
# 956-980 us -- so only the fused fc1 uses it
DEFAULT_HIP_GEMMS = ("fc1",)


def x_is_contiguous_f16(t: torch.Tensor) -> bool:
    return t.dtype == torch.float16 and t.is_contiguous()


def _default_hip_gemms() -> frozenset:
    import os

    v = os.environ.get("VLFM_VIT_GEMMS")
    if v is None:
        return frozenset(DEFAULT_HIP_GEMMS)
    v = v.strip().lower()
    if v == "all":
        return frozenset(("qkv", "proj", "fc1", "fc2"))
    if v in ("none", ""):
        return frozenset()
    names = frozenset(t.strip() for t in v.split(","))
    if not names <= {"qkv", "proj", "fc1", "fc2"}:
        raise ValueError(f"VLFM_VIT_GEMMS={v!r}: expected all, none or a comma list of qkv, proj, fc1, fc2")
    return names


class _VitBlock(nn.Module):
    def __init__(self, c: Blip2ITCConfig):
        super().__init__()
        self.heads = c.v_heads
        self.layer_norm1 = nn.LayerNorm(c.v_hidden, eps=c.v_ln_eps)
        self.qkv = nn.Linear(c.v_hidden, 3 * c.v_hidden, bias=True)  # bias = [q_bias, 0, v_bias]
        self.projection = nn.Linear(c.v_hidden, c.v_hidden)
        self.layer_norm2 = nn.LayerNorm(c.v_hidden, eps=c.v_ln_eps)
        self.fc1 = nn.Linear(c.v_hidden, c.v_mlp)
        self.fc2 = nn.Linear(c.v_mlp, c.v_hidden)

        self._packed = None
        self.hip_attention = True  # vlfm_vit_attention_f16 at the ViT-g shape (257 tokens, 88-wide heads)
        # a failing vlfm_vit_attention_f16 launch RAISES: the product has one attention backend at this geometry.  Opt out
        # explicitly (BLIP2ITM(..., strict_hip_attention=False)) to let a block switch to the library kernel with a warning
        self.strict_hip_attention = True
        # fc1 + exact GELU in one hand-written MFMA kernel once the GEMM has this many rows (below, the library's smaller
        # tiles win INSIDE the network: standalone (tools/gemm_f16_probe.py, round 5) the kernel is ahead from 8 images on -- 47
        # against 52 us for GEMM + GELU pass, 16: 86 / 95, 32: 165 / 173 -- but with the threshold at 8 images the 8-environment step
        # went from 9.47 to 9.85 ms); 0 = always the library GEMM + a separate GELU pass
        self.hip_mlp_min_rows = 32 * 257
        # which of the block's four GEMMs run on csrc/gemm_f16.hip's 8-phase kernel once the GEMM has hip_mlp_min_rows rows (the
        # others go to hipBLASLt): any of "qkv", "proj", "fc1", "fc2".  VLFM_VIT_GEMMS = all | none | a comma list overrides.
        self.hip_gemms = _default_hip_gemms()

    def pack_heads(self, multiple: int = 32) -> None:
        """Inference-time repacking for the attention kernel: the 88-wide heads of ViT-g are zero-padded to the next
        multiple of 32 (96) INSIDE the qkv / projection weights, so the fused attention runs on an MFMA-friendly head
        size (1.9x faster on gfx950) with bit-identical mathematics: padded q/k columns add exact zeros to every
        score, padded v columns produce zero outputs that meet zero projection columns; the softmax scale stays
        1/sqrt(88)."""
        h, d = self.heads, self.qkv.in_features
        hd = d // h
        hp = (hd + multiple - 1) // multiple * multiple
        if hp == hd:
            self._packed = None
            return
        w = self.qkv.weight.detach().view(3, h, hd, d)
        bq = self.qkv.bias.detach().view(3, h, hd)
        wq = torch.zeros((3, h, hp, d), dtype=w.dtype, device=w.device)
        bb = torch.zeros((3, h, hp), dtype=w.dtype, device=w.device)
        wq[:, :, :hd] = w
        bb[:, :, :hd] = bq
        pw = self.projection.weight.detach().view(d, h, hd)
        wp = torch.zeros((d, h, hp), dtype=pw.dtype, device=pw.device)
        wp[:, :, :hd] = pw
        self._packed = (wq.view(3 * h * hp, d).contiguous(), bb.view(-1).contiguous(), wp.view(d, h * hp).contiguous(),
                        hp, float(hd) ** -0.5)

    def hip_gemms_at(self, rows: int) -> frozenset:
        """The GEMMs of this block that run on csrc/gemm_f16.hip for an activation of ``rows`` rows: those of the configured set whose
        OWN operands meet the kernel's constraints (f16, contiguous, K % 64, N % 8, bias f16 or absent), once the batch is large enough
        for 256 x 256 tiles to fill the chip.  Anything else goes to hipBLASLt -- per GEMM, never an assertion inside the forward."""
        from . import ops

        if not (self.hip_mlp_min_rows and rows >= self.hip_mlp_min_rows):
            return frozenset()
        layers = {"qkv": self.qkv, "proj": self.projection, "fc1": self.fc1, "fc2": self.fc2}

        def ok(lin: nn.Linear) -> bool:
            w, b = lin.weight, lin.bias
            return (x_is_contiguous_f16(w) and w.shape[1] % 64 == 0 and w.shape[0] % 8 == 0 and w.numel() * 2 < (1 << 32)
                    and rows * max(w.shape) * 2 < (1 << 32) and (b is None or x_is_contiguous_f16(b)))

        return frozenset(n for n in self.hip_gemms if ok(layers[n]))

    def forward_deferred(self, x: torch.Tensor, c_in: torch.Tensor, c_mid: torch.Tensor) -> torch.Tensor:
        """Same block, residual adds folded into the GEMMs: ``x`` is the residual stream MINUS the bias vectors of all
        projection / fc2 layers so far (``c_in`` = their f32 sum, a constant of the network; ``c_mid`` = c_in + this
        block's projection bias).  LayerNorm(x + c) runs in one HIP kernel, the projection and fc2 GEMMs accumulate into
        ``x`` in place (beta = 1), so the block has no standalone elementwise add: 2 x (read 2, write 1) passes over the
        [B,257,1408] stream less per block.  Mathematically identical to forward(); f16 rounding of the stream differs
        (the biases are added in f32 inside the LayerNorm instead of in f16 on the stream)."""
        from . import ops

        b, n, d = x.shape
        hd = d // self.heads
        hip = self.hip_gemms_at(b * n)
        h = ops.layernorm_bias(x, c_in, self.layer_norm1.weight, self.layer_norm1.bias, self.layer_norm1.eps)
        x2 = x.view(b * n, d)
        a = None
        if self.hip_attention and n == ops.VIT_ATTENTION_TOKENS and hd in ops.VIT_ATTENTION_HEADS:
            # native head width: the HIP kernel pads 88 -> 96 in LDS / registers only, so the qkv and projection GEMMs keep
            # their original sizes and the attention output needs no transpose copy
            if "qkv" in hip:
                qkv = ops.linear_f16(h.view(b * n, d), self.qkv.weight, self.qkv.bias, "bias")
            else:
                qkv = F.linear(h.view(b * n, d), self.qkv.weight, self.qkv.bias)
            try:
                a = ops.vit_attention(qkv, b, n, self.heads, hd, float(hd) ** -0.5)
            except (RuntimeError, AssertionError, IndexError) as exc:   # e.g. the 150 KB LDS opt-in refused on this device
                if self.strict_hip_attention:
                    raise
                import warnings

                warnings.warn(f"vlfm_vit_attention_f16 unavailable ({exc}); using the library attention kernel instead")
                self.hip_attention = False
        if a is not None:
            if "proj" in hip and x2.is_contiguous():
                ops.linear_f16(a, self.projection.weight, None, "accumulate", out=x2)     # x += a Wp^T, summed in f32
            else:
                x2.addmm_(a, self.projection.weight.t())
        elif self._packed is not None:
            wq, bq, wp, hp, scale = self._packed
            q = F.linear(h.view(b * n, d), wq, bq).view(b, n, 3, self.heads, hp).permute(2, 0, 3, 1, 4)
            a = F.scaled_dot_product_attention(q[0], q[1], q[2], scale=scale).transpose(1, 2).reshape(b * n, self.heads * hp)
            x2.addmm_(a, wp.t())
        else:
            q = self.qkv(h).view(b, n, 3, self.heads, hd).permute(2, 0, 3, 1, 4)
            a = F.scaled_dot_product_attention(q[0], q[1], q[2]).transpose(1, 2).reshape(b * n, d)
            x2.addmm_(a, self.projection.weight.t())
        h = ops.layernorm_bias(x, c_mid, self.layer_norm2.weight, self.layer_norm2.bias, self.layer_norm2.eps)
        if "fc1" in hip:
            act = ops.linear_gelu(h.view(b * n, d), self.fc1.weight, self.fc1.bias)   # GELU in the GEMM epilogue
        else:
            act = F.gelu(F.linear(h, self.fc1.weight, self.fc1.bias)).view(b * n, -1)
        if "fc2" in hip and x2.is_contiguous():
            ops.linear_f16(act, self.fc2.weight, None, "accumulate", out=x2)
        else:
            x2.addmm_(act, self.fc2.weight.t())
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b, n, d = x.shape
        if self._packed is not None and self._packed[0].dtype == x.dtype:
            wq, bq, wp, hp, scale = self._packed
            qkv = F.linear(self.layer_norm1(x), wq, bq).view(b, n, 3, self.heads, hp).permute(2, 0, 3, 1, 4)
            a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], scale=scale)
            x = x + F.linear(a.transpose(1, 2).reshape(b, n, self.heads * hp), wp, self.projection.bias)
        else:
            qkv = self.qkv(self.layer_norm1(x)).view(b, n, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
            a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
            x = x + self.projection(a.transpose(1, 2).reshape(b, n, d))
        return x + self.fc2(F.gelu(self.fc1(self.layer_norm2(x))))


def _exact_split3(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """f32 weights as three f16 pieces with W = W1 + 2^-11 W2 + 2^-22 W3 (33 mantissa bits cover f32's 24)."""
    w = w.detach().float()
    w1 = w.half()
    r = (w - w1.float()) * 2048.0
    w2 = r.half()
    return w1, w2, ((r - w2.float()) * 2048.0).half()


# Pieces of the f32 weights the cross-attention K / V projection multiplies by: 3 = exact (W = W1 + 2^-11 W2 + 2^-22 W3), 2 = the
# first two, which carry 22-23 of f32's 24 significant bits (|W - W1 - 2^-11 W2| <= 2^-23 |W|: one f32 rounding of the weight, the
# size of the roundings an f32 GEMM makes in every one of its K additions).  Measured against f64 at the Q-Former's real shapes the
# two-piece form is as accurate as the f32 GEMM (tests/test_vlm_gpu.py: error <= 2 x the f32 GEMM's, <= 1e-6 of the result's
# scale) and saves a third of the projection: 5.5 -> 3.7 ms per 256 images.
KV_SPLIT_PIECES = 2


def _split_gemm(x16: torch.Tensor, pieces, bias: torch.Tensor, n_pieces: Optional[int] = None) -> torch.Tensor:
    """x16 [M,K] f16 times the split weights (transposed views [K,N]) + bias -> [M,N] f32: f16 x f16 -> f32 MFMA GEMMs
    accumulated in place, smallest term first.  Every product is exact in f32, the accumulation is f32."""
    w1t, w2t, w3t = pieces
    if (KV_SPLIT_PIECES if n_pieces is None else n_pieces) >= 3:
        o = torch.addmm(bias, x16, w3t, alpha=2.0 ** -22, out_dtype=torch.float32)
        torch.addmm(o, x16, w2t, alpha=2.0 ** -11, out_dtype=torch.float32, out=o)   # in place: no copy of the f32 output
    else:
        o = torch.addmm(bias, x16, w2t, alpha=2.0 ** -11, out_dtype=torch.float32)
    torch.addmm(o, x16, w1t, out_dtype=torch.float32, out=o)
    return o


QFORMER_SPLIT_MIN_ROWS = int(os.environ.get("VLFM_QFORMER_SPLIT_MIN_ROWS", "2048"))   # rows (images x 32 queries) from which the Q-Former's f32 Linears go to csrc/gemm_f32.hip


def _qlinear(x: torch.Tensor, lin_w: torch.Tensor, lin_b, act=None, residual=None) -> torch.Tensor:
    """A Linear of the f32 Q-Former.  Large batches on the GPU: the split-precision f32 GEMM of csrc/gemm_f32.hip (operands as f16
    pairs, f32 accumulation: f32-grade results at ~2x hipBLASLt's f32 rate, bias / exact GELU / residual in the epilogue); an operand
    beyond f16's range raises ``ops.gemm_f32_overflow_flag`` (BLIP2ITM.check_numerics reads it).  Everything else: the framework."""
    from . import ops

    if (x.is_cuda and x.dtype == torch.float32 and lin_w.dtype == torch.float32 and not torch.is_grad_enabled()
            and x.numel() // x.shape[-1] >= QFORMER_SPLIT_MIN_ROWS and ops.linear_f32_supported(x, lin_w)):
        return ops.linear_f32(x, lin_w, lin_b, act=act, residual=residual, precision="split", owner="blip2")
    y = F.linear(x, lin_w, lin_b)
    if act == "gelu":
        y = F.gelu(y)
    return y if residual is None else y + residual


class _BertAttention(nn.Module):
    def __init__(self, hidden: int, kv_hidden: int, heads: int, eps: float):
        super().__init__()
        self.heads = heads
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(kv_hidden, hidden)
        self.value = nn.Linear(kv_hidden, hidden)
        self.dense = nn.Linear(hidden, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)
        self._kv_split = None  # see _project_kv_f16
        self._qkv_fused = None  # [Wq; Wk; Wv] for self-attention on the device

    def _project_kv_f16(self, kv16: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """key(kv) and value(kv) in f32 for an f16 ``kv`` (the ViT's tokens ARE f16 values) without an f32 GEMM: the f32
        weights [Wk; Wv] are split EXACTLY into three f16 pieces, W = W1 + 2^-11 W2 + 2^-22 W3 (33 mantissa bits cover f32's
        24), and X W^T is accumulated from three f16 x f16 -> f32 MFMA GEMMs, smallest term first.  Every product x * w_i is
        exact in f32 and the accumulation is f32, i.e. the same arithmetic class as an f32 GEMM on the same operands --
        measured against f64 it is 2.5x MORE accurate than the f32 GEMM (fewer roundings) and 1.5x faster
        (tools/split_gemm_probe.py, round-5 tree); gfx950 has no TF32-like mode and f32 MFMA runs at 1/16 of the f16 rate."""
        if self._kv_split is None or self._kv_split[0].device != kv16.device:
            w1, w2, w3 = _exact_split3(torch.cat([self.key.weight, self.value.weight]))
            bias = torch.cat([self.key.bias, self.value.bias]).detach().float()
            self._kv_split = (w1.t(), w2.t(), w3.t(), bias)   # transposed views: [kv_hidden, 2 * hidden]
        o = _split_gemm(kv16.reshape(-1, kv16.shape[-1]), self._kv_split[:3], self._kv_split[3])
        o = o.view(*kv16.shape[:-1], -1)
        hid = self.key.out_features
        return o[..., :hid], o[..., hid:]

    def forward(self, x: torch.Tensor, kv: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                kv_proj: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
        """``kv_proj`` = (key(kv), value(kv)) computed by the caller (all cross-attention layers in one GEMM set)."""
        b, n, d = x.shape
        h = self.heads
        if kv_proj is not None:
            q = _qlinear(x, self.query.weight, self.query.bias).view(b, n, h, d // h).transpose(1, 2)
            k = kv_proj[0].reshape(b, -1, h, d // h).transpose(1, 2)
            v = kv_proj[1].reshape(b, -1, h, d // h).transpose(1, 2)
        elif kv is None and x.is_cuda and not self.training:
            # self-attention: query / key / value share their input -- one GEMM over [Wq; Wk; Wv] instead of three
            # small ones (every output element is the same dot product; only the launch count changes)
            if self._qkv_fused is None or self._qkv_fused[0].device != x.device or self._qkv_fused[0].dtype != x.dtype:
                self._qkv_fused = (torch.cat([self.query.weight, self.key.weight, self.value.weight]).detach(),
                                   torch.cat([self.query.bias, self.key.bias, self.value.bias]).detach())
            qkv = _qlinear(x, *self._qkv_fused).view(b, n, 3, h, d // h).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
        else:
            q = self.query(x).view(b, n, h, d // h).transpose(1, 2)
            if kv is not None and kv.dtype == torch.float16 and kv.is_cuda and self.key.weight.dtype == torch.float32:
                k, v = self._project_kv_f16(kv)
                src_len = kv.shape[1]
            else:
                src = x if kv is None else kv.to(x.dtype)
                k, v, src_len = self.key(src), self.value(src), src.shape[1]
            k = k.reshape(b, src_len, h, d // h).transpose(1, 2)
            v = v.reshape(b, src_len, h, d // h).transpose(1, 2)
        a = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        return self.LayerNorm(_qlinear(a.transpose(1, 2).reshape(b, n, d), self.dense.weight, self.dense.bias,
                                       residual=x))  # post-LN residual (BERT)


class _BertFFN(nn.Module):
    def __init__(self, hidden: int, inter: int, eps: float):
        super().__init__()
        self.up = nn.Linear(hidden, inter)
        self.down = nn.Linear(inter, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        hid = _qlinear(x, self.up.weight, self.up.bias, act="gelu")
        return self.LayerNorm(_qlinear(hid, self.down.weight, self.down.bias, residual=x))


class _QFormerLayer(nn.Module):
    def __init__(self, c: Blip2ITCConfig, idx: int):
        super().__init__()
        self.attention = _BertAttention(c.q_hidden, c.q_hidden, c.q_heads, c.q_ln_eps)
        self.crossattention = (_BertAttention(c.q_hidden, c.v_hidden, c.q_heads, c.q_ln_eps)
                               if idx % c.cross_attention_frequency == 0 else None)
        self.ffn_text = _BertFFN(c.q_hidden, c.q_mlp, c.q_ln_eps)   # HF: intermediate / output
        self.ffn_query = _BertFFN(c.q_hidden, c.q_mlp, c.q_ln_eps)  # HF: intermediate_query / output_query


class Blip2ITCModel(nn.Module):
    """The ITC graph of BLIP-2 (image branch + text branch); weights interchangeable with HF's port."""

    def __init__(self, cfg: Blip2ITCConfig):
        super().__init__()
        c = self.cfg = cfg
        n_patches = (c.image_size // c.patch_size) ** 2
        self.class_embedding = nn.Parameter(torch.zeros(1, 1, c.v_hidden))
        self.position_embedding = nn.Parameter(torch.zeros(1, n_patches + 1, c.v_hidden))
        self.patch_embedding = nn.Conv2d(3, c.v_hidden, c.patch_size, c.patch_size)
        self.blocks = nn.ModuleList([_VitBlock(c) for _ in range(c.v_layers)])
        self.post_layernorm = nn.LayerNorm(c.v_hidden, eps=c.v_ln_eps)
        self.query_tokens = nn.Parameter(torch.zeros(1, c.num_query_tokens, c.q_hidden))
        self.word_embeddings = nn.Embedding(c.vocab_size, c.q_hidden)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.q_hidden)
        self.q_layernorm = nn.LayerNorm(c.q_hidden, eps=c.q_ln_eps)
        self.q_layers = nn.ModuleList([_QFormerLayer(c, i) for i in range(c.q_layers)])
        self.vision_projection = nn.Linear(c.q_hidden, c.proj_dim)
        self.text_projection = nn.Linear(c.q_hidden, c.proj_dim)
        # inference fast path of the ViT (f16, packed heads, HIP device): residual adds folded into the GEMMs, see
        # _VitBlock.forward_deferred.  _deferred_c caches the per-layer bias sums; anything that changes weights clears it.
        self.deferred_bias = True
        self._deferred_c: Optional[torch.Tensor] = None
        self.split_kv = True  # Q-Former cross-attention K/V projections of f16 tokens via exact 3-way weight split
        self.split_kv_min_rows = 32 * 257
        self._kv_all = None

    # ---- weights ---------------------------------------------------------------------------------------------
    def init_random(self, seed: int = 0) -> "Blip2ITCModel":
        dev = self.query_tokens.device
        g = torch.Generator(device=dev).manual_seed(seed)
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=g, device=dev) * 0.02)
                else:
                    p.zero_()
            for m in self.modules():
                if isinstance(m, nn.LayerNorm):
                    m.weight.fill_(1.0)
        self.weights_changed()
        return self

    def weights_changed(self) -> None:
        """Drop everything derived from the weights (packed heads stay the caller's to redo): call after editing
        parameters in place."""
        self._deferred_c = None
        self._kv_all = None
        for layer in self.q_layers:
            layer.attention._qkv_fused = None
            if layer.crossattention is not None:
                layer.crossattention._kv_split = None

    def vision_dtype(self) -> torch.dtype:
        return self.patch_embedding.weight.dtype

    def set_precision(self, vision_dtype: torch.dtype = torch.float16) -> "Blip2ITCModel":
        """LAVIS keeps the ViT in fp16 and the Q-Former in fp32."""
        for m in (self.patch_embedding, self.blocks, self.post_layernorm):
            m.to(vision_dtype)
        self.class_embedding.data = self.class_embedding.data.to(vision_dtype)
        self.position_embedding.data = self.position_embedding.data.to(vision_dtype)
        self.weights_changed()
        return self

    def load_hf_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Load a ``Blip2ForImageTextRetrieval`` state dict (e.g. Salesforce/blip2-itm-vit-g)."""
        own = dict(self.named_parameters())

        def put(name: str, src: str) -> None:
            with torch.no_grad():
                own[name].copy_(sd[src].to(own[name].dtype).reshape(own[name].shape))

        put("class_embedding", "vision_model.embeddings.class_embedding")
        put("position_embedding", "vision_model.embeddings.position_embedding")
        put("patch_embedding.weight", "vision_model.embeddings.patch_embedding.weight")
        put("patch_embedding.bias", "vision_model.embeddings.patch_embedding.bias")
        for i in range(self.cfg.v_layers):
            s, d = f"vision_model.encoder.layers.{i}.", f"blocks.{i}."
            for a, b in [("layer_norm1", "layer_norm1"), ("layer_norm2", "layer_norm2"),
                         ("self_attn.qkv", "qkv"), ("self_attn.projection", "projection"),
                         ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")]:
                put(d + b + ".weight", s + a + ".weight")
                put(d + b + ".bias", s + a + ".bias")
        put("post_layernorm.weight", "vision_model.post_layernorm.weight")
        put("post_layernorm.bias", "vision_model.post_layernorm.bias")
        put("query_tokens", "query_tokens")
        put("word_embeddings.weight", "embeddings.word_embeddings.weight")
        put("position_embeddings.weight", "embeddings.position_embeddings.weight")
        put("q_layernorm.weight", "qformer.layernorm.weight")
        put("q_layernorm.bias", "qformer.layernorm.bias")
        for i, layer in enumerate(self.q_layers):
            s, d = f"qformer.encoder.layer.{i}.", f"q_layers.{i}."
            atts = [("attention", "attention")] + ([("crossattention", "crossattention")]
                                                   if layer.crossattention is not None else [])
            for a, b in atts:
                for hf, mine in [("attention.query", "query"), ("attention.key", "key"), ("attention.value", "value"),
                                 ("output.dense", "dense"), ("output.LayerNorm", "LayerNorm")]:
                    put(f"{d}{b}.{mine}.weight", f"{s}{a}.{hf}.weight")
                    put(f"{d}{b}.{mine}.bias", f"{s}{a}.{hf}.bias")
            for hf_i, hf_o, mine in [("intermediate", "output", "ffn_text"),
                                     ("intermediate_query", "output_query", "ffn_query")]:
                put(f"{d}{mine}.up.weight", f"{s}{hf_i}.dense.weight")
                put(f"{d}{mine}.up.bias", f"{s}{hf_i}.dense.bias")
                put(f"{d}{mine}.down.weight", f"{s}{hf_o}.dense.weight")
                put(f"{d}{mine}.down.bias", f"{s}{hf_o}.dense.bias")
                put(f"{d}{mine}.LayerNorm.weight", f"{s}{hf_o}.LayerNorm.weight")
                put(f"{d}{mine}.LayerNorm.bias", f"{s}{hf_o}.LayerNorm.bias")
        for n in ("vision_projection", "text_projection"):
            put(n + ".weight", n + ".weight")
            put(n + ".bias", n + ".bias")

    def load_lavis_state_dict(self, sd: Dict[str, torch.Tensor], vit_sd: Optional[Dict[str, torch.Tensor]] = None) -> None:
        """Load what the REFERENCE loads (blip2itm.py:29-34: LAVIS ``load_model_and_preprocess("blip2_image_text_matching",
        "pretrain")`` [ext]): ``sd`` = the ``model`` entry of ``blip2_pretrained.pth`` (Blip2Qformer: ``query_tokens``,
        ``ln_vision.*``, ``Qformer.bert.*``, ``vision_proj.*``, ``text_proj.*``, + ``itm_head`` / ``temp`` / the LM head
        ``Qformer.cls.*``, which the ITC score does not use), ``vit_sd`` = ``eva_vit_g.pth`` (the frozen EVA ViT-g LAVIS reads
        from its own file: bare names ``cls_token``, ``pos_embed``, ``patch_embed.proj.*``, ``blocks.{i}.*``; the 40th block
        and the classifier's norm / head are dropped there too, lavis/models/eva_vit.py [ext]) -- or the tower under
        ``visual_encoder.*`` inside ``sd`` when the file carries it.  Strict both ways: every parameter of this graph is fed,
        every tensor given is placed or on the explicit not-used list, shapes equal.  EVA's attention has ``q_bias`` /
        ``v_bias`` and no key bias: the qkv bias is [q_bias, 0, v_bias]."""
        c = self.cfg
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        vit = {("visual_encoder." + k): v for k, v in (vit_sd or {}).items()}
        overlap = set(vit) & set(sd)
        if overlap:
            raise KeyError(f"the vision tower is given twice (e.g. {sorted(overlap)[:2]})")
        src = {**sd, **vit}
        own = dict(self.named_parameters())
        used, fed = set(), set()

        def take(key: str) -> torch.Tensor:
            if key not in src:
                raise KeyError(f"LAVIS state dict lacks {key}" + ("" if vit_sd is not None or not key.startswith("visual_encoder.")
                                                                 else " (blip2_pretrained.pth does not carry the frozen ViT: "
                                                                      "pass eva_vit_g.pth as the vision checkpoint)"))
            used.add(key)
            return src[key]

        def put(name: str, value: torch.Tensor) -> None:
            if tuple(value.shape) != tuple(own[name].shape) and value.numel() != own[name].numel():
                raise ValueError(f"{name}: {tuple(own[name].shape)} in the graph, {tuple(value.shape)} in the file")
            with torch.no_grad():
                own[name].copy_(value.to(own[name].dtype).reshape(own[name].shape))
            fed.add(name)

        put("class_embedding", take("visual_encoder.cls_token"))
        put("position_embedding", take("visual_encoder.pos_embed"))
        put("patch_embedding.weight", take("visual_encoder.patch_embed.proj.weight"))
        put("patch_embedding.bias", take("visual_encoder.patch_embed.proj.bias"))
        for i in range(c.v_layers):
            o, d = f"visual_encoder.blocks.{i}.", f"blocks.{i}."
            for a, b in (("norm1", "layer_norm1"), ("norm2", "layer_norm2"), ("attn.proj", "projection"),
                         ("mlp.fc1", "fc1"), ("mlp.fc2", "fc2")):
                put(d + b + ".weight", take(o + a + ".weight"))
                put(d + b + ".bias", take(o + a + ".bias"))
            put(d + "qkv.weight", take(o + "attn.qkv.weight"))
            qb, vb = take(o + "attn.q_bias"), take(o + "attn.v_bias")
            put(d + "qkv.bias", torch.cat([qb, torch.zeros_like(vb), vb]))
        put("post_layernorm.weight", take("ln_vision.weight"))
        put("post_layernorm.bias", take("ln_vision.bias"))
        put("query_tokens", take("query_tokens"))
        put("word_embeddings.weight", take("Qformer.bert.embeddings.word_embeddings.weight"))
        put("position_embeddings.weight", take("Qformer.bert.embeddings.position_embeddings.weight"))
        put("q_layernorm.weight", take("Qformer.bert.embeddings.LayerNorm.weight"))
        put("q_layernorm.bias", take("Qformer.bert.embeddings.LayerNorm.bias"))
        for i, layer in enumerate(self.q_layers):
            o, d = f"Qformer.bert.encoder.layer.{i}.", f"q_layers.{i}."
            for att in ["attention"] + (["crossattention"] if layer.crossattention is not None else []):
                for theirs, mine in (("self.query", "query"), ("self.key", "key"), ("self.value", "value"),
                                     ("output.dense", "dense"), ("output.LayerNorm", "LayerNorm")):
                    put(f"{d}{att}.{mine}.weight", take(f"{o}{att}.{theirs}.weight"))
                    put(f"{d}{att}.{mine}.bias", take(f"{o}{att}.{theirs}.bias"))
            for t_i, t_o, mine in (("intermediate", "output", "ffn_text"), ("intermediate_query", "output_query", "ffn_query")):
                put(f"{d}{mine}.up.weight", take(f"{o}{t_i}.dense.weight"))
                put(f"{d}{mine}.up.bias", take(f"{o}{t_i}.dense.bias"))
                put(f"{d}{mine}.down.weight", take(f"{o}{t_o}.dense.weight"))
                put(f"{d}{mine}.down.bias", take(f"{o}{t_o}.dense.bias"))
                put(f"{d}{mine}.LayerNorm.weight", take(f"{o}{t_o}.LayerNorm.weight"))
                put(f"{d}{mine}.LayerNorm.bias", take(f"{o}{t_o}.LayerNorm.bias"))
        put("vision_projection.weight", take("vision_proj.weight"))
        put("vision_projection.bias", take("vision_proj.bias"))
        put("text_projection.weight", take("text_proj.weight"))
        put("text_projection.bias", take("text_proj.bias"))
        unfed = sorted(set(own) - fed)
        if unfed:
            raise KeyError(f"{len(unfed)} parameters of the graph were not fed, e.g. {unfed[:4]}")

        def unused_ok(k: str) -> bool:   # present in the files, not part of the ITC score (or of LAVIS' own 39-block tower)
            return (k.startswith(("Qformer.cls.", "itm_head.")) or k in ("temp", "Qformer.bert.embeddings.position_ids")
                    or k.startswith(f"visual_encoder.blocks.{c.v_layers}.")
                    or k.startswith(("visual_encoder.norm.", "visual_encoder.fc_norm.", "visual_encoder.head.")))

        left = sorted(k for k in src if k not in used and not unused_ok(k))
        if left:
            raise KeyError(f"{len(left)} tensors of the file have no place in the ITC graph, e.g. {left[:4]}")
        self.weights_changed()

    # ---- branches --------------------------------------------------------------------------------------------
    def vision_tokens(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[B,3,224,224] (or im2col patches [B,256,588] straight from the preprocess kernel) -> LayerNorm'd ViT tokens
        [B,257,1408].  The stride-14 patch convolution is evaluated as ONE GEMM over the im2col rows (MIOpen's direct
        kernels for this shape are ~100x slower than the GEMM)."""
        c = self.cfg
        if pixel_values.dim() == 4:
            b, g = pixel_values.shape[0], c.image_size // c.patch_size
            pixel_values = pixel_values.reshape(b, 3, g, c.patch_size, g, c.patch_size).permute(0, 2, 4, 1, 3, 5) \
                .reshape(b, g * g, 3 * c.patch_size * c.patch_size)
        w = self.patch_embedding.weight
        x = F.linear(pixel_values.to(w.dtype), w.view(w.shape[0], -1), self.patch_embedding.bias)
        x = torch.cat([self.class_embedding.expand(x.shape[0], -1, -1), x], dim=1) + self.position_embedding
        if self.deferred_bias and x.is_cuda and x.dtype == torch.float16 and c.v_hidden % 8 == 0 and c.v_hidden <= 2048:
            from . import ops

            c = self._deferred_constants()
            x = x.contiguous()
            for i, blk in enumerate(self.blocks):
                x = blk.forward_deferred(x, c[2 * i], c[2 * i + 1])
            return ops.layernorm_bias(x, c[2 * len(self.blocks)], self.post_layernorm.weight, self.post_layernorm.bias,
                                      self.post_layernorm.eps)
        for blk in self.blocks:
            x = blk(x)
        return self.post_layernorm(x)

    def _deferred_constants(self) -> torch.Tensor:
        """[2 L + 1][hidden] f32: running sum of the projection / fc2 bias vectors in front of every LayerNorm."""
        if self._deferred_c is None or self._deferred_c.device != self.post_layernorm.weight.device:
            rows, run = [], torch.zeros(self.cfg.v_hidden, dtype=torch.float32, device=self.post_layernorm.weight.device)
            for blk in self.blocks:
                rows.append(run.clone())
                run = run + blk.projection.bias.detach().float()
                rows.append(run.clone())
                run = run + blk.fc2.bias.detach().float()
            rows.append(run.clone())
            self._deferred_c = torch.stack(rows).contiguous()
        return self._deferred_c

    def query_features(self, image_tokens: torch.Tensor) -> torch.Tensor:
        """Q-Former query branch: [B,257,1408] -> [B,32,768] (fp32)."""
        # f16 ViT tokens stay f16 on the device: the cross-attention projects them with exact-split GEMMs
        # (_BertAttention._project_kv_f16); anything else is promoted to the Q-Former's dtype as LAVIS does (.float())
        # (three launches per projection only pay off once the GEMMs are large: >= 32 images)
        split_ok = (self.split_kv and image_tokens.dtype == torch.float16 and image_tokens.is_cuda
                    and image_tokens.shape[0] * image_tokens.shape[1] >= self.split_kv_min_rows)
        enc = image_tokens if split_ok else image_tokens.to(self.q_layernorm.weight.dtype)
        h = self.q_layernorm(self.query_tokens).expand(enc.shape[0], -1, -1)
        proj = self._project_all_kv(enc) if split_ok else None
        hid, at = self.cfg.q_hidden, 0
        for layer in self.q_layers:
            h = layer.attention(h)
            if layer.crossattention is not None:
                if proj is not None:
                    h = layer.crossattention(h, kv_proj=(proj[..., at:at + hid], proj[..., at + hid:at + 2 * hid]))
                    at += 2 * hid
                else:
                    h = layer.crossattention(h, kv=enc)
            h = layer.ffn_query(h)
        return h

    def _project_all_kv(self, tokens16: torch.Tensor) -> torch.Tensor:
        """key / value projections of EVERY cross-attention layer in one exact-split GEMM set ([B,257,1408] f16 ->
        [B,257, layers * 2 * hidden] f32, layer-major [K_0 | V_0 | K_1 | V_1 ...]): the image tokens are the same for all
        layers, and one wide GEMM runs closer to the MFMA peak than six narrow ones (_BertAttention._project_kv_f16 is
        the per-layer form)."""
        if self._kv_all is None or self._kv_all[0].device != tokens16.device:
            cross = [l.crossattention for l in self.q_layers if l.crossattention is not None]
            w = torch.cat([torch.cat([c.key.weight, c.value.weight]) for c in cross])
            bias = torch.cat([torch.cat([c.key.bias, c.value.bias]) for c in cross]).detach().float()
            w1, w2, w3 = _exact_split3(w)
            self._kv_all = (w1.t(), w2.t(), w3.t(), bias)
        o = _split_gemm(tokens16.reshape(-1, tokens16.shape[-1]), self._kv_all[:3], self._kv_all[3])
        return o.view(*tokens16.shape[:-1], -1)

    def text_feature(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Q-Former text branch: ids [B,L] -> L2-normalised text_proj(CLS) [B,256]."""
        pos = torch.arange(input_ids.shape[1], device=input_ids.device)[None]
        h = self.q_layernorm(self.word_embeddings(input_ids) + self.position_embeddings(pos))
        mask = None
        if attention_mask is not None:
            mask = (attention_mask[:, None, None, :] > 0)
        for layer in self.q_layers:
            h = layer.ffn_text(layer.attention(h, mask=mask))
        return F.normalize(self.text_projection(h[:, 0, :]), dim=-1)

    def itc_reference_head(self, query_feats: torch.Tensor, text_feat: torch.Tensor) -> torch.Tensor:
        """Plain-PyTorch fp32 reference of the ITC head (the HIP kernel in csrc/vlm_ops.hip is checked against it):
        max over queries of <normalise(vision_proj(q)), text>."""
        img = F.normalize(self.vision_projection(query_feats.float()), dim=-1)
        return torch.einsum("bqd,bd->bq", img, text_feat.float().expand(img.shape[0], -1)).max(dim=1).values


# ------------------------------------------------------------------------------------------------------ text side
_CAPTION_STRIP = re.compile(r"([.!\"()*#:;~])")


def blip_caption(text: str, max_words: int = 50) -> str:
    """LAVIS ``blip_caption`` text processor [ext]: lower-case, strip punctuation, collapse spaces, <= 50 words."""
    t = _CAPTION_STRIP.sub(" ", text.lower())
    t = re.sub(r"\s{2,}", " ", t).rstrip("\n").strip(" ")
    words = t.split(" ")
    return " ".join(words[:max_words]) if len(words) > max_words else t


class HashTokenizer:
    """Offline stand-in for the BERT WordPiece tokenizer (no vocabulary file in this environment): [CLS] + one
    deterministic id per word + [SEP].  Only used with randomly initialised weights."""

    def __init__(self, vocab_size: int, max_len: int):
        self.vocab_size, self.max_len = vocab_size, max_len

    def __call__(self, text: str) -> List[int]:
        ids = [101 % self.vocab_size]
        for w in text.split():
            hv = int.from_bytes(hashlib.sha1(w.encode()).digest()[:4], "little")
            ids.append(1000 % self.vocab_size + hv % max(1, self.vocab_size - 1000))
        ids = ids[: self.max_len - 1] + [102 % self.vocab_size]
        return ids


# ------------------------------------------------------------------------------------------------------ wrappers
class BLIP2ITM:
    """Same surface as the reference's ``BLIP2ITM`` (blip2itm.py:17-54) plus a batched entry point."""

    def __init__(self, name: str = "blip2_image_text_matching", model_type: str = "pretrain", device=None,
                 model_dir: Optional[str] = None, config: Optional[Blip2ITCConfig] = None,
                 vision_dtype: torch.dtype = torch.float16, seed: int = 0, allow_random_init: bool = False,
                 strict_hip_attention: bool = True, checkpoint: Optional[str] = None, vit_checkpoint: Optional[str] = None,
                 tokenizer_dir: Optional[str] = None) -> None:
        from ..mapping.base_map import require_gpu
        from .. import _lib

        self.device = require_gpu(device)
        _lib.lib()
        from . import ops as _ops

        _ops.use_tuned_gemms()     # the library GEMMs' per-shape solutions measured on this image (tools/tune_gemms.py), if recorded
        self.cfg = config or Blip2ITCConfig()
        self.tokenizer = None
        model_dir = model_dir or os.environ.get("BLIP2ITM_MODEL_DIR")
        checkpoint = checkpoint or os.environ.get("BLIP2_LAVIS_CHECKPOINT")
        vit_checkpoint = vit_checkpoint or os.environ.get("EVA_VIT_G_CHECKPOINT")
        if checkpoint and not model_dir:
            # the reference's own artefacts (blip2itm.py:29-34 -> LAVIS [ext]): blip2_pretrained.pth (+ eva_vit_g.pth)
            for f in (checkpoint, vit_checkpoint):
                if f and not os.path.isfile(f):
                    raise FileNotFoundError(f"BLIP-2 checkpoint {f!r} not found")
            self.model = Blip2ITCModel(self.cfg)
            ck = torch.load(checkpoint, map_location="cpu", weights_only=True)
            vit = torch.load(vit_checkpoint, map_location="cpu", weights_only=True) if vit_checkpoint else None
            self.model.load_lavis_state_dict(ck["model"] if isinstance(ck, dict) and "model" in ck else ck,
                                             vit["model"] if isinstance(vit, dict) and "model" in vit else vit)
            self.tokenizer = self._lavis_tokenizer(tokenizer_dir)
            self.weights = f"lavis:{checkpoint}" + (f" + {vit_checkpoint}" if vit_checkpoint else "")
        elif model_dir:
            self.model = Blip2ITCModel(self.cfg)
            self._load_pretrained(model_dir)
            self.weights = f"pretrained:{model_dir}"
        elif not (allow_random_init or config is not None):
            # the reference downloads its weights through LAVIS (blip2itm.py:29-34); a scorer that silently runs on random
            # weights would steer the value map with noise
            raise ValueError("BLIP2ITM needs `checkpoint` (+ `vit_checkpoint`): LAVIS' blip2_pretrained.pth and eva_vit_g.pth, what "
                             "the reference loads -- or model_dir / BLIP2ITM_MODEL_DIR (a Salesforce/blip2-itm-vit-g directory); "
                             "pass allow_random_init=True for a randomly initialised network (benchmarks / geometry tests only)")
        else:
            with torch.device(self.device):  # allocate and initialise the 1.2 B parameters directly in HBM
                self.model = Blip2ITCModel(self.cfg)
            self.model.init_random(seed)
            self.weights = "random-init"
        if self.tokenizer is None:
            self.tokenizer = HashTokenizer(self.cfg.vocab_size, self.cfg.max_txt_len)
        self.model.eval().to(self.device).set_precision(vision_dtype)
        for blk in self.model.blocks:
            blk.pack_heads()
            blk.strict_hip_attention = bool(strict_hip_attention)
        self._text_cache: Dict[str, torch.Tensor] = {}
        self._proj_t = None
        self.two_stream_min = None    # e.g. 64: run batches of at least that many images as two halves on two streams
        self._side_stream = None

    def _lavis_tokenizer(self, tokenizer_dir: Optional[str]):
        """LAVIS' Blip2Base.init_tokenizer [ext]: bert-base-uncased + the added ``[DEC]`` token (id 30522); truncation to
        ``max_txt_len``.  The vocabulary has to be on disk (``tokenizer_dir`` / ``BLIP2_TOKENIZER``, or the local transformers
        cache): real weights never run on the hash stand-in."""
        from transformers import BertTokenizer

        path = tokenizer_dir or os.environ.get("BLIP2_TOKENIZER")
        try:
            if path:
                vocab = path if os.path.isfile(path) else os.path.join(path, "vocab.txt")
                tok = BertTokenizer(vocab_file=vocab, do_lower_case=True)
            else:
                tok = BertTokenizer.from_pretrained("bert-base-uncased", local_files_only=True)
                if len(tok) < 30522:   # transformers can hand back an EMPTY tokenizer when nothing is cached
                    raise FileNotFoundError("no cached bert-base-uncased vocabulary")
        except Exception as exc:  # noqa: BLE001
            raise ValueError("BLIP-2 weights need the bert-base-uncased vocabulary: pass tokenizer_dir (or set BLIP2_TOKENIZER) "
                             "to a directory with its vocab.txt") from exc
        tok.add_special_tokens({"bos_token": "[DEC]"})
        return lambda t: tok(t, truncation=True, max_length=self.cfg.max_txt_len)["input_ids"]

    @property
    def strict_hip_attention(self) -> bool:
        return all(blk.strict_hip_attention for blk in self.model.blocks)

    @strict_hip_attention.setter
    def strict_hip_attention(self, on: bool) -> None:
        """True: a failing vlfm_vit_attention_f16 launch raises instead of switching the block to the library kernel."""
        for blk in self.model.blocks:
            blk.strict_hip_attention = bool(on)

    @property
    def attention_path(self) -> str:
        """Which attention kernel the ViT blocks run with on the fast path: "hip" (vlfm_vit_attention_f16 in every block),
        "library" (scaled_dot_product_attention in every block) or "mixed"."""
        flags = {bool(blk.hip_attention) for blk in self.model.blocks}
        return "hip" if flags == {True} else "library" if flags == {False} else "mixed"

    def mlp_path(self, n_images: int) -> str:
        """Which fc1 + GELU the ViT blocks run with at this batch size: "hip" (vlfm_gemm_f16_nt, GELU in the epilogue) or "library"
        (hipBLASLt GEMM + a GELU pass)."""
        blk = self.model.blocks[0]
        rows = n_images * ((self.cfg.image_size // self.cfg.patch_size) ** 2 + 1)
        return "hip" if "fc1" in blk.hip_gemms_at(rows) else "library"

    def gemm_path(self, n_images: int) -> str:
        """Which of a ViT block's GEMMs run on the hand-written 8-phase kernel at this batch size, e.g. "hip:fc1,fc2,proj,qkv"."""
        blk = self.model.blocks[0]
        rows = n_images * ((self.cfg.image_size // self.cfg.patch_size) ** 2 + 1)
        names = sorted(blk.hip_gemms_at(rows))
        return "hip:" + ",".join(names) if names else "library"

    def _load_pretrained(self, model_dir: str) -> None:
        from safetensors.torch import load_file

        sd: Dict[str, torch.Tensor] = {}
        for f in sorted(os.listdir(model_dir)):
            if f.endswith(".safetensors"):
                sd.update(load_file(os.path.join(model_dir, f)))
        self.model.load_hf_state_dict(sd)
        from transformers import BertTokenizer

        tok = BertTokenizer.from_pretrained(model_dir)
        self.tokenizer = lambda t: tok(t, truncation=True, max_length=self.cfg.max_txt_len)["input_ids"]

    # -- text (cached) ----------------------------------------------------------------------------------------
    @torch.inference_mode()
    def text_feature(self, txt: str) -> torch.Tensor:
        if txt not in self._text_cache:
            ids = torch.tensor([self.tokenizer(blip_caption(txt))], device=self.device)
            self._text_cache[txt] = self.model.text_feature(ids)[0].float().contiguous()
        return self._text_cache[txt]

    # -- images -----------------------------------------------------------------------------------------------
    def cosine_batch_graphed(self, images_u8: torch.Tensor, txts: Sequence[str]) -> torch.Tensor:
        """``cosine_batch`` replayed from a captured HIP graph (one per (shape, prompts) key): the ~650 kernel launches
        of a ViT-g + Q-Former forward collapse into one graph launch, which is what bounds small batches (launch-bound
        below ~16 images).  The input is copied into the graph's static buffer; the result is the graph's static output."""
        key = (tuple(images_u8.shape), tuple(txts))
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        if key not in self._graphs:
            static_in = images_u8.clone()
            for _ in range(2):  # warm-up outside capture: coefficient tables, text features, hipBLASLt workspaces
                self.cosine_batch(static_in, txts)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = self.cosine_batch(static_in, txts)
            self._graphs[key] = (g, static_in, static_out)
        g, static_in, static_out = self._graphs[key]
        static_in.copy_(images_u8)
        g.replay()
        return static_out

    @torch.inference_mode()
    def cosine_batch(self, images_u8: torch.Tensor, txts: Sequence[str]) -> torch.Tensor:
        """images_u8: [B,H,W,3] uint8 RGB on device; txts: one prompt per image (or a single shared prompt).
        Returns [B] fp32 cosines on device (no host sync)."""
        from . import ops

        B = images_u8.shape[0]
        pix = ops.preprocess_rgb(images_u8, self.cfg.image_size, self.model.vision_dtype(),
                                 patch_size=self.cfg.patch_size)
        if self.two_stream_min is not None and B >= self.two_stream_min and not torch.cuda.is_current_stream_capturing():
            # OPT-IN.  Two half batches on two HIP streams: the compute-bound GEMMs of one half overlap the memory-bound
            # LayerNorm / GELU / attention kernels of the other (+2.8 % env-steps/s at 128 images).  Off by default: the
            # large GEMMs are stream-K kernels whose workgroups wait for each other, and a probe with THREE concurrent
            # forwards hung (tools/two_stream_probe.py) -- not a risk worth 3 % in an unattended run.
            main = torch.cuda.current_stream(self.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(self.device)
            side = self._side_stream
            half = B // 2
            side.wait_stream(main)
            with torch.cuda.stream(side):
                q_hi = self.model.query_features(self.model.vision_tokens(pix[half:])).float()
            q_lo = self.model.query_features(self.model.vision_tokens(pix[:half])).float()
            main.wait_stream(side)
            q = torch.cat([q_lo, q_hi]).contiguous()
        else:
            q = self.model.query_features(self.model.vision_tokens(pix)).float().contiguous()
        if len(txts) == 1:
            text = self.text_feature(txts[0])[None].expand(B, -1).contiguous()
        else:
            assert len(txts) == B
            text = torch.stack([self.text_feature(t) for t in txts]).contiguous()
        if self._proj_t is None:
            self._proj_t = (self.model.vision_projection.weight.float().t().contiguous(),
                            self.model.vision_projection.bias.float().contiguous())
        return ops.itc_head(q, self._proj_t[0], self._proj_t[1], text)

    def check_numerics(self) -> None:
        """Raise if a split-precision f32 GEMM of the Q-Former met an operand outside f16's range since the last check (its result is
        void).  One small D2H read: the harness calls it where it synchronises anyway (episode ends, after a timed region)."""
        from . import ops

        flag = ops.gemm_f32_overflow_flag(self.device, "blip2")
        if int(flag.item()):
            flag.zero_()
            ops.split_weights_bad(self.device, "blip2")    # (a weight out of range keeps raising the flag on every later use)
            raise FloatingPointError("BLIP-2 Q-Former: an activation left f16's range inside a split-precision f32 GEMM; set "
                                     "vlfm_amd.vlm.blip2itm.QFORMER_SPLIT_MIN_ROWS = 1 << 60 to run these layers on the library's f32 GEMMs")

    def cosine(self, image: np.ndarray, txt: str) -> float:
        """blip2itm.py:37-54: one RGB frame (H,W,3) u8 + prompt -> Python float."""
        img = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)[None]
        return float(self.cosine_batch(img, [txt])[0].item())


class BLIP2ITMClient:
    """Drop-in for the reference client (blip2itm.py:57-64): same constructor and ``cosine`` signature, but the
    model lives in this process (the ``port`` argument is accepted and ignored; there is no HTTP hop, no JPEG
    round trip -- pass ``emulate_jpeg=True`` to reproduce the reference's q90 transport for A/B fidelity checks)."""

    _shared: Dict[str, BLIP2ITM] = {}

    def __init__(self, port: int = 12182, device=None, emulate_jpeg: bool = False, **model_kwargs) -> None:
        key = str(device)
        if key not in BLIP2ITMClient._shared:
            BLIP2ITMClient._shared[key] = BLIP2ITM(device=device, **model_kwargs)
        self._model = BLIP2ITMClient._shared[key]
        self._emulate_jpeg = emulate_jpeg
        self.url = f"inprocess://blip2itm (port {port} ignored)"

    def cosine(self, image: np.ndarray, txt: str) -> float:
        if self._emulate_jpeg:
            from .transport import jpeg_roundtrip

            image = jpeg_roundtrip(image)  # server_wrapper.py:57-68
        return self._model.cosine(image, txt)
