"""The YOLOv7-E6E network the reference deploys (reference: /root/reference/vlfm/vlm/yolov7.py:33-47 --
``attempt_load("data/yolov7-e6e.pt")`` then ``TracedModel``; the graph itself lives in the un-vendored WongKinYiu/yolov7
repository [ext], ``cfg/deploy/yolov7-e6e.yaml`` + ``models/common.py`` / ``models/yolo.py``).

Restated from the published architecture so that

* the module list has the checkpoint's OWN indices: ``state_dict()`` keys are ``model.{i}.conv.weight``, ``model.{i}.bn.*``,
  ``model.{i}.cv1.conv.weight`` (DownC / SPPCSPC), ``model.261.m.{j}.weight`` ..., i.e. a state dict taken from
  ``yolov7-e6e.pt`` loads with ``strict=True`` (`load_yolov7_state_dict`, strict both ways, fused or unfused Conv+BN);
* the published figures of the YOLOv7-E6E row hold: 151.7 M parameters, 843.2 GFLOPs at 1280 x 1280 (tests/test_yolov7_cpu.py
  pins both within 0.5 %), 4 detection levels at strides 8 / 16 / 32 / 64 with 3 anchors each -> 17 850 candidates at the
  reference's 448 x 640 input (yolov7.py:71-76).

Layout of the 262 modules (index: module), as in the yaml:
  0 ReOrg | 1 Conv(3*4 -> 80, 3x3) | five stages, each = DownC + E-ELAN (two ELAN branches of 2 x 1x1 + 6 x 3x3 + concat + 1x1,
  summed by a Shortcut): 2-23 (160), 24-45 (320), 46-67 (640), 68-89 (960), 90-111 (1280) | 112 SPPCSPC(640) | top-down:
  113-137 (P5, 480), 138-162 (P4, 320), 163-187 (P3, 160), each = 1x1 + upsample + 1x1 lateral + concat + E-ELAN-W |
  bottom-up: 188-210 (P4, 320), 211-233 (P5, 480), 234-256 (P6, 640), each = DownC + concat + E-ELAN-W | 257-260 3x3 output
  convs (320 / 640 / 960 / 1280) | 261 Detect.

``yolov7-e6e.pt`` itself is a pickle of the yolov7 repository's classes (``models.yolo.Model``, ``models.common.Conv`` ...).
`read_yolov7_checkpoint` opens it WITHOUT that repository: a restricted unpickler maps every class of the repository's
``models.*`` / ``utils.*`` namespaces onto an inert ``nn.Module`` shell (pickle restores ``_modules`` / ``_parameters`` /
``_buffers`` through ``__dict__``, no constructor runs), and the state dict is read off the shells.  Everything else goes through
an EXACT (module, name) allowlist -- tensor / storage rebuilders, dtypes, containers, numpy scalars, ``torch.nn.modules`` classes --
because whatever ``find_class`` returns can be CALLED by the pickle (``torch.utils.*``, ``torch.hub``, ``numpy.load`` are refused).
"""
from __future__ import annotations

import io
import math
import pickle
from typing import Dict, List, Sequence, Tuple, Union

import torch
import torch.nn as nn

ANCHORS = [[19, 27, 44, 40, 38, 94], [96, 68, 86, 152, 180, 137], [140, 301, 303, 264, 238, 542],
           [436, 615, 739, 380, 925, 792]]          # cfg/deploy/yolov7-e6e.yaml [ext]
STRIDES = [8, 16, 32, 64]
PUBLISHED = {"params_M": 151.7, "gflops_at_1280x1280": 843.2}   # yolov7 README, YOLOv7-E6E row [ext]


# ------------------------------------------------------------------------------------------------ modules (models/common.py)
class Conv(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d + SiLU, 'same' padding (models/common.py Conv [ext])."""

    def __init__(self, c1: int, c2: int, k: int = 1, s: int = 1):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)   # yolov7 sets these in Model.__init__ (initialize_weights)
        self.act = nn.SiLU(inplace=True)

    hip_conv = False   # set by use_hip_conv_()

    def forward(self, x, out=None):
        """``out``: an NHWC-memory view [B, c2, Ho, Wo] to write into (a channel slice of a concatenation buffer); only on the
        HIP path (callers check ``takes_out(x)`` first)."""
        if self.takes_hip(x):
            from .det_ops import conv_nhwc

            c = self.conv
            if x.shape[1] != self.cin_p:     # the 12-channel stem: zero channels up to the filter's padded width
                x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, self.cin_p - x.shape[1]))
            y = conv_nhwc(x, self.w_rows, self.b_rows, c.kernel_size[0], c.stride[0], "silu", out=out)
            return y if y.shape[1] == c.out_channels else y[:, :c.out_channels]
        assert out is None
        return self.act(self.bn(self.conv(x)))

    def takes_hip(self, x) -> bool:
        return self.hip_conv and x.is_cuda and x.dtype == torch.float16

    def takes_out(self, x) -> bool:
        return self.takes_hip(x) and self.conv.out_channels % 8 == 0

    def out_hw(self, h: int, w: int):
        k, s = self.conv.kernel_size[0], self.conv.stride[0]
        return (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1

    def use_hip_conv_(self) -> bool:
        """After fuse_() and .half(): evaluate this layer as an implicit GEMM on the matrix cores with bias + SiLU in the epilogue
        (det_ops.conv_nhwc -> csrc/conv_nhwc.hip).  Every Conv of the e6e graph qualifies (1x1 / 3x3, stride 1 / 2)."""
        from .det_ops import conv_nhwc_supported, pack_conv_weight
        from .ops import BiasAct

        c = self.conv
        if (isinstance(self.bn, BiasAct) and self.bn.act == "silu" and c.weight.is_cuda and c.weight.dtype == torch.float16
                and conv_nhwc_supported(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, c.groups, c.dilation)):
            self.w_rows, self.b_rows = pack_conv_weight(c.weight, self.bn.bias)
            self.cin_p = (c.in_channels + 7) // 8 * 8
            self.hip_conv = True
        return self.hip_conv

    def fuse_(self) -> None:
        """yolov7's Model.fuse() [ext] for this triple: BatchNorm folded into the convolution's weights; the folded bias and
        the SiLU become ONE in-place pass (ops.BiasAct -> vlfm_bias_act_nchw; MIOpen convolutions take no bias)."""
        from torch.nn.utils.fusion import fuse_conv_bn_eval

        from .ops import BiasAct

        if not isinstance(self.bn, nn.BatchNorm2d):
            return
        fused = fuse_conv_bn_eval(self.conv.eval(), self.bn.eval())
        bias = fused.bias.detach().clone()
        fused.bias = None
        self.conv, self.bn, self.act = fused, BiasAct(bias, "silu", inplace=True), nn.Identity()


class ReOrg(nn.Module):
    def forward(self, x):  # space to depth: (b, c, h, w) -> (b, 4c, h/2, w/2)
        return torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)


class DownC(nn.Module):
    """cat(3x3 stride-k conv of a 1x1 conv, 1x1 conv of a k x k max-pool) (models/common.py DownC [ext])."""

    def __init__(self, c1: int, c2: int, k: int = 2):
        super().__init__()
        self.cv1 = Conv(c1, c1, 1, 1)
        self.cv2 = Conv(c1, c2 // 2, 3, k)
        self.cv3 = Conv(c1, c2 // 2, 1, 1)
        self.mp = nn.MaxPool2d(kernel_size=k, stride=k)

    def forward(self, x):
        t = self.cv1(x)
        if self.cv3.takes_hip(x) and self.mp.kernel_size == 2 and self.mp.stride == 2:
            from .det_ops import maxpool2x2_nhwc

            p = maxpool2x2_nhwc(x)
        else:
            p = self.mp(x)
        if self.cv2.takes_out(t) and self.cv3.takes_out(p) and self.cv2.out_hw(*t.shape[2:]) == tuple(p.shape[2:]):
            ca, cb = self.cv2.conv.out_channels, self.cv3.conv.out_channels     # both halves straight into the concatenation
            buf = torch.empty((x.shape[0], ca + cb, p.shape[2], p.shape[3]), dtype=x.dtype, device=x.device,
                              memory_format=torch.channels_last)
            self.cv2(t, out=buf[:, :ca])
            self.cv3(p, out=buf[:, ca:])
            return buf
        return torch.cat((self.cv2(t), self.cv3(p)), dim=1)


class SPPCSPC(nn.Module):
    """CSP spatial pyramid pooling (models/common.py SPPCSPC, e = 0.5, k = (5, 9, 13) [ext])."""

    def __init__(self, c1: int, c2: int, k: Sequence[int] = (5, 9, 13)):
        super().__init__()
        c_ = c2
        self.cv1, self.cv2 = Conv(c1, c_, 1, 1), Conv(c1, c_, 1, 1)
        self.cv3, self.cv4 = Conv(c_, c_, 3, 1), Conv(c_, c_, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])
        self.cv5, self.cv6 = Conv(4 * c_, c_, 1, 1), Conv(c_, c_, 3, 1)
        self.cv7 = Conv(2 * c_, c2, 1, 1)

    def forward(self, x):
        t = self.cv3(self.cv1(x))
        if self.cv4.takes_out(t) and self.cv6.takes_out(t) and self.cv2.takes_out(x):
            c_, (B, _, h, w) = self.cv4.conv.out_channels, x.shape
            pyr = torch.empty((B, 4 * c_, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            x1 = self.cv4(t, out=pyr[:, :c_])
            for i, m in enumerate(self.m):
                pyr[:, (i + 1) * c_:(i + 2) * c_].copy_(m(x1))
            two = torch.empty((B, 2 * c_, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            self.cv6(self.cv5(pyr), out=two[:, :c_])
            self.cv2(x, out=two[:, c_:])
            return self.cv7(two)
        x1 = self.cv4(t)
        y1 = self.cv6(self.cv5(torch.cat([x1] + [m(x1) for m in self.m], 1)))
        return self.cv7(torch.cat((y1, self.cv2(x)), dim=1))


class Concat(nn.Module):
    def forward(self, xs):
        return torch.cat(xs, 1)


class Shortcut(nn.Module):
    def forward(self, xs):
        return xs[0] + xs[1]


class Detect(nn.Module):
    """Inference form of models/yolo.py Detect [ext]: per level a 1x1 conv to na * (5 + nc) channels, sigmoid, grid decode
    ``xy = (2 s - 0.5 + grid) * stride``, ``wh = (2 s)^2 * anchor``; output [B, sum(na * h * w), 5 + nc] in input pixels."""

    def __init__(self, nc: int, anchors: Sequence[Sequence[int]], ch: Sequence[int]):
        super().__init__()
        self.nc, self.no, self.nl, self.na = nc, nc + 5, len(anchors), len(anchors[0]) // 2
        a = torch.tensor(anchors, dtype=torch.float32).view(self.nl, -1, 2)
        # the checkpoint stores `anchors` in units of the level's stride and `anchor_grid` in pixels (models/yolo.py)
        self.register_buffer("anchors", a / torch.tensor(STRIDES[: self.nl], dtype=torch.float32).view(-1, 1, 1))
        self.register_buffer("anchor_grid", a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.stride = STRIDES[: self.nl]
        self.hip_rows = None

    def use_hip_conv_(self) -> int:
        """The three / four 1x1 head convolutions (255 outputs, written as 256) on csrc/conv_nhwc.hip; f16 weights on the GPU."""
        from .det_ops import pack_conv_weight

        if all(m.weight.is_cuda and m.weight.dtype == torch.float16 for m in self.m):
            self.hip_rows = [pack_conv_weight(m.weight, m.bias) for m in self.m]
        return len(self.hip_rows or ())

    def forward(self, xs):
        z = []
        for i, x in enumerate(xs):
            if self.hip_rows is not None and x.is_cuda and x.dtype == torch.float16:
                from .det_ops import conv_nhwc

                y = conv_nhwc(x, self.hip_rows[i][0], self.hip_rows[i][1], 1, 1, None)[:, :self.no * self.na]
            else:
                y = self.m[i](x)
            b, _, ny, nx = y.shape
            y = y.view(b, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).sigmoid()
            gy, gx = torch.meshgrid(torch.arange(ny, device=y.device), torch.arange(nx, device=y.device), indexing="ij")
            grid = torch.stack((gx, gy), -1).view(1, 1, ny, nx, 2).to(y.dtype)
            xy = (y[..., 0:2] * 2.0 - 0.5 + grid) * self.stride[i]
            wh = (y[..., 2:4] * 2) ** 2 * self.anchor_grid[i].to(y.dtype)
            z.append(torch.cat((xy, wh, y[..., 4:]), -1).view(b, -1, self.no))
        return torch.cat(z, 1)


# ------------------------------------------------------------------------------------------------ the layer table
Spec = Tuple[Union[int, List[int]], str, tuple]   # (from, module, args) with yolov7's relative / absolute `from` convention


def _elan(specs: List[Spec], c_in_from: int, hidden: int, mid: int, out: int, taps: List[int]) -> None:
    """One ELAN branch: two 1x1 convs on the block input, six chained 3x3 convs, concat of `taps`, 1x1 to `out`."""
    specs.append((c_in_from, "Conv", (hidden, 1, 1)))
    specs.append((c_in_from - 1, "Conv", (hidden, 1, 1)))
    for _ in range(6):
        specs.append((-1, "Conv", (mid, 3, 1)))
    specs.append((taps, "Concat", ()))
    specs.append((-1, "Conv", (out, 1, 1)))


def _e_elan(specs: List[Spec], hidden: int, mid: int, out: int, taps: List[int]) -> None:
    """E-ELAN: two ELAN branches on the same input (the second reaches back over the first: from -11 / -12), summed."""
    _elan(specs, -1, hidden, mid, out, taps)
    _elan(specs, -11, hidden, mid, out, taps)
    specs.append(([-1, -11], "Shortcut", ()))


def layer_table(nc: int = 80) -> List[Spec]:
    s: List[Spec] = [(-1, "ReOrg", ()), (-1, "Conv", (80, 3, 1))]
    back = [-1, -3, -5, -7, -8]                       # backbone ELAN: every other 3x3 + the two 1x1
    for c, out in ((64, 160), (128, 320), (256, 640), (384, 960), (512, 1280)):
        s.append((-1, "DownC", (out,)))
        _e_elan(s, c, c, out, back)
    assert len(s) == 112
    s.append((-1, "SPPCSPC", (640,)))                  # 112
    head = [-1, -2, -3, -4, -5, -6, -7, -8]            # head ELAN-W: all six 3x3 + the two 1x1
    for c, route in ((480, 89), (320, 67), (160, 45)):  # top-down: P5, P4, P3
        s.append((-1, "Conv", (c, 1, 1)))
        s.append((-1, "Upsample", ()))
        s.append((route, "Conv", (c, 1, 1)))
        s.append(([-1, -2], "Concat", ()))
        _e_elan(s, int(c * 0.8), c * 2 // 5, c, head)
    assert len(s) == 188
    for c, route in ((320, 162), (480, 137), (640, 112)):   # bottom-up: P4, P5, P6
        s.append((-1, "DownC", (c,)))
        s.append(([-1, route], "Concat", ()))
        _e_elan(s, int(c * 0.8), c * 2 // 5, c, head)
    assert len(s) == 257
    for route, c in ((187, 320), (210, 640), (233, 960), (256, 1280)):
        s.append((route, "Conv", (c, 3, 1)))
    s.append(([257, 258, 259, 260], "Detect", (nc,)))
    assert len(s) == 262
    return s


class YoloV7E6E(nn.Module):
    """models/yolo.py Model for cfg/deploy/yolov7-e6e.yaml [ext]: `self.model` is the indexed module list; forward follows
    the `from` column.  Output of the Detect layer: [B, N, 5 + nc] (yolov7's inference tensor, `pred[0]` of the reference,
    yolov7.py:79)."""

    def __init__(self, nc: int = 80, ch: int = 3, width_multiple: float = 1.0):
        """``width_multiple`` is the yaml's channel multiplier (parse_model's ``make_divisible(c2 * gw, 8)`` [ext]); yolov7-e6e
        is 1.0 -- smaller values give same-topology miniatures for tests."""
        super().__init__()
        self.nc = nc
        self.ch_in, self.width_multiple = ch, width_multiple

        def w(c: int) -> int:
            return c if width_multiple == 1.0 else max(8, int(math.ceil(c * width_multiple / 8) * 8))

        table = [(f, kind, ((w(args[0]),) + tuple(args[1:])) if kind in ("Conv", "DownC", "SPPCSPC") else args)
                 for f, kind, args in layer_table(nc)]
        chans: List[int] = []
        mods: List[nn.Module] = []
        self.froms: List[Union[int, List[int]]] = []
        for i, (f, kind, args) in enumerate(table):
            def src(j):
                return ch if (j == -1 and i == 0) else chans[j if j >= 0 else i + j]
            if kind == "ReOrg":
                m, c2 = ReOrg(), src(f) * 4
            elif kind == "Conv":
                m, c2 = Conv(src(f), *args), args[0]
            elif kind == "DownC":
                m, c2 = DownC(src(f), args[0]), args[0]
            elif kind == "SPPCSPC":
                m, c2 = SPPCSPC(src(f), args[0]), args[0]
            elif kind == "Upsample":
                m, c2 = nn.Upsample(None, 2, "nearest"), src(f)
            elif kind == "Concat":
                m, c2 = Concat(), sum(src(j) for j in f)
            elif kind == "Shortcut":
                m, c2 = Shortcut(), src(f[0])
                assert src(f[0]) == src(f[1])
            elif kind == "Detect":
                m, c2 = Detect(args[0], ANCHORS, [src(j) for j in f]), 0
            else:  # pragma: no cover
                raise ValueError(kind)
            mods.append(m)
            chans.append(c2)
            self.froms.append(f)
        self.model = nn.ModuleList(mods)
        self._out_channels = list(chans)
        self.cat_plan, self.out_slot = {}, {}
        # which outputs are read again later (yolov7's `save` list): everything else is dropped as soon as it is consumed
        self.keep = set()
        for i, f in enumerate(self.froms):
            for j in ([f] if isinstance(f, int) else f):
                if j != -1:
                    self.keep.add(j if j >= 0 else i + j)

    def fuse_(self) -> "YoloV7E6E":
        for m in self.modules():
            if isinstance(m, Conv):
                m.fuse_()
        return self

    def use_hip_conv_(self) -> int:
        """NHWC execution: every convolution on csrc/conv_nhwc.hip, the rest of the graph (pooling, concatenation, upsampling,
        the box decode) on the framework's channels_last kernels.  Call after fuse_() and .half() on the GPU; returns the
        number of layers taken."""
        self.to(memory_format=torch.channels_last)
        n = sum(int(m.use_hip_conv_()) for m in self.modules() if isinstance(m, (Conv, Detect)))
        self._plan_concats()
        return n

    def init_random(self, seed: int = 0) -> "YoloV7E6E":
        """Kaiming-uniform convolutions (PyTorch's default), BatchNorm at identity, and yolov7's Detect._initialize_biases [ext]
        (objectness prior of ~8 objects per 640-px image, class prior 0.6 / nc): without it a random-init head passes half of
        the 17 850 candidates through conf > 0.25, which no trained detector does and which would make the NMS the dominant
        cost of a benchmark run."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                    bound = math.sqrt(3.0 / fan_in)     # variance-preserving through SiLU-ish activations, 262 layers deep
                    m.weight.copy_((torch.rand(m.weight.shape, generator=g, device="cpu") * 2 - 1).mul_(bound).to(m.weight.device))
            det = self.model[-1]
            for conv, stride in zip(det.m, det.stride):
                b = conv.bias.view(det.na, -1)
                b.zero_()
                b[:, 4] = math.log(8 / (640 / stride) ** 2)
                b[:, 5:] = math.log(0.6 / (self.nc - 0.99))
        return self

    def _plan_concats(self) -> None:
        """For the NHWC path: which top-level convolutions can write straight into the buffer of the Concat that reads them (the
        ELAN blocks concatenate five convolution outputs each; torch.cat would copy every one of them once more).  A producer
        feeds at most one buffer in place; whatever else a Concat reads (an Upsample, a block output, a producer already placed
        elsewhere) is copied into its slice at the Concat."""
        self.out_slot: Dict[int, Tuple[int, int, int]] = {}      # producer index -> (concat index, channel offset, total channels)
        self.cat_plan: Dict[int, List[Tuple[int, int, int, bool]]] = {}   # concat index -> [(source, offset, channels, in place)]
        chans = self._out_channels
        for i, (m, f) in enumerate(zip(self.model, self.froms)):
            if not isinstance(m, Concat):
                continue
            srcs = [i - 1 if j == -1 else (j if j >= 0 else i + j) for j in f]
            total, off, plan = sum(chans[j] for j in srcs), 0, []
            for j in srcs:
                prod = self.model[j]
                placed = (isinstance(prod, Conv) and prod.hip_conv and prod.conv.out_channels % 8 == 0 and off % 8 == 0
                          and total % 8 == 0 and j not in self.out_slot)
                if placed:
                    self.out_slot[j] = (i, off, total)
                plan.append((j, off, chans[j], placed))
                off += chans[j]
            if any(p[3] for p in plan):
                self.cat_plan[i] = plan

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        outs: Dict[int, torch.Tensor] = {}
        bufs: Dict[int, torch.Tensor] = {}       # concatenation buffers of this forward (NHWC path)
        planned = bool(self.cat_plan) and x.is_cuda and x.dtype == torch.float16
        for i, (m, f) in enumerate(zip(self.model, self.froms)):
            if isinstance(f, int):
                inp = x if f == -1 else outs[f if f >= 0 else i + f]
            else:
                inp = [x if j == -1 else outs[j if j >= 0 else i + j] for j in f]
            if planned and i in self.out_slot:
                ci, off, total = self.out_slot[i]
                if ci not in bufs:
                    ho, wo = m.out_hw(inp.shape[2], inp.shape[3])
                    bufs[ci] = torch.empty((inp.shape[0], total, ho, wo), dtype=inp.dtype, device=inp.device,
                                           memory_format=torch.channels_last)
                x = m(inp, out=bufs[ci][:, off:off + m.conv.out_channels])
            elif planned and i in self.cat_plan:
                buf = bufs.pop(i)
                for (j, off, c, placed), t in zip(self.cat_plan[i], inp):
                    if not placed:
                        buf[:, off:off + c].copy_(t)
                x = buf
            else:
                x = m(inp)
            if i in self.keep:
                outs[i] = x
        return x


def count_parameters(model: nn.Module) -> int:
    return sum(p.numel() for p in model.parameters())


def gflops(model: nn.Module, height: int, width: int) -> float:
    """2 x MACs of every convolution for one image of height x width (what yolov7's model_info / thop reports), from the
    layer shapes alone (meta tensors: nothing is allocated or computed)."""
    total = [0.0]
    hooks = []

    def hook(m, inp, out):
        total[0] += 2.0 * out.numel() * m.kernel_size[0] * m.kernel_size[1] * (m.in_channels // m.groups)

    if isinstance(model, YoloV7E6E):      # a parameter-free twin on the meta device: same layer shapes, no memory, no kernels
        with torch.device("meta"):
            model = YoloV7E6E(model.nc, model.ch_in, model.width_multiple).eval()
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    dev = next(model.parameters()).device
    with torch.inference_mode():
        model(torch.zeros((1, 3, height, width), device=dev))
    for h in hooks:
        h.remove()
    return total[0] / 1e9


# ------------------------------------------------------------------------------------------------ weights
def expected_state_dict_keys(model: YoloV7E6E, fused: bool = False) -> List[str]:
    keys = list(model.state_dict().keys())
    if fused:  # yolov7's Model.fuse(): bn folded into conv, which gains a bias
        keys = [k for k in keys if ".bn." not in k]
        keys += [k[: -len("weight")] + "bias" for k in keys if k.endswith(".conv.weight")]
    return keys


def load_yolov7_state_dict(model: YoloV7E6E, sd: Dict[str, torch.Tensor]) -> str:
    """Load a state dict taken from ``yolov7-e6e.pt`` (``ckpt['model'].state_dict()``, before or after ``fuse()``), strict
    both ways: every tensor of the file must be consumed and every parameter / buffer of the graph must be fed, with equal
    shapes.  Returns "unfused" or "fused".  A fused file is loaded by un-fusing trivially: BatchNorm becomes the identity
    (weight 1, bias = the conv bias, mean 0, var 1 - eps) so that the graph -- and the BatchNorm folding done afterwards --
    sees one form."""
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    own = model.state_dict()
    fused = not any(".bn." in k for k in sd)
    ignorable = {k for k in sd if k.endswith("num_batches_tracked")}
    if any(k.endswith(".implicit") for k in sd):
        raise ValueError("this state dict has IDetect's implicit layers: it is a *_training.pt model, not the deploy model "
                         "yolov7-e6e.pt the reference loads (vlfm/vlm/yolov7.py:35)")
    feed: Dict[str, torch.Tensor] = {}
    used = set()
    for k, t in own.items():
        if k.endswith("num_batches_tracked"):
            if k in sd:
                feed[k] = sd[k]
                used.add(k)
            continue
        if fused and ".bn." in k:
            conv_bias = sd.get(k.split(".bn.")[0] + ".conv.bias")
            if conv_bias is None:
                raise KeyError(f"fused state dict lacks {k.split('.bn.')[0]}.conv.bias")
            used.add(k.split(".bn.")[0] + ".conv.bias")
            leaf = k.rsplit(".", 1)[1]
            if leaf == "weight":
                feed[k] = torch.ones_like(t)
            elif leaf == "bias":
                feed[k] = conv_bias.to(t.dtype)
            elif leaf == "running_mean":
                feed[k] = torch.zeros_like(t)
            else:  # running_var: the fold computes w / sqrt(var + eps)
                feed[k] = torch.full_like(t, 1.0 - 1e-3)
            continue
        if k not in sd:
            raise KeyError(f"state dict lacks {k} (is this yolov7-e6e?  {len(sd)} tensors given, {len(own)} expected)")
        if tuple(sd[k].shape) != tuple(t.shape):
            raise ValueError(f"{k}: shape {tuple(sd[k].shape)} in the file, {tuple(t.shape)} in the yolov7-e6e graph")
        feed[k] = sd[k]
        used.add(k)
    extra = sorted(set(sd) - used - ignorable)
    if extra:
        raise KeyError(f"{len(extra)} tensors of the file have no place in the yolov7-e6e graph, e.g. {extra[:4]}")
    model.load_state_dict(feed, strict=False)
    return "fused" if fused else "unfused"


class _Shell(nn.Module):
    """What a class of the yolov7 repository becomes when its pickle is opened without the repository."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("a checkpoint shell cannot be executed; take its state_dict()")


class _CheckpointUnpickler(pickle.Unpickler):
    """An EXACT allowlist, because pickle's REDUCE opcode calls whatever ``find_class`` hands back: a prefix rule such as
    "anything under torch.*" lets ``torch.utils.collect_env.run`` (a shell), ``torch.hub.load``, ``numpy.load`` ... through.
    What a yolov7 ``.pt`` (or a state-dict file) needs is: the tensor / parameter / storage rebuilders, ``torch.Size``,
    dtypes and devices, ``OrderedDict``, numpy scalars (``best_fitness``), the ``torch.nn.modules`` CLASSES (instantiated
    through ``copyreg._reconstructor`` / NEWOBJ, whose constructors only allocate) and the repository's own classes, which
    become inert shells."""

    _EXACT = {
        ("collections", "OrderedDict"), ("copyreg", "_reconstructor"), ("copy_reg", "_reconstructor"),
        ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
        ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
        ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
        ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"), ("torch.storage", "_load_from_bytes"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "dtype"), ("numpy", "ndarray"), ("_codecs", "encode"),
    }
    _EXACT.discard(("torch.storage", "_load_from_bytes"))   # (it is torch.load again: never)
    _BUILTINS = {"set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes", "slice", "complex",
                 "object"}

    def find_class(self, module: str, name: str):
        root = module.split(".")[0]
        if root in ("models", "utils"):   # the yolov7 repository's own namespaces
            return type(name, (_Shell,), {"__module__": module})
        if module in ("builtins", "__builtin__"):   # (protocol-2 pickles, which torch.save writes, spell it the old way)
            if name in self._BUILTINS:
                return super().find_class(module, name)
        elif (module, name) in self._EXACT:
            return super().find_class(module, name)
        elif module == "torch" and "." not in name:
            obj = getattr(torch, name, None)
            if isinstance(obj, torch.dtype) or (isinstance(obj, type) and name.endswith("Storage")):
                return obj
        elif module.startswith("torch.nn.modules.") and "." not in name:
            obj = super().find_class(module, name)
            if isinstance(obj, type) and issubclass(obj, nn.Module):
                return obj
        raise pickle.UnpicklingError(f"refusing {module}.{name} in a checkpoint")


class _PickleModule:
    """`pickle_module` for torch.load: the standard module with the restricted Unpickler."""

    __name__ = "vlfm_amd_checkpoint_pickle"
    Unpickler = _CheckpointUnpickler
    load = staticmethod(lambda f, **kw: _CheckpointUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _CheckpointUnpickler(io.BytesIO(b), **kw).load())
    dump, dumps, HIGHEST_PROTOCOL, PickleError, UnpicklingError = (pickle.dump, pickle.dumps, pickle.HIGHEST_PROTOCOL,
                                                                   pickle.PickleError, pickle.UnpicklingError)


def read_yolov7_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """State dict (f32) of the model inside a yolov7 ``.pt`` (what ``attempt_load`` reads: ``ckpt['ema' if ckpt.get('ema')
    else 'model']``, models/experimental.py [ext]) -- or of a file that already IS a state dict."""
    ckpt = torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)
    if isinstance(ckpt, dict) and ("model" in ckpt or "ema" in ckpt):
        obj = ckpt["ema"] if ckpt.get("ema") is not None else ckpt["model"]
    else:
        obj = ckpt
    if isinstance(obj, nn.Module):
        sd = obj.state_dict()
    elif isinstance(obj, dict) and all(torch.is_tensor(v) for v in obj.values()):
        sd = obj
    else:
        raise ValueError(f"{path!r}: neither a yolov7 checkpoint nor a state dict ({type(obj).__name__})")
    return {k: v.float() if v.is_floating_point() else v for k, v in sd.items()}
