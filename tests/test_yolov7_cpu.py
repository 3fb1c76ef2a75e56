"""The YOLOv7-E6E graph the reference deploys (/root/reference/vlfm/vlm/yolov7.py:33-47; graph and checkpoint format live in
the un-vendored WongKinYiu/yolov7 repository [ext]): the published figures of the network, the checkpoint's key layout, the
strict state-dict loader (unfused and fused files) and the reader that opens a pickled yolov7 ``.pt`` without the repository.
The real ``yolov7-e6e.pt`` cannot be downloaded here: the file formats are exercised on synthetic checkpoints with the real
key names, shapes and class paths."""
import sys
import types

import pytest
import torch
import torch.nn as nn

from vlfm_amd.vlm import yolov7_e6e as e6e


@pytest.fixture(scope="module")
def meta_model():
    with torch.device("meta"):
        return e6e.YoloV7E6E()


def test_published_figures_of_yolov7_e6e(meta_model):
    """151.7 M parameters, 843.2 GFLOPs at 1280 x 1280 (yolov7 README), each within 0.5 %; 17 850 candidates at the
    reference's 448 x 640 input (strides 8 / 16 / 32 / 64, 3 anchors)."""
    n = e6e.count_parameters(meta_model)
    assert abs(n / 1e6 - e6e.PUBLISHED["params_M"]) / e6e.PUBLISHED["params_M"] < 0.005, n
    g = e6e.gflops(meta_model, 1280, 1280)
    assert abs(g - e6e.PUBLISHED["gflops_at_1280x1280"]) / e6e.PUBLISHED["gflops_at_1280x1280"] < 0.005, g
    with torch.inference_mode():
        out = meta_model(torch.zeros((2, 3, 448, 640), device="meta"))
    assert tuple(out.shape) == (2, 17850, 85)
    assert 17850 == 3 * (56 * 80 + 28 * 40 + 14 * 20 + 7 * 10)


def test_module_indices_and_state_dict_keys_are_the_checkpoints(meta_model):
    """cfg/deploy/yolov7-e6e.yaml: 262 modules; stage boundaries, routes and the head's inputs at the yaml's indices; keys
    `model.{i}.…` as `ckpt['model'].state_dict()` has them."""
    m = meta_model.model
    assert len(m) == 262
    kinds = {i: type(m[i]).__name__ for i in range(262)}
    assert kinds[0] == "ReOrg" and kinds[1] == "Conv" and kinds[112] == "SPPCSPC" and kinds[261] == "Detect"
    assert [i for i, k in kinds.items() if k == "DownC"] == [2, 24, 46, 68, 90, 188, 211, 234]
    assert [i for i, k in kinds.items() if k == "Shortcut"] == [23, 45, 67, 89, 111, 137, 162, 187, 210, 233, 256]
    assert [i for i, k in kinds.items() if k == "Upsample"] == [114, 139, 164]
    assert meta_model.froms[115] == 89 and meta_model.froms[140] == 67 and meta_model.froms[165] == 45     # backbone routes
    assert meta_model.froms[189] == [-1, 162] and meta_model.froms[212] == [-1, 137] and meta_model.froms[235] == [-1, 112]
    assert [meta_model.froms[i] for i in (257, 258, 259, 260)] == [187, 210, 233, 256]
    assert meta_model.froms[261] == [257, 258, 259, 260]
    assert [m[i].conv.out_channels for i in (1, 12, 34, 56, 78, 100)] == [80, 160, 320, 640, 960, 1280]
    assert [m[i].conv.out_channels for i in (257, 258, 259, 260)] == [320, 640, 960, 1280]
    assert m[11].__class__.__name__ == "Concat" and meta_model.froms[11] == [-1, -3, -5, -7, -8]
    assert meta_model.froms[125] == [-1, -2, -3, -4, -5, -6, -7, -8] and m[126].conv.in_channels == 6 * 192 + 2 * 384
    sd = meta_model.state_dict()
    for k, shape in (("model.1.conv.weight", (80, 12, 3, 3)), ("model.1.bn.running_var", (80,)),
                     ("model.2.cv1.conv.weight", (80, 80, 1, 1)), ("model.2.cv2.conv.weight", (80, 80, 3, 3)),
                     ("model.2.cv3.bn.weight", (80,)), ("model.112.cv7.conv.weight", (640, 1280, 1, 1)),
                     ("model.112.cv5.conv.weight", (640, 2560, 1, 1)), ("model.260.conv.weight", (1280, 640, 3, 3)),
                     ("model.261.m.0.weight", (255, 320, 1, 1)), ("model.261.m.3.bias", (255,)),
                     ("model.261.anchors", (4, 3, 2)), ("model.261.anchor_grid", (4, 1, 3, 1, 1, 2))):
        assert tuple(sd[k].shape) == shape, k
    assert not any(k.startswith(("model.0.", "model.11.", "model.23.", "model.114.")) for k in sd)   # parameter-free modules
    assert len(e6e.expected_state_dict_keys(meta_model, fused=True)) < len(sd)


def _tiny(seed=0):
    torch.manual_seed(seed)
    net = e6e.YoloV7E6E(width_multiple=0.1).init_random(seed).eval()
    with torch.no_grad():   # non-trivial BatchNorm statistics, so that a wrong fold would show
        for mod in net.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.1, 0.1); mod.running_var.uniform_(0.5, 1.5)
    return net


def test_miniature_runs_and_fusing_keeps_the_function():
    net = _tiny()
    x = torch.rand(1, 3, 128, 192)
    with torch.inference_mode():
        want = net(x)
        assert want.shape == (1, 3 * (16 * 24 + 8 * 12 + 4 * 6 + 2 * 3), 85)
        got = _tiny().fuse_()(x)
    assert not any(isinstance(m, nn.BatchNorm2d) for m in _tiny().fuse_().modules())
    assert torch.allclose(got, want, atol=2e-4, rtol=2e-4), float((got - want).abs().max())
    # the decode: xy inside the padded image, wh positive, scores in (0, 1)
    assert (want[..., 2:4] > 0).all() and (want[..., 4:] > 0).all() and (want[..., 4:] < 1).all()


def test_state_dict_loader_is_strict_both_ways_and_takes_fused_files():
    src = _tiny(1)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    dst = e6e.YoloV7E6E(width_multiple=0.1).eval()
    assert e6e.load_yolov7_state_dict(dst, dict(sd)) == "unfused"
    x = torch.rand(1, 3, 64, 64)
    with torch.inference_mode():
        want = src(x)
        assert torch.equal(dst(x), want)
    # DataParallel prefix and num_batches_tracked are tolerated
    e6e.load_yolov7_state_dict(e6e.YoloV7E6E(width_multiple=0.1), {"module." + k: v for k, v in sd.items()})
    missing = dict(sd); missing.pop("model.34.conv.weight")
    with pytest.raises(KeyError, match="model.34.conv.weight"):
        e6e.load_yolov7_state_dict(e6e.YoloV7E6E(width_multiple=0.1), missing)
    extra = dict(sd); extra["model.262.conv.weight"] = torch.zeros(1)
    with pytest.raises(KeyError, match="no place"):
        e6e.load_yolov7_state_dict(e6e.YoloV7E6E(width_multiple=0.1), extra)
    wrong = dict(sd); wrong["model.1.conv.weight"] = torch.zeros(8, 12, 1, 1)
    with pytest.raises(ValueError, match="model.1.conv.weight"):
        e6e.load_yolov7_state_dict(e6e.YoloV7E6E(width_multiple=0.1), wrong)
    training = dict(sd); training["model.261.ia.0.implicit"] = torch.zeros(1, 8, 1, 1)
    with pytest.raises(ValueError, match="training"):
        e6e.load_yolov7_state_dict(e6e.YoloV7E6E(width_multiple=0.1), training)
    # a state dict taken AFTER yolov7's fuse(): conv.weight + conv.bias, no bn.* -- same function
    from torch.nn.utils.fusion import fuse_conv_bn_eval

    fused_sd = {}
    for name, mod in src.named_modules():
        if isinstance(mod, e6e.Conv):
            f = fuse_conv_bn_eval(mod.conv, mod.bn)
            fused_sd[name + ".conv.weight"], fused_sd[name + ".conv.bias"] = f.weight.detach(), f.bias.detach()
    fused_sd.update({k: v for k, v in sd.items() if k.startswith("model.261.")})
    dst2 = e6e.YoloV7E6E(width_multiple=0.1).eval()
    assert e6e.load_yolov7_state_dict(dst2, fused_sd) == "fused"
    with torch.inference_mode():
        got = dst2(x)
    assert torch.allclose(got, want, atol=2e-4, rtol=2e-4), float((got - want).abs().max())
    assert sorted(fused_sd) == sorted(e6e.expected_state_dict_keys(e6e.YoloV7E6E(width_multiple=0.1), fused=True))


def test_a_pickled_yolov7_checkpoint_opens_without_the_yolov7_repository(tmp_path):
    """`yolov7-e6e.pt` = torch.save({'model': <models.yolo.Model>, 'ema': None, ...}).  Built here with stand-in classes
    registered under the repository's module paths, saved, the stand-ins REMOVED from sys.modules (as on a machine without
    the repository), then read back: same tensors, loadable into the graph; anything outside tensors / containers / the
    repository's namespaces is refused."""
    src = _tiny(2)
    names = {"models": types.ModuleType("models"), "models.yolo": types.ModuleType("models.yolo"),
             "models.common": types.ModuleType("models.common")}

    def shell(mod, name):
        cls = type(name, (nn.Module,), {"__module__": mod})
        setattr(names[mod], name, cls)
        return cls

    Model = shell("models.yolo", "Model")
    kinds = {}
    sys.modules.update(names)
    try:
        def rebuild(mod: nn.Module) -> nn.Module:   # the same tree under the repository's class paths
            cname = type(mod).__name__
            if isinstance(mod, (nn.Conv2d, nn.BatchNorm2d, nn.SiLU, nn.MaxPool2d, nn.Upsample, nn.ModuleList)) and \
                    type(mod).__module__.startswith("torch"):
                out = mod
                if isinstance(mod, nn.ModuleList):
                    out = nn.ModuleList([rebuild(c) for c in mod])
                return out
            where = "models.yolo" if cname == "Detect" else "models.common"
            if cname not in kinds:
                kinds[cname] = shell(where, cname)
            cls = kinds[cname]
            out = cls()
            for k, v in mod._buffers.items():
                out.register_buffer(k, v)
            for k, c in mod._modules.items():
                out.add_module(k, rebuild(c))
            return out

        model = Model()
        model.add_module("model", nn.Sequential(*[rebuild(c) for c in src.model]))
        path = str(tmp_path / "yolov7-e6e.pt")
        torch.save({"model": model.half(), "ema": None, "epoch": -1, "optimizer": None, "training_results": None}, path)
    finally:
        for k in names:
            sys.modules.pop(k, None)
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=False)     # the plain way needs the repository
    sd = e6e.read_yolov7_checkpoint(path)
    want = src.float().state_dict()
    assert sorted(sd) == sorted(want)
    assert all(torch.equal(sd[k].float(), want[k].half().float() if want[k].is_floating_point() else want[k]) for k in want)
    dst = e6e.YoloV7E6E(width_multiple=0.1)
    assert e6e.load_yolov7_state_dict(dst, sd) == "unfused"
    # a plain state-dict file goes through the same reader
    torch.save(want, str(tmp_path / "sd.pt"))
    assert sorted(e6e.read_yolov7_checkpoint(str(tmp_path / "sd.pt"))) == sorted(want)

    class Evil:
        def __reduce__(self):
            import os

            return (os.system, ("echo pwned",))

    torch.save({"model": Evil()}, str(tmp_path / "evil.pt"))
    with pytest.raises(Exception, match="refusing"):
        e6e.read_yolov7_checkpoint(str(tmp_path / "evil.pt"))


@pytest.mark.parametrize("target", ["torch.utils.collect_env.run", "torch.hub.load", "torch.load", "numpy.load",
                                    "builtins.getattr", "builtins.eval", "torch.storage._load_from_bytes", "os.system"])
def test_the_checkpoint_reader_refuses_callables_inside_torch_and_numpy(tmp_path, target):
    """a prefix rule ("torch.*", "numpy*") lets REDUCE call torch.utils.collect_env.run(<shell command>).  The
    allowlist is exact: each of these globals must be refused BEFORE anything is called (the marker file must not appear)."""
    import importlib

    mod, name = target.rsplit(".", 1)
    fn = getattr(importlib.import_module(mod), name)
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (fn, (f"touch {marker}",))

    torch.save({"model": Evil()}, str(tmp_path / "evil.pt"))
    with pytest.raises(Exception, match="refusing"):
        e6e.read_yolov7_checkpoint(str(tmp_path / "evil.pt"))
    assert not marker.exists()
