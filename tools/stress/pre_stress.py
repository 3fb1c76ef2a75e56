"""BLIP-2 / SAM / YOLO preprocessing against PIL and the oracle on RANDOM image sizes (odd sizes, upscaling, extreme aspect ratios)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from PIL import Image
from vlfm_amd.vlm import ops, det_ops
from oracle.ref_detect import resize_area_u8
dev = torch.device("cuda:0")
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
mean = torch.tensor(ops.CLIP_MEAN).view(3, 1, 1); std = torch.tensor(ops.CLIP_STD).view(3, 1, 1)
smean = torch.tensor(ops.SAM_MEAN).view(3, 1, 1); sstd = torch.tensor(ops.SAM_STD).view(3, 1, 1)
for seed in range(a, b):
    rng = np.random.default_rng(seed)
    H = int(rng.choice([rng.integers(8, 64), rng.integers(64, 400), rng.integers(400, 1100)]))
    W = int(rng.choice([rng.integers(8, 64), rng.integers(64, 400), rng.integers(400, 1400)]))
    n = int(rng.integers(1, 4))
    kind = seed % 4
    imgs = (rng.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8) if kind else
            np.broadcast_to(((np.arange(W)[None, :, None] * 7 + np.arange(H)[:, None, None] * 13) % 256).astype(np.uint8), (n, H, W, 3)).copy())
    if kind == 2: imgs[:] = np.where(imgs > 127, 255, 0)            # saturated edges: bicubic over/undershoot must clip like PIL
    t = torch.from_numpy(imgs).to(dev)
    try:
        out = ops.preprocess_rgb(t, 224, torch.float32).cpu()
        for i in range(n):
            pil = np.asarray(Image.fromarray(imgs[i]).resize((224, 224), Image.BICUBIC))
            want = (torch.from_numpy(pil.copy()).permute(2, 0, 1).float().div(255) - mean) / std
            if not torch.equal(out[i], want):
                bad += 1; print("blip2 preprocess differs", seed, (H, W), float((out[i] - want).abs().max())); break
        got, (oh, ow) = ops.preprocess_sam(t)
        for i in range(n):
            pil = np.asarray(Image.fromarray(imgs[i]).resize((ow, oh), Image.BILINEAR))
            want = (torch.from_numpy(pil.copy()).permute(2, 0, 1).float() - smean) / sstd
            if not (torch.equal(got[i, :, :oh, :ow].cpu(), want) and bool((got[i, :, oh:, :] == 0).all()) and bool((got[i, :, :, ow:] == 0).all())):
                bad += 1; print("sam preprocess differs", seed, (H, W), (oh, ow)); break
        if H >= 448 and W >= 640:        # INTER_AREA is the shrinking case the detector uses (yolov7.py:71-75)
            r = det_ops.resize_area(t, 448, 640, torch.float32).cpu()
            for i in range(n):
                want = torch.from_numpy(resize_area_u8(imgs[i], 640, 448)).permute(2, 0, 1).float() / 255.0
                if not torch.equal(r[i], want):
                    bad += 1; print("resize_area differs", seed, (H, W)); break
    except Exception as e:
        bad += 1; print("seed", seed, (H, W), "RAISED", type(e).__name__, str(e)[:200])
print(f"preprocessing seeds {a}..{b - 1}: {bad} failed")
