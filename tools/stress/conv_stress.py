"""vlfm_conv_nhwc_f16 on RANDOM shapes (every tile configuration pick_cfg can choose, ragged pixel / channel tails, stride 2 on odd sizes,
channel counts that are multiples of 8 but not of 64, batch 1..5) against torch's f32 convolution of the same f16 operands."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util, numpy as np, torch
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_conv_nhwc_gpu.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
from vlfm_amd.vlm import det_ops
dev = torch.device("cuda:0")
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([1, 3])); s = int(rng.choice([1, 1, 2])) if k == 3 else 1
    cin = int(rng.choice([8, 12, 16, 24, 40, 64, 72, 80, 128, 160, 192, 256, 320, 384, 480, 512, 640, 960, 1280]))
    cout = int(rng.choice([8, 16, 40, 64, 72, 80, 85, 128, 160, 192, 255, 256, 320, 384, 512, 640]))
    H, W = int(rng.integers(1, 70)), int(rng.integers(1, 90))
    B = int(rng.integers(1, 6))
    if cin * cout * k * k * H * W * B > 6e10: H, W = max(1, H // 4), max(1, W // 4)
    if not det_ops.conv_nhwc_supported(cin, cout, k, s):
        continue
    try:
        t.test_conv_nhwc_matches_torch_f32(dev, B, cin, cout, k, s, H, W)
    except AssertionError as e:
        bad += 1; print("seed", seed, (B, cin, cout, k, s, H, W), "FAILED", str(e)[:200])
    except Exception as e:
        bad += 1; print("seed", seed, (B, cin, cout, k, s, H, W), "RAISED", type(e).__name__, str(e)[:200])
print(f"conv_nhwc random shapes, seeds {a}..{b - 1}: {bad} failed")
