"""The tile shapes of csrc/conv_nhwc.hip (a 1x1 convolution IS an NT GEMM) on the ViT-g GEMM shapes of the headline step, next to
hipBLASLt: is the two-workgroups-per-CU 128x128 form that wins on the detector's layers a candidate for the ViT?  (Measured: no --
785-857 TFLOP/s against 968-1184.)  Usage: python tools/conv_as_gemm_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vlfm_amd.vlm import det_ops, ops
dev = torch.device("cuda:0")
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
M = 256 * 257
with torch.inference_mode():
    for name, K, N in (("fc1", 1408, 6144), ("qkv", 1408, 4224), ("proj", 1408, 1408), ("fc2", 6144, 1408)):
        x = torch.randn(256, K, 257, 1, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(N, K, 1, 1, device=dev, dtype=torch.float16) / K ** 0.5
        rows, bias = det_ops.pack_conv_weight(w, torch.randn(N, device=dev))
        x2 = x.permute(0, 2, 3, 1).reshape(M, K)
        flop = 2.0 * M * K * N
        ts = []
        for cfg in (0, 1, 3):
            os.environ["VLFM_CONV_CFG"] = str(cfg)
            ts.append(timed(lambda: det_ops.conv_nhwc(x, rows, bias, 1, 1, None)))
        os.environ.pop("VLFM_CONV_CFG")
        w2 = w.reshape(N, K).contiguous()
        t_lib = timed(lambda: torch.nn.functional.linear(x2, w2, bias))
        print(f"{name:5s} K={K} N={N}: conv kernel as GEMM 256x256 {flop/ts[0]/1e12:6.0f}  256x128 {flop/ts[1]/1e12:6.0f}  128x128 {flop/ts[2]/1e12:6.0f} TFLOP/s | hipBLASLt {flop/t_lib/1e12:6.0f}", flush=True)
