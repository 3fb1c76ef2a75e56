"""One fc1-shaped problem through the library GEMM and through vlfm_gemm_f16_nt, for tools/gemm_pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from vlfm_amd.vlm import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, K = 256 * 257, 6144, 1408
x = torch.nn.functional.layer_norm(torch.randn(M, K, device=dev), (K,)).half()
w = ((torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5).half(); b = torch.zeros(N, device=dev).half()
for _ in range(4):
    F.linear(x, w, b); ops.linear_gelu(x, w, b)
torch.cuda.synchronize()
