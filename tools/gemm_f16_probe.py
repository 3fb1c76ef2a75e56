"""vlfm_gemm_f16_nt (hand-written MFMA GEMMs of csrc/gemm_f16.hip) against PyTorch / hipBLASLt: correctness on ragged shapes, a
race screen (the 8-phase kernel keeps LDS-DMA loads in flight across barriers: a wrong wait count shows up as RARE wrong tiles), and
interleaved A/B timing against the library on the four ViT-g GEMM shapes at 256 images, random operands.  (Round 6: one kernel, the
persistent 8-phase schedule; the superseded variants of rounds 2-5 and their A/B numbers: DESIGN.md 6c / 6g, profiles/r05_gemm_probe*.)

    python tools/gemm_f16_probe.py [--quick] [--rounds 5]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vlfm_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument("--quick", action="store_true"); ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
_lib.build()
L = _lib.lib()
dev = torch.device("cuda:0")
VARIANTS = [7]
NAMES = {0: "ping-pong", 1: "lock-step", 2: "8-phase", 3: "8-phase-bal", 4: "8p-1bar", 5: "8p-bal-1bar", 6: "w4", 7: "8p-persist"}
def ours(x, w, b, epi, variant, out=None):
    M, K = x.shape; N = w.shape[0]
    c = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=dev)
    _lib.check(L.vlfm_gemm_f16_nt(x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, c.data_ptr(), M, N, K, epi,
                                  torch.cuda.current_stream().cuda_stream), "gemm")
    return c
torch.manual_seed(0)
bad = 0
# ---- correctness: asymmetric data, M and N tails, short and long K (1, 2, 3 K-tiles exercise the prologue / tail wait counts)
for (M, N, K) in [(300, 264, 128), (512, 512, 64), (256, 256, 192), (1000, 776, 1408), (130, 8, 64), (2500, 1408, 192), (777, 4224, 320),
                  (1028, 1408, 6144)]:
    x = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); b = torch.randn(N, device=dev).half()
    r0 = torch.randn(M, N, device=dev).half()
    ref = x.double() @ w.double().t() + b.double()
    for v in VARIANTS:
        for epi in (0, 1, 2):
            if epi == 2 and v < 2: continue
            want = F.gelu(ref) if epi == 1 else ref + r0.double() if epi == 2 else ref
            got = ours(x, w, b, epi, v, out=r0.clone() if epi == 2 else None).double()
            err = (got - want).abs().max().item(); scale = want.abs().max().item()
            ok = err <= 2e-3 * max(scale, 1) + 2e-3
            bad += not ok
            print(f"M={M} N={N} K={K} epi={epi} {NAMES[v]:11s}: max|err| {err:.3e} (max|ref| {scale:.2f})", "OK" if ok else "WRONG")
# ---- race screen: the same problem many times, bitwise against the first result (and that against f64)
for v in [v for v in VARIANTS if v >= 2]:  # (every kernel that keeps loads in flight across barriers)
    for (M, N, K) in [(4096, 4224, 1408), (2048, 1408, 6144), (8192, 6144, 1408)]:
        x = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); b = torch.randn(N, device=dev).half()
        first = ours(x, w, b, 0, v).clone()
        ref = (x.float() @ w.float().t() + b.float())
        e0 = (first.float() - ref).abs().max().item()
        diff = 0
        n_rep = 5 if args.quick else 40
        for _ in range(n_rep):
            diff += int((ours(x, w, b, 0, v) != first).sum().item())
        ok = diff == 0 and e0 < 0.05 * max(1.0, ref.abs().max().item() / 50)
        bad += not ok
        print(f"race screen {NAMES[v]:11s} M={M} N={N} K={K}: {n_rep} repeats, {diff} differing elements, max|first - f32 ref| {e0:.3e}", "OK" if ok else "WRONG")
# ---- timing on the real shapes (256 images), interleaved rounds, random operands
def t_once(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
M = 256 * 257
for name, N, K, epi in [("fc1+gelu", 6144, 1408, 1), ("qkv", 4224, 1408, 0), ("proj+=", 1408, 1408, 2), ("fc2+=", 1408, 6144, 2)]:
    x = torch.randn(M, K, device=dev).half() * (0.5 if epi != 2 else 0.2); w = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev).half()
    acc = torch.randn(M, N, device=dev).half()
    fl = 2.0 * M * N * K
    arms = {}
    if epi == 1: arms["library+gelu"] = lambda: F.gelu(F.linear(x, w, b))
    if epi == 0: arms["library"] = lambda: F.linear(x, w, b)
    if epi == 2: arms["library addmm_"] = lambda: acc.addmm_(x, w.t())
    for v in VARIANTS:
        if epi == 2 and v < 2: continue
        arms[NAMES[v]] = (lambda v=v: ours(x, w, b if epi != 2 else None, epi, v, out=acc if epi == 2 else None))
    for fn in arms.values():
        for _ in range(3): fn()
    res = {k: [] for k in arms}
    for _ in range(args.rounds):
        for k, fn in arms.items():
            res[k].append(t_once(fn, 5 if args.quick else 10))
    line = f"{name:9s} M={M} N={N} K={K}: "
    for k, ts in res.items():
        ts = sorted(ts); med = ts[len(ts) // 2]
        line += f"{k} {med*1e6:7.1f} us ({fl/med/1e15:.3f} PF, min {ts[0]*1e6:.1f}) | "
    print(line)
if not args.quick:
    # crossover against the library at small batches (fc1 + GELU), best 8-phase variant vs library
    for imgs in (1, 8, 16, 32, 64, 128):
        M = imgs * 257
        x = (torch.randn(M, 1408, device=dev) * 0.5).half(); w = (torch.randn(6144, 1408, device=dev) * 0.03).half(); b = torch.randn(6144, device=dev).half()
        t_lib = t_once(lambda: F.gelu(F.linear(x, w, b)), 30)
        line = f"fc1+gelu at {imgs:3d} images: library {t_lib*1e6:7.1f} us"
        for v in VARIANTS:
            line += f"  {NAMES[v]} {t_once(lambda: ours(x, w, b, 1, v), 30)*1e6:7.1f} us"
        print(line)
print("PROBE", "FAILED" if bad else "PASSED", f"({bad} wrong)")
