"""Where does a workgroup of gemm_f16_8p_kernel spend its time, and how do the workgroups of a launch line up on the 256 CUs?
Diagnostic build (-DVLFM_PHASE_TIMING: only gemm_f16.hip is recompiled); every workgroup stamps entry / prologue done / main loop
done / epilogue done (stores drained) on the 100 MHz wall clock.
    python tools/gemm_stamp_probe.py [variant]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vlfm_amd import _lib
_lib.build()
csrc = os.path.join(ROOT, "vlfm_amd", "csrc")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
obj, out = os.path.join(ROOT, "gpurun_out", "gemm_f16_phase.o"), os.path.join(ROOT, "gpurun_out", "libvlfm_amd_gemm_phase.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DVLFM_PHASE_TIMING",
                       "-c", os.path.join(csrc, "gemm_f16.hip"), "-o", obj])
objs = [os.path.join(csrc, "build", f) for f in os.listdir(os.path.join(csrc, "build")) if f.endswith(".o") and not f.startswith("gemm_f16.")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + objs)
os.environ["VLFM_LIB_PATH"] = out
import importlib
importlib.reload(_lib)
import numpy as np, torch
L = _lib.lib()
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "3"
os.environ["VLFM_GEMM_VARIANT"] = variant
M = 256 * 257
for name, N, K, epi in [("fc1+gelu", 6144, 1408, 1), ("qkv", 4224, 1408, 0), ("proj+=", 1408, 1408, 2), ("fc2+=", 1408, 6144, 2)]:
    x = torch.randn(M, K, device=dev).half() * 0.3; w = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev).half()
    c = torch.randn(M, N, device=dev).half()
    nwg = ((M + 255) // 256) * ((N + 255) // 256)
    for rep in range(3):
        _lib.check(L.vlfm_gemm_f16_nt(x.data_ptr(), w.data_ptr(), b.data_ptr() if epi != 2 else None, c.data_ptr(), M, N, K, epi,
                                      torch.cuda.current_stream().cuda_stream), "gemm")
        torch.cuda.synchronize()
    n = min(nwg, 8192)
    buf = np.zeros(4 * n, np.int64)
    L.vlfm_debug_gemm_clocks(ctypes.c_void_p(buf.ctypes.data), n)
    t = buf.reshape(n, 4).astype(np.float64) * 0.01          # us
    t0 = t[:, 0].min()
    pro, loop, epi_t, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    start = t[:, 0] - t0
    end = t[:, 3] - t0
    first = start < 3.0                                       # the first wave of workgroups
    print(f"{name:9s} N={N} K={K} variant {variant}: {nwg} workgroups ({n} stamped), launch span {end.max():.1f} us")
    print(f"   per workgroup (median / p10 / p90 us): prologue {np.median(pro):.2f} / {np.percentile(pro,10):.2f} / {np.percentile(pro,90):.2f}   "
          f"main loop {np.median(loop):.2f} / {np.percentile(loop,10):.2f} / {np.percentile(loop,90):.2f} ({np.median(loop) / (K // 64):.3f} us per K-tile)   "
          f"epilogue {np.median(epi_t):.2f} / {np.percentile(epi_t,10):.2f} / {np.percentile(epi_t,90):.2f}   total {np.median(tot):.2f}")
    print(f"   first wave ({int(first.sum())} workgroups): prologue {np.median(pro[first]):.2f}  main loop {np.median(loop[first]):.2f}  epilogue {np.median(epi_t[first]):.2f}")
    bb = np.zeros(1024, np.int64)
    L.vlfm_debug_gemm_barriers(ctypes.c_void_p(bb.ctypes.data))
    for g in (0, 1):
        arr, rel = bb[g * 512:g * 512 + 512:2][:60], bb[g * 512 + 1:g * 512 + 512:2][:60]
        work = arr[1:] - rel[:-1]          # release of barrier k -> arrival at barrier k + 1: this wavefront's own segment
        wait = rel - arr                   # time spent inside the barrier
        print(f"   workgroup 300, wavefront {4 * g}: own segment cycles (release -> next arrival): " + " ".join(str(int(v)) for v in work[2:50]))
        print(f"   workgroup 300, wavefront {4 * g}: cycles waiting in the barrier:               " + " ".join(str(int(v)) for v in wait[3:51]))
    # rounds: how many workgroups are running at time T
    ts = np.linspace(0, end.max(), 60)
    running = [(int(((start <= T) & (end > T)).sum())) for T in ts]
    print("   workgroups in flight over the launch (60 samples): " + " ".join(str(r) for r in running))
    # half tiles (last n column when N % 256 <= 128): they are the LAST indices within each XCD
    short = tot < 0.7 * np.median(tot)
    print(f"   short workgroups (< 0.7 x median): {int(short.sum())}, their median total {np.median(tot[short]) if short.any() else 0:.2f} us; "
          f"idle tail: last start {start.max():.1f} us, busy CU-time / (256 x span) = {tot.sum() / (256 * end.max()):.3f}")
