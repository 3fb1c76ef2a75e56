"""Where does value_map_update_fused_kernel spend its time?  Builds a diagnostic library with -DVLFM_PHASE_TIMING
(only value_map.hip is recompiled, the other objects come from vlfm_amd/csrc/build) and prints the per-phase microseconds
of workgroup (0, 0), averaged over steps, for a few batch geometries.
    python tools/vm_phase_probe.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "vlfm_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "libvlfm_amd_vmphase.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
from vlfm_amd import _lib
_lib.build()
obj = os.path.join(ROOT, "gpurun_out", "value_map_phase.o")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DVLFM_PHASE_TIMING",
                       "-c", os.path.join(csrc, "value_map.hip"), "-o", obj])
objs = [os.path.join(csrc, "build", f) for f in os.listdir(os.path.join(csrc, "build")) if f.endswith(".o") and not f.startswith("value_map")]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + objs)
os.environ["VLFM_LIB_PATH"] = out
import importlib
importlib.reload(_lib)
import numpy as np, torch
from vlfm_amd.harness import BatchedEpisodes
names = ["keys->vertices", "flattened raster", "key hand-back", "resolve+visible+bbox", "dst box (1 lane)", "mask (written&~explored)", "fuse tiles"]
TARGET = int(os.environ.get("VLFM_VM_TARGET_WGS", "256"))     # (read once per process by the library: one process per value)
for (E, H, W, sync, wgs) in [(256, 480, 640, False, TARGET), (128, 480, 640, False, TARGET), (64, 480, 640, False, TARGET), (16, 720, 1280, True, TARGET),
                             (8, 480, 640, False, TARGET), (1, 480, 640, False, TARGET)]:
    sim = BatchedEpisodes(E, device=torch.device("cuda:0"), use_blip2=False, overlap=False, height=H, width=W, sync_explored=sync)
    sim.fast_forward(60)
    acc = np.zeros(7); n = 0; tile = np.zeros(3); spans = []
    _lib.lib().vlfm_profile_enable(1)
    for _ in range(20):
        sim.fast_forward(1); torch.cuda.synchronize()
        buf = np.zeros(16, np.int64)
        _lib.lib().vlfm_debug_vm_phase_clocks(ctypes.c_void_p(buf.ctypes.data))
        d = np.diff(buf[:8]) * 0.01
        if (d >= 0).all() and d.sum() < 1e4:
            acc += d; n += 1
            tile += np.diff(buf[8:12]) * 0.01
        sp = np.zeros((2048, 2), np.int64)
        _lib.lib().vlfm_debug_vm_span_clocks(ctypes.c_void_p(sp.ctypes.data))
        sp = sp[sp[:, 1] > sp[:, 0]]
        if len(sp):
            # per workgroup: its own first -> last stamp; and over the launch: first stamp of any workgroup -> last stamp of any
            spans.append((np.percentile((sp[:, 1] - sp[:, 0]) * 0.01, [50, 90, 100]), (sp[:, 1].max() - sp[:, 0].min()) * 0.01,
                          (sp[:, 0].max() - sp[:, 0].min()) * 0.01, len(sp)))
    ms, cnt = _lib.profile_read("value_map_update_fused_kernel")
    _lib.lib().vlfm_profile_enable(0)
    if spans:
        pct = np.mean([x[0] for x in spans], axis=0)
        print(f"E={E} {W}x{H} sync={sync}: {spans[-1][3]} workgroups, own first->last stamp 50/90/100 %: {pct[0]:.1f} / {pct[1]:.1f} / {pct[2]:.1f} us; "
              f"earliest first stamp -> latest last stamp {np.mean([x[1] for x in spans]):.1f} us; spread of the first stamps {np.mean([x[2] for x in spans]):.1f} us")
    print(f"E={E} {W}x{H} sync={sync} target_wgs={wgs}: kernel {ms * 1e3:.1f} us; workgroup (0,0): " + ", ".join(f"{nm}={a / max(n, 1):.1f}" for nm, a in zip(names, acc)) + f" (sum {acc.sum() / max(n, 1):.1f} us); wave 0, last batch: taps {tile[0] / max(n, 1):.2f}")
    del sim
