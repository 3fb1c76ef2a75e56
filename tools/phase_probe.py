"""Where do the obstacle-pipeline kernels spend their time?  Needs the diagnostic build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DVLFM_PHASE_TIMING \
          -o gpurun_out/libvlfm_amd_phase.so vlfm_amd/csrc/*.hip vlfm_amd/csrc/host.cpp
    VLFM_LIB_PATH=$PWD/gpurun_out/libvlfm_amd_phase.so python tools/phase_probe.py 8
Prints per-phase microseconds of workgroup 0 (100 MHz wall clock) averaged over steps."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vlfm_amd import _lib
if "VLFM_LIB_PATH" not in os.environ:   # build the diagnostic library here: only obstacle_map.hip is recompiled
    _lib.build()
    csrc = os.path.join(ROOT, "vlfm_amd", "csrc")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    obj, out = os.path.join(ROOT, "gpurun_out", "obstacle_map_phase.o"), os.path.join(ROOT, "gpurun_out", "libvlfm_amd_phase.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-DVLFM_PHASE_TIMING"] + ([f"-DVLFM_CUT_SKIP={os.environ['VLFM_CUT_SKIP']}"] if "VLFM_CUT_SKIP" in os.environ else [])
                          + ["-c", os.path.join(csrc, "obstacle_map.hip"), "-o", obj])
    objs = [os.path.join(csrc, "build", f) for f in os.listdir(os.path.join(csrc, "build"))
            if f.endswith(".o") and not f.startswith("obstacle_map")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + objs)
    os.environ["VLFM_LIB_PATH"] = out
    import importlib
    importlib.reload(_lib)
import numpy as np, torch
from vlfm_amd.harness import BatchedEpisodes
E = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sim = BatchedEpisodes(E, device=torch.device("cuda:0"), use_blip2=False, overlap=False)
sim.fast_forward(int(sys.argv[2]) if len(sys.argv) > 2 else 150)
BLOCK = int(sys.argv[3]) if len(sys.argv) > 3 else 0
_lib.lib().vlfm_debug_phase_block(BLOCK)
acc = np.zeros((3, 16)); n = 0
spans = np.zeros((3, 1024))
walk = np.zeros(3)
w0 = np.zeros(3, np.int64); _lib.lib().vlfm_debug_walk_stats(ctypes.c_void_p(w0.ctypes.data))  # reset
for _ in range(20):
    sim.step(); torch.cuda.synchronize()
    buf = np.zeros((3, 16), np.int64)
    _lib.lib().vlfm_debug_phase_clocks(ctypes.c_void_p(buf.ctypes.data))
    d = np.diff(buf, axis=1) * 0.01   # 100 MHz ticks -> us
    d[(d < 0) | (d > 1e5)] = 0
    acc[:, :15] += d; n += 1
    f1 = np.zeros((3, 1024), np.int64); l1 = np.zeros((3, 1024), np.int64)
    _lib.lib().vlfm_debug_wg_spans(ctypes.c_void_p(f1.ctypes.data), ctypes.c_void_p(l1.ctypes.data)); spans += (l1 - f1) * 0.01
    w = np.zeros(3, np.int64); _lib.lib().vlfm_debug_walk_stats(ctypes.c_void_p(w.ctypes.data)); walk += w
names = {0: ["cone raster", "window+masks", "obst contours", "shadow pts", "line ends", "cut lines", "visible contours+pick", "fill", "dilate+OR"],
         1: ["window copy", "scan+walk", "offset+pick", "publish"],
         2: ["dilate5 full planes", "small-unexplored filter", "window copy", "scan+walk", "offset", "bad flags", "pieces", "midpoints"]}
for k, nm in names.items():
    print(["fog_of_war", "explored_select", "frontier"][k], " ".join(f"{a}={acc[k, i] / n:.0f}us" for i, a in enumerate(nm)))
for k, nm in enumerate(["fog_of_war", "explored_select", "frontier"]):
    sp = spans[k, :E] / n
    print(f"{nm}: first-to-last stamp per workgroup: block {BLOCK} {sp[BLOCK]:.0f} us, mean {sp.mean():.0f} us, slowest block {int(sp.argmax())} {sp.max():.0f} us")
print(f"border walks, all envs and scans: {walk[0] * 0.01 / n / E:.0f} us per env-step inside follow_border, "
      f"{walk[1] / n / E:.0f} emitted points, {walk[2] / n / E:.1f} contours per env-step")
wc = np.zeros(16, np.int64); _lib.lib().vlfm_debug_parallel_walk_clocks(ctypes.c_void_p(wc.ctypes.data))
dw = np.diff(wc[:8]) * 0.01
if wc[12] == 2:   # the LDS-table form (border_parallel.h, round 4): stamps 0..5 of wg_build_rank_lds
    print("LDS-table follower, last build of workgroup 0 (frontier kernel): " + " ".join(f"{nm}={v:.0f}us" for nm, v in zip(
        ["border mask+pixel list", "states per pixel+scan", "successors+init", "ranking (all borders)", "lengths+clear labels",
         "first search", "first border out", "further borders + last search"], np.diff(wc[:9]) * 0.01)),
        f"| states={wc[13]} border pixels={wc[14]}")
else:
    print("parallel follower, last call of workgroup 0 (frontier kernel): " + " ".join(f"{nm}={v:.0f}us" for nm, v in zip(
        ["count+scan", "ranks", "successors", "(scan/short walk)", "init", "ranking", "emit"], dw)),
        f"| tables ok={wc[12]} states={wc[13]} border pixels={wc[14]}")
paths = np.zeros(4, np.int64); _lib.lib().vlfm_walk_path_counters(ctypes.c_void_p(paths.ctypes.data), 0)
print(f"borders from the LDS tables {paths[0]}, by one lane although the LDS tables existed {paths[2]}, windows with LDS tables {paths[3]} "
      f"(since the process started: {sim.steps_done if hasattr(sim, 'steps_done') else '?'} steps x {E} envs x 4 scans)")
print("frontier counts (frontiers, overflow, contours, chain points) env 0:", sim.obstacles.counts[0].tolist())
st = np.zeros((E, 8), np.int32)
ob = sim.obstacles
_lib.lib().vlfm_obstacle_status(ctypes.c_void_p(ob.scratch.data_ptr()), ob.n_envs, ob.size, ob.CAP_PTS, ob.CAP_CONTOURS, ctypes.c_void_p(st.ctypes.data))
print("fog status env 0 (overflow, obstacle contours, shadow lines, -) + select status (overflow, contours, chosen, refilled):", st[0].tolist())
for k, nm in enumerate(["fog_of_war", "explored_select", "frontier"]):
    sp = spans[k, :E] / n
    order = np.argsort(-sp)[:6]
    print(f"{nm}: slowest workgroups (block: us, fog+select status of the LAST step, frontier counts): " + "; ".join(
        f"{int(b)}: {sp[b]:.0f} {st[b].tolist()} {sim.obstacles.counts[b].tolist()}" for b in order))
    print(f"{nm}: span percentiles 50/90/99: {np.percentile(sp, 50):.0f} / {np.percentile(sp, 90):.0f} / {np.percentile(sp, 99):.0f} us")
bb = sim.obstacles.bbox.cpu().numpy() if hasattr(sim.obstacles, "bbox") else None
if bb is not None:
    hh, ww = bb[:, 1] - bb[:, 0] + 1, bb[:, 3] - bb[:, 2] + 1
    print(f"explored bounding boxes: rows {hh.min()}..{hh.max()} (mean {hh.mean():.0f}), cols {ww.min()}..{ww.max()} (mean {ww.mean():.0f})")
