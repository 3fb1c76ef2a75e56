#!/usr/bin/env python
"""bench.py -- env-steps/s of the VLFM perception + value-map hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python bench.py --gpus 8 --steps 30 --warmup 5            # re-executes itself under torch.distributed.run, 8 ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                 # what the driver does; same code path from there on

One "step" = one ITMPolicyV2 perception+mapping step for every resident environment (BASELINE.json metric
"env-steps/s (VLM+value-map update), 640x480 RGB-D"): batched in-process BLIP-2 ITC cosine at the full ViT-g/14
geometry (random-init weights: no checkpoints offline), one depth-ingest pass, ObstacleMap update, ValueMap.update_map
and frontier scoring (sort_waypoints) -- see vlfm_amd/harness.py.  Observations are synthetic and already resident in
HBM when the timed region starts; before the warm-up every episode is advanced by --preroll map-only steps so that the
timed steps see a mid-episode map (explored area, obstacle planes, contour lengths), not an empty world.  Episodes are
independent: rank r owns the contiguous block of environments [r * envs, (r + 1) * envs) (weak scaling, --envs per GPU
fixed); the only collectives are the two metric all-reduces at the end (RCCL).

Prints ONE JSON line (rank 0) with the driver's keys plus (one roofline vocabulary: `achieved` = SURVEY.md 8d's ALGORITHMIC bytes of
the launch(es) / mean launch duration, `frac` = achieved / peak, `traffic` = HBM bytes per launch from the committed PMC runs or null;
launch durations are HIP events on the dispatch itself, every n-th launch inside the timed region):
  roofline             SURVEY 8d's value-map-update row (depth image + template + confidence / value RMW of the window) over the SUM
                       of the two launches that execute it: depth_ingest_kernel (column maxima, shared with the obstacle map) +
                       value_map_update_fused_kernel -- north_star's ">= 50 % of the HBM roofline on the map-fusion kernel"
  roofline_cfg5        the same row at BASELINE configs[4]'s per-GPU geometry (16 envs, 1280x720, full-map explored sync)
  roofline_depth_pass  the depth pass alone (4 H W bytes per observation)
  roofline_mfma        the ViT-g fc1 + GELU GEMM (csrc/gemm_f16.hip) against the dense f16 MFMA peak
  small_batch / full_step   the reference's own geometries (configs[1], [2], [3], [4] per GPU) and the PCIe-inclusive rate
  cpu_baseline         the oracle (NumPy + C restatement of the maps + the same ITC graph in fp32) on the host cores, bounded sample
DESIGN.md section 5 has the prose; the full record of a run goes to --detail.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md
# the full step's scripted detector head and episode shape (vlfm_amd/harness.py:ScriptedSightings): episodes of 12 initialisation
# turns + 60..179 search steps + 30 steps with the target in view (0.8 of them carry a surviving detection), then "arrival" = the
# episode ends and the next one starts in place; 1/16 of the search steps carry a distractor the filters must drop
SIGHTING_SCRIPT = dict(in_view_rate=0.8, distractor_rate=0.0625, search_min=60, search_span=120, nav_steps=30)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=256,
                    help="environments resident per GPU (weak scaling).  BASELINE configs[3]/[4] use 8 and 16 per GPU; "
                         "288 GB of HBM holds far more, and the batched BLIP-2 forward and the map kernels only reach "
                         "their efficient regime at >= 64 (the 8/GPU, 16/GPU-HD and 1/GPU figures are reported alongside)")
    ap.add_argument("--preroll", type=int, default=150,
                    help="map-only steps every episode is advanced by before the warm-up (mid-episode map state)")
    ap.add_argument("--no-small", action="store_true", help="skip the side measurements (small batches, config 5, PCIe)")
    ap.add_argument("--no-full", action="store_true", help="skip the configs[2] full-step side runs")
    ap.add_argument("--only-full", action="store_true", help="side runs: only the configs[2] full-step legs (tuning aid)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-blip2", action="store_true", help="map kernels only (NOT the headline metric)")
    ap.add_argument("--no-obstacle", action="store_true")
    ap.add_argument("--sync-explored", action="store_true", help="config 5: value map synchronised with explored area")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2, help="BLIP-2 fp32 forwards timed on the host")
    ap.add_argument("--no-overlap", action="store_true", help="run the obstacle pipeline on the BLIP-2 stream")
    ap.add_argument("--kernel-event-every", type=int, default=5,
                    help="HIP events around every n-th launch of each map kernel in the timed region (roofline.achieved)")
    ap.add_argument("--host-profile", action="store_true", help="cProfile the timed region (stderr), for tuning")
    ap.add_argument("--host-busy-cores", type=float, default=None,
                    help="--dry-run: host cores one rank keeps busy (default: the figure measured by the last committed "
                         "1-GPU run, profiles/host_busy.json); the dry run asserts ranks x this <= the cgroup CPU quota")
    ap.add_argument("--detail", default=os.path.join("gpurun_out", "bench_detail.json"),
                    help="where the FULL record goes (every leg, per-kernel tables, the prose): the printed line carries "
                         "numbers and short enum strings only and names this file")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher + process group + barriers + metric all-reduce around a stub step: no GPU work "
                         "(backend gloo when no GPU is visible) -- the multi-rank plumbing test of tests/")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relaunch_under_torchrun(gpus: int) -> None:
    """`python bench.py --gpus N` without a torchrun environment: become `python -m torch.distributed.run ... bench.py
    <same arguments>`, one rank per GPU, rendezvous on 127.0.0.1.  Never returns."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def usable_cores() -> int:
    """Cores this process may actually burn: affinity mask, capped by the cgroup CPU quota (cpu.max) when one is set --
    spinning more threads than the quota allows only gets the whole process throttled for the rest of the period."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_map_runs(height: int, width: int, with_obstacle: bool, warm: int, runs: int, per_run: int, env_id: int = 0,
                  stages: bool = False):
    """The oracle's maps on ONE core: ``warm`` untimed steps (they populate the confidence-mask cache, value_map.py:37),
    then ``runs`` x ``per_run`` timed steps of the same episode.  Returns (seconds per step of each run, stage seconds)."""
    import numpy as np

    from oracle.ref_obstacle_map import RefObstacleMap
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.synthetic import (MAX_DEPTH, MIN_DEPTH, camera_intrinsics, depth_from_profile, integrate, plan_actions,
                                    tf_of, wall_profile)

    fx, fy, fov = camera_intrinsics(width)
    # the same rooms-and-pillars episode the GPU harness steps through (environment env_id's tour offset)
    poses = integrate(plan_actions(1000))[(37 * env_id) % 500:]
    rng = np.random.Generator(np.random.PCG64(7 + env_id))
    cursor = [0]

    class env:  # noqa: N801 -- one observation per call, like SyntheticEnv.observe
        @staticmethod
        def observe():
            x, y, k = poses[cursor[0]]
            cursor[0] += 1
            return (depth_from_profile(wall_profile(x, y, k, width), height), tf_of(x, y, k), rng.uniform(0.15, 0.45, size=1))

    vm = RefValueMap(1, use_max_confidence=False)
    om = RefObstacleMap(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5) if with_obstacle else None
    stage_s: dict = {}
    restore = []
    if stages:
        import oracle.ref_obstacle_map as rom_
        import oracle.ref_value_map as rvm_

        def timed(mod, name, label):
            fn = getattr(mod, name)

            def wrapper(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    stage_s[label] = stage_s.get(label, 0.0) + time.perf_counter() - t0

            setattr(mod, name, wrapper)
            restore.append((mod, name, fn))

        timed(rvm_.RefValueMap, "_process_local_data", "depth profile + polygon cut")
        timed(rvm_, "rotate_about_centre", "rotate (warpAffine)")
        timed(rvm_, "paste_centred", "place")
        timed(rvm_.RefValueMap, "_fuse_new_data", "fuse (full-map NumPy passes)")
        timed(rvm_.RefValueMap, "sort_waypoints", "sort_waypoints")
        timed(rom_, "fill_small_holes", "fill_small_holes")
        timed(rom_, "unproject", "unproject")
        timed(rom_, "apply_tf", "transform_points")
        timed(rom_, "reveal_fog_of_war", "reveal_fog_of_war")
        timed(rom_, "detect_frontier_waypoints", "detect_frontier_waypoints")

    def one():
        depth, tf, values = env.observe()
        t0 = time.perf_counter()
        if om is not None:
            om.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
        vm.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fov)
        if om is not None and len(om.frontiers):
            vm.sort_waypoints(om.frontiers, 0.5)
        return time.perf_counter() - t0

    for _ in range(warm):
        one()
    stage_s.clear()
    per = []
    for _ in range(runs):
        per.append(float(np.sum([one() for _ in range(per_run)]) / per_run))
    for mod, name, fn in restore:
        setattr(mod, name, fn)
    return per, {k: v / (runs * per_run) for k, v in stage_s.items()}


def _cpu_worker(job):
    height, width, with_obstacle, env_id = job
    import torch

    torch.set_num_threads(1)
    per, _ = _cpu_map_runs(height, width, with_obstacle, warm=10, runs=1, per_run=40, env_id=env_id)
    return per[0]


def cpu_baseline(args, with_blip2: bool):
    """Reference-faithful CPU path on the host cores (SURVEY.md 8d protocol): oracle maps (NumPy + oracle/libcvport.so)
    single-process on 1 core -- the reference is single-env, single-threaded (vlfm_objectnav_hm3d.yaml:34,46-48) -- 20
    warm-up + 5 runs x 40 timed steps, median of the 5; the same on every usable core at once (one env per process) for
    the whole-box figure; and the same BLIP-2 ITC graph in fp32 through PyTorch-CPU (the reference's own fallback device,
    vlfm/vlm/blip2itm.py:26-27)."""
    import multiprocessing

    import numpy as np
    import torch

    from vlfm_amd.synthetic import rgb_frame

    quota = usable_cores()
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(1)
    per, stage = _cpu_map_runs(args.height, args.width, not args.no_obstacle, warm=20, runs=5, per_run=40, stages=True)
    map_s = float(np.median(per))
    # whole box: one single-env process per usable core, all running at once
    try:
        ctx = multiprocessing.get_context("spawn")
        t0 = time.perf_counter()
        with ctx.Pool(quota) as pool:
            per_proc = pool.map(_cpu_worker, [(args.height, args.width, not args.no_obstacle, 100 + i) for i in range(quota)])
        box = {"processes": quota, "maps_env_steps_per_s": round(float(sum(1.0 / p for p in per_proc)), 2),
               "ms_per_step_per_process": round(float(np.median(per_proc)) * 1e3, 2),
               "wall_s": round(time.perf_counter() - t0, 1)}
    except Exception as exc:  # noqa: BLE001
        box = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    blip_s = 0.0
    threads = min(quota, 16)  # more intra-op threads than this only oversubscribes the fp32 GEMMs
    if with_blip2:
        from PIL import Image

        from vlfm_amd.vlm.blip2itm import Blip2ITCConfig, Blip2ITCModel
        from vlfm_amd.vlm.ops import CLIP_MEAN, CLIP_STD

        torch.set_num_threads(threads)
        model = Blip2ITCModel(Blip2ITCConfig()).eval()
        with torch.no_grad():  # throughput does not depend on the weight values: cheap deterministic fill, no RNG
            for p in model.parameters():
                if p.dim() > 1:
                    p.copy_(((torch.arange(p.numel()) % 23).float() * 0.002 - 0.022).view(p.shape))
                else:
                    p.fill_(0.0)
            for m_ in model.modules():
                if isinstance(m_, torch.nn.LayerNorm):
                    m_.weight.fill_(1.0)
        rng = np.random.default_rng(0)
        ids = torch.randint(0, 30000, (1, 9))
        mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
        std = torch.tensor(CLIP_STD).view(3, 1, 1)
        times = []
        with torch.inference_mode():
            text = model.text_feature(ids)
            for i in range(args.cpu_steps + 1):
                rgb = rgb_frame(rng, args.height, args.width)
                t0 = time.perf_counter()
                pil = np.asarray(Image.fromarray(rgb).resize((224, 224), Image.BICUBIC))
                pix = ((torch.from_numpy(pil.copy()).permute(2, 0, 1).float().div(255) - mean) / std)[None]
                float(model.itc_reference_head(model.query_features(model.vision_tokens(pix)), text)[0])
                if i > 0:
                    times.append(time.perf_counter() - t0)
        blip_s = float(np.mean(times))
        torch.set_num_threads(1)
    total = map_s + blip_s
    return {
        "value": round(1.0 / total, 3), "unit": "env-steps/s", "cores": threads if with_blip2 else 1, "kind": "port",
        "sample": (f"1 env: 20 warm-up + 5 runs x 40 timed map steps (oracle ValueMap{'' if args.no_obstacle else '+ObstacleMap'}"
                   f"+sort_waypoints) on 1 core, median of the 5 runs = {map_s * 1e3:.1f} ms/step"
                   + (f"; + {args.cpu_steps} BLIP-2 ITC fp32 forwards on {threads} torch threads ({blip_s * 1e3:.0f} ms/frame)"
                      if with_blip2 else " (BLIP-2 leg skipped)")),
        "host": {"affinity_cores": affinity, "cgroup_quota_cores": quota},
        "maps_1core_env_steps_per_s": round(1.0 / map_s, 2),
        "maps_1core_runs_ms": [round(p * 1e3, 2) for p in per],
        "whole_box": box,
        "map_stage_ms_per_step": {k: round(v * 1e3, 3) for k, v in sorted(stage.items(), key=lambda kv: -kv[1])},
    }


# ------------------------------------------------------------------------------------------------ roofline helpers
MAP_KERNELS = ("depth_ingest_kernel", "depth_scatter_kernel", "depth_ingest_scatter_kernel", "fill_small_holes_kernel",
               "hole_scatter_kernel", "value_map_update_fused_kernel", "sort_waypoints_kernel", "resample_h_kernel", "resample_v_norm_kernel",
               "itc_head_kernel", "navigable_kernel", "fog_of_war_kernel", "explored_select_kernel",
               "frontier_prepare_kernel", "frontier_kernel")


LAUNCHES_TIMED = {}


def read_kernel_ms():
    from vlfm_amd import _lib

    out = {}
    for k in MAP_KERNELS:
        ms, n = _lib.profile_read(k)
        if n:
            out[k] = ms
            LAUNCHES_TIMED[k] = n
    return out


def load_pmc():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return {}


def survey_bytes(H, W, sync):
    """SURVEY.md 8d, algorithmic bytes per env-step (f32 maps, C = 1): the depth image once; the value-map row = depth + template +
    confidence RMW + value RMW of the 201 x 201 window (or explored plane + full-map RMW with the explored-area sync)."""
    T, S = 2 * int(5.0 * 20) + 1, 1000
    depth = 4 * H * W
    window = 4 * T * T + 16 * T * T
    full = 4 * T * T + S * S + 16 * S * S
    return {"depth": depth, "update": full if sync else window, "row": depth + (full if sync else window)}


def map_roofline(kms, E, H, W, sync, pmc):
    """One vocabulary for every HBM-priced figure: `achieved` = SURVEY 8d's ALGORITHMIC bytes of the launch / its mean duration
    (GB/s), `frac` = achieved / 8 TB/s, `traffic` = HBM bytes per launch from the committed PMC runs (profiles/pmc_traffic.json) or
    None.  Keys: the two kernels of the value-map row, and `row` = SURVEY 8d's value-map-update row (a6 + a9 + a10) priced over the
    SUM of the two launches that execute it (depth pass: column maxima, shared with the obstacle map; fused update) -- the figure
    north_star's ">= 50 % of the HBM roofline on the map-fusion kernel" is read against."""
    b = survey_bytes(H, W, sync)
    tag = f"@E={E},{W}x{H}" + (",sync" if sync else "")
    depth_k = next((k for k in ("depth_ingest_scatter_kernel", "depth_ingest_kernel") if k in kms), None)
    upd_k = "value_map_update_fused_kernel" if "value_map_update_fused_kernel" in kms else None

    def rec(kernel, nbytes, ms, traffic):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "launch_ms": round(ms, 5),
                "algorithmic_bytes_per_launch": int(nbytes)}

    def traffic_of(k):
        p = pmc.get(k + tag)
        return int(p["bytes_per_launch"]) if p else None

    out = {}
    if depth_k:
        out["depth_pass"] = rec(depth_k, E * b["depth"], kms[depth_k], traffic_of(depth_k))
    if upd_k:
        out["update"] = rec(upd_k, E * b["update"], kms[upd_k], traffic_of(upd_k))
    if depth_k and upd_k:
        t = [traffic_of(depth_k), traffic_of(upd_k)]
        out["row"] = rec(f"{depth_k} + {upd_k}", E * b["row"], kms[depth_k] + kms[upd_k], sum(t) if None not in t else None)
    return out


# ------------------------------------------------------------------------------------------------ dry run
def dry_run(args) -> None:
    """The multi-rank plumbing without any GPU work: same launcher, same process-group init, same barriers and metric
    all-reduces as the real run around a stub step (tests/test_distributed_cpu.py drives this with --gpus 2 on CPU)."""
    import torch

    from vlfm_amd import distributed as D

    rank, local_rank, world = D.world()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    use_gpu = torch.cuda.is_available()
    device = torch.device(f"cuda:{local_rank}") if use_gpu else torch.device("cpu")
    D.init("nccl" if use_gpu else "gloo", device if use_gpu else None)
    ids = D.shard_env_ids(rank, world, args.envs)
    D.barrier(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    D.barrier(device)
    elapsed = time.perf_counter() - t0
    elapsed_max, (env_steps, id_sum) = D.reduce_metrics(elapsed, [float(len(ids) * args.steps), float(sum(ids))], device)
    # host headroom: every rank drives its GPU from one Python thread; N ranks on one node must fit the CPU quota or the
    # cgroup throttles all of them (it did at 16 CPUs: tools/stall_probe.py).  The per-rank figure is measured, not assumed:
    # the 1-GPU run writes it into its JSON line (`host.busy_cores_per_rank`) and into profiles/host_busy.json.
    busy = args.host_busy_cores
    if busy is None:
        try:
            busy = float(json.load(open(os.path.join(ROOT, "profiles", "host_busy.json")))["busy_cores_per_rank"])
        except (OSError, ValueError, KeyError):
            busy = None
    quota = usable_cores()
    if busy is not None:
        assert world * busy <= quota, (f"{world} ranks x {busy:.2f} busy host cores each = {world * busy:.1f} > the CPU quota of "
                                       f"{quota} cores: the ranks would throttle one another")
    if rank == 0:
        print(json.dumps({"metric": "env-steps/s (DRY RUN: stub step, no GPU work)", "dry_run": True,
                          "host": {"busy_cores_per_rank": busy, "ranks": world, "cgroup_quota_cores": quota,
                                   "fits": None if busy is None else bool(world * busy <= quota)},
                          "value": round(env_steps / elapsed_max, 2), "unit": "env-steps/s", "n_gpus": world,
                          "ranks": world, "backend": "nccl (RCCL)" if use_gpu else "gloo", "steps": args.steps,
                          "warmup": args.warmup, "global_envs": args.envs * world, "env_id_checksum": id_sum,
                          "scaling": "weak"}), flush=True)
    D.shutdown()



# ------------------------------------------------------------------------------------------------ the printed line
LINE_BUDGET = 6000   # bytes; the driver reads the LAST 8 KB of stdout -- round 4's 30 KB line could not be parsed


def write_detail(out: dict, path: str):
    """The full record (prose, per-kernel tables, every leg) as a JSON FILE; returns the path written, or None."""
    try:
        full = path if os.path.isabs(path) else os.path.join(ROOT, path)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, "w") as f:
            json.dump(out, f, indent=1)
        return path
    except OSError:
        return None


def _num(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def _short_roofline(r):
    """Contract keys + the few figures a reader needs; numbers and enum strings only."""
    if not isinstance(r, dict):
        return None
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "algorithmic_bytes_per_launch",
            "launches_timed", "flop_per_launch")
    return {k: _num(r[k]) for k in keep if k in r}


def _leg(rec):
    if not isinstance(rec, dict):
        return None if rec is None else "FAILED"
    o = {"value": rec.get("value"), "ms_per_step": rec.get("ms_per_step")}
    dr = rec.get("detector_roofline")
    if isinstance(dr, dict):
        o["conv_mfma_frac"] = dr.get("frac")
    if "nms_candidates_per_frame" in rec:
        o["nms_candidates_per_frame"] = rec["nms_candidates_per_frame"]
    return o


def compact_line(out: dict) -> str:
    """ONE JSON line of at most LINE_BUDGET bytes from the full record: the driver's keys, `roofline` (+ the depth pass and the
    dominant MFMA kernel), `cpu_baseline`, one {value, ms_per_step} per side leg.  Everything else lives in `detail`."""
    c = out.get("config", {})
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": c.get("workload_short", c.get("workload")), "envs_per_gpu": c.get("envs_per_gpu"),
                      "global_envs": c.get("global_envs"), "rgbd": c.get("rgbd"), "map": c.get("map"),
                      "parallelism": c.get("parallelism_short", c.get("parallelism")),
                      "tuned_gemms": c.get("tuned_gemms"), "vit_gemm": c.get("vit_gemm"),
                      "attention": c.get("attention_short")}
    line["roofline"] = _short_roofline(out.get("roofline"))
    for k in ("roofline_depth_pass", "roofline_mfma"):
        if out.get(k):
            line[k] = _short_roofline(out[k])
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = ({"error": str(cb["error"])[:120]} if "error" in cb else
                                {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                 "kind": cb.get("kind"), "sample": cb.get("sample_short", str(cb.get("sample"))[:160]),
                                 "maps_1core": cb.get("maps_1core_env_steps_per_s"),
                                 "maps_whole_box": (cb.get("whole_box") or {}).get("maps_env_steps_per_s")})
    sb = out.get("small_batch")
    if isinstance(sb, dict):
        small = {}
        for k, v in sb.items():
            if k.startswith("envs_per_gpu="):
                tag = "envs=" + k.split("=")[1].split(" ")[0]
            elif k.startswith("configs[4]"):
                tag = "cfg5_16env_1280x720_sync"
            elif k.startswith("pcie_inclusive"):
                tag = "pcie_inclusive"
            elif k.endswith("(FAILED)"):
                tag = k[:40]
            else:
                continue   # per-phase kernel tables: detail only
            small[tag] = _leg(v)
            if tag == "cfg5_16env_1280x720_sync" and isinstance(v, dict) and v.get("roofline"):
                line["roofline_cfg5"] = _short_roofline(v["roofline"])     # SURVEY 8d's own ">= 4 TB/s at config 5" row
        line["small_batch"] = small
    fs = out.get("full_step")
    if isinstance(fs, dict):
        legs = {}
        for k, v in fs.items():
            if not k.startswith("configs[2] full step"):
                continue
            det = "gdino" if "GroundingDINO" in k else "yolov7_e6e"
            legs[f"{det},envs={k.rsplit('=', 1)[1]}"] = _leg(v)
        line["full_step"] = {"unit": "env-steps/s", **legs}
    h = out.get("host")
    if isinstance(h, dict):
        line["host"] = {"busy_cores_per_rank": h.get("busy_cores_per_rank"), "cgroup_quota_cores": h.get("cgroup_quota_cores")}
    line["detail"] = out.get("detail")
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > LINE_BUDGET:   # never print an unreadable line: shed the side legs, keep the contract
        for k in ("small_batch", "host", "roofline_mfma", "roofline_depth_pass", "roofline_cfg5", "full_step"):
            line.pop(k, None)
            txt = json.dumps(line, separators=(",", ":"))
            if len(txt) <= LINE_BUDGET:
                break
    assert "\n" not in txt
    return txt


# ------------------------------------------------------------------------------------------------ main
def tuned_gemms_applied() -> bool:
    """Whether vlfm_amd/tunableop_results.csv was accepted by this box's libraries (on another image it is silently ignored and
    the library GEMMs run hipBLASLt's own heuristic)."""
    try:
        from vlfm_amd.vlm import ops

        return bool(ops.use_tuned_gemms())
    except Exception:  # noqa: BLE001
        return False


def thread_cpu_seconds():
    """{tid: (comm, user+system CPU seconds)} of this process's threads (/proc/self/task): who is busy on the host."""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            raw = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        comm = raw[raw.index("(") + 1:raw.rindex(")")]
        f = raw[raw.rindex(")") + 2:].split()
        out[int(tid)] = (comm, (int(f[11]) + int(f[12])) / tick)
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)
    if args.dry_run:
        return dry_run(args)
    import torch
    import torch.distributed as dist

    from vlfm_amd import _lib
    from vlfm_amd import distributed as D

    rank, local_rank, world = D.world()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU"
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    # before the first stream exists: the runtime fixes a queue's wait policy when it creates the queue
    _lib.host_wait_blocking(device)
    D.init("nccl", device)  # backend "nccl" is RCCL on ROCm; only the metric all-reduces use it
    torch.set_num_threads(1)  # the GPU leg has no CPU tensor math; idle OpenMP spinners would only eat the CPU quota

    from vlfm_amd.harness import BatchedEpisodes

    have_obstacle = not args.no_obstacle
    common = dict(device=device, height=args.height, width=args.width, obstacle=have_obstacle,
                  overlap=not args.no_overlap)
    sim = BatchedEpisodes(args.envs, env_offset=D.shard_env_ids(rank, world, args.envs)[0],
                          use_blip2=not args.no_blip2, sync_explored=args.sync_explored, **common)
    if sim.blip2 is not None:
        sim.blip2.strict_hip_attention = True  # a silent library fallback would change what is being measured

    def barrier():
        D.barrier(device)

    sim.fast_forward(args.preroll)
    sim.prepare(args.warmup + args.steps)   # rooms world: the window's depth frames are rendered before the clock starts
    for _ in range(args.warmup):
        sim.step()
    first_timed_step = sim.t
    # dispatch-timestamp events on every `--kernel-event-every`-th launch of each libvlfm_amd kernel: a timed dispatch keeps
    # the runtime's completion thread (and the thread waiting for the stream) awake for the whole step - measured with
    # tools/host_busy_probe.py: every launch timed = 164.5 ms/step and 2.0 busy host cores, none = 160.8 ms and 1.0
    _lib.lib().vlfm_profile_enable(args.kernel_event_every)
    prof = None
    if args.host_profile:
        import cProfile

        prof = cProfile.Profile()
    barrier()
    t0 = time.perf_counter()
    cpu0 = time.process_time()   # CPU seconds of ALL threads of this rank (Python thread + HIP runtime helpers)
    threads0 = thread_cpu_seconds()
    if prof is not None:
        prof.enable()
    for _ in range(args.steps):
        sim.step()
    if prof is not None:
        prof.disable()
    barrier()
    elapsed = time.perf_counter() - t0
    host_cpu = time.process_time() - cpu0
    threads1 = thread_cpu_seconds()
    by_thread = sorted(((round((threads1[t][1] - threads0.get(t, (None, 0.0))[1]) / elapsed, 3), threads1[t][0])
                        for t in threads1), reverse=True)[:4]
    sim.check()  # a capacity overflow or an off-map obstacle point inside the timed region fails the benchmark
    if prof is not None and rank == 0:
        import pstats

        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(35)

    elapsed_max, (env_steps, host_cpu_sum, host_busy_sum) = D.reduce_metrics(
        elapsed, [float(args.envs * args.steps), host_cpu, host_cpu / elapsed], device)

    torch.cuda.synchronize(device)
    kms = read_kernel_ms()
    gemm_kernel = "gemm_f16_8pp_kernel<1>"
    gemm_ms, gemm_n = _lib.profile_read(gemm_kernel)
    _lib.lib().vlfm_profile_enable(0)
    if rank == 0:
        H, W, E = args.height, args.width, args.envs
        # north_star's roofline target is the map-fusion kernel: SURVEY 8d's value-map row over the two launches that execute it
        rf = map_roofline(kms, E, H, W, args.sync_explored, load_pmc())
        roofline = dict(rf.get("row") or rf.get("update") or {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS,
                                                               "unit": "GB/s", "frac": None, "traffic": None})
        roofline["update_kernel"] = rf.get("update")
        roofline["launches_timed"] = LAUNCHES_TIMED.get("value_map_update_fused_kernel", 0)
        roofline["all_kernels_ms"] = {k: round(v, 5) for k, v in kms.items()}
        roofline_depth = rf.get("depth_pass")
        # the kernel the STEP is made of: the ViT-g's fc1 + GELU GEMM on csrc/gemm_f16.hip (the other three GEMMs of a block are
        # hipBLASLt's and carry no events of ours), timed by the same dispatch events, priced against the dense f16 MFMA peak
        roofline_mfma = None
        if sim.blip2 is not None:
            gms, gn = gemm_ms, gemm_n
            if gn:
                cfg = sim.blip2.cfg
                rows = E * ((cfg.image_size // cfg.patch_size) ** 2 + 1)
                flop = 2.0 * rows * cfg.v_mlp * cfg.v_hidden
                tf = flop / (gms * 1e-3) / 1e12
                roofline_mfma = {"bound": "mfma", "kernel": gemm_kernel + " (ViT-g fc1 + erf-GELU, f16 in / f32 accumulate)",
                                 "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                                 "traffic": None, "launch_ms": round(gms, 5), "launches_timed": gn, "flop_per_launch": int(flop),
                                 "meaning": "2 M N K of the GEMM (the GELU's ~8 operations per output are not counted) / mean launch "
                                            "time / 2.5 PFLOP/s dense f16; 39 launches per step = ~28 % of the step's GPU time"}
        out = {
            "metric": "env-steps/s (VLM+value-map update), 640x480 RGB-D",
            "value": round(env_steps / elapsed_max, 2), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 ViT-g (f32 accumulate) + f32 Q-Former (split-f16, 22-bit); f32 confidence + f64 value maps", "data": "synthetic",
            "ranks": world, "backend": "nccl (RCCL)" if dist.is_initialized() else "single process, no process group",
            "config": {"workload": ("configs[1] step (BLIP-2 ITC cosine + ValueMap fusion"
                                    + (" + ObstacleMap update" if have_obstacle else "")
                                    + " + sort_waypoints) for every resident env, envs sharded over GPUs as in configs[3]"
                                    if not args.no_blip2 else "MAP KERNELS ONLY (no VLM) -- not the headline metric"),
                       "workload_short": ("configs[1] step x resident envs: BLIP-2 ITC + ValueMap"
                                          + ("+ObstacleMap" if have_obstacle else "") + "+sort_waypoints"
                                          if not args.no_blip2 else "MAP KERNELS ONLY (not the headline metric)"),
                       "parallelism_short": f"env-sharded x{world}",
                       "tuned_gemms": bool(tuned_gemms_applied()),
                       "vit_gemm": (None if sim.blip2 is None else sim.blip2.gemm_path(E) if hasattr(sim.blip2, "gemm_path")
                                    else "fc1:" + sim.blip2.mlp_path(E)),
                       "attention_short": (sim.blip2.attention_path if sim.blip2 is not None else None),
                       "envs_per_gpu": E, "global_envs": E * world, "rgbd": f"{W}x{H}", "map": "1000x1000 @ 20 px/m",
                       "world": ("rooms-and-pillars world ray-cast per environment (vlfm_amd/synthetic.py), tour offset 37*env mod 500"
                                 if sim.rooms is not None else "per-frame random wall profiles (SURVEY 8d)"),
                       "frontiers_per_env_last_step": sim.frontier_stats(),
                       "episode_steps_timed": [first_timed_step, first_timed_step + args.steps - 1],
                       "blip2": "ViT-g/14 39 blocks + Q-Former 12 layers, random-init" if not args.no_blip2 else None,
                       "attention_path": (sim.blip2.attention_path if sim.blip2 is not None else None),
                       "fc1_gelu_path": (sim.blip2.mlp_path(E) if sim.blip2 is not None else None),
                       "parallelism": f"env-sharded x{world} (contiguous blocks), metric all-reduce only"},
            "roofline": roofline,
            "roofline_depth_pass": roofline_depth,
            "roofline_mfma": roofline_mfma,
            # 8-GPU readiness without the node: how much host CPU one rank needs, against what the container may use
            "host": {"cpu_s_per_step_per_rank": round(host_cpu_sum / world / args.steps, 5),
                     "busy_cores_per_rank": round(host_busy_sum / world, 3), "busy_cores_all_ranks": round(host_busy_sum, 3),
                     "busiest_threads_rank0": [{"comm": c, "busy_cores": b} for b, c in by_thread],
                     "cgroup_quota_cores": usable_cores(),
                     "ranks_that_fit_the_quota": int(usable_cores() / max(host_busy_sum / world, 1e-9)),
                     "note": "process_time of all threads of a rank over the timed region / wall time; `python bench.py "
                             "--gpus 8 --dry-run` asserts 8 x busy_cores_per_rank <= the quota"},
        }
        if world == 1 and not args.no_blip2 and args.envs == 256:
            try:   # the figure the dry run checks against (committed with the profiles of the round)
                json.dump({"busy_cores_per_rank": round(host_busy_sum / world, 3), "envs_per_gpu": E,
                           "cpu_s_per_step": round(host_cpu_sum / world / args.steps, 5)},
                          open(os.path.join(ROOT, "gpurun_out", "host_busy.json"), "w"))
            except OSError:
                pass
        if world == 1 and not args.no_small:
            side = side_legs(args, sim, device, common)
            # BASELINE configs[2] -- the reference's FULL ITMPolicyV2 step -- is its own top-level record; the legs are listed ONCE
            full = {k: side.pop(k) for k in [k for k in side if k.startswith("configs[2] full step")]}
            out["small_batch"] = side
            if full:
                out["full_step"] = {
                    "what": "BASELINE configs[2]: detector + MobileSAM + ObjectPointCloudMap + BLIP-2 ITC + ObstacleMap + ValueMap + "
                            "frontier selection / object goal + PointNav for every resident env, one GPU; parity of this object "
                            "against the single-environment ITMPolicyV2 restatement: tests/test_full_step_gpu.py",
                    "unit": "env-steps/s", **full}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, with_blip2=not args.no_blip2)
            except Exception as exc:  # noqa: BLE001 -- reported, never fatal for the headline line
                out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        out["detail"] = write_detail(out, args.detail)
        print(compact_line(out), flush=True)
    D.shutdown()


def side_legs(args, sim, device, common):
    """The reference's own geometries next to the headline; a failure is recorded, never raised."""
    import torch

    from vlfm_amd import _lib
    from vlfm_amd.harness import BatchedEpisodes

    side = {}

    def leg(name, fn):
        try:
            fn()
        except Exception as exc:  # noqa: BLE001
            side[name + " (FAILED)"] = f"{type(exc).__name__}: {exc}"[:300]
        torch.cuda.synchronize(device)

    def timed(s, warm, n):
        s.prepare(warm + n)
        for _ in range(warm):
            s.step()
        torch.cuda.synchronize(device)
        ts = time.perf_counter()
        for _ in range(n):
            s.step()
        torch.cuda.synchronize(device)
        s.check()
        return (time.perf_counter() - ts) / n

    blip2 = sim.blip2

    def leg_small():
        if blip2 is None:
            return
        for e_small in ((128, 8, 1) if args.envs != 128 else (8, 1)):
            small = BatchedEpisodes(e_small, blip2=blip2, **common)
            small.fast_forward(args.preroll)
            dt = timed(small, 3, 20)
            side[f"envs_per_gpu={e_small}" + (" (configs[3] per-GPU geometry)" if e_small == 8 else
                                             " (configs[1])" if e_small == 1 else "")] = {
                "value": round(e_small / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 3)}
            if e_small == 1:
                side["envs_per_gpu=1 (configs[1])"]["parts"] = one_env_parts(small)
            del small

    def one_env_parts(small):
        """Where the single-environment step goes: each part alone, synchronised on its own."""
        rgb = small.rgb_pool[0]

        def alone(fn, n=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(device)
            ts = time.perf_counter()
            for _ in range(n):
                fn()
                torch.cuda.synchronize(device)
            return (time.perf_counter() - ts) / n * 1e3

        vlm = alone(lambda: (small.blip2.cosine_batch_graphed(rgb, small.prompts) if small.graph_blip2
                             else small.blip2.cosine_batch(rgb, small.prompts)))
        _lib.lib().vlfm_profile_enable(1)
        maps = alone(lambda: small.fast_forward(1))
        kms1 = read_kernel_ms()
        _lib.lib().vlfm_profile_enable(0)
        return {"blip2_itc_ms": round(vlm, 3), "blip2_path": "HIP-graph replay" if small.graph_blip2 else "eager",
                "maps_only_step_ms": round(maps, 3), "map_kernels_sum_ms": round(sum(kms1.values()), 4),
                "map_kernels_ms": {k: round(v, 5) for k, v in kms1.items()},
                "host_device_syncs_per_step": 2,
                "note": "maps_only_step = depth ingest + obstacle pipeline + value update + sort_waypoints with a stub cosine, incl. "
                        "its two D2H read-backs (frontier list, frontier values) and the host-side pose / window bookkeeping; "
                        "maps_only_step - map_kernels_sum = launch gaps + the two syncs + host work; batch-1 HBM floor of the ViT "
                        "is ~0.35 ms (2 GB of f16 weights at 6 TB/s)"}

    def leg_cfg5():
        # BASELINE configs[4] per GPU: 16 envs, 1280x720, value map synchronised with the explored area (full-map mode):
        # north_star's "HBM-bound map-fusion stress".  With BLIP-2 (env-steps/s) and the map kernels' own times.
        hd = dict(common, height=720, width=1280)
        s5 = BatchedEpisodes(16, blip2=blip2, use_blip2=blip2 is not None, sync_explored=True, **hd)
        s5.fast_forward(args.preroll)
        s5.prepare(23)
        for _ in range(3):
            s5.step()
        _lib.lib().vlfm_profile_enable(1)
        dt = timed(s5, 0, 20)
        kms5 = read_kernel_ms()
        _lib.lib().vlfm_profile_enable(0)
        rf5 = map_roofline(kms5, 16, 720, 1280, True, load_pmc())
        side["configs[4] per-GPU geometry: 16 envs, 1280x720, explored-area sync"] = {
            "value": round(16 / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 3),
            "roofline": rf5.get("row"), "roofline_kernels": rf5,
            "all_kernels_ms": {k: round(v, 5) for k, v in kms5.items()}}
        del s5

    def leg_episode_phases():
        # how the map kernels' launch times move through a 500-step episode (explored area and contours grow)
        ph = BatchedEpisodes(min(args.envs, 64), use_blip2=False, **common)
        table = {}
        for target in (25, 250, 475):
            ph.fast_forward(target - 5 - ph.t)
            _lib.lib().vlfm_profile_enable(1)
            ph.fast_forward(10)
            torch.cuda.synchronize(device)
            table[f"episode steps {target - 5}-{target + 4}"] = {k: round(v, 5) for k, v in read_kernel_ms().items()}
            _lib.lib().vlfm_profile_enable(0)
        ph.check()
        side[f"map kernel launch ms by episode phase, {ph.E} envs"] = table
        del ph

    def leg_host():
        # what the rate becomes when the frames arrive as HOST buffers every step (the reference's API hands over
        # numpy arrays): same workload, depth + rgb uploaded from pinned memory inside the timed region
        if blip2 is None:
            return
        host = BatchedEpisodes(args.envs, blip2=blip2, host_inputs=True, **common)
        host.fast_forward(min(args.preroll, 20))
        dt = timed(host, 2, 8)
        side[f"pcie_inclusive, envs_per_gpu={args.envs}"] = {
            "value": round(args.envs / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 3),
            "upload_bytes_per_step": int(args.envs * (4 * args.height * args.width + 3 * args.height * args.width))}
        del host

    def full_step(n_envs, detector, tag, warm, n):
        from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
        from vlfm_amd.vlm.sam import MobileSAM

        from vlfm_amd.harness import ScriptedSightings

        # the reference's full step (base_objectnav_policy.py:106-150,285-356): detector on every frame -> class / confidence
        # filters -> MobileSAM per surviving box -> ObjectPointCloudMap.update_map per mask + update_explored per step ->
        # initialise / explore / navigate -> PointNav.  Random-init networks cannot decide the workload, so a scripted
        # detector HEAD and episode script do (vlfm_amd/harness.py:ScriptedSightings, SIGHTING_SCRIPT above)
        sight = ScriptedSightings(height=args.height, width=args.width, **SIGHTING_SCRIPT)
        full = BatchedEpisodes(n_envs, blip2=blip2, detector=detector, sam=MobileSAM(device=device, allow_random_init=True),
                               select_frontiers=True, object_maps=True, sightings=sight, scripted_masks=True,
                               pointnav=WrappedPointNavResNetPolicy(None, device=device, n_envs=n_envs,
                                                                    discrete_actions=True), **common)
        full.fast_forward(min(args.preroll, 40))
        full.warm_up_segmenter()       # (every batch size the segmenter can meet, before the clock starts)
        for k_ in list(full.object_stats):
            full.object_stats[k_] = {m: 0 for m in full.object_stats[k_]} if isinstance(full.object_stats[k_], dict) else 0
        conv_launches = int(getattr(detector, "hip_convs", 0))
        if conv_launches:
            _lib.lib().vlfm_profile_enable(7)    # every 7th convolution launch: walks through all 244 layers over the timed steps
        from vlfm_amd.vlm import det_ops as det_ops__

        det_ops__.NMS_STATS.update(frames=0, candidates=0, boxes_in=0)
        dt = timed(full, warm, n)
        st = full.object_stats
        steps_all = max(st["env_steps"], 1)
        rec = {
            "value": round(n_envs / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 3), "timed_steps": n,
            "controller": "PointNav ResNet-18-GN + LSTM, random-init, discrete head",
            "detector": getattr(detector, "description", detector.weights),
            "detector_head": f"scripted episodes {SIGHTING_SCRIPT}: 12 initialisation turns, 60-179 search steps (distractors only: "
                             "wrong class / low confidence, dropped by the filters), 30 steps with the target in view (0.8 of them "
                             "carry a confidence-0.9 detection), then arrival = episode end, the next episode starts in place; "
                             f"expected {sight.mean_detections_per_env_step():.3f} surviving detections per env-step; environments are "
                             "spread over all phases; the network's forward runs on every frame and is timed; " + (
                                 "GroundingDINO: its random logits are not used (all 900 queries of an untrained network pass the "
                                 "box threshold, so the timed post-processing sees MORE than a trained one's)"
                                 if tag else
                                 "YOLOv7: the head's candidates (24 jittered boxes per sighting) are written into the raw prediction "
                                 "and go through the detector's own non_max_suppression / scale_coords (harness._inject_candidates)"),
            "segmenter": "MobileSAM (TinyViT-5M) random-init: one forward per surviving box (frame re-encoded per box, as the "
                         "reference does); the mask handed on is the box's inscribed ellipse",
            "object_map": "ObjectPointCloudMap.update_map per mask (erosion 5, back-projection, 5000-point subsample, DBSCAN eps "
                          "0.2 / 100 on csrc/object_cloud.hip) + update_explored per env-step",
            "measured_over_warmup_and_timed_steps": {
                "surviving_detections_per_env_step": round(st["detections"] / steps_all, 4),
                "masks_segmented_per_env_step": round(st["masks"] / steps_all, 4),
                "object_cloud_updates_per_env_step": round(st["cloud_updates"] / steps_all, 4),
                "episodes_ended": st.get("episodes_ended", 0),
                "mode_mix": {m: round(c / steps_all, 3) for m, c in st["modes"].items()}}}
        from vlfm_amd.vlm import det_ops as det_ops_

        if det_ops_.NMS_STATS["frames"]:
            # what the timed post-processing really chews on: the network's (random-init) logits pass the 0.25 objectness gate for
            # this many of the E6E head's 17 850 boxes per frame (the reference: yolov7.py:91-99)
            rec["nms_candidates_per_frame"] = round(det_ops_.NMS_STATS["candidates"] / det_ops_.NMS_STATS["frames"], 1)
            rec["nms_boxes_per_frame_in"] = det_ops_.NMS_STATS["boxes_in"]
        if conv_launches:
            ms, timed_n = _lib.profile_read("conv_nhwc_kernel")
            _lib.lib().vlfm_profile_enable(0)
            if timed_n:
                flop = detector.gflops * 1e9 * n_envs                      # 2 x MACs of every convolution, all frames
                achieved = flop / (ms * 1e-3 * conv_launches) / 1e12       # mean launch x launches per forward
                rec["detector_roofline"] = {
                    "bound": "mfma", "kernel": "conv_nhwc_kernel (csrc/conv_nhwc.hip), f16 in / f32 accumulate",
                    "achieved": round(achieved, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4),
                    "launches_per_forward": conv_launches, "launches_timed": timed_n, "mean_launch_ms": round(ms, 5),
                    "gflop_per_frame": round(detector.gflops, 1),
                    "note": "HIP events on every 7th launch; the layer mix is what it is: 64-channel 3x3 layers are LDS-bound "
                            "(~0.6 PFLOP/s), the 80 / 160-channel 1x1 layers at 224x320 / 112x160 HBM-bound (DESIGN.md 6d)"}
        side[f"configs[2] full step{tag}, envs_per_gpu={n_envs}"] = rec
        del full

    def leg_full():
        if blip2 is None:
            return
        from vlfm_amd.vlm.yolov7 import YOLOv7

        det = YOLOv7(device=device, allow_random_init=True)
        for n_envs in (8, 64, 128):
            full_step(n_envs, det, "", 3, 30)

    def leg_gdino():
        # the open-vocabulary detector the config names (what the reference uses for non-COCO targets): GroundingDINO at
        # its real geometry (HF Swin-T + BERT-base, random-init), HIP MsDeformAttn
        if blip2 is None:
            return
        from vlfm_amd.vlm.grounding_dino import GroundingDINO

        det = GroundingDINO(device=device, allow_random_init=True)
        for n_envs in (8, 64):
            full_step(n_envs, det, " with GroundingDINO", 2, 30)

    if not args.only_full:
        leg("small batches", leg_small)
        leg("config 5", leg_cfg5)
        leg("episode phases", leg_episode_phases)
        leg("pcie_inclusive", leg_host)
    if not args.no_full:
        leg("configs[2] full step", leg_full)
        leg("configs[2] full step with GroundingDINO", leg_gdino)
    return side


if __name__ == "__main__":
    main()
