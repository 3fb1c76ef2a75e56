#!/usr/bin/env python
"""bench.py -- env-steps/s of the VLFM perception + value-map hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one ITMPolicyV2 perception+mapping step for every resident environment (BASELINE.json metric
"env-steps/s (VLM+value-map update), 640x480 RGB-D"): batched in-process BLIP-2 ITC cosine at the full ViT-g/14
geometry (random-init weights: no checkpoints offline), one depth-ingest pass, ObstacleMap update (when built),
ValueMap.update_map and frontier scoring (sort_waypoints) -- see vlfm_amd/harness.py.  Observations are synthetic and
already resident in HBM when the timed region starts.  Episodes are independent: env e lives on rank e mod N
(weak scaling, --envs per GPU fixed), the only collective is the final metric all-reduce.

Prints ONE JSON line (rank 0) with the driver's keys plus:
  roofline      dominant HIP map kernel: algorithmic bytes per launch / mean launch time (HIP events on the launch
                stream, inside the timed region) vs the 8 TB/s HBM peak
  cpu_baseline  the reference-faithful CPU path (oracle/ NumPy+C restatement of the maps + the same ITC graph in fp32
                on the host cores), timed on rank 0 at N=1 on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=256,
                    help="environments resident per GPU (weak scaling).  BASELINE configs[3]/[4] use 8 and 16 per GPU; "
                         "288 GB of HBM holds far more, and the batched BLIP-2 forward and the map kernels only reach "
                         "their efficient regime at >= 64 (the 8/GPU and 1/GPU figures are reported alongside)")
    ap.add_argument("--no-small", action="store_true", help="skip the 8-env and 1-env side measurements")
    ap.add_argument("--no-full", action="store_true",
                    help="skip the configs[2] side run (8 envs: BLIP-2 + detector + MobileSAM + maps + frontier selection; the "
                         "detector/segmenter are random-init stand-ins of the YOLOv7-E6E / MobileSAM class: a side figure)")
    ap.add_argument("--with-full", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-blip2", action="store_true", help="map kernels only (NOT the headline metric)")
    ap.add_argument("--no-obstacle", action="store_true")
    ap.add_argument("--sync-explored", action="store_true", help="config 5: value map synchronised with explored area")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true", help="run the obstacle pipeline on the BLIP-2 stream")
    ap.add_argument("--host-profile", action="store_true", help="cProfile the timed region (stderr), for tuning")
    return ap.parse_args()


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


def cpu_baseline(args, with_blip2: bool):
    """Reference-faithful CPU path on the host cores: oracle maps (NumPy + oracle/libcvport.so) and the same BLIP-2
    ITC graph in fp32 through PyTorch-CPU (the reference's own fallback device, vlfm/vlm/blip2itm.py:26-27)."""
    import numpy as np
    import torch

    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, SyntheticEnv, camera_intrinsics, rgb_frame

    cores = min(usable_cores(), 16)  # more intra-op threads than this only oversubscribes the fp32 GEMMs
    torch.set_num_threads(cores)
    fov = camera_intrinsics(args.width)[2]
    env = SyntheticEnv(0, args.height, args.width)
    vm = RefValueMap(1, use_max_confidence=False)
    om = None
    try:
        from oracle.ref_obstacle_map import RefObstacleMap

        if not args.no_obstacle:
            om = RefObstacleMap(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)
    except ImportError:
        om = None
    fx, fy, _ = camera_intrinsics(args.width)
    # per-stage CPU times: wrap the oracle's stage functions with wall-clock accumulators
    stage_s: dict = {}

    def timed(mod, name, label):
        fn = getattr(mod, name)

        def wrapper(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                stage_s[label] = stage_s.get(label, 0.0) + time.perf_counter() - t0

        setattr(mod, name, wrapper)
        return fn

    import oracle.ref_obstacle_map as rom_
    import oracle.ref_value_map as rvm_

    restore = [(rvm_.RefValueMap, "_process_local_data", timed(rvm_.RefValueMap, "_process_local_data", "depth profile + polygon cut")),
               (rvm_, "rotate_about_centre", timed(rvm_, "rotate_about_centre", "rotate (warpAffine)")),
               (rvm_, "paste_centred", timed(rvm_, "paste_centred", "place")),
               (rvm_.RefValueMap, "_fuse_new_data", timed(rvm_.RefValueMap, "_fuse_new_data", "fuse (full-map NumPy passes)")),
               (rvm_.RefValueMap, "sort_waypoints", timed(rvm_.RefValueMap, "sort_waypoints", "sort_waypoints")),
               (rom_, "fill_small_holes", timed(rom_, "fill_small_holes", "fill_small_holes")),
               (rom_, "unproject", timed(rom_, "unproject", "unproject")),
               (rom_, "apply_tf", timed(rom_, "apply_tf", "transform_points")),
               (rom_, "reveal_fog_of_war", timed(rom_, "reveal_fog_of_war", "reveal_fog_of_war")),
               (rom_, "detect_frontier_waypoints", timed(rom_, "detect_frontier_waypoints", "detect_frontier_waypoints"))]
    # maps: warm-up populates the confidence-mask cache, then a bounded timed sample
    n_map = 60
    obs = [env.observe() for _ in range(n_map + 5)]
    t_map = 0.0
    for i, (depth, tf, values) in enumerate(obs):
        t0 = time.perf_counter()
        if om is not None:
            om.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
        vm.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fov)
        if om is not None and len(om.frontiers):
            vm.sort_waypoints(om.frontiers, 0.5)
        dt = time.perf_counter() - t0
        if i >= 5:
            t_map += dt
        else:
            stage_s.clear()  # warm-up steps do not count
    for owner, name, fn in restore:
        setattr(owner, name, fn)
    map_s = t_map / n_map
    blip_s = 0.0
    if with_blip2:
        from vlfm_amd.vlm.blip2itm import Blip2ITCConfig, Blip2ITCModel
        from vlfm_amd.vlm.ops import CLIP_MEAN, CLIP_STD
        from PIL import Image

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
    total = map_s + blip_s
    return {
        "value": round(1.0 / total, 3), "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": (f"1 env, {n_map} timed map steps (oracle ValueMap{'+ObstacleMap' if om is not None else ''}, "
                   f"{map_s * 1e3:.1f} ms/step on 1 core)"
                   + (f" + {args.cpu_steps} BLIP-2 ITC fp32 forwards on {cores} torch threads ({blip_s * 1e3:.0f} ms/frame)"
                      if with_blip2 else " (BLIP-2 leg skipped)")),
        "maps_only_env_steps_per_s": round(1.0 / map_s, 2),
        "map_stage_ms_per_step": {k: round(v / n_map * 1e3, 3) for k, v in sorted(stage_s.items(), key=lambda kv: -kv[1])},
    }


def main():
    args = parse_args()
    import numpy as np
    import torch

    from vlfm_amd import distributed as D

    rank, local_rank, world = D.world()
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    D.init("nccl", device)  # backend "nccl" is RCCL on ROCm; only the metric all-reduces use it
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.set_num_threads(1)  # the GPU leg has no CPU tensor math; idle OpenMP spinners would only eat the CPU quota

    from vlfm_amd.harness import BatchedEpisodes

    have_obstacle = os.path.exists(os.path.join(ROOT, "vlfm_amd", "mapping", "obstacle_map.py")) and not args.no_obstacle
    sim = BatchedEpisodes(args.envs, device=device, height=args.height, width=args.width,
                          env_offset=D.shard_env_ids(rank, world, args.envs)[0],
                          use_blip2=not args.no_blip2, obstacle=have_obstacle, sync_explored=args.sync_explored,
                          overlap=not args.no_overlap)

    def barrier():
        D.barrier(device)

    for _ in range(args.warmup):
        sim.step()
    from vlfm_amd import _lib

    _lib.lib().vlfm_profile_enable(1)  # HIP events around every kernel launch of libvlfm_amd, on its launch stream
    prof = None
    if args.host_profile:
        import cProfile

        prof = cProfile.Profile()
    barrier()
    t0 = time.perf_counter()
    if prof is not None:
        prof.enable()
    for _ in range(args.steps):
        sim.step()
    if prof is not None:
        prof.disable()
    barrier()
    elapsed = time.perf_counter() - t0
    if prof is not None and rank == 0:
        import pstats

        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(35)

    elapsed_max, (env_steps,) = D.reduce_metrics(elapsed, [float(args.envs * args.steps)], device)

    torch.cuda.synchronize(device)
    kms = {}
    for kname in ("depth_ingest_kernel", "depth_scatter_kernel", "depth_ingest_scatter_kernel", "fill_small_holes_kernel",
                  "hole_scatter_kernel",
                  "visible_mask_kernel", "value_map_fuse_kernel", "sort_waypoints_kernel",
                  "mask_unexplored_kernel", "resample_h_kernel", "resample_v_norm_kernel", "itc_head_kernel",
                  "navigable_kernel", "fog_of_war_kernel", "explored_select_kernel", "frontier_kernel"):
        ms, n = _lib.profile_read(kname)
        if n:
            kms[kname] = ms
    _lib.lib().vlfm_profile_enable(0)
    if rank == 0:
        H, W, E = args.height, args.width, args.envs
        T = 2 * int(5.0 * 20) + 1
        S = 1000
        # Algorithmic bytes per launch (DESIGN.md "Kernels"), E observations per launch:
        #   depth ingest     reads every depth texel once: 4*H*W  (+ <= H*W scattered obstacle bytes, not counted)
        #   map fusion       template read 4*T^2 + conf RMW 8*T^2 + value RMW 8*C*T^2  (C = 1)
        #   mask unexplored  (obstacle-synchronised mode) explored S^2 + conf RMW 8*S^2 + value RMW 8*C*S^2
        per_kernel = {}

        def entry(name, nbytes):
            if name in kms:
                gbs = nbytes / (kms[name] * 1e-3) / 1e9
                per_kernel[name] = {"algorithmic_bytes_per_launch": nbytes, "launch_ms": round(kms[name], 5),
                                    "achieved": round(gbs, 1), "frac": round(gbs / 8000.0, 4)}

        entry("depth_ingest_kernel", E * 4 * H * W)
        entry("depth_scatter_kernel", E * 4 * H * W)
        # one pass that replaces BOTH of the reference's reads of the depth image (column maximum for the value map,
        # unprojection for the obstacle map): SURVEY.md 8d prices them at 4*H*W each
        entry("depth_ingest_scatter_kernel", E * 8 * H * W)
        entry("value_map_fuse_kernel", E * (4 * T * T + 8 * T * T + 8 * T * T))
        entry("mask_unexplored_kernel", E * (S * S + 8 * S * S + 8 * S * S))
        # the kernel BASELINE.json's north_star names for the roofline target is the map-fusion kernel
        name = "value_map_fuse_kernel"
        head = per_kernel[name]
        roofline = {"bound": "hbm", "kernel": name, "achieved": head["achieved"], "peak": 8000.0, "unit": "GB/s",
                    "frac": head["frac"], "traffic": None,
                    "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"],
                    "launch_ms": head["launch_ms"], "hbm_kernels": per_kernel,
                    "all_kernels_ms": {k: round(v, 5) for k, v in kms.items()}}
        # HBM traffic of the roofline kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, corrected
        # as /opt/skills/guides/MI355X_MICROARCH.md prescribes): collected offline by tools/pmc_traffic.sh on the GPU box
        # and committed under profiles/; valid only for the same E / geometry
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            rec = pmc.get(f"value_map_fuse_kernel@E={E},{W}x{H}")
            if rec:
                traffic = rec["bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        roofline["traffic"] = traffic
        out = {
            "metric": "env-steps/s (VLM+value-map update), 640x480 RGB-D",
            "value": round(env_steps / elapsed_max, 2), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 maps / f16 ViT-g + f32 Q-Former", "data": "synthetic",
            "config": {"workload": ("configs[1] step (BLIP-2 ITC cosine + ValueMap fusion"
                                    + (" + ObstacleMap update" if have_obstacle else "")
                                    + " + sort_waypoints) for every resident env, envs sharded over GPUs as in configs[3]"
                                    if not args.no_blip2 else "MAP KERNELS ONLY (no VLM) -- not the headline metric"),
                       "envs_per_gpu": E, "global_envs": E * world, "rgbd": f"{W}x{H}", "map": "1000x1000 @ 20 px/m",
                       "blip2": "ViT-g/14 39 blocks + Q-Former 12 layers, random-init" if not args.no_blip2 else None,
                       "parallelism": f"env-sharded x{world}, metric all-reduce only"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_small and not args.no_blip2:
            # the reference's own geometry: configs[3] keeps 8 envs per GPU, configs[1] a single env (batch 1)
            side = {}

            def leg(name, fn):
                """A side measurement must never cost the headline: a failure is recorded, not raised."""
                try:
                    fn()
                except Exception as exc:  # noqa: BLE001
                    side[name + " (FAILED)"] = f"{type(exc).__name__}: {exc}"[:300]
                torch.cuda.synchronize(device)

            def leg_small():
                for e_small in ((128, 8, 1) if args.envs != 128 else (8, 1)):
                    small = BatchedEpisodes(e_small, device=device, height=args.height, width=args.width,
                                            blip2=sim.blip2, obstacle=have_obstacle, overlap=not args.no_overlap)
                    for _ in range(3):
                        small.step()
                    torch.cuda.synchronize(device)
                    ts = time.perf_counter()
                    n_small = 20
                    for _ in range(n_small):
                        small.step()
                    torch.cuda.synchronize(device)
                    dt = (time.perf_counter() - ts) / n_small
                    side[f"envs_per_gpu={e_small}"] = {"value": round(e_small / dt, 2), "unit": "env-steps/s",
                                                       "ms_per_step": round(dt * 1e3, 3)}
                    del small

            leg("small batches", leg_small)

            def leg_host():
                # what the rate becomes when the frames arrive as HOST buffers every step (the reference's API hands over
                # numpy arrays): same workload, depth + rgb uploaded from pinned memory inside the timed region
                host = BatchedEpisodes(args.envs, device=device, height=args.height, width=args.width, blip2=sim.blip2,
                                       obstacle=have_obstacle, overlap=not args.no_overlap, host_inputs=True)
                for _ in range(2):
                    host.step()
                torch.cuda.synchronize(device)
                ts = time.perf_counter()
                for _ in range(8):
                    host.step()
                torch.cuda.synchronize(device)
                dt = (time.perf_counter() - ts) / 8
                side[f"pcie_inclusive, envs_per_gpu={args.envs}"] = {
                    "value": round(args.envs / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 3),
                    "upload_bytes_per_step": int(args.envs * (4 * args.height * args.width + 3 * args.height * args.width))}
                del host

            leg("pcie_inclusive", leg_host)

            def leg_full():
                from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
                from vlfm_amd.vlm.sam import MobileSAM
                from vlfm_amd.vlm.yolov7 import YOLOv7

                full = BatchedEpisodes(8, device=device, height=args.height, width=args.width, blip2=sim.blip2,
                                       obstacle=have_obstacle, overlap=not args.no_overlap,
                                       detector=YOLOv7(device=device), sam=MobileSAM(device=device),
                                       select_frontiers=True,
                                       pointnav=WrappedPointNavResNetPolicy(None, device=device, n_envs=8,
                                                                            discrete_actions=True))
                for _ in range(3):
                    full.step()
                torch.cuda.synchronize(device)
                ts = time.perf_counter()
                for _ in range(20):
                    full.step()
                torch.cuda.synchronize(device)
                dt = (time.perf_counter() - ts) / 20
                side["configs[2] full step, envs_per_gpu=8"] = {
                    "value": round(8 / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 3),
                    "controller": "PointNav ResNet-18-GN + LSTM, random-init, discrete head",
                    "detector": full.detector.weights, "segmenter": "MobileSAM (TinyViT-5M) random-init, 1 box for every "
                                                                    "4th env-step"}
                del full

            def leg_gdino():
                # ... and with the open-vocabulary detector the config names (what the reference uses for non-COCO
                # targets): GroundingDINO at its real geometry (HF Swin-T + BERT-base, random-init), HIP MsDeformAttn
                from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
                from vlfm_amd.vlm.grounding_dino import GroundingDINO
                from vlfm_amd.vlm.sam import MobileSAM

                full = BatchedEpisodes(8, device=device, height=args.height, width=args.width, blip2=sim.blip2,
                                       obstacle=have_obstacle, overlap=not args.no_overlap,
                                       detector=GroundingDINO(device=device), sam=MobileSAM(device=device),
                                       select_frontiers=True,
                                       pointnav=WrappedPointNavResNetPolicy(None, device=device, n_envs=8,
                                                                            discrete_actions=True))
                for _ in range(2):
                    full.step()
                torch.cuda.synchronize(device)
                ts = time.perf_counter()
                for _ in range(10):
                    full.step()
                torch.cuda.synchronize(device)
                dt = (time.perf_counter() - ts) / 10
                side["configs[2] full step with GroundingDINO, envs_per_gpu=8"] = {
                    "value": round(8 / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 3),
                    "detector": "GroundingDINO (HF Swin-T + BERT-base geometry, 172 M parameters) " + full.detector.weights}
                del full

            if not args.no_full:
                leg("configs[2] full step", leg_full)
                leg("configs[2] full step with GroundingDINO", leg_gdino)
            out["small_batch"] = side
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, with_blip2=not args.no_blip2)
            except Exception as exc:  # noqa: BLE001 -- reported, never fatal for the headline line
                out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        print(json.dumps(out), flush=True)
    D.shutdown()


if __name__ == "__main__":
    main()
