"""The ONE line bench.py prints must be readable by the driver: a single line of JSON well under the 8 KB stdout tail the driver
keeps, with the contract's keys.  Round 4's line had grown to 30 KB of duplicated prose and `BENCH_r04.parsed` was null; the full
record now goes to a file (``--detail``) and the printed line carries numbers and short enum strings only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "full_step")


def full_record():
    """A complete record as main() assembles it: round 4's committed 30 KB record (every leg present, all the prose), with the
    full-step legs split off `small_batch` the way main() does now."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_e256.json")))
    side = rec["small_batch"]
    for k in [k for k in side if k.startswith("configs[2] full step")]:
        side.pop(k)
    rec["config"].update(workload_short="configs[1] step x resident envs: BLIP-2 ITC + ValueMap+ObstacleMap+sort_waypoints",
                         parallelism_short="env-sharded x1", tuned_gemms=True, vit_gemm="hip8p:fc1,fc2,qkv,proj",
                         attention_short="hip")
    for leg in rec["full_step"].values():
        if isinstance(leg, dict):
            leg["nms_candidates_per_frame"] = 8925.0
    rec["roofline_mfma"] = {"bound": "mfma", "kernel": "gemm_f16_8p_kernel<1>", "achieved": 1234.5, "peak": 2500.0,
                            "unit": "TFLOP/s", "frac": 0.4938, "traffic": None, "launch_ms": 0.9213, "launches_timed": 234,
                            "flop_per_launch": 1138341937152, "meaning": "x" * 700}
    rec["detail"] = "gpurun_out/bench_detail.json"
    return rec


def test_line_is_short_single_and_complete():
    rec = full_record()
    assert len(json.dumps(rec)) > 15000            # the input really is the long form
    line = bench.compact_line(rec)
    assert "\n" not in line and "\r" not in line
    assert len(line.encode()) < bench.LINE_BUDGET <= 6000
    back = json.loads(line)
    for k in CONTRACT:
        assert k in back, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert back["config"]["workload"] and back["config"]["tuned_gemms"] is True
    assert back["value"] == rec["value"] and back["ms_per_step"] == rec["ms_per_step"] and back["steps"] == rec["steps"]
    assert back["detail"] == "gpurun_out/bench_detail.json"
    # every full-step leg once, as numbers
    legs = [k for k in back["full_step"] if k != "unit"]
    assert sorted(legs) == sorted(["yolov7_e6e,envs=8", "yolov7_e6e,envs=64", "yolov7_e6e,envs=128", "gdino,envs=8", "gdino,envs=64"])
    for k in legs:
        assert set(back["full_step"][k]) <= {"value", "ms_per_step", "conv_mfma_frac", "nms_candidates_per_frame"}
        assert back["full_step"][k]["nms_candidates_per_frame"] == 8925.0
    assert not any(k.startswith("configs[2]") for k in back["small_batch"])
    # no prose anywhere: every string in the line is short
    def strings(o):
        if isinstance(o, str):
            yield o
        elif isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
    assert max(len(t) for t in strings(back)) <= 170


def test_line_survives_failed_legs_and_a_huge_record():
    rec = full_record()
    rec["cpu_baseline"] = {"error": "RuntimeError: " + "y" * 1000}
    rec["small_batch"]["config 5 (FAILED)"] = "HipError: " + "z" * 300
    rec["full_step"]["configs[2] full step, envs_per_gpu=64"] = None
    for i in range(400):                                   # a record that would not fit: the side legs are shed, the contract stays
        rec["small_batch"][f"envs_per_gpu={1000 + i} (synthetic)"] = {"value": 1.0, "ms_per_step": 2.0}
    line = bench.compact_line(rec)
    assert len(line.encode()) < bench.LINE_BUDGET
    back = json.loads(line)
    for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "dtype", "roofline", "cpu_baseline"):
        assert k in back, k


def test_detail_file_round_trip(tmp_path):
    rec = full_record()
    path = bench.write_detail(rec, str(tmp_path / "sub" / "bench_detail.json"))
    assert path and json.load(open(path))["roofline"]["hbm_kernels"] == rec["roofline"]["hbm_kernels"]
