#!/bin/bash
set -u
mkdir -p gpurun_out/r5t
O=gpurun_out/r5t
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 900 python bench.py --steps 10 --warmup 3 --only-full --no-cpu-baseline --detail $O/detail_$tag.json > $O/bench_$tag.txt 2>$O/bench_$tag.err; python - <<PY
import json
d=json.load(open("$O/detail_$tag.json"))
print("$tag", d["value"], (d.get("roofline_mfma") or {}).get("achieved"), {k.replace("configs[2] full step","fs")[-34:]:(v["value"]) for k,v in d["full_step"].items() if isinstance(v,dict)})
PY
}
run prio0_a VLFM_SIDE_PRIORITY=0
run prio1_a VLFM_SIDE_PRIORITY=-1
run prio0_b VLFM_SIDE_PRIORITY=0
run prio1_b VLFM_SIDE_PRIORITY=-1
run prio0_c VLFM_SIDE_PRIORITY=0
run prio1_c VLFM_SIDE_PRIORITY=-1
