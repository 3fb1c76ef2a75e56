#!/bin/bash
set -u
mkdir -p gpurun_out/r5h
O=gpurun_out/r5h
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 900 python bench.py --steps 10 --warmup 3 --only-full --no-cpu-baseline --detail $O/detail_$tag.json > $O/bench_$tag.txt 2>$O/bench_$tag.err; python - <<PY
import json
d=json.load(open("$O/detail_$tag.json"))
print("$tag", d["value"], {k.replace("configs[2] full step","fs")[-34:]:(v["value"]) for k,v in d["full_step"].items() if isinstance(v,dict)})
PY
}
run default A=1
run fc1only VLFM_VIT_GEMMS=fc1
run nobeside VLFM_VLM_BESIDE=0
run beside64 VLFM_VLM_BESIDE=64
run default2 A=1
