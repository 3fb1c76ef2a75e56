#!/bin/bash
# persistent 8-phase kernel inside the network: headline at 256 envs, interleaved A/B
set -u
R=$(pwd); O=gpurun_out/r5x; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-small --no-full --no-cpu-baseline --detail $O/d_$name.json > $O/b_$name.txt 2>$O/e_$name.txt
  python - "$O/b_$name.txt" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["ms_per_step"], d["config"].get("vit_gemm"))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
run base_$rep VLFM_GEMM_VARIANT=3
run p_fc1_$rep VLFM_GEMM_VARIANT=7 VLFM_VIT_GEMMS=fc1
run p_fc1proj_$rep VLFM_GEMM_VARIANT=7 VLFM_VIT_GEMMS=fc1,proj
run p_all_$rep VLFM_GEMM_VARIANT=7 VLFM_VIT_GEMMS=qkv,proj,fc1
done
