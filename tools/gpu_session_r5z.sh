#!/bin/bash
# split ragged round of the persistent walk: GEMM tests, packed-GELU identity test, probe, headline A/B against the unsplit walk
set -u
R=$(pwd); O=gpurun_out/r5z; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_f16_gpu.py tests/test_gelu_packed_gpu.py -x -q -m gpu > $O/pytest_gemm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gemm.txt
tail -3 $O/pytest_gemm.txt
timeout 600 python tools/gemm_f16_probe.py --rounds 7 --variants 3,7 > $O/probe.txt 2>&1
grep "^fc1\|^qkv \|^proj\|^fc2\|PROBE" $O/probe.txt | grep -v sweep | cut -c1-330
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-small --no-full --no-cpu-baseline --detail $O/d_$name.json > $O/b_$name.txt 2>$O/e_$name.txt
  python - "$O/b_$name.txt" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["ms_per_step"], d["config"].get("vit_gemm"), d.get("roofline_mfma",{}).get("launch_ms"))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
run nosplit_$rep VLFM_GEMM_NO_SPLIT=1
run split_$rep VLFM_GEMM_VARIANT=7
done
