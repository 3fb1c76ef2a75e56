#!/bin/bash
# round-4 session 2: full-step harness parity, f32 GEMM forms (numerics + rate), rewritten roofline block, full-step legs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm_f32_gpu.py -x -q --timeout 500 > $O/pytest_gemm.log 2>&1; echo "rc=$?" >> $O/pytest_gemm.log; tail -15 $O/pytest_gemm.log
(timeout 300 python tools/gemm_f32_probe.py 2>&1 | grep -v amdgpu.ids) > $O/gemm_f32_probe.txt; cat $O/gemm_f32_probe.txt
timeout 600 python -m pytest tests/test_full_step_gpu.py tests/test_object_map_gpu.py tests/test_policy_step_gpu.py -x -q --timeout 500 > $O/pytest_full_step.log 2>&1; echo "rc=$?" >> $O/pytest_full_step.log
tail -30 $O/pytest_full_step.log
timeout 900 python bench.py --no-cpu-baseline --steps 10 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s2/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d['roofline'], indent=0)[:1500])
print(json.dumps(d.get('roofline_depth_pass'), indent=0)[:900])
print(json.dumps(d.get('full_step'), indent=0)[:6000])
print(json.dumps(d['small_batch'].get('envs_per_gpu=1 (configs[1])'), indent=0)[:2500])
PY
tail -5 $O/bench.err
