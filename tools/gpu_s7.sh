#!/bin/bash
# round-4 session 7: wide-tile split GEMM, fusion layer isolated, Q-Former on split GEMMs, headline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_f32_gpu.py tests/test_gdino_fast_gpu.py tests/test_vlm_gpu.py tests/test_harness_gpu.py -q --timeout 800 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
(timeout 300 python tools/gemm_f32_probe.py 2>&1 | grep -v amdgpu.ids) > $O/gemm_f32_probe.txt; cat $O/gemm_f32_probe.txt
(timeout 300 python tools/gdino_profile_probe.py 64 4 2>&1 | grep -v amdgpu.ids) > $O/gdino_graph_wall.txt; tail -2 $O/gdino_graph_wall.txt
timeout 900 python bench.py --no-cpu-baseline --no-small --steps 10 > $O/bench_head.json 2> $O/bench_head.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s7/bench_head.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'])
PY
tail -3 $O/bench_head.err
