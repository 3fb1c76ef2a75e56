#!/bin/bash
# round-4 session 11: workgroup-scope fences in explored_select / frontier: exactness + kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
timeout 1200 python -m pytest tests/test_obstacle_map_gpu.py tests/test_golden_gpu.py tests/test_properties_gpu.py tests/test_obstacle_prims_gpu.py tests/test_obstacle_windows_gpu.py tests/test_harness_gpu.py tests/test_policy_step_gpu.py -q --timeout 900 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
(timeout 400 python tools/phase_probe.py 256 150 2>&1 | grep -v amdgpu.ids) > $O/phase_e256.txt; head -8 $O/phase_e256.txt
timeout 600 python bench.py --no-blip2 --no-small --no-cpu-baseline --steps 20 > $O/bench_maps.json 2> $O/bench_maps.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s11/bench_maps.json').read().strip().splitlines()[-1])
print('maps only', d['value'], d['ms_per_step']); print(d['roofline']['all_kernels_ms'])
PY
timeout 600 python bench.py --no-blip2 --envs 16 --height 720 --width 1280 --sync-explored --no-small --no-cpu-baseline --steps 20 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s11/bench_cfg5.json').read().strip().splitlines()[-1])
print('cfg5 maps only', d['value'], d['ms_per_step']); print(d['roofline']['all_kernels_ms'])
PY
