#!/bin/bash
# round-4 session 8: split depth pass (exactness + A/B timing), GDINO with SDPA / vectorised post-processing, SAM on split GEMMs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
timeout 1200 python -m pytest tests/test_obstacle_map_gpu.py tests/test_golden_gpu.py tests/test_properties_gpu.py tests/test_obstacle_prims_gpu.py tests/test_gdino_fast_gpu.py tests/test_detect_gpu.py tests/test_sam_ops_gpu.py tests/test_full_step_gpu.py -q --timeout 900 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
(timeout 300 python tools/ingest_probe.py 2>&1 | grep -v amdgpu.ids) > $O/ingest_split.txt; cat $O/ingest_split.txt
(VLFM_INGEST_SINGLE_PASS=1 timeout 300 python tools/ingest_probe.py 2>&1 | grep -v amdgpu.ids) > $O/ingest_single.txt; cat $O/ingest_single.txt
(timeout 300 python tools/gdino_profile_probe.py 64 4 2>&1 | grep -v amdgpu.ids) > $O/gdino_graph_wall.txt; tail -2 $O/gdino_graph_wall.txt
(timeout 300 python tools/gdino_sections_probe.py 64 1 split 2>&1 | grep -v amdgpu.ids) > $O/gdino_sections_fast.txt; head -12 $O/gdino_sections_fast.txt
(timeout 200 python tools/sam_probe.py 32 2>&1 | grep -v amdgpu.ids | tail -4) > $O/sam_probe.txt; cat $O/sam_probe.txt
timeout 900 python bench.py --envs 8 --steps 3 --warmup 2 --preroll 20 --no-cpu-baseline --only-full > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s8/bench_full.json').read().strip().splitlines()[-1])
for k,v in d.get('full_step',{}).items():
    if isinstance(v, dict): print(k, v['value'], v['ms_per_step'])
for k,v in d['small_batch'].items():
    if 'FAILED' in k: print(k, v)
PY
tail -3 $O/bench_full.err
