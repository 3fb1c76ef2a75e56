#!/bin/bash
# round-4 session 6: re-associated fusion layer, device post-processing, SAM batch buckets; GDINO + full-step numbers
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
timeout 900 python -m pytest tests/test_gdino_fast_gpu.py tests/test_detect_gpu.py tests/test_full_step_gpu.py -q --timeout 800 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
(timeout 300 python tools/gdino_profile_probe.py 64 4 2>&1 | grep -v amdgpu.ids) > $O/gdino_graph_wall.txt; tail -3 $O/gdino_graph_wall.txt
(timeout 300 python tools/gdino_sections_probe.py 64 1 split 2>&1 | grep -v amdgpu.ids) > $O/gdino_sections_fast.txt; cat $O/gdino_sections_fast.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_gd -o p -- python $R/tools/gdino_profile_probe.py 64 3 > $R/$O/gdino_probe.log 2>&1
cd $R
python tools/rocprof_tail.py /tmp/prof_gd/p_results.db $(cat /tmp/gdino_window_ms) 40 > $O/r04_gdino_b64_fast.txt 2>&1
head -44 $O/r04_gdino_b64_fast.txt
timeout 900 python bench.py --envs 8 --steps 3 --warmup 2 --preroll 20 --no-cpu-baseline --only-full > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s6/bench_full.json').read().strip().splitlines()[-1])
for k,v in d.get('full_step',{}).items():
    if isinstance(v, dict): print(k, v['value'], v['ms_per_step'], v.get('measured_over_warmup_and_timed_steps'))
for k,v in d['small_batch'].items():
    if 'FAILED' in k: print(k, v)
PY
tail -3 $O/bench_full.err
