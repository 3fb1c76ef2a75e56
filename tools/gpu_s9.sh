#!/bin/bash
# round-4 session 9: operand planes in the split GEMM (numerics, rate), GDINO at 8 and 64 frames, headline, full step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=gpurun_out/s9; mkdir -p $O
timeout 1200 python -m pytest tests/test_gemm_f32_gpu.py tests/test_gdino_fast_gpu.py tests/test_vlm_gpu.py tests/test_sam_ops_gpu.py tests/test_detect_gpu.py -q --timeout 900 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
(timeout 300 python tools/gemm_f32_probe.py 2>&1 | grep -v amdgpu.ids) > $O/gemm_f32_probe.txt; cat $O/gemm_f32_probe.txt
(timeout 300 python tools/gdino_profile_probe.py 64 4 2>&1 | grep -v amdgpu.ids) > $O/gdino_wall_b64.txt; tail -1 $O/gdino_wall_b64.txt
(timeout 300 python tools/gdino_profile_probe.py 8 6 2>&1 | grep -v amdgpu.ids) > $O/gdino_wall_b8.txt; tail -1 $O/gdino_wall_b8.txt
(timeout 300 python tools/gdino_sections_probe.py 8 1 split 2>&1 | grep -v amdgpu.ids) > $O/gdino_sections_b8.txt; head -14 $O/gdino_sections_b8.txt
(timeout 200 python tools/sam_probe.py 32 2>&1 | grep -v amdgpu.ids | tail -2) > $O/sam_probe.txt; cat $O/sam_probe.txt
timeout 900 python bench.py --no-cpu-baseline --steps 10 --only-full > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s9/bench.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'])
for k,v in d.get('full_step',{}).items():
    if isinstance(v, dict): print(k, v['value'], v['ms_per_step'])
for k,v in d['small_batch'].items():
    if 'FAILED' in k: print(k, v)
PY
tail -3 $O/bench.err
