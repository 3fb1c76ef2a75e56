#!/bin/bash
# round-4 session 5: host profile of the full step (what makes the object-map stage slow), fused deformable sampling, graph'd GDINO wall
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
(timeout 600 python tools/full_step_profile_probe.py 64 12 2>&1 | grep -v amdgpu.ids) > $O/full_step_host_profile_e64.txt; head -70 $O/full_step_host_profile_e64.txt
timeout 900 python -m pytest tests/test_gdino_fast_gpu.py -q --timeout 800 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
(timeout 300 python tools/gdino_profile_probe.py 64 4 2>&1 | grep -v amdgpu.ids) > $O/gdino_graph_wall.txt; tail -3 $O/gdino_graph_wall.txt
(timeout 300 python tools/gdino_sections_probe.py 64 1 split 2>&1 | grep -v amdgpu.ids) > $O/gdino_sections_fast.txt; cat $O/gdino_sections_fast.txt
