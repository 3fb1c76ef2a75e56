#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
(timeout 400 python tools/phase_probe.py 256 150 2>&1 | grep -v amdgpu.ids) > $O/phase_e256.txt; cat $O/phase_e256.txt
(timeout 300 python tools/phase_probe.py 1 150 2>&1 | grep -v amdgpu.ids) > $O/phase_e1.txt; cat $O/phase_e1.txt
(timeout 300 python tools/phase_probe.py 16 150 2>&1 | grep -v amdgpu.ids) > $O/phase_e16.txt; cat $O/phase_e16.txt
