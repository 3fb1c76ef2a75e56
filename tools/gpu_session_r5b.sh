#!/bin/bash
# round 5, session B: 8-phase GEMM after the epilogue rewrite + half-tiles-last order
set -u
mkdir -p gpurun_out/r5b
O=gpurun_out/r5b
export TMPDIR=/tmp
timeout 900 python tools/gemm_f16_probe.py --variants 0,2,3 > $O/gemm_probe.txt 2>&1; echo "probe rc=$?" >> $O/gemm_probe.txt
timeout 600 python -m pytest tests/test_gemm_f16_gpu.py -x -q -m gpu > $O/pytest_gemm.txt 2>&1; echo "rc=$?" >> $O/pytest_gemm.txt
timeout 600 python -m pytest tests/test_vlm_gpu.py tests/test_full_step_gpu.py -x -q -m gpu > $O/pytest_vlm.txt 2>&1; echo "rc=$?" >> $O/pytest_vlm.txt
for cfg in all fc1; do
  VLFM_VIT_GEMMS=$cfg timeout 600 python bench.py --steps 20 --warmup 5 --no-small --no-cpu-baseline --detail $O/detail_$cfg.json > $O/bench_$cfg.txt 2>$O/bench_$cfg.err; echo "rc=$?" >> $O/bench_$cfg.txt
done
grep -v "epi=" $O/gemm_probe.txt | tail -20; tail -3 $O/pytest_gemm.txt; tail -3 $O/pytest_vlm.txt; for cfg in all fc1; do cut -c1-200 $O/bench_$cfg.txt; done
