#!/bin/bash
set -u
mkdir -p gpurun_out/r5d
O=gpurun_out/r5d
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gemm_f16_gpu.py -x -q -m gpu > $O/pytest_gemm.txt 2>&1; echo "rc=$?" >> $O/pytest_gemm.txt
timeout 900 python tools/gemm_f16_probe.py --variants 0,2,3 > $O/gemm_probe.txt 2>&1; echo "probe rc=$?" >> $O/gemm_probe.txt
timeout 300 python tools/gemm_stamp_probe.py 3 > $O/stamps_v3.txt 2>&1
for cfg in all fc1; do
  VLFM_GEMM_VARIANT=3 VLFM_VIT_GEMMS=$cfg timeout 600 python bench.py --steps 20 --warmup 5 --no-small --no-cpu-baseline --detail $O/detail_$cfg.json > $O/bench_$cfg.txt 2>$O/bench_$cfg.err; echo "rc=$?" >> $O/bench_$cfg.txt
done
tail -3 $O/pytest_gemm.txt; grep -v "epi=" $O/gemm_probe.txt | tail -18; cat $O/stamps_v3.txt | grep -v "in flight"; for cfg in all fc1; do cut -c1-200 $O/bench_$cfg.txt; done
