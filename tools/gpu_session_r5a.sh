#!/bin/bash
# round 5, session A: the 8-phase GEMM (correctness, race screen, A/B timing), its tests, and the driver's bench command with the new
# short line.  Everything lands in gpurun_out/r5a/.
set -u
mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
export TMPDIR=/tmp
timeout 900 python tools/gemm_f16_probe.py > $O/gemm_probe.txt 2>&1; echo "probe rc=$?" >> $O/gemm_probe.txt
timeout 600 python -m pytest tests/test_gemm_f16_gpu.py tests/test_gemm_f32_gpu.py -x -q -m gpu > $O/pytest_gemm.txt 2>&1; echo "rc=$?" >> $O/pytest_gemm.txt
timeout 600 python -m pytest tests/test_vlm_gpu.py -x -q -m gpu -k "vit or blip2 or mlp" > $O/pytest_vlm.txt 2>&1; echo "rc=$?" >> $O/pytest_vlm.txt
# headline only, three GEMM configurations (fc1 on the new kernel = default; all four; none)
for cfg in fc1 all none; do
  VLFM_VIT_GEMMS=$cfg timeout 600 python bench.py --steps 20 --warmup 5 --no-small --no-cpu-baseline --detail $O/detail_$cfg.json > $O/bench_$cfg.txt 2>$O/bench_$cfg.err; echo "rc=$?" >> $O/bench_$cfg.txt
done
# the driver's command, as it is
( time timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/bench_detail.json ) > $O/bench_driver.txt 2>$O/bench_driver.err; echo "rc=$?" >> $O/bench_driver.txt
tail -c 300 $O/gemm_probe.txt; tail -3 $O/pytest_gemm.txt; tail -3 $O/pytest_vlm.txt; for cfg in fc1 all none; do tail -c 600 $O/bench_$cfg.txt | cut -c1-400; done; wc -c $O/bench_driver.txt
