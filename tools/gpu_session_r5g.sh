#!/bin/bash
set -u
mkdir -p gpurun_out/r5g
O=gpurun_out/r5g
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 --detail $O/bench_detail.json ) > $O/bench_driver.txt 2>$O/bench_driver.err; echo "rc=$?" >> $O/bench_driver.txt
tail -5 $O/pytest_gpu.txt; cat $O/bench_driver.txt
