#!/bin/bash
# persistent 8-phase kernel (variant 7): parity + race screen, then A/B against variant 3 and the library
set -u
R=$(pwd); O=gpurun_out/r5w; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm_f16_gpu.py -x -q -m gpu -k "7 or rejects" > $O/pytest_gemm.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gemm.txt
tail -3 $O/pytest_gemm.txt
timeout 600 python tools/gemm_f16_probe.py --quick --rounds 7 --variants 3,7 > $O/probe.txt 2>&1
grep -v "^M=" $O/probe.txt | cut -c1-400 | tail -14
