#!/bin/bash
# tools/gpu_sweep.sh "<map-only env counts>" "<BLIP env counts>" -- run on the GPU box via gpurun: gpu tests, map-only
# sweeps, BLIP sweeps.  Output -> gpurun_out/
mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -30) > gpurun_out/pytest_gpu.log
for E in ${1:-8 64 128}; do
  timeout 100 python bench.py --envs $E --steps 20 --warmup 3 --no-cpu-baseline --no-blip2 2>gpurun_out/err_maps_e$E.txt | tail -1 > gpurun_out/sweep_maps_e$E.json
done
timeout 100 python bench.py --envs 16 --height 720 --width 1280 --sync-explored --steps 20 --warmup 3 --no-cpu-baseline --no-blip2 2>&1 | tail -1 > gpurun_out/sweep_maps_cfg5.json
for E in ${2:-8 32 64}; do
  timeout 200 python bench.py --envs $E --steps 8 --warmup 2 --no-cpu-baseline 2>gpurun_out/err_blip_e$E.txt | tail -1 > gpurun_out/sweep_blip_e$E.json
done
tail -3 gpurun_out/pytest_gpu.log
