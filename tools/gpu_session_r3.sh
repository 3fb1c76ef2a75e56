#!/bin/bash
# Round-3 evidence session: full GPU tests, default bench, rocprofv3 kernel stats (headline + config 5), PMC traffic of the
# map kernels, SQ counters of the depth pass.  Everything lands under gpurun_out/s_r3/; the summaries are copied to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TMPDIR=/tmp
O=gpurun_out/s_r3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 400 $O/bench_default.json
cp gpurun_out/host_busy.json $O/host_busy.json 2>/dev/null   # the plain run's figure (the rocprofv3 runs below rewrite the file with their own overhead)
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_default -o p -- python $R/bench.py --no-small --no-cpu-baseline --steps 10 > $R/$O/prof_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_cfg5 -o p -- python $R/bench.py --no-blip2 --envs 16 --height 720 --width 1280 --sync-explored --no-small --no-cpu-baseline > $R/$O/prof_cfg5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_maps -o p -- python $R/bench.py --no-blip2 --no-small --no-cpu-baseline --steps 20 > $R/$O/prof_maps.log 2>&1
cd $R
python tools/rocprof_summary.py $O/prof_default/p_results.db $O/r03_bench_e256_kernel_stats.csv > $O/summary_default.log 2>&1
python tools/rocprof_summary.py $O/prof_cfg5/p_results.db $O/r03_cfg5_maps_kernel_stats.csv > $O/summary_cfg5.log 2>&1
python tools/rocprof_summary.py $O/prof_maps/p_results.db $O/r03_maps_e256_kernel_stats.csv > $O/summary_maps.log 2>&1
VLFM_COMMIT=${VLFM_COMMIT:-unknown} bash tools/pmc_traffic.sh 256 640 480 > $O/pmc_e256.log 2>&1
VLFM_COMMIT=${VLFM_COMMIT:-unknown} bash tools/pmc_traffic.sh 16 1280 720 sync > $O/pmc_cfg5.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
bash tools/pmc_sq.sh ingest depth_ingest -- python $R/tools/ingest_probe.py 4 > /dev/null 2>&1; cp gpurun_out/pmc_sq_ingest.txt $O/r03_ingest_sq_pmc.txt
# yolov7-e6e on conv_nhwc.hip: per-layer-shape table, tile-shape sweep, rocprofv3 kernel stats of the forward alone
(timeout 200 python tools/yolo_layer_probe.py 64 2>&1 | grep -v amdgpu.ids) > $O/r03_yolo_e6e_layers_b64.txt 2>&1
(timeout 200 python tools/yolo_layer_probe.py 128 2>&1 | grep -v amdgpu.ids) > $O/r03_yolo_e6e_layers_b128.txt 2>&1
(timeout 200 python tools/conv_nhwc_probe.py 64 sweep 2>&1 | grep -v amdgpu.ids) > $O/r03_conv_nhwc_tile_sweep_b64.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_yolo -o p -- python $R/tools/yolo_probe.py 64 hip-only > $R/$O/prof_yolo.log 2>&1
cd $R
python tools/rocprof_summary.py $O/prof_yolo/p_results.db $O/r03_yolo_e6e_b64_kernel_stats.csv > $O/summary_yolo.log 2>&1
# MobileSAM: steady-state kernel table of 32 frames, and the full step's parts at 128 environments
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_sam -o p -- python $R/tools/sam_probe.py 32 > $R/$O/sam_probe.log 2>&1
cd $R
python tools/rocprof_tail.py /tmp/prof_sam/p_results.db 200 30 > $O/r03_mobile_sam_b32_steady_state.txt 2>&1
(timeout 300 python tools/full_step_parts_probe.py 128 2>&1 | grep -v amdgpu.ids | tail -6) > $O/r03_full_step_parts_e128.txt 2>&1
find gpurun_out -name "*.db" -size +20M -delete
ls $O
