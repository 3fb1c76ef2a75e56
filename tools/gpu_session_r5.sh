#!/bin/bash
# Round-5 evidence session (the ONE at the end of the round): full GPU tests, the driver's bench command (the short line + the detail
# file), rocprofv3 kernel stats (headline, config 5, maps), PMC traffic of the map kernels, the GEMM probe / stamps / SQ counters,
# value-map phases, depth pass, GroundingDINO and MobileSAM walls.  Everything lands under gpurun_out/s_r5/; the summaries are copied to
# profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TMPDIR=/tmp
O=gpurun_out/s_r5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 --detail $O/bench_detail.json ) > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench_line.json
cp gpurun_out/host_busy.json $O/host_busy.json 2>/dev/null
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_default -o p -- python $R/bench.py --no-small --no-cpu-baseline --steps 10 --detail $O/prof_default_detail.json > $R/$O/prof_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_cfg5 -o p -- python $R/bench.py --no-blip2 --envs 16 --height 720 --width 1280 --sync-explored --no-small --no-cpu-baseline --detail $O/prof_cfg5_detail.json > $R/$O/prof_cfg5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_maps -o p -- python $R/bench.py --no-blip2 --no-small --no-cpu-baseline --steps 20 --detail $O/prof_maps_detail.json > $R/$O/prof_maps.log 2>&1
cd $R
python tools/rocprof_summary.py $O/prof_default/p_results.db $O/r05_bench_e256_kernel_stats.csv > $O/summary_default.log 2>&1
python tools/rocprof_summary.py $O/prof_cfg5/p_results.db $O/r05_cfg5_maps_kernel_stats.csv > $O/summary_cfg5.log 2>&1
python tools/rocprof_summary.py $O/prof_maps/p_results.db $O/r05_maps_e256_kernel_stats.csv > $O/summary_maps.log 2>&1
VLFM_COMMIT=${VLFM_COMMIT:-unknown} bash tools/pmc_traffic.sh 256 640 480 > $O/pmc_e256.log 2>&1
VLFM_COMMIT=${VLFM_COMMIT:-unknown} bash tools/pmc_traffic.sh 16 1280 720 sync > $O/pmc_cfg5.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
(timeout 600 python tools/gemm_f16_probe.py --variants 0,3,7 2>&1 | grep -v amdgpu.ids) > $O/r05_gemm_probe.txt
(timeout 300 python tools/gemm_stamp_probe.py 3 2>&1 | grep -v amdgpu.ids | grep -v "in flight") > $O/r05_gemm_stamps.txt
(bash tools/gemm_pmc.sh 2>&1 | grep -v amdgpu.ids) > $O/r05_gemm_pmc.txt
(timeout 300 python tools/vm_phase_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400) > $O/r05_vm_phase_probe.txt
(timeout 300 python tools/ingest_probe.py 2>&1 | grep -v amdgpu.ids) > $O/r05_ingest_probe.txt
(timeout 300 python tools/gdino_profile_probe.py 64 4 2>&1 | grep -v amdgpu.ids | tail -1) > $O/r05_gdino_wall.txt
(timeout 300 python tools/gdino_profile_probe.py 8 6 2>&1 | grep -v amdgpu.ids | tail -1) >> $O/r05_gdino_wall.txt
(timeout 500 python tools/full_step_parts_probe.py 128 2>&1 | grep -v amdgpu.ids | tail -6) > $O/r05_full_step_parts_e128.txt
(timeout 500 python tools/full_step_parts_probe.py 8 2>&1 | grep -v amdgpu.ids | tail -6) > $O/r05_full_step_parts_e8.txt
(timeout 200 python tools/sam_probe.py 32 2>&1 | grep -v amdgpu.ids | tail -1) > $O/r05_mobile_sam_b32.txt
(timeout 300 python tools/phase_probe.py 256 150 2>&1 | grep -v amdgpu.ids | tail -24) > $O/r05_obstacle_phase_probe.txt
# only summaries travel back (gpurun copies at most 64 MiB): traces, counter databases and the diagnostic libraries stay behind
find gpurun_out -name "*.db" -delete
rm -rf $O/prof_default $O/prof_cfg5 $O/prof_maps gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/gemm_pmc_[0-9]
rm -f gpurun_out/*.so gpurun_out/*.o
du -sh gpurun_out
ls $O
