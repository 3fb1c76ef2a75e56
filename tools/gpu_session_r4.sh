#!/bin/bash
# Round-4 evidence session (ONE per round): full GPU tests, default bench, rocprofv3 kernel stats (headline, config 5, maps), PMC
# traffic of the map kernels, GroundingDINO kernel table after the fused forward, f32 GEMM probe, host profile of the full step.
# Everything lands under gpurun_out/s_r4/; the summaries are copied to profiles/ by hand afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TMPDIR=/tmp
O=gpurun_out/s_r4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 300 $O/bench_default.json
cp gpurun_out/host_busy.json $O/host_busy.json 2>/dev/null
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_default -o p -- python $R/bench.py --no-small --no-cpu-baseline --steps 10 > $R/$O/prof_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_cfg5 -o p -- python $R/bench.py --no-blip2 --envs 16 --height 720 --width 1280 --sync-explored --no-small --no-cpu-baseline > $R/$O/prof_cfg5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_maps -o p -- python $R/bench.py --no-blip2 --no-small --no-cpu-baseline --steps 20 > $R/$O/prof_maps.log 2>&1
cd $R
python tools/rocprof_summary.py $O/prof_default/p_results.db $O/r04_bench_e256_kernel_stats.csv > $O/summary_default.log 2>&1
python tools/rocprof_summary.py $O/prof_cfg5/p_results.db $O/r04_cfg5_maps_kernel_stats.csv > $O/summary_cfg5.log 2>&1
python tools/rocprof_summary.py $O/prof_maps/p_results.db $O/r04_maps_e256_kernel_stats.csv > $O/summary_maps.log 2>&1
VLFM_COMMIT=${VLFM_COMMIT:-unknown} bash tools/pmc_traffic.sh 256 640 480 > $O/pmc_e256.log 2>&1
VLFM_COMMIT=${VLFM_COMMIT:-unknown} bash tools/pmc_traffic.sh 16 1280 720 sync > $O/pmc_cfg5.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# GroundingDINO: kernel table of the fused forward at 64 frames, wall per call at 64 and 8 frames, sections
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_gd -o p -- python $R/tools/gdino_profile_probe.py 64 3 > $R/$O/gdino_probe_rocprof.log 2>&1
cd $R
python tools/rocprof_tail.py /tmp/prof_gd/p_results.db $(cat /tmp/gdino_window_ms) 45 > $O/r04_gdino_b64_after.txt 2>&1
(timeout 300 python tools/gdino_profile_probe.py 64 4 2>&1 | grep -v amdgpu.ids | tail -1) > $O/r04_gdino_wall.txt
(timeout 300 python tools/gdino_profile_probe.py 8 6 2>&1 | grep -v amdgpu.ids | tail -1) >> $O/r04_gdino_wall.txt
(timeout 300 python tools/gdino_sections_probe.py 64 1 split 2>&1 | grep -v amdgpu.ids) > $O/r04_gdino_sections_b64.txt
(timeout 300 python tools/gemm_f32_probe.py 2>&1 | grep -v amdgpu.ids) > $O/r04_gemm_f32_probe.txt
(timeout 400 python tools/full_step_profile_probe.py 64 12 2>&1 | grep -v amdgpu.ids | head -60) > $O/r04_full_step_host_profile_e64.txt
(timeout 500 python tools/full_step_parts_probe.py 128 2>&1 | grep -v amdgpu.ids | tail -6) > $O/r04_full_step_parts_e128.txt
(timeout 200 python tools/sam_probe.py 32 2>&1 | grep -v amdgpu.ids | tail -1) > $O/r04_mobile_sam_b32.txt
(timeout 300 python tools/phase_probe.py 256 150 2>&1 | grep -v amdgpu.ids | tail -24) > $O/r04_obstacle_phase_probe_lds_tables.txt
find gpurun_out -name "*.db" -size +20M -delete
ls $O
