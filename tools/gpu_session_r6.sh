#!/bin/bash
# Round-6 evidence session: full GPU tests, the driver's bench command (the short line + the detail file), rocprofv3 kernel stats
# (headline, config 5, maps), PMC traffic of the map kernels and of the attention kernel, the attention probes (whole / parts compiled
# out / phase stamps), the GEMM probe, value-map phases, depth pass (+ its SQ counters), the obstacle pipeline early and late in an
# episode, the batch-1 / 8-env BLIP-2 kernel tables, MobileSAM's steady state, YOLOv7's per-layer table, GroundingDINO and full-step
# walls.  Everything lands under gpurun_out/s_r6/; the summaries are copied to profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TMPDIR=/tmp
O=gpurun_out/s_r6; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 --detail $O/bench_detail.json ) > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench_line.json
cp gpurun_out/host_busy.json $O/host_busy.json 2>/dev/null
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_default -o p -- python $R/bench.py --no-small --no-cpu-baseline --steps 10 --detail $O/prof_default_detail.json > $R/$O/prof_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_cfg5 -o p -- python $R/bench.py --no-blip2 --envs 16 --height 720 --width 1280 --sync-explored --no-small --no-cpu-baseline --detail $O/prof_cfg5_detail.json > $R/$O/prof_cfg5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_maps -o p -- python $R/bench.py --no-blip2 --no-small --no-cpu-baseline --steps 20 --detail $O/prof_maps_detail.json > $R/$O/prof_maps.log 2>&1
cd $R
python tools/rocprof_summary.py $O/prof_default/p_results.db $O/r06_bench_e256_kernel_stats.csv > $O/summary_default.log 2>&1
python tools/rocprof_summary.py $O/prof_cfg5/p_results.db $O/r06_cfg5_maps_kernel_stats.csv > $O/summary_cfg5.log 2>&1
python tools/rocprof_summary.py $O/prof_maps/p_results.db $O/r06_maps_e256_kernel_stats.csv > $O/summary_maps.log 2>&1
VLFM_COMMIT=${VLFM_COMMIT:-unknown} bash tools/pmc_traffic.sh 256 640 480 > $O/pmc_e256.log 2>&1
VLFM_COMMIT=${VLFM_COMMIT:-unknown} bash tools/pmc_traffic.sh 16 1280 720 sync > $O/pmc_cfg5.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# ---- attention
( echo "# tools/vit_attn_probe.py (standalone calls, random qkv)"; timeout 300 python tools/vit_attn_probe.py 256 128 64 16 8 1 2>&1 | grep "D=88"
  echo; echo "# tools/vit_attn_stub_probe.py 256: the kernel with parts compiled out (PA_STUB, csrc/vit_attention.hip)"
  timeout 600 python tools/vit_attn_stub_probe.py 256 2>&1 | grep images
  echo; echo "# reading qkv as [B][H][3][S][88] (-DPA_HEAD_MAJOR, timing experiment), and with a head-major output too (-DPA_CONTIG_OUT)"
  PA_STUBS=0,4,5 timeout 600 python tools/vit_attn_stub_probe.py 256 -DPA_HEAD_MAJOR 2>&1 | grep images
  PA_STUBS=0,4 timeout 600 python tools/vit_attn_stub_probe.py 256 -DPA_HEAD_MAJOR -DPA_CONTIG_OUT 2>&1 | grep images
  echo; echo "# no waits, no barriers (-DPA_FREE_RUN): the memory side's issue rate"
  PA_STUBS=4,5 timeout 600 python tools/vit_attn_stub_probe.py 256 -DPA_FREE_RUN 2>&1 | grep images
  echo; echo "# tools/vit_attn_phase_probe.py 256"; timeout 300 python tools/vit_attn_phase_probe.py 256 2>&1 | grep -v amdgpu.ids ) > $O/r06_vit_attention_probes.txt 2>&1
( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pmc_att_$c; timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_att_$c -o pmc -- python $R/tools/vit_attn_probe.py 256 > /dev/null 2>&1; (cd $R && python tools/pmc_sq.py /tmp/pmc_att_$c vit_attention); done ) > $O/r06_vit_attention_pmc.txt 2>&1
# ---- GEMM, value-map update, depth pass
(timeout 600 python tools/gemm_f16_probe.py 2>&1 | grep -v amdgpu.ids) > $O/r06_gemm_probe.txt
(timeout 300 python tools/vm_phase_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400) > $O/r06_vm_phase_probe.txt
(timeout 300 python tools/ingest_probe.py 2>&1 | grep -v amdgpu.ids) > $O/r06_ingest_probe.txt
(bash tools/kernel_pmc.sh ingest depth_ingest_kernel -- python $R/tools/ingest_probe.py 0 256 480 640 2>&1 | grep -v amdgpu.ids) > $O/r06_ingest_sq_pmc.txt
# ---- obstacle pipeline early / late in the episode
(timeout 300 python tools/phase_probe.py 256 150 2>&1 | grep -v amdgpu.ids | tail -24 | cut -c1-600) > $O/r06_obstacle_phase_probe.txt
(timeout 300 python tools/phase_probe.py 64 470 2>&1 | grep -v amdgpu.ids | tail -24 | cut -c1-600) > $O/r06_obstacle_phase_probe_step470.txt
# ---- small batches, the other networks
( for e in 1 8; do rm -rf /tmp/prof_b$e; rocprofv3 --kernel-trace -d /tmp/prof_b$e -o p -- python tools/blip2_small_batch_probe.py $e 60 2>&1 | grep "E="; python tools/rocprof_tail.py /tmp/prof_b$e/p_results.db $((e == 1 ? 200 : 400)) 24 | cut -c1-150; done
  for e in 1 8; do python tools/blip2_small_batch_probe.py $e 100 2>&1 | grep "E="; done ) > $O/r06_blip2_small_batch_kernels.txt 2>&1
( rm -rf /tmp/prof_sam; rocprofv3 --kernel-trace -d /tmp/prof_sam -o p -- python tools/sam_probe.py 32 2>&1 | grep MobileSAM; python tools/rocprof_tail.py /tmp/prof_sam/p_results.db 225 30 | cut -c1-150 ) > $O/r06_mobile_sam_b32.txt 2>&1
(timeout 600 python tools/yolo_layer_probe.py 128 2>&1 | grep -v amdgpu.ids | head -40) > $O/r06_yolo_layers_b128.txt
(timeout 300 python tools/gdino_profile_probe.py 64 4 2>&1 | grep -v amdgpu.ids | tail -1) > $O/r06_gdino_wall.txt
(timeout 300 python tools/gdino_profile_probe.py 8 6 2>&1 | grep -v amdgpu.ids | tail -1) >> $O/r06_gdino_wall.txt
(timeout 500 python tools/full_step_parts_probe.py 128 2>&1 | grep -v amdgpu.ids | tail -6) > $O/r06_full_step_parts_e128.txt
(timeout 500 python tools/full_step_parts_probe.py 8 2>&1 | grep -v amdgpu.ids | tail -6) > $O/r06_full_step_parts_e8.txt
# only summaries travel back (gpurun copies at most 64 MiB): traces, counter databases and the diagnostic libraries stay behind
find gpurun_out -name "*.db" -delete
rm -rf $O/prof_default $O/prof_cfg5 $O/prof_maps gpurun_out/pmc_fetch gpurun_out/pmc_write
rm -f gpurun_out/*.so gpurun_out/*.o
du -sh gpurun_out
ls $O
