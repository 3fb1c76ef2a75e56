#!/bin/bash
# Round-2 evidence session: full GPU tests, default bench, rocprofv3 kernel stats (headline + config 5), PMC traffic, GEMM counters.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/prof_default -o p -- python $R/bench.py --no-small --no-cpu-baseline --steps 10 > $R/$O/prof_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_cfg5 -o p -- python $R/bench.py --no-blip2 --envs 16 --height 720 --width 1280 --sync-explored --no-small --no-cpu-baseline > $R/$O/prof_cfg5.log 2>&1
cd $R
python tools/rocprof_summary.py $O/prof_default/p_results.db $O/r02_bench_e256_kernel_stats.csv > $O/summary_default.log 2>&1
python tools/rocprof_summary.py $O/prof_cfg5/p_results.db $O/r02_cfg5_maps_kernel_stats.csv > $O/summary_cfg5.log 2>&1
bash tools/pmc_traffic.sh 256 640 480 > $O/pmc_e256.log 2>&1
bash tools/pmc_traffic.sh 16 1280 720 sync > $O/pmc_cfg5.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
bash tools/gemm_pmc.sh > $O/gemm_pmc.log 2>&1
find gpurun_out -name "*.db" -size +20M -delete
ls $O
