#!/bin/bash
# Round-2 GPU session 1: full GPU test suite, default bench, A/B of the value-map update (single launch vs 3 launches),
# rocprofv3 kernel stats, PMC traffic of the fused kernel.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for mode in 0 1; do
  VLFM_VM_SPLIT=$mode timeout 300 python bench.py --no-blip2 --envs 256 --no-small --no-cpu-baseline > $O/maps_e256_split$mode.json 2>> $O/ab.err
  VLFM_VM_SPLIT=$mode timeout 300 python bench.py --no-blip2 --envs 16 --height 720 --width 1280 --sync-explored --no-small --no-cpu-baseline > $O/maps_cfg5_split$mode.json 2>> $O/ab.err
  VLFM_VM_SPLIT=$mode timeout 300 python bench.py --no-blip2 --envs 8 --no-small --no-cpu-baseline > $O/maps_e8_split$mode.json 2>> $O/ab.err
done
cd /tmp
for mode in 0 1; do
  VLFM_VM_SPLIT=$mode timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_cfg5_split$mode -o p -- python $GRAFT_REPO_ROOT/bench.py --no-blip2 --envs 16 --height 720 --width 1280 --sync-explored --no-small --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_cfg5_split$mode.log 2>&1
done
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_default -o p -- python $GRAFT_REPO_ROOT/bench.py --no-small --no-cpu-baseline --steps 10 > $GRAFT_REPO_ROOT/$O/prof_default.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh 256 640 480 > $O/pmc_e256.log 2>&1
bash tools/pmc_traffic.sh 16 1280 720 sync > $O/pmc_cfg5.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $O -name "*.db" -size +20M -delete
ls -la $O
