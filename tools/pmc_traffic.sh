#!/bin/bash
# tools/pmc_traffic.sh [E] [W] [H] [sync] -- on the GPU box: HBM traffic of the map kernels from rocprofv3 PMC counters.
# FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), with
# --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Output: gpurun_out/pmc_{fetch,write}/*.db,
# then tools/pmc_traffic.py turns them into profiles/pmc_traffic.json (key "<kernel>@E=..,WxH[,sync]").
E=${1:-256}; W=${2:-640}; H=${3:-480}; SYNC=${4:-}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out
cd /tmp
EXTRA=""
[ -n "$SYNC" ] && EXTRA="--sync-explored"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$REPO/gpurun_out/pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rm -rf $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python $REPO/bench.py --envs $E --width $W --height $H $EXTRA \
      --steps 6 --warmup 2 --preroll 150 --no-cpu-baseline --no-small --no-blip2 > $REPO/gpurun_out/pmc_$c.log 2>&1
done
cd $REPO && python tools/pmc_traffic.py $E $W $H $SYNC
