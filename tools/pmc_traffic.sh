#!/bin/bash
# tools/pmc_traffic.sh [E] -- on the GPU box: HBM traffic of the map kernels from rocprofv3 PMC counters.
# FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), with
# --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Output: gpurun_out/pmc_{fetch,write}/*.db,
# then tools/pmc_traffic.py turns them into profiles/pmc_traffic.json.
E=${1:-128}
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=/root/repo/gpurun_out/pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rm -rf $d
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $d -o pmc -- python /root/repo/bench.py --envs $E --steps 6 --warmup 2 \
      --no-cpu-baseline --no-small --no-blip2 > /root/repo/gpurun_out/pmc_$c.log 2>&1
done
ls -la /root/repo/gpurun_out/pmc_fetch /root/repo/gpurun_out/pmc_write
