#!/bin/bash
# gpurun with retries while no slot / box is free (exit code 3 = nothing charged).  usage: tools/gpurun_retry.sh TIMEOUT 'command' OUTFILE
T=$1; CMD=$2; OUT=$3
python -c "import __graft_entry__ as g; g.build(); from vlfm_amd import _lib; _lib.lib()" > /dev/null 2>&1 || { echo 'libvlfm_amd.so does not build / load here: not launching' > $OUT; exit 9; }
for i in $(seq 1 40); do
  gpurun --timeout $T -- "$CMD" > $OUT 2>&1; rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $OUT; then exit $rc; fi
  sleep 45
done
exit 3
