#!/bin/bash
# batch-1 BLIP-2 forward: kernel table of the steady state (graph replay)
set -u
R=$(pwd); O=gpurun_out/r5r; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/b1_probe.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["R"])
import torch
torch.set_num_threads(1)
from vlfm_amd.vlm.blip2itm import BLIP2ITM
dev = torch.device("cuda:0")
m = BLIP2ITM(device=dev, allow_random_init=True)
img = torch.randint(0, 256, (1, 480, 640, 3), dtype=torch.uint8, device=dev)
p = ["Seems like there is a chair ahead."]
for _ in range(3): m.cosine_batch_graphed(img, p)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): m.cosine_batch_graphed(img, p)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"BLIP-2 batch 1, graph replay: {dt / 50 * 1e3:.3f} ms per call; window {dt * 1e3:.0f} ms")
open("/tmp/b1_window_ms", "w").write(str(int(dt * 1e3)))
PY
cd /tmp
R=$R timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 -o p -- python /tmp/b1_probe.py > $R/$O/b1_rocprof.log 2>&1
cd $R
python tools/rocprof_tail.py /tmp/prof_b1/p_results.db $(cat /tmp/b1_window_ms) 40 > $O/b1_kernels.txt 2>&1
grep "BLIP-2 batch 1" $O/b1_rocprof.log; head -45 $O/b1_kernels.txt | cut -c1-175
