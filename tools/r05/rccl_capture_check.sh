#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_rccl_capture; mkdir -p $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29585 RANK=0 WORLD_SIZE=1
for i in 1 2 3; do timeout 200 python tests/rccl_single_worker.py 2>$O/err_$i.txt | grep RCCL_OK || { echo "run $i failed"; grep -v "^frame" $O/err_$i.txt | tail -8; }; done
# the same capture in the default (global) mode, to see the failure the helper avoids
for i in 1 2 3; do
timeout 200 python - <<'PY' > $O/global_$i.txt 2>&1; echo "global-mode control run $i: rc $?"; grep -h "survived\|not permitted\|terminate" $O/global_$i.txt | head -3
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29583")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 20, device=dev)
for _ in range(4):
    dist.all_reduce(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    x.mul_(2.0)
    time.sleep(1.0)
g.replay(); torch.cuda.synchronize()
print("global-mode capture survived", float(x[0]), flush=True)
dist.destroy_process_group()
PY
done
