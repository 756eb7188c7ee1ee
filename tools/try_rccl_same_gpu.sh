#!/bin/bash
# Can two ranks on ONE GPU form an RCCL communicator?  (rehearsal attempt; expected to be refused)
cd $GRAFT_REPO_ROOT
export NCCL_DEBUG=WARN
for r in 0 1; do
  RANK=$r WORLD_SIZE=2 LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout -k 5 90 \
    python examples/multi_gpu.py --k 6 --n 20000 --shared-device > gpurun_out/rccl_same_$r.log 2>&1 &
done
wait
tail -5 gpurun_out/rccl_same_0.log; echo ---; tail -5 gpurun_out/rccl_same_1.log
