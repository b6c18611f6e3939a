#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
# can two ranks share GPU 0 under RCCL?  (only to exercise the N>1 code path of bench.py on a 1-GPU box)
HIPX_ALL_RANKS_DEVICE0=1 NCCL_DEBUG=WARN timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --grid 64 > gpurun_out/bench12_2rank.log 2>&1
echo "exit $?" >> gpurun_out/bench12_2rank.log
tail -25 gpurun_out/bench12_2rank.log
