#!/bin/bash
# the mailbox communicator: sharded solver as N processes on the box's GPU(s); prints rank 0's JSON line per run
export HSA_ENABLE_IPC_MODE_LEGACY=0
p=29700
for cfg in "2 map" "2 features" "4 map" "4 features" "8 features"; do
  set -- $cfg; p=$((p+1))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$1 --master-addr 127.0.0.1 --master-port $p tests/_multirank_worker.py $2 p2p 2>/dev/null | grep "^{"
done
