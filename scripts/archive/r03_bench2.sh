#!/bin/bash
# bench.py's N > 1 path on a one-GPU box: 2 (and 4) ranks sharing the device through the mailbox communicator
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port $((29800+n)) bench.py --gpus $n --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_n$n.json 2> gpurun_out/bench_n$n.err
  echo "N=$n rc=$?"; tail -3 gpurun_out/bench_n$n.err | cut -c1-300
  python - $n <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/bench_n{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print("ms/step", d["ms_per_step"], "value", d["value"], "n_gpus", d["n_gpus"], "scaling", d["scaling"], d["multi_gpu"])
except Exception as e:
    print("no JSON", e)
PY
done
