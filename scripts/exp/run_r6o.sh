mkdir -p gpurun_out/r6o
timeout -k 10 500 python scripts/soak_api.py 240 601 4 2>&1 | tail -1 | cut -c1-400 > gpurun_out/r6o/api4.txt
SOAK_BIG=1 timeout -k 10 500 python scripts/soak_api.py 240 602 4 2>&1 | tail -1 | cut -c1-400 > gpurun_out/r6o/api4_big.txt
timeout -k 10 400 python scripts/soak_schedule.py 180 603 2>&1 | tail -1 | cut -c1-400 > gpurun_out/r6o/schedule.txt
timeout -k 10 400 python scripts/soak_stdsort.py 1500 604 2>&1 | tail -1 > gpurun_out/r6o/stdsort.txt
cat gpurun_out/r6o/*.txt
