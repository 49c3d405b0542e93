mkdir -p gpurun_out/r6k
fails=0
for s in 401 402 403 404 405 406 407 408 409 410 411 412; do
timeout -k 5 200 python scripts/soak_api.py 6 $s 4 > gpurun_out/r6k/soak4_fix_$s.txt 2>&1 || { fails=$((fails+1)); grep MISMATCH gpurun_out/r6k/soak4_fix_$s.txt | cut -c1-150 | head -2; tail -2 gpurun_out/r6k/soak4_fix_$s.txt | cut -c1-200; }
done; echo "fixed library: $fails of 12 failed"
timeout -k 5 300 python scripts/soak_stdsort.py 300 5 2>&1 | tail -1
