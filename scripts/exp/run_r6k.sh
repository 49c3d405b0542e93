mkdir -p gpurun_out/r6k
fails=0
for s in 501 502 503; do
timeout -k 5 400 python scripts/soak_api.py 40 $s 4 > gpurun_out/r6k/soak4_long_$s.txt 2>&1 || { fails=$((fails+1)); grep -A3 MISMATCH gpurun_out/r6k/soak4_long_$s.txt | cut -c1-250 | head -5; }
tail -2 gpurun_out/r6k/soak4_long_$s.txt | cut -c1-300
done
SOAK_BIG=1 timeout -k 5 600 python scripts/soak_api.py 60 504 4 > gpurun_out/r6k/soak4_big.txt 2>&1 || { fails=$((fails+1)); grep -A3 MISMATCH gpurun_out/r6k/soak4_big.txt | cut -c1-250 | head -5; }
tail -2 gpurun_out/r6k/soak4_big.txt | cut -c1-300
echo "failed: $fails"
