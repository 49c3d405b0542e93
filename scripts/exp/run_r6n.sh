timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do for v in knnold -; do
if [ $v = - ]; then L=$PWD/m-loam_amd/lib/libmloam_hip.so; else L=$PWD/m-loam_amd/lib_ab/$v/libmloam_hip.so; fi
MLOAM_HIP_LIB=$L python bench.py --no-cpu-baseline --no-supplementary --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', 'step', d['ms_per_step'], 'frac', r['frac'], 'knn us', r['avg_kernel_us'], d.get('kernel_us_per_launch'))" | cut -c1-330
done; done
