MLOAM_HIP_LIB=$PWD/m-loam_amd/lib_ab/keepregs/libmloam_hip.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_residency.py -x -q -m gpu -k "scan2map or lm or residency or loop or barrier or frame" 2>&1 | tail -2
for rep in 1 2 3; do for v in - keepregs; do
if [ $v = - ]; then L=$PWD/m-loam_amd/lib/libmloam_hip.so; else L=$PWD/m-loam_amd/lib_ab/$v/libmloam_hip.so; fi
MLOAM_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('scan2map') or {}
print('$v', 's2m', s.get('ms_per_frame'), s.get('ms_per_frame_pipelined'), 'frame', d['frame']['ms_per_frame'])"
done; done
