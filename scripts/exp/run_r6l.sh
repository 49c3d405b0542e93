for rep in 1 2; do for v in - tagsleep0 tagsleep4 tagsleep12; do
if [ $v = - ]; then L=$PWD/m-loam_amd/lib/libmloam_hip.so; else L=$PWD/m-loam_amd/lib_ab/$v/libmloam_hip.so; fi
MLOAM_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 50 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('scan2map') or {}
print('$v', 's2m', s.get('ms_per_frame'), s.get('ms_per_frame_pipelined'), 'frame', d['frame']['ms_per_frame'])"
done; done
