timeout -k 5 600 python scripts/soak_stdsort.py 300 91 2>&1 | tail -2
MLOAM_HIP_LIB=$PWD/m-loam_amd/lib/libmloam_hip_dbg.so timeout -k 5 300 python scripts/stageclock_sort.py 2>&1 | grep -E "stored|heap sorts|recursion done \(this|slowest rings \(rec"
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "voxel or sort or thin or downsample or extract" 2>&1 | tail -3
REPS_OUTER=2 bash scripts/ab_thin.sh - 2>&1 | cut -c1-130
