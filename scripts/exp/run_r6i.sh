timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -x -q -m gpu -k "fps or good_feature or selection or methods" 2>&1 | tail -5
GF_ONLY=fps timeout -k 5 300 python scripts/gfbench.py 2>&1 | tail -1
GF_ONLY=fps MLOAM_HIP_LIB=$PWD/m-loam_amd/lib_ab/fpsstats/libmloam_hip.so timeout -k 5 300 python scripts/gfbench.py 2>&1 | sort | uniq -c | sort -rn | head -5 | cut -c1-700
