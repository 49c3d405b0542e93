mkdir -p gpurun_out/r6i
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fps or good_feature" 2>&1 | tail -5
GF_ONLY=fps timeout -k 5 300 python scripts/gfbench.py 2>&1 | tail -2
GF_ONLY=fps MLOAM_HIP_LIB=$PWD/m-loam_amd/lib_ab/fpsstats/libmloam_hip.so timeout -k 5 300 python scripts/gfbench.py 2>&1 | sort | uniq -c | sort -rn | head -6
