timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fps or good_feature" 2>&1 | tail -2
for i in 1 2 3; do
GF_ONLY=fps MLOAM_HIP_LIB=$PWD/m-loam_amd/lib_ab/fppnoxor/libmloam_hip.so timeout -k 5 300 python scripts/gfbench.py 2>&1 | tail -1 | cut -c1-60
GF_ONLY=fps timeout -k 5 300 python scripts/gfbench.py 2>&1 | tail -1 | cut -c1-60
done
