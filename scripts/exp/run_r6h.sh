mkdir -p gpurun_out/r6h
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fps or good_feature" 2>&1 | tail -15 > gpurun_out/r6h/fps_tests.txt
timeout 600 python scripts/gfbench.py > gpurun_out/r6h/gfbench.txt 2>&1
MLH_FPS_DENSE=1 timeout 600 python scripts/gfbench.py 2>&1 | grep fps > gpurun_out/r6h/gfbench_dense.txt
timeout 900 python scripts/soak_stdsort.py 300 77 2>&1 | tail -3 > gpurun_out/r6h/soak_stdsort.txt
REPS_OUTER=3 bash scripts/ab_thin.sh - leafwide2048 > /dev/null 2>&1; cp gpurun_out/ab_thin/result.txt gpurun_out/r6h/ab_thin.txt
cat gpurun_out/r6h/fps_tests.txt gpurun_out/r6h/gfbench.txt gpurun_out/r6h/gfbench_dense.txt gpurun_out/r6h/soak_stdsort.txt; cut -c1-120 gpurun_out/r6h/ab_thin.txt
