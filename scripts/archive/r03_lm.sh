#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_golden.py tests/test_gpu_edge_cases.py tests/test_gpu_facade.py -q -m gpu -k "scan2map or golden or lm or track or rccl or facade or gf or good" 2>&1 | tail -8
timeout 250 python scripts/exp/s2m_time.py 2>&1 | tail -2
timeout 200 python scripts/trackbench.py 2>&1 | tail -4
timeout 300 python scripts/framebench.py 2>&1 | grep "one launch set, both kinds" 
