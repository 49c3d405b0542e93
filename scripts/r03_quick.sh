#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_parity_fullsize.py -q -m gpu -k "scan2map or golden or rccl or track" 2>&1 | grep -E "passed|failed" | tail -2
timeout 250 python scripts/exp/s2m_time.py 2>&1 | tail -3
