#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_parity_fullsize.py tests/test_gpu_facade.py -q -m gpu -k "voxel or extract or facade or downsample or sort" 2>&1 | tail -4
timeout 300 python scripts/framebench.py 2>&1 | grep "one launch set, both kinds" 
