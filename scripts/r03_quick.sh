#!/bin/bash
python -m pytest tests/test_gpu_edge_cases.py -q -m gpu -k "submitted or chained" 2>&1 | tail -8
timeout 300 python scripts/framebench.py 2>&1 | grep "one launch set, both kinds" 
timeout 300 python scripts/framebench.py 2>&1 | grep "one launch set, both kinds" 
