#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "avg" 2>&1 | tail -25
