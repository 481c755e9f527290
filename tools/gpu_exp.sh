#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "c_abi" 2>&1 | tail -5
