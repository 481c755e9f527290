#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_csv.py -q -x --timeout 900 -k "seven" 2>&1 | tail -n 40
