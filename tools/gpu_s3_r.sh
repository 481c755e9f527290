#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/host_stream_bench.py 134217728 16777216 2>&1 | tail -3
timeout 600 python tools/host_stream_bench.py 134217728 1048576 2>&1 | tail -1
