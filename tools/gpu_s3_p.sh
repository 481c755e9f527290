#!/bin/bash
export TMPDIR=/tmp
timeout 300 python tools/prof_query.py cfg3 67108864 0 2>&1 | grep "PAWG" | sort -k4 -n | head -40
