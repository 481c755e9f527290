#!/bin/bash
export TMPDIR=/tmp
for wl in cfg3 headline; do timeout 300 python tools/prof_query.py $wl 134217728 1 agg.partition_mode=258 2>&1 | grep "PA p=" | tail -12; done
