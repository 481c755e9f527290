export TMPDIR=/tmp
for a in "oneterm" "threecol" "product" "oneterm scan.fast=0"; do set -- $a; echo "== $a"; timeout 200 python tools/prof_query.py $1 1e9 4 $2 2>&1 | grep -v amdgpu.ids | tail -2; done
