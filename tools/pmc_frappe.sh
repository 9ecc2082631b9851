#!/bin/bash
# usage: tools/pmc_frappe.sh <tag>   FETCH_SIZE and WRITE_SIZE of every kernel of the frappe step, one counter per pass
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$c -- python3 $R/bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline > $R/gpurun_out/${tag}_$c.log 2>&1 || exit 1
  (cd $R && python3 tools/pmc_report.py gpurun_out/${tag}_$c > gpurun_out/${tag}_$c.txt 2>&1; head -9 gpurun_out/${tag}_$c.txt)
  find $R/gpurun_out/${tag}_$c -name '*.csv' -size +2M -delete
done
