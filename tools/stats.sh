#!/bin/bash
# usage: tools/stats.sh <outdir> [bench args]   (run on the GPU box from the repo root)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${CFFM_PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
cd $GRAFT_REPO_ROOT && python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/$out/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print('%-70s calls %5s avg_us %8.2f  %5s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:5]))
PY
