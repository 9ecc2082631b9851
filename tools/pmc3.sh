#!/bin/bash
# usage: tools/pmc3.sh <outdir> <counters...>   PMC pass over the fused frappe train step (bench.py --quick, 30 steps)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${CFFM_PROF_TIMEOUT:-300} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --quick > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/pmc_report.py gpurun_out/$out
