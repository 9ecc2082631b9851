#!/bin/bash
# usage: tools/pmc4.sh <outdir> <bench args...> -- <counters...>   PMC pass over an arbitrary bench.py invocation
out=$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${CFFM_PROF_TIMEOUT:-300} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/bench.py "${args[@]}" > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/pmc_report.py gpurun_out/$out
