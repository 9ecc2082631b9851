#!/bin/bash
# usage: tools/pmc2.sh <outdir> <script args...> -- <counters...>
out=$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${CFFM_PROF_TIMEOUT:-300} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/tools/stage.py "${args[@]}" > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
