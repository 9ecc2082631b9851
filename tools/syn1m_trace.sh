#!/bin/bash
# usage: tools/syn1m_trace.sh <tag>   kernel trace of 3 stress-shape steps (GPU box, repo root) -> gpurun_out/<tag>.md
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${CFFM_PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --workload syn1m --steps 3 --warmup 1 --quick > $R/gpurun_out/$tag.log 2>&1
cd $R && python3 tools/stats_md.py gpurun_out/$tag "$tag" > gpurun_out/$tag.md; head -16 gpurun_out/$tag.md; tail -2 gpurun_out/$tag.log | cut -c1-300
find gpurun_out/$tag -name '*kernel_trace.csv' -delete
