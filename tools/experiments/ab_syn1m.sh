#!/bin/bash
# A/B helper (GPU box): the wide parity cases, then the kernel trace of three stress-shape steps.  usage: tools/experiments/ab_syn1m.sh <tag>
tag=${1:-ab}
R=$GRAFT_REPO_ROOT
K="f32 or f16-d32 or f16-k8 or f20 or f33 or cfg4"
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "$K" > gpurun_out/${tag}_par.log 2>&1 || { tail -30 gpurun_out/${tag}_par.log; exit 1; }
tail -1 gpurun_out/${tag}_par.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_kt -- python3 $R/bench.py --workload syn1m --steps 3 --warmup 1 --quick > $R/gpurun_out/${tag}_kt.log 2>&1
cd $R
python3 tools/stats_md.py gpurun_out/${tag}_kt "$tag" > gpurun_out/${tag}_kt.md 2>&1
head -12 gpurun_out/${tag}_kt.md
find gpurun_out/${tag}_kt -name "*kernel_trace.csv" -size +8M -delete
grep '^{' gpurun_out/${tag}_kt.log | tail -1 | cut -c1-200
