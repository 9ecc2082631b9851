#!/bin/bash
# usage: tools/pmc_frappe_sq.sh <tag>   SQ counters of every kernel of the frappe step, two passes of four counters
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_A -- python3 $R/bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline > $R/gpurun_out/${tag}_A.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_B -- python3 $R/bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline > $R/gpurun_out/${tag}_B.log 2>&1 || exit 1
cd $R
for t in A B; do python3 tools/pmc_report.py gpurun_out/${tag}_$t > gpurun_out/${tag}_$t.txt 2>&1; grep -E "kernel|fwd_all|conv01|bwd_top|update_all" gpurun_out/${tag}_$t.txt; find gpurun_out/${tag}_$t -name '*.csv' -size +2M -delete; done
