#!/bin/bash
# usage: tools/pmc_gather.sh <tag>   counters of the roofline kernel (tools/roofline_only.py = bench.py's gather_roofline):
# SQ passes (what the waves do), FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md: TCC slots), GRBM clock
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  n=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$n -- python3 $R/tools/roofline_only.py > $R/gpurun_out/${tag}_$n.log 2>&1 || return 1
  (cd $R && python3 tools/pmc_report.py gpurun_out/${tag}_$n > gpurun_out/${tag}_$n.txt 2>&1; grep -E "kernel|gather" gpurun_out/${tag}_$n.txt)
  find $R/gpurun_out/${tag}_$n -name '*.csv' -size +2M -delete
}
run A SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES &&
run B SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE &&
run F FETCH_SIZE &&
run W WRITE_SIZE
