#!/bin/bash
# Round-2 evidence run (GPU box, from the repo root): kernel traces of the contract workload and of the stress shape, PMC
# passes (counters in their own runs, --kernel-trace only) for the stress-shape kernels and for the gather kernels.
# usage: tools/profile_r02.sh <tag>
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { out=$1; shift; timeout -k 10 ${CFFM_PROF_TIMEOUT:-600} rocprofv3 "$@" > $R/gpurun_out/$out.log 2>&1; echo "$out rc=$?"; }
# 1) kernel trace, frappe (contract workload) and the stress shape
run ${tag}_kt_frappe --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_kt_frappe -- python3 $R/bench.py --steps 50 --warmup 5 --blocks 1 --no-cpu-baseline
run ${tag}_kt_syn1m --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_kt_syn1m -- python3 $R/bench.py --workload syn1m --steps 3 --warmup 1 --quick
# 2) PMC passes over the stress shape (4 SQ counters per pass; FETCH_SIZE and WRITE_SIZE in passes of their own)
run ${tag}_pmcA --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmcA -- python3 $R/bench.py --workload syn1m --steps 2 --warmup 1 --quick
run ${tag}_pmcB --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmcB -- python3 $R/bench.py --workload syn1m --steps 2 --warmup 1 --quick
run ${tag}_pmcC --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmcC -- python3 $R/bench.py --workload syn1m --steps 2 --warmup 1 --quick
run ${tag}_pmcD --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmcD -- python3 $R/bench.py --workload syn1m --steps 2 --warmup 1 --quick
# 3) gather kernels: HBM traffic of the roofline launches inside the default bench command
run ${tag}_gF --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_gF -- python3 $R/bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline
run ${tag}_gW --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_gW -- python3 $R/bench.py --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline
cd $R
for d in ${tag}_kt_frappe ${tag}_kt_syn1m; do python3 tools/stats_md.py gpurun_out/$d "$d" > gpurun_out/$d.md 2>&1; done
for d in ${tag}_pmcA ${tag}_pmcB ${tag}_pmcC ${tag}_pmcD ${tag}_gF ${tag}_gW; do python3 tools/pmc_report.py gpurun_out/$d > gpurun_out/$d.txt 2>&1; done
# keep the merged output small: the raw csv of the big traces is not needed
find gpurun_out/${tag}_* -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null
du -sh gpurun_out/${tag}_* | tail -12
