#!/bin/bash
# Round-4 evidence run (GPU box, from the repo root): the default bench line, kernel traces of the contract workload and of the
# stress shape (bf16x3 loops and, for the A/B, the fp32 MFMA loops), PMC passes for the stress-shape kernels (counters in runs of
# their own, --kernel-trace only), and the wide parity cases under both conv loops with their worst err / bound per tensor.
# usage: tools/profile_r04.sh <tag>
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
cd $R && python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
run() { out=$1; shift; timeout -k 10 ${CFFM_PROF_TIMEOUT:-600} rocprofv3 "$@" > $R/gpurun_out/$out.log 2>&1; echo "$out rc=$?"; }
run ${tag}_kt_frappe --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_kt_frappe -- python3 $R/bench.py --steps 50 --warmup 5 --blocks 1 --no-cpu-baseline
run ${tag}_kt_syn1m --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_kt_syn1m -- python3 $R/bench.py --workload syn1m --steps 3 --warmup 1 --quick
run ${tag}_pmcA --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmcA -- python3 $R/bench.py --workload syn1m --steps 2 --warmup 1 --quick
run ${tag}_pmcB --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmcB -- python3 $R/bench.py --workload syn1m --steps 2 --warmup 1 --quick
cd $R
CFFM_CONV_FP32=1 python3 bench.py --workload syn1m --steps 3 --warmup 1 --quick > gpurun_out/${tag}_syn1m_fp32.json 2>/dev/null
python3 bench.py --workload syn1m --steps 3 --warmup 1 --quick > gpurun_out/${tag}_syn1m_b3.json 2>/dev/null
for d in ${tag}_kt_frappe ${tag}_kt_syn1m; do python3 tools/stats_md.py gpurun_out/$d "$d" > gpurun_out/$d.md 2>&1; done
for d in ${tag}_pmcA ${tag}_pmcB; do python3 tools/pmc_report.py gpurun_out/$d > gpurun_out/$d.txt 2>&1; done
# the wide parity cases under both conv loops
K='f32 or f16-d32 or f16-k8 or f20 or f33 or cfg4'
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "$K" > gpurun_out/${tag}_par_b3.log 2>&1; tail -1 gpurun_out/${tag}_par_b3.log
cp gpurun_out/parity_worst.json gpurun_out/${tag}_parity_worst_b3.json
CFFM_CONV_FP32=1 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "$K" > gpurun_out/${tag}_par_fp32.log 2>&1; tail -1 gpurun_out/${tag}_par_fp32.log
cp gpurun_out/parity_worst.json gpurun_out/${tag}_parity_worst_fp32.json
find gpurun_out/${tag}_* -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null
du -sh gpurun_out/${tag}_* | tail -12
