#!/bin/bash
# Same-box kernel traces of the frappe step under the two library builds of ab_old_new.sh (variables exported in the shell, the
# program after `--` is python3 itself).  Prints the four step kernels' average durations per leg.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
trace() { tag=$1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abt_$tag -- python3 $R/bench.py --workload frappe --steps 300 --warmup 30 --blocks 1 --quick > $R/gpurun_out/abt_$tag.log 2>&1 || { echo "$tag trace failed"; tail -5 $R/gpurun_out/abt_$tag.log; exit 1; }
  python3 - $R/gpurun_out/abt_$tag $tag <<'PY'
import csv, glob, sys
d, tag = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
want = ('fwd_all_kernel', 'bwd_top_kernel', 'conv01_bwd_kernel', 'update_all_kernel')
out = {}
for r in csv.DictReader(open(f)):
    for w in want:
        if w in r['Name']: out[w] = (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3)
print(tag, ' '.join('%s n=%d avg %.2f min %.2f' % (w.replace('_kernel', ''), *out[w]) for w in want), '| sum avg %.2f us' % sum(out[w][1] for w in want))
PY
  find $R/gpurun_out/abt_$tag -name "*kernel_trace.csv" -delete
}
for rep in 1 2; do
  ( export CFFM_HIP_LIB=$R/tools/experiments/ab_lib/libcffm_hip.so CFFM_HOST_LIB_DIR=$R/tools/experiments/ab_lib; trace old$rep ) || exit 1
  ( trace new$rep ) || exit 1
done
