run() { tag=$1; shift; env "$@" python3 bench.py --quick --blocks 5 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
run base X=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run noscratchreclaim HSA_NO_SCRATCH_RECLAIM=1
run base2 X=1
