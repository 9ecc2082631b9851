#!/bin/bash
# Same-box A/B of two builds of the library (GPU box): tools/experiments/ab_lib/ = conv.hip of commit 404b4a7 (before the late
# round-4 changes of DESIGN 3.5) linked with HEAD's other objects, cffm_amd/lib/ = HEAD.  Each leg goes through its OWN pybind11
# module and ABORTS unless the libcffm_hip image it mapped is the one it is meant to time (CFFM_HIP_LIB alone is not enough: the
# pybind11 module binds to the library next to it).
# Recipe for the other library (here, not on the box):
#   git show <commit>:cffm_amd/csrc/conv.hip > cffm_amd/csrc/conv_old.hip      # must sit in csrc/: common.hpp includes ../../include
#   hipcc <CXXFLAGS of the Makefile> -c cffm_amd/csrc/conv_old.hip -o build/ab/conv_old.o && rm cffm_amd/csrc/conv_old.hip
#   hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v '^build/conv.o$') build/ab/conv_old.o -o tools/experiments/ab_lib/libcffm_hip.so
#   g++ ... pybind_module.cpp -o tools/experiments/ab_lib/_cffm_pybind$(python3-config --extension-suffix) -Ltools/experiments/ab_lib -lcffm_hip -Wl,-rpath,'$ORIGIN'
#   cp cffm_amd/lib/libcffm_libfm.so tools/experiments/ab_lib/       (build/ is NOT sent to the GPU box; tools/experiments/ab_lib/*.so is, and is git-ignored)
set -u
R=$GRAFT_REPO_ROOT
leg() { # tag, expected library path fragment, forbidden fragment, env...
  tag=$1; want=$2; forbid=$3; shift 3
  env "$@" TAG=$tag WANT=$want FORBID=$forbid ROOT=$R python3 - <<'PY' || exit 1
import os, sys, json
R = os.environ['ROOT']; sys.path.insert(0, R)
import torch
from cffm_amd import hip
hip.fast()
maps = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libcffm_hip' in l or '_cffm_pybind' in l})
rel = [os.path.relpath(m, R) for m in maps]
ok = any(os.environ['WANT'] in m for m in rel) and not any(os.environ['FORBID'] in m for m in rel if 'libcffm_hip' in m)
if not ok or hip.binding_name() != 'pybind11':
    sys.exit('%s leg mapped the wrong library: %s (binding %s)' % (os.environ['TAG'], rel, hip.binding_name()))
import subprocess
out = {}
for wl, extra in (('syn1m', ['--steps', '3', '--warmup', '1']), ('frappe', ['--blocks', '5'])):
    p = subprocess.run([sys.executable, os.path.join(R, 'bench.py'), '--workload', wl, '--quick'] + extra, capture_output=True, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    if p.returncode != 0 or not lines:
        sys.exit('%s leg: bench.py %s failed (rc %d): %s' % (os.environ['TAG'], wl, p.returncode, p.stderr[-400:]))
    out[wl] = json.loads(lines[-1])['ms_per_step']
print(os.environ['TAG'], json.dumps(out), 'mapped', rel)
PY
}
for rep in 1 2; do
  leg old tools/experiments/ab_lib/libcffm_hip.so cffm_amd/lib/libcffm_hip.so CFFM_HIP_LIB=$R/tools/experiments/ab_lib/libcffm_hip.so CFFM_HOST_LIB_DIR=$R/tools/experiments/ab_lib
  leg new cffm_amd/lib/libcffm_hip.so tools/experiments/ab_lib/libcffm_hip.so X=1
done
