#!/usr/bin/env python3
"""Memory-event skeleton of one kernel out of hipcc's assembly: global / flat / scratch loads and stores, LDS DMA, s_waitcnt vmcnt,
barriers and MFMA runs in program order (line numbers inside the kernel).  What it is for: finding latency chains the source does
not show - a load that hipcc sank next to its use (load, vmcnt(0), use, load, ...), a vmcnt(0) inside a predicated store block
(which also waits for the store in front of it), spills on the critical path.

LIMITS (learned the hard way): this is the STATIC code.  (1) It lists branches the workload never takes - the four
"load x3, wait, store x5" blocks of conv01_bwd_kernel are the `!first` side of the layer-0 slab update, dead in the fused step where a
workgroup holds one example; the staging loop it shows in fwd_all_kernel is the non-`early` path.  Read the branch conditions before
believing a chain is executed.  (2) A chain that is executed may still be covered by other wavefronts: of three gains predicted from
counting round trips here, none came out as predicted and one was a loss.  Use it to find candidates; decide with
tools/experiments/ab_old_new_trace.sh (old and new library on one box).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -S --cuda-device-only cffm_amd/csrc/conv.hip -o /tmp/conv.s
    tools/isa_waits.py /tmp/conv.s fwd_all_kernelILi3ELi8ELi3E [first_line last_line]
"""
import re, sys

def main():
    path, pat = sys.argv[1], sys.argv[2]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
    s = open(path).read()
    names = [m.group(1) for m in re.finditer(r'\n(_Z\w+):\s+; @', s) if pat in m.group(1)]
    if not names:
        sys.exit('no kernel matches ' + pat)
    for nm in names:
        i = s.index('\n' + nm + ':'); j = s.index('.Lfunc_end', i)
        lines = s[i:j].split('\n')
        ev = []
        for k, l in enumerate(lines):
            m = re.match(r'\s+(global_load_\w+|flat_load_\w+|flat_store_\w+|global_store_\w+|global_atomic_\w+|scratch_\w+|'
                         r's_waitcnt[^;\n]*vmcnt[^;\n]*|s_barrier|v_mfma\w+)', l)
            if m:
                t = m.group(1).strip()
                ev.append((k, 'MFMA' if t.startswith('v_mfma') else t))
        out, prev, cnt, k0 = [], None, 0, 0
        for k, t in ev:
            if t == prev:
                cnt += 1
            else:
                if prev: out.append((k0, prev, cnt))
                prev, cnt, k0 = t, 1, k
        if prev: out.append((k0, prev, cnt))
        meta = re.search(r'\.name:\s+' + nm + r'.*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', s, re.S)
        print('%s  (%d lines, vgpr %s, spilled %s)' % (nm, len(lines), meta.group(1), meta.group(2)))
        for k0, t, c in out:
            if lo <= k0 <= hi: print('%6d  %s%s' % (k0, t, '' if c == 1 else ' x %d' % c))

if __name__ == '__main__':
    main()
