#!/usr/bin/env python3
"""Static per-phase instruction histogram of nn_coop_kernel_one<false> from the ISA of the measurement build
(-DVISMA_COOP_DEBUG_PHASES=2 --save-temps: the clock stamps of grid_coop_probe.h are the phase borders).  Runs where
hipcc is (no GPU needed).   python tools/isa_histogram.py > profiles/rNN_coop_isa_histogram.txt"""
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {9: 'A done', 10: 'past barrier 1', 0: 'query taken over', 1: 'rows asked for', 2: 'list written', 3: 'chunks worked off',
         4: 'merged', 5: 'f64 winner ranked', 11: 'search done', 12: 'past last barrier', 13: 'phase C done', 6: 'rounds done',
         7: 'partial row stored', 8: 'fold done', None: '(wave span stamp)'}


def klass(op):
    if op.startswith('v_'):
        return 'VALU'
    if op.startswith(('s_load', 's_memrealtime')):
        return 'SMEM'
    if op.startswith(('s_waitcnt', 's_barrier', 's_cbranch', 's_branch', 's_nop', 's_sleep', 's_endpgm')):
        return 'ctrl'
    if op.startswith('s_'):
        return 'SALU'
    if op.startswith('ds_'):
        return 'LDS'
    if op.startswith(('global_', 'scratch_', 'flat_', 'buffer_')):
        return 'VMEM'
    return 'other'


def main():
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
                               "-w", "-DVISMA_COOP_DEBUG_PHASES=2", "-x", "hip", "-c", os.path.join(ROOT, "visma_amd", "csrc", "grid_coop.hip"),
                               "-o", os.path.join(d, "g.o"), "--save-temps=obj"])
        lines = open(os.path.join(d, "grid_coop-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_ZN5visma18nn_coop_kernel_oneILb0E'))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    segs, cur, label = [], collections.Counter(), None
    for i, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
            continue
        op = t.split()[0]
        if op == 's_memrealtime':
            k = None
            for j in range(i, min(i + 14, len(body))):
                m = re.search(r'g_coop_marksE@rel32@lo\+(\d+)', body[j])
                if m:
                    k = (int(m.group(1)) - 4) // 8
                    break
            segs.append((NAMES.get(k, 'mark %s' % k), cur))
            cur = collections.Counter()
            continue
        cur[klass(op)] += 1
        if op.startswith('v_') and 'f64' in op:
            cur['f64'] += 1
    segs.append(('end of kernel', cur))
    print("nn_coop_kernel_one<false>, measurement build: STATIC instruction counts of the code that PRECEDES each clock stamp in")
    print("program order (a loop body counts once; block placement shuffles the logical order; each stamp's own address")
    print("arithmetic -- ~8 SALU / VALU -- is included).  Dynamic counts per launch: profiles/rNN_c4_kernel_pmc_summary.csv.")
    print("%-32s %6s %6s %6s %6s %6s %6s" % ("code before the stamp", "VALU", "f64", "SALU", "LDS", "VMEM", "ctrl"))
    for name, c in segs:
        print("%-32s %6d %6d %6d %6d %6d %6d" % (name, c['VALU'], c['f64'], c['SALU'], c['LDS'], c['VMEM'], c['ctrl']))


if __name__ == "__main__":
    main()
