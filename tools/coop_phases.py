#!/usr/bin/env python3
"""GPU probe: where a wave of the warm-started search spends its cycles (measurement build
-DVISMA_COOP_DEBUG_PHASES, built to a side library).   python tools/coop_phases.py [ns nt]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visma_amd import build  # noqa: E402

SIDE = os.path.join(ROOT, "visma_amd", "lib", "libvisma_icp_spans.so" if "--spans" in sys.argv else "libvisma_icp_phases.so")
if not os.path.exists(SIDE) or "--rebuild" in sys.argv:
    build.build_lib(force=True, defines=("VISMA_COOP_DEBUG_PHASES=2" if "--spans" in sys.argv else "VISMA_COOP_DEBUG_PHASES=1",), out=SIDE)
os.environ["VISMA_ICP_LIB"] = SIDE
from visma_amd import _lib, synth  # noqa: E402

NAMES = ["src+slot", "bounds+prev, prune", "list written", "chunks", "merge", "f64 winner", "out+moments", "partial row", "fold"]


def main():
    a = [int(x) for x in sys.argv[1:] if x.isdigit()]
    ns, nt = (a + [262144, 4194304])[:2] if len(a) >= 2 else (262144, 4194304)
    src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
    c = _lib.Context(0)
    c.set_clouds_f64(src, tgt)
    c.set_nn_mode(_lib.NN_GRID)
    c.iterate(np.eye(4), r, 6)
    L = _lib.load()
    out = (C.c_ulonglong * 16)()
    L.visma_debug_coop_phases(out, 1)
    steps = 10
    c.iterate(np.eye(4), r, steps)
    L.visma_debug_coop_phases(out, 0)
    nwg = (ns + 255) // 256 * steps
    tot = sum(out[k] for k in range(9))
    print("ns=%d nt=%d: cycles per workgroup (wave 0) and share; 100 MHz counter" % (ns, nt))
    for k in range(9):
        print("  %-22s %9.1f ticks = %6.2f us  %5.1f%%" % (NAMES[k], out[k] / nwg, out[k] / nwg / 100.0, 100.0 * out[k] / max(tot, 1)))
    print("  total %.2f us" % (tot / nwg / 100.0))
    # when the waves of the LAST launch started and reached the block reduction (100 MHz clock)
    nw = min((ns + 255) // 256, 2048) * 4
    sp = (C.c_ulonglong * (2 * nw))()
    L.visma_debug_coop_spans(sp, 2 * nw)
    a = np.array(sp[:], dtype=np.int64).reshape(nw, 2)
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    q = [0, 10, 50, 90, 99, 100]
    print("  wave start  us, percentiles %s: %s" % (q, np.percentile(st, q).round(2).tolist()))
    print("  wave done   us, percentiles %s: %s" % (q, np.percentile(en, q).round(2).tolist()))
    print("  wave length us, percentiles %s: %s" % (q, np.percentile(en - st, q).round(2).tolist()))
    ln = en - st
    blk = np.arange(nw) // 4
    print("  mean / max wave length by XCD (block & 7): " + ", ".join("%d: %.1f/%.1f" % (x, ln[(blk & 7) == x].mean(), ln[(blk & 7) == x].max()) for x in range(8)))
    seq = blk >> 3
    nb = max(int(seq.max()) + 1, 1)
    print("  mean wave length by dispatch order within the XCD (eighths): " + ", ".join("%.1f" % ln[(seq * 8 // nb) == e].mean() for e in range(8)))
    print("  XCD 6, by sixteenths of its query range: " + ", ".join("%.1f" % ln[((blk & 7) == 6) & ((seq * 16 // nb) == e)].mean() for e in range(16)))
    if "--spans" in sys.argv:
        mk = (C.c_ulonglong * (16 * nw))()
        L.visma_debug_coop_marks(mk, 16 * nw)
        m = np.array(mk[:], dtype=np.int64).reshape(nw, 16)[:, :7]
        prev = a[:, 0]
        first, last = (seq * 8 // nb) < 2, (seq * 8 // nb) >= 6
        print("  phase durations per wave, us: median all | first-dispatched quarter | last-dispatched quarter | max")
        for k in range(7):
            d = (m[:, k] - prev) / 100.0
            ok = m[:, k] > 0
            print("    %-22s %6.2f | %6.2f | %6.2f | %6.2f" % (NAMES[k], np.median(d[ok]), np.median(d[ok & first]), np.median(d[ok & last]), d[ok].max()))
            prev = np.where(ok, m[:, k], prev)
    late = np.argsort(en)[-8:]
    print("  last waves: " + ", ".join("w%d %.2f-%.2f" % (w, st[w], en[w]) for w in late))


if __name__ == "__main__":
    main()
