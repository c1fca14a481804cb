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
    T0, _ = c.iterate(np.eye(4), r, int(os.environ.get('PROBE_WARMUP', '6')))
    if os.environ.get('PROBE_CONTINUE', '1') != '1':
        T0 = np.eye(4)                                       # (restart from the identity: the jump defeats the certificates)
    L = _lib.load()
    out = (C.c_ulonglong * 16)()
    L.visma_debug_coop_phases(out, 1)
    steps = 10
    c.iterate(T0, r, steps)
    L.visma_debug_coop_phases(out, 0)
    nwg = (ns + 255) // 256 * steps
    tot = sum(out[k] for k in range(9))
    print("ns=%d nt=%d: cycles per workgroup (wave 0) and share; 100 MHz counter" % (ns, nt))
    for k in range(9):
        print("  %-22s %9.1f ticks = %6.2f us  %5.1f%%" % (NAMES[k], out[k] / nwg, out[k] / nwg / 100.0, 100.0 * out[k] / max(tot, 1)))
    print("  total %.2f us" % (tot / nwg / 100.0))
    # when the waves of the LAST launch started and reached the block reduction (100 MHz clock)
    nw = min((ns + 255) // 256, 2048) * 4
    # (waves per workgroup of the kernel under test)
    WPB = int(os.environ.get('PROBE_WPB', 4))
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
    blk = np.arange(nw) // WPB
    print("  mean / max wave length by XCD (block & 7): " + ", ".join("%d: %.1f/%.1f" % (x, ln[(blk & 7) == x].mean(), ln[(blk & 7) == x].max()) for x in range(8)))
    seq = blk >> 3
    nb = max(int(seq.max()) + 1, 1)
    print("  mean wave length by dispatch order within the XCD (eighths): " + ", ".join("%.1f" % ln[(seq * 8 // nb) == e].mean() for e in range(8)))
    print("  XCD 6, by sixteenths of its query range: " + ", ".join("%.1f" % ln[((blk & 7) == 6) & ((seq * 16 // nb) == e)].mean() for e in range(16)))
    if "--spans" in sys.argv:
        mk = (C.c_ulonglong * (16 * nw))()
        L.visma_debug_coop_marks(mk, 16 * nw)
        m = np.array(mk[:], dtype=np.int64).reshape(nw, 16)
        first, last = (seq * 8 // nb) < 2, (seq * 8 // nb) >= 6
        order = [(9, "A: src+state, certificate"), (10, "barrier 1"), (0, "S query taken over"), (1, "S rows asked for"),
                 (2, "S list written (+2 barriers)"), (3, "S chunks worked off"), (4, "S merged (+1 barrier)"), (5, "S f64 winner"),
                 (11, "S outputs"), (12, "last barrier"), (13, "C moments"), (7, "partial row"), (8, "fold")]
        searching = m[:, 11] > 0
        print("  waves that searched: %d of %d" % (int(searching.sum()), nw))
        print("  phase END times since the wave's start and durations, us: median end | median dur all | first quarter | last quarter | max dur  (S = searching waves only)")
        prev = a[:, 0].copy()
        for k, name in order:
            ok = m[:, k] > 0
            if not ok.any():
                continue
            d = (m[:, k] - prev) / 100.0
            e = (m[:, k] - a[:, 0]) / 100.0
            print("    %-28s %6.2f | %6.2f | %6.2f | %6.2f | %6.2f   (%d waves)" % (name, np.median(e[ok]), np.median(d[ok]), np.median(d[ok & first]) if (ok & first).any() else -1, np.median(d[ok & last]) if (ok & last).any() else -1, d[ok].max(), int(ok.sum())))
            prev = np.where(ok, m[:, k], prev)
        # where the waves ran: HW_ID = wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
        hw, xcc = m[:, 14], m[:, 15] & 0xF
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
        wv = np.arange(nw) % WPB
        print("  SIMD of wave k of a workgroup (rows: wave 0..3; columns: SIMD 0..3): " +
              " | ".join(" ".join("%4d" % int(((wv == k) & (simd == s_)).sum()) for s_ in range(4)) for k in range(min(WPB, 4))))
        print("  block b -> XCC: " + " ".join("%d" % int(xcc[WPB * b]) for b in range(16)))
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        for b in (0, 8, 16, 24, 256, 512, 768):
            if WPB * b < nw:
                print("    block %4d: xcc %d se %d sh %d cu %2d, simd of waves %s" % (b, xcc[WPB * b], se[WPB * b], sh[WPB * b], cu[WPB * b], simd[WPB * b:WPB * b + WPB].tolist()))
        # blocks sharing a CU with block 0 of XCC 0
        same = np.flatnonzero(cuid[::WPB] == cuid[0])
        print("    blocks on the CU of block 0: %s" % same[:12].tolist())
        # searching waves per (CU, SIMD)
        key = cuid[searching] * 4 + simd[searching]
        if key.size:
            cnts = np.bincount(np.unique(key, return_inverse=True)[1])
            print("    searching waves per (CU, SIMD) that has any: mean %.2f max %d; histogram %s" % (cnts.mean(), cnts.max(), np.bincount(cnts).tolist()))
        # the slowest searching waves: where did they lose their time?
        if searching.any():
            sd = np.where(searching, m[:, 11], 0)
            worst = np.argsort(sd)[-6:]
            names = [n for _, n in order]
            print("    slowest searching waves (durations per phase, us):")
            for w in worst:
                prevt = a[w, 0]
                parts = []
                for k, name in order[:9]:
                    if m[w, k] > 0:
                        parts.append("%s %.2f" % (name.split()[1] if name.startswith("S ") else name.split(":")[0], (m[w, k] - prevt) / 100.0))
                        prevt = m[w, k]
                print("      wave %5d (block %4d, xcc %d): " % (w, w // WPB, int(xcc[w])) + " | ".join(parts))
        t0g = a[:, 0].min()
        for k, name in ((9, "A done"), (10, "past barrier 1"), (11, "search done"), (12, "past barrier 2"), (7, "partial row stored"), (8, "fold / ticket done")):
            ok = m[:, k] > 0
            if ok.any():
                print("    launch clock, %-20s percentiles %s: %s" % (name, q, np.percentile((m[ok, k] - t0g) / 100.0, q).round(2).tolist()))
    late = np.argsort(en)[-8:]
    print("  last waves: " + ", ".join("w%d %.2f-%.2f" % (w, st[w], en[w]) for w in late))


if __name__ == "__main__":
    main()
