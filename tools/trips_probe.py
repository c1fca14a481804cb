#!/usr/bin/env python3
"""Measurement build only (VISMA_GRID_DEBUG_TRIPS): batch trips per query / per wave at C4.
   VISMA_ICP_LIB=visma_amd/lib/libvisma_icp_dbg.so python tools/trips_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visma_amd import _lib, synth
ns, nt = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (262144, 4194304)
src, tgt, T_gt, r = synth.make_pair(ns, nt, motion="radius")
c = _lib.Context(0); c.set_clouds_f64(src, tgt); c.set_nn_mode(_lib.NN_GRID)
T, _ = c.iterate(np.eye(4), r, 6)
c.nn_pass(T, r)
code = c.correspondence_index()
trips = code & 127; matched = (code >> 7) & 1; wmax = (code >> 8) & 255; wmax_m = (code >> 16) & 255
print("queries", len(code), "matched", matched.mean())
print("trips per query: mean %.2f  p50 %d p90 %d p99 %d max %d" % (trips.mean(), *np.percentile(trips, [50, 90, 99]), trips.max()))
print("trips unmatched: mean %.2f" % trips[matched == 0].mean(), " matched: mean %.2f" % trips[matched == 1].mean())
print("wave max (mean over queries = over waves): %.2f   if only matched lanes counted: %.2f" % (wmax.mean(), wmax_m.mean()))
print("hist trips", np.bincount(trips)[:20])
print("hist wave max", np.bincount(wmax)[:24] // 64)
print("hist wave max (matched only)", np.bincount(wmax_m)[:24] // 64)
