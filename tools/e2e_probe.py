import os, sys, time
sys.path.insert(0, "/root/repo")
if os.environ.get("E2E_TRACE", "1") != "0":
    os.environ["VISMA_ICP_UPLOAD_TRACE"] = "1"      # (the trace drains the stream inside the upload: E2E_TRACE=0 times the real thing)
import numpy as np
from visma_amd import _lib, synth
for ns, nt, r, it in ((262144, 4194304, None, 30), (5000, 20000, 0.075, 20)):
    src, tgt, T, rr = synth.make_pair(ns, nt, motion="radius" if r is None else "fixed")
    r = r or rr
    c = _lib.Context(0)
    c.set_clouds_f64(src, tgt); c.run(None, r, it, 0, 0)
    for k in range(5):
        t0 = time.perf_counter(); c.set_clouds_f64(src, tgt); t1 = time.perf_counter(); res = c.run(None, r, it, 0, 0); t2 = time.perf_counter()
        print("ns=%d nt=%d upload %.3f ms run %.3f ms" % (ns, nt, (t1-t0)*1e3, (t2-t1)*1e3), flush=True)
    c.set_profiling(1); c.get_timing(reset=True)
    c.set_clouds_f64(src, tgt); res = c.run(None, r, it, 0, 0)
    tm = c.get_timing(reset=True)
    print({k: round(v, 4) for k, v in tm.items() if isinstance(v, float) and v and not k.startswith("tile")}, tm["nn_launches"], tm["aux_launches"])
