"""Time config C3: 12 objects x 24 yaw initialisations = 288 small ICPs."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visma_amd import _lib, synth

ctx = _lib.Context(0)
rng = np.random.default_rng(3)
objs = []
for i in range(12):
    ns = int(rng.integers(4000, 40000)); nt = ns // 2
    src, tgt, _, _ = synth.make_pair(ns, nt, seed_t=300 + i, seed_s=400 + i)
    objs.append((src, tgt))
probs = []
for src, tgt in objs:
    for k in range(24):
        probs.append((src, tgt, synth.make_T(synth.rot_y(2 * np.pi * k / 24), [0, 0, 0]), 0.02))
for label, dev in (("all 288 in flight (batched device loop)", None), ("one at a time (host loop)", False)):
    ctx.set_device_loop(dev)
    ctx.run_batch(probs[:24], max_iter=30)
    t0 = time.time()
    res = ctx.run_batch(probs, max_iter=30)
    dt = time.time() - t0
    its = sum(r.iterations for r in res)
    print(json.dumps(dict(mode=label, problems=len(probs), seconds=dt, total_iterations=its,
                          icp_iterations_per_s=its / dt, source_points=sum(len(p[0]) for p in probs))))
# the same workload as 12 yaw sweeps (shared clouds per object)
ctx.set_device_loop(None)
t0 = time.time(); its = 0
for src, tgt in objs:
    ctx.set_clouds_f64(src, tgt)
    best, lvl, per = ctx.run_yaw_sweep(24, 0.02)
    its += sum(p.iterations for p in per)
dt = time.time() - t0
print(json.dumps(dict(mode="12 x visma_icp_run_yaw_sweep(24)", seconds=dt, total_iterations=its, icp_iterations_per_s=its / dt)))
