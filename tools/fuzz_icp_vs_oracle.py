"""Distribution of the final-transform error of the GPU path against the f64 oracle (the CPU
restatement of the reference algorithm) over random registrations.  usage: fuzz_icp_vs_oracle.py [N] [seed] [f32|auto|f64]
Test infrastructure: imports oracle/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visma_amd import _lib, synth
from oracle.oracle import Oracle, Ref

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
# the checker: the compiled reference itself (oracle/_ref, Open3D's RegistrationICP on its KD-tree)
# when it is there -- much faster than the brute-force restatement -- else the restatement
o = Ref() if (Ref.available() and os.environ.get("FUZZ_CHECKER", "ref") == "ref") else Oracle()
print("checker:", type(o).__name__)
ctx = _lib.Context(0)
ctx.set_search_precision(sys.argv[3] if len(sys.argv) > 3 else "f32")      # f32 | auto | f64
errs = []
kernels = {}
for it in range(N):
    ns = int(rng.integers(500, 8000)); nt = int(rng.integers(2000, 40000))
    src, tgt, T_gt, r = synth.make_pair(ns, nt, seed_t=int(rng.integers(1 << 30)), seed_s=int(rng.integers(1 << 30)),
                                        noise=10.0 ** rng.uniform(-4, -2.5), motion="radius")
    r *= rng.uniform(0.7, 3.0)
    off = rng.standard_normal(3) * rng.choice([0.0, 1.0, 10.0])
    src = src + off; tgt = tgt + off
    init = synth.make_T(synth.rot_y(rng.uniform(-0.02, 0.02)), rng.standard_normal(3) * r * 0.3)
    iters = int(rng.integers(1, 40))
    want = o.registration_icp(src, tgt, r, init=init, max_iter=iters)
    ctx.set_clouds_f64(src, tgt)
    got = ctx.run(init, r, iters, 1e-6, 1e-6)
    e = synth.rel_frobenius(got.transformation_, want.T)
    errs.append(e)
    kernels[ctx.search_kernel_used()] = kernels.get(ctx.search_kernel_used(), 0) + 1
    if e > 1e-5 or got.num_correspondences != want.k:
        print("it=%d ns=%d nt=%d r=%.4g iters=%d/%d: rel %.3g  K %d vs %d" % (it, ns, nt, r, got.iterations, iters, e, got.num_correspondences, want.k))
errs = np.array(errs)
print("N=%d  median %.2g  p99 %.2g  max %.2g  (tolerance 1e-5); search kernel of the last pass: %s" % (N, np.median(errs), np.quantile(errs, 0.99), errs.max(), kernels))
