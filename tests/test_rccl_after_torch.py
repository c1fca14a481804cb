"""A host program that has imported PyTorch carries PyTorch's bundled RCCL under the soname the library's
own dlopen would ask for: the library binds the system's copy to itself (RTLD_LOCAL | RTLD_DEEPBIND), so a
one-rank communicator must come up -- and all-reduce -- in a process where torch was imported first
(what the ranks of `bench.py --gpus N` are)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r"""
import sys
sys.path.insert(0, %r)
import torch                      # first: its lib/ directory is on the loader's list from here on
import torch.distributed          # (the c10d bindings pull the bundled RCCL in)
import numpy as np
from visma_amd import _lib, synth
src, tgt, T_gt, r = synth.make_pair(4000, 16000, motion="radius")
one = _lib.Context(0); one.set_clouds_f64(src, tgt)
w = one.run(None, r, 8, 0.0, 0.0)
c = _lib.Context(0); c.set_clouds_f64(src, tgt)
c.comm_init(0, 1, _lib.comm_unique_id())
for loop in (False, True):
    c.set_device_loop(loop)
    g = c.run(None, r, 8, 0.0, 0.0)
    assert g.num_correspondences == w.num_correspondences
    assert synth.rel_frobenius(g.transformation_, w.transformation_) < 1e-12, synth.rel_frobenius(g.transformation_, w.transformation_)
print("RCCL-AFTER-TORCH-OK", torch.__version__)
"""


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_the_library_communicator_comes_up_in_a_process_that_imported_torch(lib):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", PROG % ROOT], capture_output=True, text=True, timeout=500, env=env)
    assert p.returncode == 0 and "RCCL-AFTER-TORCH-OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
