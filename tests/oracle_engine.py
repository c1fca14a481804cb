"""An engine for visma_icp_create_with_engine backed by the CPU oracle.

TEST SEAM ONLY: lets the CPU test-suite run the product's host driver
(centring, ICP loop, solves, stop test, sharding, all-reduce hook) without a
GPU, with the oracle's kernel-specification functions (vk_*) standing in for
the HIP kernels.  The product never constructs this.
"""
import ctypes as C

import numpy as np

from visma_amd import _lib


class OracleEngine:
    def __init__(self, oracle, grid=True):
        self.o = oracle
        self.grid = grid
        self.src = self.tgt = self.nrm = None
        self.idx = None
        self.d2 = None
        self.T32 = None
        self.r2f = None
        self.calls = {"nn": 0, "reduce": 0}

        def set_source(_u, p, n):
            self.src = np.ctypeslib.as_array(p, shape=(max(n, 1), 4))[:n].copy()
            return 0

        def set_target(_u, p, n):
            self.tgt = np.ctypeslib.as_array(p, shape=(max(n, 1), 4))[:n].copy()
            return 0

        def set_normals(_u, p, n):
            self.nrm = np.ctypeslib.as_array(p, shape=(max(n, 1), 4))[:n].copy()
            return 0

        def nn_pass(_u, T, r):
            Tc = np.ctypeslib.as_array(T, shape=(16,)).copy()
            self.T32 = Tc[:12].astype(np.float32)
            self.r2f = np.float32(r * r)
            if len(self.tgt) == 0:
                self.idx = np.full(len(self.src), -1, np.int32)
                self.d2 = np.full(len(self.src), self.r2f, np.float32)
            else:
                _, self.idx, self.d2 = self.o.k_nn_pass(self.src, self.tgt, self.T32, self.r2f,
                                                       grid=self.grid)
            self.calls["nn"] += 1
            return 0

        def reduce(_u, T, off, plane, out):
            Tc = np.ctypeslib.as_array(T, shape=(16,)).copy()
            off = np.ctypeslib.as_array(off, shape=(3,)).copy()
            if plane:
                st = self._plane_stats(Tc, off)
            else:
                st = self.o.k_reduce_stats(self.src, self.tgt if len(self.tgt) else np.zeros((1, 4), np.float32),
                                           self.idx, Tc, offset=off)
            np.ctypeslib.as_array(out, shape=(38,))[:] = st
            self.calls["reduce"] += 1
            return 0

        def get_corr(_u, idx, d2):
            n = len(self.src)
            if n:
                np.ctypeslib.as_array(idx, shape=(n,))[:] = self.idx
                np.ctypeslib.as_array(d2, shape=(n,))[:] = self.d2
            return 0

        self._cbs = [_lib.ENG_SET(set_source), _lib.ENG_SET(set_target), _lib.ENG_SET(set_normals),
                     _lib.ENG_NN(nn_pass), _lib.ENG_REDUCE(reduce), _lib.ENG_CORR(get_corr)]
        self.table = _lib.CEngine(*self._cbs)

    def _plane_stats(self, Tc, off):
        """Point-to-plane rows in f64 from the fp32 clouds (same layout as the kernel)."""
        m = self.idx >= 0
        s = self.src[m, :3].astype(np.float64)
        p = s @ Tc.reshape(4, 4)[:3, :3].T + Tc.reshape(4, 4)[:3, 3] + off
        q = self.tgt[self.idx[m], :3].astype(np.float64) + off
        n = self.nrm[self.idx[m], :3].astype(np.float64)
        corr = np.stack([np.arange(len(p)), np.arange(len(p))], 1).astype(np.int32)
        JTJ, JTr, r2 = self.o.jtj_jtr(p, q, corr, tgt_normals=n)
        st = np.zeros(38)
        st[0] = len(p); st[1] = float(((p - q) ** 2).sum())   # Registration.cpp:65-68: the NN distance, not r2
        st[2:23] = JTJ[np.triu_indices(6)]
        st[23:29] = JTr
        return st

    def context(self):
        return _lib.Context(engine=self.table)
