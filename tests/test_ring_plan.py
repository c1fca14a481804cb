"""Host logic of the ring search (visma_amd/csrc/grid_ring.hip, grid.hip: grid_plan_ring / ring_visiting_order) through the
two context-free entry points of include/visma_icp_testing.h -- no device needed: the cell table the library plans for a
radius that is large against the point spacing, and the order in which a query visits the rows around its own."""
import ctypes as C

import numpy as np
import pytest

from visma_amd import _lib


@pytest.fixture(scope="module")
def L():
    lib = _lib.load()
    lib.visma_icp_plan_ring_grid.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_double, C.c_double,
                                             C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.visma_icp_ring_visiting_order.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_short), C.POINTER(C.c_short),
                                                  C.POINTER(C.c_float), C.POINTER(C.c_int)]
    return lib


def plan(L, mn, mx, r, cell):
    a = (C.c_float * 3)(*mn)
    b = (C.c_float * 3)(*mx)
    dims = (C.c_int * 3)()
    h = C.c_double(0.0)
    rings = C.c_int(-1)
    assert L.visma_icp_plan_ring_grid(a, b, r, cell, dims, C.byref(h), C.byref(rings)) == 0
    return list(dims), h.value, rings.value


def order(L, rings):
    n = C.c_int(0)
    cap = (2 * rings + 1) ** 2 if 1 <= rings <= 64 else 8
    dy = np.zeros(cap, np.int16)
    dz = np.zeros(cap, np.int16)
    base = np.zeros(cap, np.float32)
    rc = L.visma_icp_ring_visiting_order(rings, cap, dy.ctypes.data_as(C.POINTER(C.c_short)), dz.ctypes.data_as(C.POINTER(C.c_short)),
                                         base.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n))
    return rc, n.value, dy, dz, base


@pytest.mark.parametrize("rings", [1, 2, 5, 22, 64])
def test_the_visiting_order_holds_every_row_once_nearest_first(L, rings):
    rc, n, dy, dz, base = order(L, rings)
    side = 2 * rings + 1
    assert rc == 0 and n == side * side
    assert len({(int(a), int(b)) for a, b in zip(dy, dz)}) == n                       # every offset of the square, once
    assert np.abs(dy).max() == rings and np.abs(dz).max() == rings
    want = np.maximum(np.abs(dy.astype(int)) - 1, 0) ** 2 + np.maximum(np.abs(dz.astype(int)) - 1, 0) ** 2
    assert np.array_equal(base, want.astype(np.float32))
    assert np.all(np.diff(base) >= 0)                                                 # what the walk's stop test relies on
    assert (dy[0], dz[0]) == (0, 0) and base[8] == 0 and (rings == 1 or base[9] > 0)  # the query's row, then its eight neighbours
    # the lower bound is a lower bound: any point of row (dy, dz) against any point of the query's row of cells
    assert np.all(want <= dy.astype(int) ** 2 + dz.astype(int) ** 2)


def test_rings_outside_the_table_are_refused(L):
    assert order(L, 0)[0] != 0 and order(L, 65)[0] != 0 and order(L, -3)[0] != 0


def test_the_plan_decouples_the_cell_from_the_radius(L):
    mn, mx = (-1.5, -1.5, -1.5), (1.5, 1.5, 1.5)
    # the literal workload: r = 0.15, cells of 7.5 mm wished
    dims, h, rings = plan(L, mn, mx, 0.15, 0.0075)
    assert abs(h - 0.0075) < 1e-6 and rings == int(np.ceil(0.15 * 1.001 / h)) + 1 == 22
    assert dims == [int(np.floor(3.0 / h)) + 1] * 3 and np.prod(dims) <= 64 * 1024 * 1024
    # a wished edge that is not smaller than the radius: the ordinary plan (cells >= 1.001 r, no rings)
    dims, h, rings = plan(L, mn, mx, 0.15, 0.2)
    assert rings == 0 and h >= 0.15 * 1.001 * (1 - 1e-6)
    dims, h, rings = plan(L, mn, mx, 0.15, float("nan"))
    assert rings == 0
    # a radius of thousands of wished cells: the edge grows until the radius spans at most 64 rings
    dims, h, rings = plan(L, mn, mx, 1.0, 1e-4)
    assert 2 <= rings <= 64 and h >= 1.0 * 1.001 / 62 * (1 - 1e-6)
    # a table that would not fit (more than 2048 cells per axis / 64 M cells): the edge grows by 1.26 until it does
    dims, h, rings = plan(L, (0, 0, 0), (100.0, 100.0, 100.0), 2.0, 0.04)
    assert max(dims) <= 2048 and np.prod([float(d) for d in dims]) <= 64 * 1024 * 1024 and h > 0.04 and 1 <= rings <= 64
    # a flat target (a plane: no extent along z) and an empty box
    dims, h, rings = plan(L, (0, 0, 0), (1.0, 1.0, 0.0), 0.1, 0.01)
    assert dims[2] == 1 and dims[0] == dims[1] == 101 and rings == 12
    dims, h, rings = plan(L, (0, 0, 0), (0.0, 0.0, 0.0), 0.1, 0.01)
    assert dims == [1, 1, 1] and rings == 12
