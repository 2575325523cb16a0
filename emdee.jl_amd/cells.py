"""Cells / update_cells! -- host mirror of src/cells.jl (which the reference does not even load:
src/EmDee.jl:3-7 has no include("cells.jl"); its data model and cell-id convention are kept)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .device import check_array, context_for, precision_of


class Cells:
    """Cells(r, L, cutoff; ndiv=2) -- src/cells.jl:176-194.

    M = floor(ndiv L / cutoff) cells per dimension (src/cells.jl:36); index[i] is the 1-based cell
    of atom i, 1 + vx + M vy + M^2 vz (src/cells.jl:85,181); population[c] counts atoms per cell.
    The reference's head/next linked lists are replaced by start/order arrays (counting sort);
    `r` is an (N, 3) GPU tensor [= Julia 3xN]."""

    def __init__(self, r, L, cutoff, ndiv=2):
        N = r.shape[0]
        check_array(r, "r", N, 3)
        self.N, self.L, self.cutoff, self.ndiv = N, float(L), float(cutoff), int(ndiv)
        self._ctx = context_for(r.device)
        self._device = r.device
        h = C.c_void_p()
        _lib.call("emdee_cells_create", self._ctx.handle, N, self.L, self.cutoff, self.ndiv, precision_of(r), C.byref(h))
        self._handle = h
        self._dtype = r.dtype
        m = C.c_int32()
        _lib.call("emdee_cells_M", h, C.byref(m))
        self.M = m.value
        update_cells_(self, r, L)

    def _array(self, which, n):
        ptrs = [C.c_void_p() for _ in range(4)]
        _lib.call("emdee_cells_arrays", self._handle, *[C.byref(p) for p in ptrs])
        out = torch.empty(n, dtype=torch.int32, device=self._device)
        if n:
            _lib.call("emdee_memcpy_d2d", self._ctx.handle, C.c_void_p(out.data_ptr()), ptrs[which], 4 * n)
        return out

    @property
    def index(self):
        return self._array(0, self.N)

    @property
    def population(self):
        return self._array(1, self.M ** 3)

    @property
    def start(self):
        return self._array(2, self.M ** 3 + 1)

    @property
    def order(self):
        return self._array(3, self.N)

    def close(self):
        if self._handle is not None:
            _lib.call("emdee_cells_destroy", self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def update_cells_(cells, r, L):
    """update_cells!(cells, r, L) -- src/cells.jl:196-222.  Re-bins every atom in O(N); the
    reference's incremental linked-list edit (clean_cells!/collect_baskets!/renew_cells!) has a race
     and costs O(cells x moved)."""
    check_array(r, "r", cells.N, 3, cells._dtype, cells._device)
    if float(L) != cells.L:
        raise ValueError("L changed: build new Cells")
    _lib.call("emdee_cells_update", cells._handle, C.c_void_p(r.data_ptr()))
    return None
