"""FORCES/ENERGIES/VIRIALS, nonbonded_computation_tiles, compute_nonbonded!, naively_compute_nonbonded!
-- host mirror of src/nonbonded.jl.  Julia's `f!` is spelled `f_` (torch's in-place convention).

Argument order and meaning follow the reference.  Device arrays are torch GPU tensors:
positions/forces (N, 3) [= Julia 3xN column-major], energies/virials (N,), atoms (N, 2) float32
[= Vector{LJAtom}], all float32 (the reference's type) or all float64 (north-star precision).
"""
import ctypes as C

import torch

from . import _lib
from .device import check_array, context_for, precision_of

# src/nonbonded.jl:12-14
FORCES = 1 << 0
ENERGIES = 1 << 1
VIRIALS = 1 << 2

# src/nonbonded.jl:16 is WARPSIZE = 32; a CDNA4 wavefront is 64 lanes and tiles are 64 x 64
WAVESIZE = 64


class Val:
    """Val(bitmask) -- the reference selects outputs at compile time through Val{bitmask}
    (src/nonbonded.jl:45,111); here the kernels are pre-instantiated for the 7 masks."""

    def __init__(self, value):
        self.value = int(value)


def _mask(bitmask):
    return bitmask.value if isinstance(bitmask, Val) else int(bitmask)


class NeighborTiles:
    """What nonbonded_computation_tiles(N) returns: the object compute_nonbonded! iterates over.

    The reference builds the n(n+1)/2 tile pairs of the all-pairs matrix (src/nonbonded.jl:18-26),
    which is O(N^2) memory (3.9 GB at 10^6 atoms).  This handle is the O(N) replacement: cell
    binning + a full neighbour list with a skin, created lazily for the dtype/device of the first
    compute_nonbonded_ call and rebuilt inside it when an atom has moved more than skin/2.
    """

    def __init__(self, N, skin=0.3):
        if N < 0:
            raise ValueError("N must be >= 0")
        self.N, self.skin = int(N), float(skin)
        self._handle = None
        self._key = None
        self._ctx = None
        self._excl, self._p14, self._scale14 = None, None, 1.0

    def __len__(self):
        n = -(-self.N // WAVESIZE)
        return n * (n + 1) // 2

    def _get(self, ctx, precision):
        key = (id(ctx), precision)
        if self._handle is not None and self._key != key:
            self.close()
        if self._handle is None:
            h = C.c_void_p()
            _lib.call("emdee_nbr_create", ctx.handle, self.N, self.skin, precision, C.byref(h))
            self._handle, self._key, self._ctx = h, key, ctx
            self._apply_pairs()
        return self._handle

    # -- exclusions and 1-4 pairs (include/emdee_hip.h: emdee_nbr_set_exclusions / emdee_nbr_set_pairs14; the hooks of the
    # reference's force-field file, src/modelling.jl:197-200, which its own hot path never consumes)
    @staticmethod
    def _pairs(pairs):
        if pairs is None:
            return None
        t = torch.as_tensor(pairs)
        if t.numel() == 0:
            return None
        if t.dim() != 2 or t.shape[1] != 2:
            raise ValueError("pairs: an (n, 2) array of atom indices")
        return t.to(dtype=torch.int32).contiguous()

    def _apply_pairs(self):
        if self._handle is None:
            return
        dev = self._ctx.device
        for name, t, extra in (("emdee_nbr_set_exclusions", self._excl, ()), ("emdee_nbr_set_pairs14", self._p14, (float(self._scale14),))):
            if t is None:
                _lib.call(name, self._handle, None, 0, *extra)
            else:
                td = t.to(dev)
                _lib.call(name, self._handle, C.c_void_p(td.data_ptr()), int(td.shape[0]), *extra)

    def set_exclusions_(self, pairs):
        """pairs (n, 2) of atom indices that contribute nothing (bonded neighbours); None / empty clears the table."""
        self._excl = self._pairs(pairs)
        self._apply_pairs()

    def set_pairs14_(self, pairs, lj14scale):
        """pairs (n, 2) whose pair terms count lj14scale times (NonbondedTable.lj14scale of a force-field file); None clears."""
        self._p14, self._scale14 = self._pairs(pairs), float(lj14scale)
        self._apply_pairs()

    def stats(self):
        """dict(builds, listed, max_count, capacity) of the current list (blocking)."""
        if self._handle is None:
            return dict(builds=0, listed=0, max_count=0, capacity=0)
        b, l, m, c = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        _lib.call("emdee_nbr_stats", self._handle, C.byref(b), C.byref(l), C.byref(m), C.byref(c))
        return dict(builds=b.value, listed=l.value, max_count=m.value, capacity=c.value)

    def count_pairs(self):
        """Pairs with r^2 < rc^2 in the current list, each counted once (blocking)."""
        if self._handle is None:
            return 0
        n = C.c_int64()
        _lib.call("emdee_nbr_count_pairs", self._handle, C.byref(n))
        return n.value

    def neighbor_lists(self):
        """The current list as caller ids: (counts (N,) int32, neighbors (N, capacity) int32), rows in list order
        (verification accessor: the library itself stores 16-bit tile-local slots)."""
        import torch
        cap = max(self.stats()["capacity"], 1)
        dev = self._ctx.device
        counts = torch.zeros(self.N, dtype=torch.int32, device=dev)
        nb = torch.full((self.N, cap), -1, dtype=torch.int32, device=dev)
        _lib.call("emdee_nbr_list", self._handle, C.c_void_p(counts.data_ptr()), C.c_void_p(nb.data_ptr()), cap)
        return counts, nb

    def close(self):
        if self._handle is not None:
            _lib.call("emdee_nbr_destroy", self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AllPairsTiles:
    """The reference's own tile semantics: every minimum-image pair, 64 x 64 tiles, O(N^2).
    mode = LITERAL reproduces compute_tile! exactly, including full LJ beyond rc."""

    def __init__(self, N, mode=_lib.LITERAL):
        self.N, self.mode = int(N), int(mode)

    def __len__(self):
        n = -(-self.N // WAVESIZE)
        return n * (n + 1) // 2


def nonbonded_computation_tiles(N, skin=0.3, all_pairs=False, mode=_lib.LITERAL):
    """nonbonded_computation_tiles(N) -- src/nonbonded.jl:18-26."""
    return AllPairsTiles(N, mode) if all_pairs else NeighborTiles(N, skin)


def _check_outputs(forces, energies, virials, positions, atoms, mask):
    N = positions.shape[0]
    dev, dt = positions.device, positions.dtype
    check_array(positions, "positions", N, 3)
    check_array(atoms, "atoms", N, 2, torch.float32, dev)
    if mask & FORCES:
        check_array(forces, "forces", N, 3, dt, dev)
    if mask & ENERGIES:
        check_array(energies, "energies", N, None, dt, dev)
    if mask & VIRIALS:
        check_array(virials, "virials", N, None, dt, dev)
    return N


def _ptr(t, selected=True):
    return C.c_void_p(t.data_ptr()) if (selected and t is not None) else None


def compute_nonbonded_(forces, energies, virials, positions, L, tiles, model, atoms, bitmask):
    """compute_nonbonded!(forces, energies, virials, positions, L, tiles, model, atoms, Val(bitmask))
    -- src/nonbonded.jl:109-120.  Selected outputs are overwritten; work is enqueued on the current
    stream and not synchronised (as in the reference)."""
    mask = _mask(bitmask)
    if not 0 <= mask <= 7:
        raise ValueError("bitmask must combine FORCES | ENERGIES | VIRIALS")
    N = _check_outputs(forces, energies, virials, positions, atoms, mask)
    if N != tiles.N:
        raise ValueError("tiles were built for N = %d, positions hold %d atoms" % (tiles.N, N))
    ctx = context_for(positions.device)
    prec = precision_of(positions)
    args = (_ptr(forces, mask & FORCES), _ptr(energies, mask & ENERGIES), _ptr(virials, mask & VIRIALS),
            _ptr(positions), float(L))
    if isinstance(tiles, AllPairsTiles):
        _lib.call("emdee_compute_nonbonded_tiles", ctx.handle, *args, N, _lib.model_c(model), _ptr(atoms), mask,
                  tiles.mode, prec)
    else:
        _lib.call("emdee_compute_nonbonded", ctx.handle, *args, tiles._get(ctx, prec), _lib.model_c(model),
                  _ptr(atoms), mask, prec)
    return None


def naively_compute_nonbonded_(forces, energies, virials, positions, L, model, atoms, mode=_lib.LITERAL):
    """naively_compute_nonbonded!(forces, energies, virials, positions, L, model, atoms)
    -- src/nonbonded.jl:122-155: the plain all-pairs double loop.  The reference runs it on the host;
    here it is a device kernel with one thread per atom (the product has no CPU path)."""
    N = _check_outputs(forces, energies, virials, positions, atoms, FORCES | ENERGIES | VIRIALS)
    ctx = context_for(positions.device)
    _lib.call("emdee_compute_nonbonded_naive", ctx.handle, _ptr(forces), _ptr(energies), _ptr(virials), _ptr(positions),
              float(L), N, _lib.model_c(model), _ptr(atoms), int(mode), precision_of(positions))
    return None
