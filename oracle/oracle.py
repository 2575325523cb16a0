"""ctypes loader for the CPU oracle (oracle/emdee_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the product package.  PARITY
UNPINNED in the golden-vector sense (see emdee_oracle.h).

Array conventions mirror the reference: positions/forces are 3xN column-major
in Julia == (N, 3) C-contiguous numpy arrays here (xyz interleaved).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libemdee_oracle.so")

FORCES, ENERGIES, VIRIALS = 1, 2, 4
LITERAL, CUTOFF = 0, 1


class Model32(C.Structure):
    _fields_ = [("rc2", C.c_float), ("rs2", C.c_float), ("inv_delta2", C.c_float)]


class Model64(C.Structure):
    _fields_ = [("rc2", C.c_double), ("rs2", C.c_double), ("inv_delta2", C.c_double)]


class Atom(C.Structure):
    _fields_ = [("half_sigma", C.c_float), ("twice_sqrt_eps", C.c_float)]


ATOM_DTYPE = np.dtype([("half_sigma", np.float32), ("twice_sqrt_eps", np.float32)])


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ("emdee_oracle.c", "oracle_impl.inc", "emdee_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libemdee_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        # (EMDEE_ORACLE_LIB: another build of the same sources -- `make -C oracle asan`, tests/test_oracle.py::test_oracle_under_asan)
        path = os.environ.get("EMDEE_ORACLE_LIB")
        if not path:
            build()
            path = _LIB
        L = C.CDLL(path)
        p = C.c_void_p
        L.orc_model_f32.argtypes = [C.c_double, C.c_double, C.POINTER(Model32)]
        L.orc_model_f64.argtypes = [C.c_double, C.c_double, C.POINTER(Model64)]
        L.orc_lj_atom.argtypes = [C.c_double, C.c_double, C.POINTER(Atom)]
        L.orc_interaction_f32.argtypes = [C.c_float, C.POINTER(Model32), Atom, Atom, C.c_int,
                                          C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_interaction_f64.argtypes = [C.c_double, C.POINTER(Model64), Atom, Atom, C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_naive_f32.argtypes = [C.c_int32, p, C.c_float, C.POINTER(Model32), p, C.c_int, p, p, p]
        L.orc_naive_f64.argtypes = [C.c_int32, p, C.c_double, C.POINTER(Model64), p, C.c_int, p, p, p]
        L.orc_cells_per_dimension.argtypes = [C.c_double, C.c_double, C.c_int32]
        L.orc_cells_per_dimension.restype = C.c_int32
        L.orc_cells_f64.argtypes = [C.c_int32, p, C.c_double, C.c_double, C.c_int32, p, p]
        L.orc_cells_f64.restype = C.c_int32
        L.orc_cells_f32.argtypes = [C.c_int32, p, C.c_double, C.c_double, C.c_int32, p, p]
        L.orc_cells_f32.restype = C.c_int32
        L.orc_neighbor_list_f64.argtypes = [C.c_int32, p, C.c_double, C.c_double, p, p]
        L.orc_neighbor_list_f64.restype = C.c_int64
        L.orc_nonbonded_cells_f64.argtypes = [C.c_int32, p, C.c_double, C.POINTER(Model64), p, C.c_int, p, p, p]
        L.orc_nonbonded_cells_f32.argtypes = [C.c_int32, p, C.c_float, C.POINTER(Model32), p, C.c_int, p, p, p]
        L.orc_verlet_f64.argtypes = [C.c_int32, p, p, C.c_double, C.POINTER(Model64), p, p, C.c_double,
                                     C.c_int32, C.c_int, C.c_int, p, p, p, p]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def model(cutoff, switch, dtype=np.float64):
    """LennardJonesModel(cutoff, switch) -- src/lennard_jones.jl:6-11."""
    if np.dtype(dtype) == np.float32:
        m = Model32()
        lib().orc_model_f32(float(cutoff), float(switch), C.byref(m))
    else:
        m = Model64()
        lib().orc_model_f64(float(cutoff), float(switch), C.byref(m))
    return m


def lj_atoms(eps, sigma, n=None):
    """Array of LennardJonesAtom(eps, sigma) -- src/lennard_jones.jl:13-18."""
    eps = np.atleast_1d(np.asarray(eps, dtype=np.float64))
    sigma = np.atleast_1d(np.asarray(sigma, dtype=np.float64))
    if n is not None:
        eps = np.broadcast_to(eps, (n,))
        sigma = np.broadcast_to(sigma, (n,))
    out = np.empty(eps.shape[0], dtype=ATOM_DTYPE)
    # same IEEE operations as orc_lj_atom (sqrt and the casts are correctly rounded)
    out["half_sigma"] = (0.5 * sigma).astype(np.float32)
    out["twice_sqrt_eps"] = (2.0 * np.sqrt(eps)).astype(np.float32)
    return out


def interaction(r2, mdl, ai, aj, mode=LITERAL):
    """interaction(r2, model, atom_i, atom_j) -> (E, W) -- src/lennard_jones.jl:25-42."""
    A, B = Atom(*[float(x) for x in ai]), Atom(*[float(x) for x in aj])
    if isinstance(mdl, Model32):
        e, w = C.c_float(), C.c_float()
        lib().orc_interaction_f32(float(np.float32(r2)), C.byref(mdl), A, B, mode, C.byref(e), C.byref(w))
    else:
        e, w = C.c_double(), C.c_double()
        lib().orc_interaction_f64(float(r2), C.byref(mdl), A, B, mode, C.byref(e), C.byref(w))
    return e.value, w.value


def naive(pos, L, mdl, atoms, mode=LITERAL):
    """naively_compute_nonbonded! -- src/nonbonded.jl:122-155. Returns (forces, energies, virials)."""
    f32 = isinstance(mdl, Model32)
    dt = np.float32 if f32 else np.float64
    pos = np.ascontiguousarray(pos, dtype=dt)
    atoms = np.ascontiguousarray(atoms, dtype=ATOM_DTYPE)
    n = pos.shape[0]
    f, e, w = np.empty((n, 3), dt), np.empty(n, dt), np.empty(n, dt)
    fn = lib().orc_naive_f32 if f32 else lib().orc_naive_f64
    fn(n, _ptr(pos), float(L), C.byref(mdl), _ptr(atoms), mode, _ptr(f), _ptr(e), _ptr(w))
    return f, e, w


def nonbonded_cells(pos, L, mdl, atoms, nthreads=0):
    """O(N) CUTOFF-mode evaluation (cell list, OpenMP). Returns (forces, energies, virials)."""
    f32 = isinstance(mdl, Model32)
    dt = np.float32 if f32 else np.float64
    pos = np.ascontiguousarray(pos, dtype=dt)
    atoms = np.ascontiguousarray(atoms, dtype=ATOM_DTYPE)
    n = pos.shape[0]
    f, e, w = np.empty((n, 3), dt), np.empty(n, dt), np.empty(n, dt)
    fn = lib().orc_nonbonded_cells_f32 if f32 else lib().orc_nonbonded_cells_f64
    fn(n, _ptr(pos), float(L), C.byref(mdl), _ptr(atoms), int(nthreads), _ptr(f), _ptr(e), _ptr(w))
    return f, e, w


def cells(pos, L, cutoff, ndiv=2):
    """Cells(r, L, cutoff; ndiv) index/population -- src/cells.jl:36,176-181. Returns (M, index, population)."""
    f32 = np.asarray(pos).dtype == np.float32
    dt = np.float32 if f32 else np.float64
    pos = np.ascontiguousarray(pos, dtype=dt)
    n = pos.shape[0]
    M = max(1, lib().orc_cells_per_dimension(float(L), float(cutoff), int(ndiv)))
    index = np.empty(n, np.int32)
    pop = np.empty(M ** 3, np.int32)
    fn = lib().orc_cells_f32 if f32 else lib().orc_cells_f64
    fn(n, _ptr(pos), float(L), float(cutoff), int(ndiv), _ptr(index), _ptr(pop))
    return M, index, pop


def neighbor_list(pos, L, rlist):
    """Full neighbour list (CSR) of minimum-image pairs with r < rlist. Returns (offsets[N+1], nbrs)."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n = pos.shape[0]
    off = np.empty(n + 1, np.int64)
    total = lib().orc_neighbor_list_f64(n, _ptr(pos), float(L), float(rlist), _ptr(off), None)
    if total < 0:
        raise ValueError("rlist must be <= L/2")
    nb = np.empty(max(total, 1), np.int32)
    lib().orc_neighbor_list_f64(n, _ptr(pos), float(L), float(rlist), _ptr(off), _ptr(nb))
    return off, nb[:total]


def verlet(x, v, L, mdl, atoms, dt, nsteps, inv_mass=None, use_cells=True, nthreads=0):
    """Build-defined velocity-Verlet. Returns dict with x, v, f, epot, ekin, virial."""
    x = np.array(x, dtype=np.float64, order="C")
    v = np.array(v, dtype=np.float64, order="C")
    atoms = np.ascontiguousarray(atoms, dtype=ATOM_DTYPE)
    n = x.shape[0]
    im = None if inv_mass is None else np.ascontiguousarray(inv_mass, dtype=np.float64)
    ep, ek, vir = (np.empty(nsteps + 1) for _ in range(3))
    f = np.empty((n, 3))
    lib().orc_verlet_f64(n, _ptr(x), _ptr(v), float(L), C.byref(mdl), _ptr(atoms), _ptr(im), float(dt),
                         int(nsteps), int(bool(use_cells)), int(nthreads), _ptr(ep), _ptr(ek), _ptr(vir), _ptr(f))
    return dict(x=x, v=v, f=f, epot=ep, ekin=ek, virial=vir)


def langevin_normals(seed, step, ident):
    """The three N(0,1) numbers of (seed, step, atom id)."""
    out = (C.c_double * 3)()
    f = lib().orc_langevin_normals
    f.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_double)]
    f.restype = None
    f(int(seed), int(step), int(ident), out)
    return np.array(out[:])


def verlet_langevin(x, v, L, mdl, atoms, dt, nsteps, gamma, temperature, seed, step0=0, ids=None, inv_mass=None,
                    use_cells=True, nthreads=0):
    """velocity-Verlet with the Langevin O step (see emdee_oracle.h). Returns dict with x, v, f, epot, ekin, virial."""
    x = np.array(x, dtype=np.float64, order="C")
    v = np.array(v, dtype=np.float64, order="C")
    atoms = np.ascontiguousarray(atoms, dtype=ATOM_DTYPE)
    n = x.shape[0]
    im = None if inv_mass is None else np.ascontiguousarray(inv_mass, dtype=np.float64)
    idv = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
    ep, ek, vir = (np.empty(nsteps + 1) for _ in range(3))
    f = np.empty((n, 3))
    fn = lib().orc_verlet_langevin_f64
    p = C.c_void_p
    fn.argtypes = [C.c_int32, p, p, C.c_double, C.POINTER(Model64), p, p, C.c_double, C.c_int32, C.c_int, C.c_int,
                   C.c_double, C.c_double, C.c_uint64, C.c_uint64, p, p, p, p, p]
    fn.restype = None
    fn(n, _ptr(x), _ptr(v), float(L), C.byref(mdl), _ptr(atoms), _ptr(im), float(dt), int(nsteps),
       int(bool(use_cells)), int(nthreads), float(gamma), float(temperature), int(seed), int(step0), _ptr(idv),
       _ptr(ep), _ptr(ek), _ptr(vir), _ptr(f))
    return dict(x=x, v=v, f=f, epot=ep, ekin=ek, virial=vir)


def max_threads():
    return lib().orc_max_threads()
