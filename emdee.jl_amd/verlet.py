"""Velocity-Verlet on the device.  Build-defined: the reference has no integrator (SURVEY.md 8a
row a16); the API follows EmDee's operator style (constructor + `!` mutators spelled `_`)."""
import ctypes as C

import torch

from . import _lib
from .device import check_array, context_for, precision_of
from .nonbonded import ENERGIES, FORCES, VIRIALS

KERNELS = {"lj_force_nbr": 0, "verlet_kick_drift": 1, "rebuild": 2, "verlet_kick": 3, "lj_force_nbr_fused_step": 4,
           # decomposed steps: the fused launches over interior bricks (or all bricks, in-order form), over boundary bricks, and the
           # halo (pack -> exchange -> unpack) on the stream it runs on
           "fused_step_interior": 5, "fused_step_boundary": 6, "halo": 7}


class VelocityVerlet:
    """v += (dt/2m) f ; x += dt v ; f = F(x) ; v += (dt/2m) f, with the state kept in cell order in
    HBM between steps.  positions (n_owned + n_ghost, 3), velocities (n_owned, 3), atoms
    (n_owned + n_ghost, 2) float32 [LJAtom], inv_mass (n_owned,) or None (m = 1).

    Single-GPU, reference-shaped box: VelocityVerlet(x, v, L, model, atoms).
    Domain-decomposed: pass lo/lengths/periodic and n_ghost, and drive kick_drift_/forces_/kick_."""

    def __init__(self, positions, velocities, L, model, atoms, skin=0.3, inv_mass=None, lo=None, lengths=None,
                 periodic=None, n_ghost=0):
        n_total = positions.shape[0]
        n_owned = n_total - int(n_ghost)
        dev, dt = positions.device, positions.dtype
        self.n_owned, self.n_ghost, self.device, self.dtype = n_owned, int(n_ghost), dev, dt
        self.model = model
        self._ctx = context_for(dev)
        lo = [0.0, 0.0, 0.0] if lo is None else [float(v) for v in lo]
        lengths = [float(L)] * 3 if lengths is None else [float(v) for v in lengths]
        periodic = [1, 1, 1] if periodic is None else [int(bool(v)) for v in periodic]
        self.lo, self.lengths, self.periodic = lo, lengths, periodic
        h = C.c_void_p()
        _lib.call("emdee_md_create", self._ctx.handle, (C.c_double * 3)(*lo), (C.c_double * 3)(*lengths),
                  (C.c_int32 * 3)(*periodic), _lib.model_c(model), float(skin), precision_of(positions), C.byref(h))
        self._handle = h
        self.set_state_(positions, velocities, atoms, inv_mass, n_ghost)

    def set_state_(self, positions, velocities, atoms, inv_mass=None, n_ghost=0):
        n_total = positions.shape[0]
        n_owned = n_total - int(n_ghost)
        check_array(positions, "positions", n_total, 3, self.dtype, self.device)
        check_array(velocities, "velocities", n_owned, 3, self.dtype, self.device)
        check_array(atoms, "atoms", n_total, 2, torch.float32, self.device)
        if inv_mass is not None:
            check_array(inv_mass, "inv_mass", n_owned, None, self.dtype, self.device)
        self.n_owned, self.n_ghost = n_owned, int(n_ghost)
        self._langevin_ids = None                      # the library forgets the id array with the old state
        _lib.call("emdee_md_set_state", self._handle, n_owned, int(n_ghost), C.c_void_p(positions.data_ptr()),
                  C.c_void_p(velocities.data_ptr()), C.c_void_p(atoms.data_ptr()),
                  C.c_void_p(inv_mass.data_ptr()) if inv_mass is not None else None)

    # -- whole steps (single domain)
    def step_(self, nsteps, dt, rebuild_every=0):
        _lib.call("emdee_md_step", self._handle, int(nsteps), float(dt), int(rebuild_every))

    # -- split step (domain-decomposed driver: kick_drift_ -> halo exchange -> forces_ -> kick_)
    def kick_drift_(self, dt, kick=0.5):
        """v += kick dt f/m ; x += dt v.  kick = 1.0 fuses the previous step's closing half kick."""
        _lib.call("emdee_md_kick_drift", self._handle, float(dt), float(kick))

    def forces_(self, bitmask=FORCES, phase=0):
        """phase 0: all; 1: interior bricks (no ghost in their tile); 2: boundary bricks."""
        _lib.call("emdee_md_forces", self._handle, int(bitmask), int(phase))

    def kick_(self, dt):
        _lib.call("emdee_md_kick", self._handle, float(dt))

    def fused_step_(self, dt, kick=1.0, phase=0):
        """One inner step as one kernel (force + kick + drift, positions ping-ponged).  Returns False, having
        done nothing, when the LDS-tiled kernels are not in use for this box."""
        ok = C.c_int32(0)
        _lib.call("emdee_md_fused_step", self._handle, float(dt), float(kick), int(phase), C.byref(ok))
        return bool(ok.value)

    def needs_rebuild(self):
        f = C.c_int32()
        _lib.call("emdee_md_needs_rebuild", self._handle, C.byref(f))
        return bool(f.value)

    def rebuild_(self):
        _lib.call("emdee_md_rebuild", self._handle)

    def pack_positions(self, ids, shifts, codes=None, out=None):
        """Positions of caller-order ids (int32 GPU tensor) plus shifts[codes[k]] (flat list of 3-vectors;
        codes None: every atom uses shifts[0:3]) -> (n, 3) buffer for the halo exchange."""
        n = ids.shape[0]
        if out is None:
            out = torch.empty((n, 3), dtype=self.dtype, device=self.device)
        flat = [float(s) for s in shifts]
        _lib.call("emdee_md_pack_positions", self._handle, C.c_void_p(ids.data_ptr()),
                  C.c_void_p(codes.data_ptr()) if codes is not None else None, n,
                  (C.c_double * len(flat))(*flat), len(flat) // 3, C.c_void_p(out.data_ptr()))
        return out

    def unpack_ghosts_(self, buf, first):
        _lib.call("emdee_md_unpack_ghosts", self._handle, C.c_void_p(buf.data_ptr()), int(first), buf.shape[0])

    # -- state back in caller order
    def state(self, positions=True, velocities=True, forces=True, energies=False, virials=False):
        n, nt = self.n_owned, self.n_owned + self.n_ghost
        mk = lambda rows, cols: torch.empty((rows, cols) if cols else (rows,), dtype=self.dtype, device=self.device)
        out = dict(positions=mk(nt, 3) if positions else None, velocities=mk(n, 3) if velocities else None,
                   forces=mk(n, 3) if forces else None, energies=mk(n, 0) if energies else None,
                   virials=mk(n, 0) if virials else None)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.call("emdee_md_get_state", self._handle, p(out["positions"]), p(out["velocities"]), p(out["forces"]),
                  p(out["energies"]), p(out["virials"]))
        return out

    def totals(self):
        """(potential energy, kinetic energy, virial sum) over owned atoms; fp64 reduction (blocking)."""
        out = (C.c_double * 3)()
        _lib.call("emdee_md_energies", self._handle, out)
        return out[0], out[1], out[2]

    def observables(self, volume=None, n_atoms=None):
        """Instantaneous thermodynamic state from the on-device reductions, reduced units:
        T = 2 KE / (3 N - 3) (centre-of-mass momentum removed), P = (2 KE + sum W) / (3 V) -- the per-atom virials
        w_i already hold half of every pair's -r dE/dr, so sum W is the pair virial.  Decomposed runs pass the
        global volume / atom count and sum the totals over ranks first (DecomposedVerlet.observables)."""
        ep, ek, vir = self.totals()
        n = self.n_owned if n_atoms is None else n_atoms
        v = self.lengths[0] * self.lengths[1] * self.lengths[2] if volume is None else volume
        return dict(potential=ep, kinetic=ek, virial=vir, temperature=2.0 * ek / max(3 * n - 3, 1),
                    pressure=(2.0 * ek + vir) / (3.0 * v), density=n / v)

    def nbr_stats(self):
        b, l, m, c = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        _lib.call("emdee_md_nbr_stats", self._handle, C.byref(b), C.byref(l), C.byref(m), C.byref(c))
        return dict(builds=b.value, listed=l.value, max_count=m.value, capacity=c.value)

    def count_pairs(self):
        n = C.c_int64()
        _lib.call("emdee_md_count_pairs", self._handle, C.byref(n))
        return n.value

    def neighbor_lists(self):
        """The current list as caller ids: (counts (n_owned,), neighbors (n_owned, capacity)) int32, list order."""
        cap = max(self.nbr_stats()["capacity"], 1)
        counts = torch.zeros(self.n_owned, dtype=torch.int32, device=self.device)
        nb = torch.full((self.n_owned, cap), -1, dtype=torch.int32, device=self.device)
        _lib.call("emdee_md_nbr_list", self._handle, C.c_void_p(counts.data_ptr()), C.c_void_p(nb.data_ptr()), cap)
        return counts, nb

    def profile_(self, enable=True):
        _lib.call("emdee_md_profile", self._handle, int(bool(enable)))

    def kernel_time(self, kernel):
        """(total device ms, launches) of a kernel since profile_(True): HIP events on the stream."""
        ms, k = C.c_double(), C.c_int64()
        _lib.call("emdee_md_kernel_time", self._handle, KERNELS[kernel] if isinstance(kernel, str) else int(kernel),
                  C.byref(ms), C.byref(k))
        return ms.value, k.value

    # -- Langevin thermostat
    def set_langevin_(self, gamma, temperature, seed=0, first_step=0):
        """gamma > 0: every later step does v = c1 v + c2 sqrt(T/m) xi between its kick and its drift
        (include/emdee_hip.h: emdee_md_set_langevin); gamma <= 0 switches the thermostat off."""
        _lib.call("emdee_md_set_langevin", self._handle, float(gamma), float(temperature), int(seed) & (2 ** 64 - 1),
                  int(first_step))

    def set_langevin_ids_(self, ids):
        """int64 ids (device, caller order, owned atoms) keying the noise; None = the caller index.  Must be set
        again after every set_state_."""
        if ids is not None:
            check_array(ids, "ids", self.n_owned, None, torch.int64, self.device)
        self._langevin_ids = ids
        _lib.call("emdee_md_set_langevin_ids", self._handle, C.c_void_p(ids.data_ptr()) if ids is not None else None)

    def langevin_normals(self, seed, step, ids):
        """The thermostat's generator on its own: (len(ids), 3) float64 tensor of N(0,1) numbers."""
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty((ids.shape[0], 3), dtype=torch.float64, device=self.device)
        _lib.call("emdee_md_langevin_normals", self._handle, int(seed) & (2 ** 64 - 1), int(step),
                  C.c_void_p(ids.data_ptr()), int(ids.shape[0]), C.c_void_p(out.data_ptr()))
        return out

    # -- exclusions and 1-4 pairs (include/emdee_hip.h; set after the state is loaded, undivided boxes)
    def set_exclusions_(self, pairs):
        t = torch.as_tensor(pairs if pairs is not None else []).reshape(-1, 2).to(device=self.device, dtype=torch.int32).contiguous()
        _lib.call("emdee_md_set_exclusions", self._handle, C.c_void_p(t.data_ptr()) if t.numel() else None, int(t.shape[0]))

    def set_pairs14_(self, pairs, lj14scale):
        t = torch.as_tensor(pairs if pairs is not None else []).reshape(-1, 2).to(device=self.device, dtype=torch.int32).contiguous()
        _lib.call("emdee_md_set_pairs14", self._handle, C.c_void_p(t.data_ptr()) if t.numel() else None, int(t.shape[0]), float(lj14scale))

    def close(self):
        if self._handle is not None:
            _lib.call("emdee_md_destroy", self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
