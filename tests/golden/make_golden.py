#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/.

The reference (Julia + CUDA.jl) cannot run in this image and stores no expected
outputs, so these vectors come from restatements that are INDEPENDENT of the C
oracle under oracle/ (this script never imports it):

  kat_interaction.json   interaction() of src/lennard_jones.jl:25-42 evaluated in exact
                         rational arithmetic (fractions.Fraction) at the given binary
                         inputs -- closed-form LJ + quintic switch, LITERAL semantics.
  lj_sample_expected.npz all-pairs fp64 numpy restatement of src/nonbonded.jl:122-155 on the
                         reference's own fixture test/data/lj_sample.xyz (copied here as data)
                         with the reference test's parameters L=10, rc=3, rs=2.5, eps=sigma=1
                         (test/runtests.jl:58); positions rounded to Float32 first as
                         CUDA.cu does (test/runtests.jl:22). LITERAL and CUTOFF modes.
  fcc864_expected.npz    the same restatement (CUTOFF) on the synthetic 864-atom fcc box
                         (BASELINE.json configs[0]) + a 100-step velocity-Verlet trajectory.
  mix500_expected.npz    two-species box (Lorentz-Berthelot through the LJAtom encoding).

Run from the repository root:  python tests/golden/make_golden.py
"""
import importlib.util
import json
import os
import sys
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

spec = importlib.util.spec_from_file_location("synthetic", os.path.join(ROOT, "emdee.jl_amd", "synthetic.py"))
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)


# ---------------------------------------------------------------- exact KATs
def exact_interaction(r2, rc2, rs2, idl2, hs_i, te_i, hs_j, te_j):
    """Exact value of the reference formula at binary inputs (all given as floats)."""
    r2, rc2, rs2, idl2 = (Fraction(float(v)) for v in (r2, rc2, rs2, idl2))
    hs_i, te_i, hs_j, te_j = (Fraction(float(np.float32(v))) for v in (hs_i, te_i, hs_j, te_j))
    sigma = hs_i + hs_j
    s2 = sigma * sigma / r2
    s6 = s2 ** 3
    e4s6 = te_i * te_j * s6
    E = e4s6 * (s6 - 1)
    W = 6 * e4s6 * (2 * s6 - 1)
    x = (r2 - rs2) * idl2

    def sign(t):
        return (t > 0) - (t < 0)

    x = x * Fraction(1, 2) * (sign(x) - sign(x - 1))
    g = 1 + x ** 3 * (15 * x - 6 * x * x - 10)
    mgr = 60 * x * x * (1 - 2 * x + x * x) * idl2 * r2
    return E * g, W * g + E * mgr


def make_kats():
    rows = []

    def add(rc, rs, ai, aj, r=None, r2=None):
        rc2, rs2 = rc * rc, rs * rs
        idl2 = 1.0 / (rc2 - rs2)
        if r2 is None:
            r2 = r * r
        hs_i, te_i = np.float32(0.5 * ai[1]), np.float32(2.0 * np.sqrt(ai[0]))
        hs_j, te_j = np.float32(0.5 * aj[1]), np.float32(2.0 * np.sqrt(aj[0]))
        E, W = exact_interaction(r2, rc2, rs2, idl2, hs_i, te_i, hs_j, te_j)
        rows.append(dict(rc=rc, rs=rs, r2=float(r2), rc2=rc2, rs2=rs2, inv_delta2=idl2,
                         half_sigma_i=float(hs_i), twice_sqrt_eps_i=float(te_i),
                         half_sigma_j=float(hs_j), twice_sqrt_eps_j=float(te_j),
                         E=float(E), W=float(W), beyond_cutoff=bool(r2 >= rc2)))

    A, B = (1.0, 1.0), (0.5, 0.88)
    for r in (0.95, 1.0, 1.5, 2.5, 2.6, 2.75, 2.9, 2.99, 3.5):
        add(3.0, 2.5, A, A, r=r)
    add(3.0, 2.5, A, A, r2=2.0 ** (1.0 / 3.0))                        # r = 2^(1/6): E = -1, W = 0
    for r in (2.0, 2.25, 2.4, 1.1, 2.499):
        add(2.5, 2.0, A, A, r=r)
    for r in (0.94, 1.0, 3.25, 3.1):
        add(3.5, 3.0, A, B, r=r)
    for r in (0.9, 1.2, 3.4):
        add(3.5, 3.0, B, B, r=r)
    with open(os.path.join(HERE, "kat_interaction.json"), "w") as fh:
        json.dump(dict(doc="exact rational evaluation of src/lennard_jones.jl:25-42 (LITERAL); "
                           "CUTOFF mode returns (0,0) where beyond_cutoff", rows=rows), fh, indent=1)
    return rows


# ------------------------------------------------- independent numpy all-pairs
def numpy_all_pairs(pos, L, rc, rs, eps, sigma, mode):
    """fp64 restatement of src/nonbonded.jl:122-155 + src/lennard_jones.jl:25-42, vectorised
    over the full NxN matrix (each pair visited from both sides, self excluded)."""
    pos = np.asarray(pos, dtype=np.float64)
    n = pos.shape[0]
    hs = np.float32(0.5 * np.asarray(sigma, dtype=np.float64)).astype(np.float64)
    te = np.float32(2.0 * np.sqrt(np.asarray(eps, dtype=np.float64))).astype(np.float64)
    rc2, rs2 = rc * rc, rs * rs
    idl2 = 1.0 / (rc2 - rs2)
    s = pos / L
    f = np.zeros((n, 3)); e = np.zeros(n); w = np.zeros(n)
    blk = 256
    for lo in range(0, n, blk):
        hi = min(n, lo + blk)
        d = s[lo:hi, None, :] - s[None, :, :]
        rv = L * (d - np.rint(d))
        r2 = np.einsum("ijk,ijk->ij", rv, rv)
        idx = np.arange(lo, hi)
        r2[idx - lo, idx] = 1.0                                   # placeholder on the diagonal
        sg = hs[lo:hi, None] + hs[None, :]
        s2 = sg * sg / r2
        s6 = s2 * s2 * s2
        e4s6 = te[lo:hi, None] * te[None, :] * s6
        E = e4s6 * (s6 - 1.0)
        W = 6.0 * e4s6 * (2.0 * s6 - 1.0)
        x = (r2 - rs2) * idl2
        x = x * 0.5 * (np.sign(x) - np.sign(x - 1.0))
        x2 = x * x
        g = 1.0 + x * x2 * (15.0 * x - 6.0 * x2 - 10.0)
        mgr = 60.0 * x2 * (1.0 - 2.0 * x + x2) * idl2 * r2
        Eg = E * g
        Wg = W * g + E * mgr
        keep = np.ones_like(r2, dtype=bool)
        keep[idx - lo, idx] = False
        if mode == "cutoff":
            keep &= r2 < rc2
        Eg = np.where(keep, Eg, 0.0)
        Wg = np.where(keep, Wg, 0.0)
        f[lo:hi] = np.einsum("ij,ijk->ik", Wg / r2, rv)
        e[lo:hi] = 0.5 * Eg.sum(axis=1)
        w[lo:hi] = 0.5 * Wg.sum(axis=1)
    return f, e, w


def read_xyz(path):
    with open(path) as fh:
        n = int(fh.readline())
        fh.readline()
        return np.array([[float(t) for t in fh.readline().split()[1:4]] for _ in range(n)])


def make_lj_sample():
    xyz = read_xyz(os.path.join(HERE, "lj_sample.xyz"))
    pos = xyz.astype(np.float32).astype(np.float64)               # CUDA.cu(xyz_data) => Float32
    out = {}
    for mode in ("literal", "cutoff"):
        f, e, w = numpy_all_pairs(pos, 10.0, 3.0, 2.5, np.ones(800), np.ones(800), mode)
        out["forces_" + mode], out["energies_" + mode], out["virials_" + mode] = f, e, w
        print("lj_sample", mode, "sumE", repr(e.sum()), "sumW", repr(w.sum()), "max|F|", np.abs(f).max())
    np.savez_compressed(os.path.join(HERE, "lj_sample_expected.npz"), **out)


def numpy_verlet(x, v, L, rc, rs, eps, sigma, dt, nsteps):
    x, v = x.copy(), v.copy()
    f, e, w = numpy_all_pairs(x, L, rc, rs, eps, sigma, "cutoff")
    ep, ek, vir = [e.sum()], [0.5 * np.sum(v * v)], [w.sum()]
    for _ in range(nsteps):
        v += 0.5 * dt * f
        x += dt * v
        f, e, w = numpy_all_pairs(x, L, rc, rs, eps, sigma, "cutoff")
        v += 0.5 * dt * f
        ep.append(e.sum()); ek.append(0.5 * np.sum(v * v)); vir.append(w.sum())
    return x, v, f, np.array(ep), np.array(ek), np.array(vir)


def make_fcc864():
    pos, L = synthetic.fcc_positions(6)
    vel = synthetic.velocities(pos.shape[0])
    ones = np.ones(pos.shape[0])
    f, e, w = numpy_all_pairs(pos, L, 2.5, 2.0, ones, ones, "cutoff")
    x1, v1, f1, ep, ek, vir = numpy_verlet(pos, vel, L, 2.5, 2.0, ones, ones, 0.005, 100)
    print("fcc864 sumE", repr(e.sum()), "drift", (ep[-1] + ek[-1]) / (ep[0] + ek[0]) - 1.0)
    np.savez_compressed(os.path.join(HERE, "fcc864_expected.npz"), L=L, pos_checksum=pos.sum(), vel_checksum=np.abs(vel).sum(),
                        forces=f, energies=e, virials=w, x100=x1, v100=v1, f100=f1, epot=ep, ekin=ek, virial=vir)


def make_mix500():
    pos, L = synthetic.fcc_positions(5)                           # 500 atoms, L = 8.55 > 2*3.5
    types = synthetic.mixture_types(pos.shape[0])
    eps, sigma = synthetic.mixture_parameters(types)
    f, e, w = numpy_all_pairs(pos, L, 3.5, 3.0, eps, sigma, "cutoff")
    print("mix500 sumE", repr(e.sum()), "nB", int(types.sum()))
    np.savez_compressed(os.path.join(HERE, "mix500_expected.npz"), L=L, types=types, forces=f, energies=e, virials=w)


M64 = (1 << 64) - 1


def py_langevin_normals(seed, step, ident):
    """Independent restatement (Python integers + math) of the thermostat's counter-based normals."""
    import math

    def mix(z):
        z &= M64
        z ^= z >> 30; z = (z * 0xBF58476D1CE4E5B9) & M64
        z ^= z >> 27; z = (z * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    base = mix(seed + 0x9E3779B97F4A7C15 * ident)
    s2 = mix(base ^ ((0xD1B54A32D192ED03 * (step + 1)) & M64))
    r = [mix(s2 + 0x9E3779B97F4A7C15 * (j + 1)) for j in range(4)]
    u1, v2 = ((r[0] >> 11) + 1) / 2.0 ** 53, (r[1] >> 11) / 2.0 ** 53
    u3, v4 = ((r[2] >> 11) + 1) / 2.0 ** 53, (r[3] >> 11) / 2.0 ** 53
    a, b = math.sqrt(-2.0 * math.log(u1)), math.sqrt(-2.0 * math.log(u3))
    return [a * math.cos(2.0 * math.pi * v2), a * math.sin(2.0 * math.pi * v2), b * math.cos(2.0 * math.pi * v4)]


def numpy_verlet_langevin(x, v, L, rc, rs, eps, sigma, dt, nsteps, gamma, temperature, seed):
    import math
    x, v = x.copy(), v.copy()
    c1 = math.exp(-gamma * dt)
    c2 = math.sqrt(1.0 - c1 * c1)
    f, e, w = numpy_all_pairs(x, L, rc, rs, eps, sigma, "cutoff")
    for s in range(nsteps):
        xi = np.array([py_langevin_normals(seed, s, i) for i in range(x.shape[0])])
        v += 0.5 * dt * f
        v = c1 * v + c2 * math.sqrt(temperature) * xi
        x += dt * v
        f, e, w = numpy_all_pairs(x, L, rc, rs, eps, sigma, "cutoff")
        v += 0.5 * dt * f
    return x, v, f


def make_langevin():
    cases = [(0, 0, 0), (1234, 0, 0), (1234, 1, 0), (1234, 0, 1), (0x5EED, 99, 863), (2 ** 63 + 5, 10 ** 9, 10 ** 8)]
    kat = [dict(seed=a, step=b, id=c, normals=py_langevin_normals(a, b, c)) for a, b, c in cases]
    pos, L = synthetic.fcc_positions(6)
    vel = synthetic.velocities(pos.shape[0])
    ones = np.ones(pos.shape[0])
    x1, v1, f1 = numpy_verlet_langevin(pos, vel, L, 2.5, 2.0, ones, ones, 0.005, 10, 2.0, 0.7, 0x5EED)
    with open(os.path.join(HERE, "kat_langevin.json"), "w") as fh:
        json.dump(dict(doc="counter-based normals of the Langevin thermostat (python ints + math) and 10 thermostatted "
                           "steps of the 864-atom fcc box (gamma 2, T 0.7, seed 0x5EED, dt 0.005): checksums",
                       normals=kat, x10_sum=float(np.sum(x1)), v10_abs_sum=float(np.abs(v1).sum()),
                       f10_abs_sum=float(np.abs(f1).sum()), v10_first=v1[0].tolist()), fh, indent=1)
    print("langevin kat", kat[1]["normals"], "ekin after 10 steps", 0.5 * float(np.sum(v1 * v1)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "langevin":
        make_langevin()
        sys.exit(0)
    make_kats()
    make_lj_sample()
    make_fcc864()
    make_mix500()
    make_langevin()
