"""Deterministic synthetic Lennard-Jones boxes -- host side, numpy only.

The reference ships no generator (its one hot-path input is test/data/lj_sample.xyz);
these boxes are build-defined.  One generator feeds the GPU path, the CPU oracle
and the golden fixtures, so inputs are bit-identical by construction.

  * fcc lattice, n^3 cells x 4 atoms, a = (4/rho)^(1/3), L = n a, origin 0, cubic, periodic
  * jitter: every coordinate += 0.1 sigma (u - 1/2), u = uniform(seed, 3 i + d)
  * velocities: Box-Muller normals from the same counter stream at counters >= 3 N,
    centre-of-mass momentum removed, rescaled to exactly T* (3N-3 degrees of freedom), m = 1
  * binary mixture: type = mix(seed2 ^ i) & 1; A: eps=1, sigma=1; B: eps=0.5, sigma=0.88
"""
import numpy as np

SEED = 0x5EED
SEED_TYPES = 0x7A9E5
RHO = 0.8
JITTER = 0.1
FCC_BASIS = np.array([[0.0, 0.0, 0.0], [0.5, 0.5, 0.0], [0.5, 0.0, 0.5], [0.0, 0.5, 0.5]])

# fcc cells per side for the BASELINE.json configs (SURVEY.md 8 notation)
CONFIG_CELLS = {"C1": 6, "C2": 63, "C3": 136, "target": 293}

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """splitmix64 output function of the uint64 array x (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = np.asarray(x, dtype=np.uint64) + _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform(seed, counters):
    """u in [0,1): (splitmix64(seed ^ counter) >> 11) * 2^-53."""
    k = np.asarray(counters, dtype=np.uint64) ^ np.uint64(seed)
    return (splitmix64(k) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def fcc_box(n, rho=RHO):
    """(N, L, a) of an n^3-cell fcc box at reduced density rho."""
    a = (4.0 / rho) ** (1.0 / 3.0)
    return 4 * n ** 3, n * a, a


def fcc_positions(n, rho=RHO, jitter=JITTER, seed=SEED, chunk=1 << 22):
    """Jittered fcc positions, (N, 3) float64 (same memory layout as Julia's 3xN). Returns (pos, L)."""
    N, L, a = fcc_box(n, rho)
    pos = np.empty((N, 3), dtype=np.float64)
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        i = np.arange(lo, hi, dtype=np.int64)
        cell, b = i >> 2, i & 3
        cx, cy, cz = cell % n, (cell // n) % n, cell // (n * n)
        base = np.stack([cx, cy, cz], axis=1).astype(np.float64) + FCC_BASIS[b]
        ctr = (3 * i)[:, None] + np.arange(3, dtype=np.int64)[None, :]
        pos[lo:hi] = base * a + jitter * (uniform(seed, ctr) - 0.5)
    return pos, L


def velocities(N, temperature=1.0, seed=SEED):
    """Maxwell-Boltzmann velocities (m = 1), zero total momentum, exactly T*."""
    m = 3 * N
    half = (m + 1) // 2
    k = np.arange(half, dtype=np.int64)
    u1 = uniform(seed, 3 * N + 2 * k)
    u2 = uniform(seed, 3 * N + 2 * k + 1)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))           # 1-u1 in (0,1]
    g = np.empty(2 * half, dtype=np.float64)
    g[0::2] = r * np.cos(2.0 * np.pi * u2)
    g[1::2] = r * np.sin(2.0 * np.pi * u2)
    v = g[:m].reshape(N, 3).copy()
    v -= v.mean(axis=0)
    ke = 0.5 * np.sum(v * v)
    dof = max(3 * N - 3, 1)
    v *= np.sqrt(temperature * dof / (2.0 * ke))
    return v


def fcc_block(ncells, block_lo, block_n, rho=RHO, jitter=JITTER, seed=SEED):
    """Atoms of the sub-block of fcc cells [block_lo, block_lo + block_n) of a box of ncells = (nx, ny, nz)
    cells (one rank's brick of a decomposed run).  Global atom id = 4 (cx + nx (cy + ny cz)) + b, so the
    jitter stream does not depend on the decomposition; for a cubic box and the whole box as the block
    this is fcc_positions(n) up to atom order.  Returns (pos (n, 3), gid (n,) int64, lengths (3,))."""
    nx, ny, nz = (int(v) for v in ncells)
    a = (4.0 / rho) ** (1.0 / 3.0)
    bx, by, bz = (int(v) for v in block_n)
    lx, ly, lz = (int(v) for v in block_lo)
    cz, cy, cx = np.meshgrid(np.arange(lz, lz + bz), np.arange(ly, ly + by), np.arange(lx, lx + bx), indexing="ij")
    cell = (cx + nx * (cy + ny * cz)).reshape(-1).astype(np.int64)
    cxyz = np.stack([cx.reshape(-1), cy.reshape(-1), cz.reshape(-1)], axis=1).astype(np.float64)
    gid = (4 * cell[:, None] + np.arange(4, dtype=np.int64)[None, :]).reshape(-1)
    base = (cxyz[:, None, :] + FCC_BASIS[None, :, :]).reshape(-1, 3)
    ctr = (3 * gid)[:, None] + np.arange(3, dtype=np.int64)[None, :]
    pos = base * a + jitter * (uniform(seed, ctr) - 0.5)
    return pos, gid, np.array([nx * a, ny * a, nz * a])


def raw_normals(gid, N_global, seed=SEED):
    """The Box-Muller normals velocities() assigns to atoms gid of an N_global-atom box, before the
    centre-of-mass and temperature corrections (which need global sums)."""
    gid = np.asarray(gid, dtype=np.int64)
    m = (3 * gid)[:, None] + np.arange(3, dtype=np.int64)[None, :]
    k, odd = m >> 1, (m & 1).astype(bool)
    u1 = uniform(seed, 3 * N_global + 2 * k)
    u2 = uniform(seed, 3 * N_global + 2 * k + 1)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return np.where(odd, r * np.sin(2.0 * np.pi * u2), r * np.cos(2.0 * np.pi * u2))


def mixture_types(N, seed=SEED_TYPES):
    """0/1 species labels, ~50:50 (config C5). N: atom count, or an array of global atom ids."""
    ids = np.arange(N, dtype=np.uint64) if np.isscalar(N) else np.asarray(N).astype(np.uint64)
    return (splitmix64(ids ^ np.uint64(seed)) & np.uint64(1)).astype(np.int32)


def mixture_parameters(types, eps=(1.0, 0.5), sigma=(1.0, 0.88)):
    """Per-atom (eps, sigma) arrays for a binary mixture; Lorentz-Berthelot mixing is
    carried by the LJAtom encoding (src/lennard_jones.jl:13,29-30)."""
    eps = np.asarray(eps, dtype=np.float64)[types]
    sigma = np.asarray(sigma, dtype=np.float64)[types]
    return eps, sigma
