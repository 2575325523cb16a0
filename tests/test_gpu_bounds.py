"""The bounds-checked device build (emdee.jl_amd/csrc: make BOUNDS=1 -> libemdee_hip_bounds.so; SURVEY.md section 5, "debug build
with bounds asserts"): every device-side access whose index comes from data goes through EMDEE_BOUND (csrc/common.hpp) --
an index outside its capacity is not used, a sticky device word records the site, and every C-ABI call from then on
returns EMDEE_ERR_OVERFLOW with the site in emdee_last_error().  Round 3's out-of-range write in the padded ghost pack
ended as SIGABRT of the host process; under this build it is a message.

The library is chosen when the package is imported (EMDEE_HIP_LIB), so everything here runs in child processes."""
import os
import subprocess
import sys
import textwrap

import pytest

from .conftest import ROOT

pytestmark = pytest.mark.gpu

BOUNDS_LIB = os.path.join(ROOT, "emdee.jl_amd", "libemdee_hip_bounds.so")


def _env(**extra):
    env = dict(os.environ)
    env["EMDEE_HIP_LIB"] = BOUNDS_LIB
    env.update(extra)
    return env


HOT_BOX = textwrap.dedent("""
    import sys
    import numpy as np, torch
    sys.path.insert(0, %r)
    import __graft_entry__ as g
    E = g.load_package()
    assert E._lib.LIB_PATH.endswith("libemdee_hip_bounds.so"), E._lib.LIB_PATH
    syn, dev = E.synthetic, torch.device("cuda", 0)
    pos, gid, lengths = syn.fcc_block((8,) * 3, (0, 0, 0), (8,) * 3)
    pos = pos[np.argsort(gid)]
    N, L = pos.shape[0], float(lengths[0])
    vel = syn.raw_normals(np.arange(N), N)
    vel -= vel.mean(axis=0)
    vel *= 1.6 * np.sqrt((3 * N - 3) / np.sum(vel * vel))          # hot: rebuilds every few steps, ghost counts grow
    atoms = E.lennard_jones_atoms(1.0, 1.0, N)
    world = 8
    try:
        dd = E.DomainDecomposition([L] * 3, E.domain.rank_grid(world), E.LennardJonesModel(2.5, 2.0), skin=0.3, device=dev)
        for r in range(world):
            mine = np.arange(r, N, world)
            dd.set_atoms_(r, E.cu(pos[mine], dev), E.cu(vel[mine], dev), E.cu(atoms[mine], dev), torch.from_numpy(mine.astype(np.int64)).to(dev))
        dd.load_()
        dd.step_(17, 0.005, 0)
        dd.step_(15, 0.005, 3)
        e = dd.totals()
        print("FINISHED redone=%%d energy=%%.12g" %% (dd.rebuild_stats()["redone"], e[0] + e[1]))
    except E.EmDeeError as err:
        print("REPORTED code=%%d %%s" %% (err.code, err))
""") % ROOT


def _run(code, env, timeout=300):
    return subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_out_of_range_index_is_reported_not_used():
    """EMDEE_DD_GHOST_CAP=exact makes ghost messages overflow (tests/test_gpu_dd.py); EMDEE_BOUNDS_INJECT=ghost_pack puts
    back what k_dd_pack_ghost_rows_padded did before commit 2d85cf1 -- the capacity-sized send list indexed by the counts.
    Under the bounds build that run ends with an error message naming the site; without the injection the same run
    finishes, with the rebuilds redone, at the energy of the product library."""
    assert os.path.exists(BOUNDS_LIB), "libemdee_hip_bounds.so is built by __graft_entry__.build() (make BOUNDS=1)"
    bad = _run(HOT_BOX, _env(EMDEE_DD_GHOST_CAP="exact", EMDEE_BOUNDS_INJECT="ghost_pack"))
    assert bad.returncode == 0, (bad.stdout[-400:], bad.stderr[-800:])          # no abort, no fault: a Python exception
    line = [l for l in bad.stdout.splitlines() if l.startswith(("REPORTED", "FINISHED"))][-1]
    assert line.startswith("REPORTED"), line
    assert "device bounds check (dd: ghost send list)" in line and "outside capacity" in line, line
    good = _run(HOT_BOX, _env(EMDEE_DD_GHOST_CAP="exact"))
    assert good.returncode == 0, (good.stdout[-400:], good.stderr[-800:])
    gline = [l for l in good.stdout.splitlines() if l.startswith(("REPORTED", "FINISHED"))][-1]
    assert gline.startswith("FINISHED") and "redone=0" not in gline, gline
    env = dict(os.environ)
    env.pop("EMDEE_HIP_LIB", None)
    prod = _run(HOT_BOX.replace('assert E._lib.LIB_PATH.endswith("libemdee_hip_bounds.so"), E._lib.LIB_PATH', "pass"),
                dict(env, EMDEE_DD_GHOST_CAP="exact"))
    pline = [l for l in prod.stdout.splitlines() if l.startswith(("REPORTED", "FINISHED"))][-1]
    assert pline == gline, (pline, gline)                                        # same states, bit for bit, with the checks in


def test_dd_and_parity_suites_under_the_bounds_build():
    """The decomposition and neighbour-set suites once more, every kernel with its bounds checks in: green, i.e. no access
    of the product path is out of range on these inputs (hot boxes, overflowing messages, empty and dense domains, two
    species, fp32).  The multi-process RCCL tests and the 10^7-atom boxes are left to the product build."""
    assert os.path.exists(BOUNDS_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_dd.py", "tests/test_gpu_parity2.py", "-x", "-q", "-m", "gpu",
                        "-k", "not rccl and not full_size and not ten_million and not bounds", "-p", "no:cacheprovider",
                        "--timeout", "500"], env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=800)
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail + r.stderr[-600:]
    assert " passed" in tail and "failed" not in tail, tail
