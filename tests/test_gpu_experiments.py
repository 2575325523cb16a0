"""The laboratory behind `make EXPERIMENTS=1` (emdee.jl_amd/csrc/Makefile -> libemdee_hip_exp.so): measured alternatives that
lost -- the transposed build, the 4-lane build, near/far rows and the far-class skip, the two-phase build with per-lane
candidate loops -- and the ablation switches that measured them.  The product library instantiates none of those kernels
and REFUSES their switches with a message (a run that believes it measures a variant must not silently measure the
default); under the experiments build each variant still produces the default build's neighbour SET, entry for entry.

The library is chosen when the package is imported (EMDEE_HIP_LIB), so the experiments build runs in a child process."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from .conftest import ROOT

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

EXP_LIB = os.path.join(ROOT, "emdee.jl_amd", "libemdee_hip_exp.so")
REMOVED = ["EMDEE_TBUILD", "EMDEE_BUILD4", "EMDEE_BUILD_ALG", "EMDEE_BUILD_NEARFAR", "EMDEE_FAR_SKIP", "EMDEE_DEBUG_RC2_SCALE",
           "EMDEE_NO_BRICK_TABLES", "EMDEE_NO_PREMUL", "EMDEE_BRICK_VARIANT", "EMDEE_STRIDE"]

SMALL = textwrap.dedent("""
    import sys
    import numpy as np, torch
    sys.path.insert(0, %r)
    import __graft_entry__ as g
    E = g.load_package()
    dev = torch.device("cuda", 0)
    x, L = E.synthetic.fcc_positions(8)
    N = x.shape[0]
    try:
        tiles = E.nonbonded_computation_tiles(N, skin=0.3)
        f = torch.zeros((N, 3), dtype=torch.float64, device=dev)
        E.compute_nonbonded_(f, None, None, E.cu(x, dev), L, tiles, E.LennardJonesModel(2.5, 2.0), E.cu(E.lennard_jones_atoms(1.0, 1.0, N), dev), E.Val(E.FORCES))
        print("RAN", float(f.abs().max()))
    except E.EmDeeError as err:
        print("REFUSED code=%%d %%s" %% (err.code, err))
""") % ROOT


@pytest.mark.parametrize("switch", REMOVED)
def test_the_product_library_refuses_experiment_switches(switch):
    env = dict(os.environ)
    env.pop("EMDEE_HIP_LIB", None)
    env[switch] = "1"
    r = subprocess.run([sys.executable, "-c", SMALL], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-600:]
    line = [l for l in r.stdout.splitlines() if l.startswith(("RAN", "REFUSED"))][-1]
    assert line.startswith("REFUSED") and switch in line and "experiment switch" in line and "EXPERIMENTS=1" in line, line


def test_experiment_variants_keep_the_neighbour_set():
    """This file's `under_the_experiments_build` cases once more in a child process that loads libemdee_hip_exp.so."""
    assert os.path.exists(EXP_LIB), "libemdee_hip_exp.so is built by __graft_entry__.build() (make EXPERIMENTS=1)"
    env = dict(os.environ, EMDEE_HIP_LIB=EXP_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_experiments.py", "-x", "-q", "-m", "gpu", "-k", "under_the_experiments_build",
                        "-p", "no:cacheprovider", "--timeout", "500"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail + r.stderr[-600:]
    assert " passed" in tail and "failed" not in tail and "skipped" not in tail, tail


def _rows(counts, nb):
    counts, nb = counts.cpu().numpy(), nb.cpu().numpy()
    return [np.sort(nb[i, :counts[i]]) for i in range(counts.shape[0])]


VARIANTS = {"transposed": {"EMDEE_TBUILD": "1"}, "four_lanes": {"EMDEE_BUILD4": "1"}, "per_lane_loops": {"EMDEE_BUILD_ALG": "2"},
            "ballot": {"EMDEE_BUILD_ALG": "1"}, "near_far": {"EMDEE_BUILD_NEARFAR": "1"},
            "far_skip": {"EMDEE_BUILD_NEARFAR": "1", "EMDEE_NEAR_DELTA": "0.12", "EMDEE_FAR_SKIP": "1"},
            "no_brick_tables": {"EMDEE_NO_BRICK_TABLES": "1"}, "brick_variant_1": {"EMDEE_BRICK_VARIANT": "1"}}


@pytest.mark.parametrize("case", ["fcc_jitter_f64", "random_gas_f64", "fcc_jitter_f32"])
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_neighbour_set_under_the_experiments_build(emdee, oracle, case, variant, monkeypatch):
    """Same neighbour SET as the default build (and, in fp64, as the oracle), same forces to rounding -- for every build the
    rounds measured and set aside (profiles/r02 .. r05): k_brick_build_t (candidates of an own cell in the lanes' registers),
    four lanes per atom, per-lane candidate loops, the ballot build, near entries first, the far class skipped while nobody
    has moved delta / 2 (a short trajectory: the skip must give the sums of the plain rows), no per-brick tables, a tuning
    brick shape."""
    E = emdee
    if not E._lib.LIB_PATH.endswith("libemdee_hip_exp.so"):
        pytest.skip("runs in the child process of test_experiment_variants_keep_the_neighbour_set")
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(23)
    rc, rs, skin = 2.5, 2.0, 0.3
    dtype = np.float32 if case.endswith("f32") else np.float64
    if case.startswith("fcc"):
        x, L = E.synthetic.fcc_positions(14)
        x = x + rng.normal(0.0, 0.1, size=x.shape)
    else:
        L, N = 9 * 2.8, 14000
        x = rng.uniform(0.0, L, size=(N, 3))
    x = x.astype(dtype)
    N = x.shape[0]
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    atoms = E.lennard_jones_atoms(1.0, 1.0, N)
    out = {}
    for name in ("default", variant):
        if name != "default":
            for k, v in VARIANTS[variant].items():
                monkeypatch.setenv(k, v)
        tiles = E.nonbonded_computation_tiles(N, skin=skin)
        f = torch.zeros((N, 3), dtype=tdt, device=dev)
        E.compute_nonbonded_(f, None, None, E.cu(x, dev), L, tiles, E.LennardJonesModel(rc, rs), E.cu(atoms, dev), E.Val(E.FORCES))
        out[name] = (_rows(*tiles.neighbor_lists()), f.cpu().numpy())
        traj = None
        if case == "fcc_jitter_f64" and variant in ("far_skip", "near_far"):
            v0 = E.synthetic.velocities(N)
            md = E.VelocityVerlet(E.cu(x, dev), E.cu(v0, dev), L, E.LennardJonesModel(rc, rs), E.cu(atoms, dev), skin=skin)
            md.step_(12, 0.005)
            traj = (md.state()["positions"].cpu().numpy(), md.totals())
        out[name] += (traj,)
    for k in VARIANTS[variant]:
        monkeypatch.delenv(k)
    got, ref = out[variant][0], out["default"][0]
    assert sum(len(r) for r in got) > 10 * N
    differing = [i for i in range(N) if not np.array_equal(got[i], ref[i])]
    # (fp32 boxes: the fp32 distance test IS the definition of the listed set, and the near/far build tests coordinates scaled
    # by k -- a pair within an ulp of r_list may fall on the other side; both sets are valid lists, nothing inside r_c differs)
    allowed = 8 if (dtype == np.float32 and variant in ("near_far", "far_skip")) else 0
    assert len(differing) <= allowed, "%d rows differ from the default build (first: %s)" % (len(differing), differing[:4])
    for i in differing:
        assert len(np.setxor1d(got[i], ref[i])) <= 2
    if dtype == np.float64:
        off, nb = oracle.neighbor_list(x, L, rc + skin)
        for i in range(N):
            assert np.array_equal(got[i], np.sort(nb[off[i]:off[i + 1]])), "row %d differs from the oracle" % i
    fa, fb = out[variant][1], out["default"][1]
    finite = np.isfinite(fb).all(axis=1)                              # (a random gas has pairs at r -> 0)
    assert np.abs(fa[finite] - fb[finite]).max() <= (1e-9 if dtype == np.float64 else 2e-3) * max(1.0, np.abs(fb[finite]).max())
    if out[variant][2] is not None:
        (xa, ea), (xb, eb) = out[variant][2], out["default"][2]
        assert np.abs(xa - xb).max() < 1e-9 and ea[0] == pytest.approx(eb[0], rel=1e-10) and ea[1] == pytest.approx(eb[1], rel=1e-10)
