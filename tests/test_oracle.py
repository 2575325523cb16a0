"""Pins the CPU oracle (oracle/) before anything trusts it.

The reference stores no golden outputs (test/runtests.jl holds one differential
test), so the oracle is pinned by restatements that are independent of it:
exact rational known answers, a numpy fp64 restatement on the reference's own
fixture, and the values recorded at survey time.
"""
import json
import os

import numpy as np
import pytest

from .conftest import GOLDEN


def _kat_rows():
    with open(os.path.join(GOLDEN, "kat_interaction.json")) as fh:
        return json.load(fh)["rows"]


def _model64(orc, row):
    m = orc.Model64()
    m.rc2, m.rs2, m.inv_delta2 = row["rc2"], row["rs2"], row["inv_delta2"]
    return m


def test_interaction_exact_rational_kats_f64(oracle):
    """interaction() vs closed-form values in exact rational arithmetic (src/lennard_jones.jl:25-42)."""
    for row in _kat_rows():
        m = _model64(oracle, row)
        ai = (row["half_sigma_i"], row["twice_sqrt_eps_i"])
        aj = (row["half_sigma_j"], row["twice_sqrt_eps_j"])
        E, W = oracle.interaction(row["r2"], m, ai, aj, oracle.LITERAL)
        scale = max(abs(row["E"]), abs(row["W"]), 1e-300)
        assert abs(E - row["E"]) <= 1e-12 * scale + 1e-15, row
        assert abs(W - row["W"]) <= 1e-12 * scale + 1e-13, row
        Ec, Wc = oracle.interaction(row["r2"], m, ai, aj, oracle.CUTOFF)
        if row["beyond_cutoff"]:
            assert (Ec, Wc) == (0.0, 0.0)          # CUTOFF drops r2 >= rc2 (Q1/Q2)
        else:
            assert (Ec, Wc) == (E, W)              # identical inside the cutoff


def test_interaction_survey_values(oracle):
    """The survey-time table of SURVEY.md 8(c) (a second, earlier restatement)."""
    m = oracle.model(3.0, 2.5)
    a = (0.5, 2.0)
    table = {0.95: (1.960974656887615e+00, 5.618067529064674e+01), 1.0: (0.0, 24.0),
             1.5: (-3.203365942785746e-01, -1.737043246569233e+00),
             2.5: (-1.631689113600000e-02, -9.749869363200002e-02),
             2.6: (-1.229538236539820e-02, -1.169719886315101e-01),
             2.75: (-5.006131934194279e-03, -1.247273985325205e-01),
             2.9: (-4.679297634217142e-04, -3.778420084297349e-02),
             2.99: (-5.589954148723520e-07, -4.983949716412851e-04)}
    for r, (E0, W0) in table.items():
        E, W = oracle.interaction(r * r, m, a, a, oracle.LITERAL)
        assert E == pytest.approx(E0, rel=1e-10, abs=1e-15)
        assert W == pytest.approx(W0, rel=1e-10, abs=1e-12)
    E, W = oracle.interaction(2.0 ** (1.0 / 3.0), m, a, a)         # r = 2^(1/6): minimum of LJ
    assert E == pytest.approx(-1.0, rel=1e-14) and abs(W) < 1e-13
    E, W = oracle.interaction(3.5 * 3.5, m, a, a, oracle.LITERAL)  # Q1: full LJ beyond rc
    assert E == pytest.approx(-2.1747802e-03, rel=1e-6) and W == pytest.approx(-1.3041579e-02, rel=1e-6)
    m2 = oracle.model(2.5, 2.0)
    for r, (E0, W0) in {2.0: (-6.152343750000000e-02, -3.632812500000000e-01),
                        2.25: (-1.688593301999643e-02, -3.570558867551361e-01),
                        2.4: (-1.509251577777293e-03, -1.018174201003912e-01)}.items():
        E, W = oracle.interaction(r * r, m2, a, a)
        assert E == pytest.approx(E0, rel=1e-10) and W == pytest.approx(W0, rel=1e-10)


def test_interaction_is_a_potential_and_its_virial(oracle):
    """A property of the reference's pair function that no restatement of its lines can satisfy by accident: the second
    value returned by interaction() (src/lennard_jones.jl:25-42: W g + E (-r g')) is the virial -r dE/dr of the first
    (E g, the switched Lennard-Jones energy), on both sides of rs and right up to rc, for like and unlike atoms; and
    the naive double loop's force on an atom is minus the gradient of the total energy it reports
    (src/nonbonded.jl:136-145).  Checked by Richardson-extrapolated central differences of the oracle's own energies."""
    def dE_dr(m, ai, aj, r, h=1e-4):
        E = lambda q: oracle.interaction(q * q, m, ai, aj, oracle.LITERAL)[0]
        d1 = (E(r + h) - E(r - h)) / (2 * h)
        d2 = (E(r + 2 * h) - E(r - 2 * h)) / (4 * h)
        return (4 * d1 - d2) / 3                     # O(h^4)
    for rc, rs in ((3.0, 2.5), (2.5, 2.0), (3.5, 3.0)):
        m = oracle.model(rc, rs)
        for ai, aj in (((0.5, 2.0), (0.5, 2.0)), ((0.45, 2.2), (0.6, 1.7)), ((0.5, 2.0), (0.55, 0.0))):
            for r in np.concatenate([np.linspace(0.92, rs - 0.01, 7), np.linspace(rs + 0.01, rc - 0.01, 9)]):
                E, W = oracle.interaction(r * r, m, ai, aj, oracle.LITERAL)
                want = -r * dE_dr(m, ai, aj, r)
                assert W == pytest.approx(want, rel=2e-7, abs=2e-9), (rc, rs, ai, aj, r)
    # forces of the double loop = -grad of its total energy (a few atoms of a small random box, one coordinate each)
    rng = np.random.default_rng(3)
    N, L = 40, 8.0
    x = rng.uniform(0, L, size=(N, 3))
    eps, sig = rng.uniform(0.6, 1.4, N), rng.uniform(0.85, 1.1, N)
    atoms = oracle.lj_atoms(eps, sig)
    m = oracle.model(3.0, 2.5)
    f, e, w = oracle.naive(x, L, m, atoms, oracle.CUTOFF)
    tot = lambda y: oracle.naive(y, L, m, atoms, oracle.CUTOFF)[1].sum()
    h = 1e-5
    for i, d in ((0, 0), (7, 1), (23, 2), (39, 0)):
        xp, xm = x.copy(), x.copy()
        xp[i, d] += h; xm[i, d] -= h
        assert f[i, d] == pytest.approx(-(tot(xp) - tot(xm)) / (2 * h), rel=1e-6, abs=1e-7)
    # and the per-atom virials add up to sum over pairs of r . f = -sum_i x_i . f_i only up to the images: check the pair form
    assert w.sum() == pytest.approx(sum(0.5 * oracle.interaction(float(np.sum((dx - L * np.rint(dx / L)) ** 2)), m,
                                        (atoms["half_sigma"][i], atoms["twice_sqrt_eps"][i]), (atoms["half_sigma"][j], atoms["twice_sqrt_eps"][j]),
                                        oracle.CUTOFF)[1] for i in range(N) for j in range(N) if i != j
                                        for dx in [x[i] - x[j]]), rel=1e-12)


def test_interaction_f32_follows_reference_precision(oracle):
    """Float32 instantiation (the reference's precision) agrees with the exact values to fp32 rounding."""
    for row in _kat_rows():
        m = oracle.Model32()
        m.rc2, m.rs2, m.inv_delta2 = row["rc2"], row["rs2"], row["inv_delta2"]
        ai = (row["half_sigma_i"], row["twice_sqrt_eps_i"])
        aj = (row["half_sigma_j"], row["twice_sqrt_eps_j"])
        E, W = oracle.interaction(row["r2"], m, ai, aj, oracle.LITERAL)
        scale = max(abs(row["E"]), abs(row["W"]))
        # the switch window amplifies the Float32 rounding of x by ~30 g'(x): 2e-4 of the larger term
        assert abs(E - row["E"]) <= 2e-4 * scale + 1e-7
        assert abs(W - row["W"]) <= 2e-4 * scale + 1e-6


def test_clamp_quirks_q1_q2(oracle):
    """x<0 -> 0, x>1 -> 0 (g = 1 beyond rc), x == 1 -> 0.5, x == 0 -> 0 (src/lennard_jones.jl:37)."""
    m = oracle.model(3.0, 2.5)
    a = (0.5, 2.0)
    lj = lambda r2: (4 * (r2 ** -6 - r2 ** -3), 24 * (2 * r2 ** -6 - r2 ** -3))
    E, W = oracle.interaction(4.0, m, a, a)                        # r = 2 < rs: pure LJ
    assert (E, W) == pytest.approx(lj(4.0), rel=1e-14)
    E, W = oracle.interaction(16.0, m, a, a, oracle.LITERAL)       # r = 4 > rc: pure LJ again (Q1)
    assert (E, W) == pytest.approx(lj(16.0), rel=1e-14)
    E, W = oracle.interaction(9.0, m, a, a, oracle.LITERAL)        # r2 == rc2: x = 0.5 -> g = 0.5 (Q2)
    assert E == pytest.approx(0.5 * lj(9.0)[0], rel=1e-12)
    assert oracle.interaction(9.0, m, a, a, oracle.CUTOFF) == (0.0, 0.0)   # strict r2 < rc2


@pytest.mark.parametrize("mode", ["literal", "cutoff"])
def test_naive_f64_on_reference_fixture(oracle, lj_sample, golden, mode):
    """All-pairs oracle vs the independent numpy restatement on test/data/lj_sample.xyz."""
    g = golden["lj_sample_expected"]
    m = oracle.model(3.0, 2.5)
    atoms = oracle.lj_atoms(1.0, 1.0, 800)
    f, e, w = oracle.naive(lj_sample.astype(np.float64), 10.0, m, atoms,
                           oracle.LITERAL if mode == "literal" else oracle.CUTOFF)
    np.testing.assert_allclose(f, g["forces_" + mode], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(e, g["energies_" + mode], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(w, g["virials_" + mode], rtol=1e-10, atol=1e-10)


def test_reference_fixture_totals_match_survey(oracle, lj_sample):
    """SURVEY.md 8(c) totals on lj_sample.xyz (survey-time numpy restatement)."""
    m = oracle.model(3.0, 2.5)
    atoms = oracle.lj_atoms(1.0, 1.0, 800)
    x = lj_sample.astype(np.float64)
    f, e, w = oracle.naive(x, 10.0, m, atoms, oracle.LITERAL)
    assert e.sum() == pytest.approx(-4466.170949619774, rel=1e-12)
    assert w.sum() == pytest.approx(-2000.6794145109884, rel=1e-12)
    assert f[0] == pytest.approx([-10.64328702, -3.4714277, -16.41018877], rel=1e-8)
    assert e[0] == pytest.approx(-5.592611750406278, rel=1e-12)
    f, e, w = oracle.naive(x, 10.0, m, atoms, oracle.CUTOFF)
    assert e.sum() == pytest.approx(-4292.184050996729, rel=1e-12)
    assert w.sum() == pytest.approx(-957.3125855282784, rel=1e-12)
    assert e[0] == pytest.approx(-5.369534210366058, rel=1e-12)
    assert w[0] == pytest.approx(1.6144006748865198, rel=1e-12)
    assert np.abs(f.sum(axis=0)).max() < 1e-9                       # Newton's third law


def test_reference_test_bound_f32_vs_f64(oracle, lj_sample):
    """The reference's own bound (test/runtests.jl:39-41): two implementations agree to 1e-4 in Float32."""
    m32, m64 = oracle.model(3.0, 2.5, np.float32), oracle.model(3.0, 2.5)
    atoms = oracle.lj_atoms(1.0, 1.0, 800)
    f32, e32, w32 = oracle.naive(lj_sample, 10.0, m32, atoms, oracle.LITERAL)
    f64, e64, w64 = oracle.naive(lj_sample.astype(np.float64), 10.0, m64, atoms, oracle.LITERAL)
    assert np.abs(f32 - f64).max() < 1e-4 * max(1.0, np.abs(f64).max())
    assert np.abs(e32 - e64).max() < 1e-4
    assert np.abs(w32 - w64).max() < 2e-4


def test_cell_path_equals_all_pairs_cutoff(oracle, lj_sample, golden):
    """O(N) cell-list oracle == all-pairs CUTOFF oracle (the structure of the reference's differential test)."""
    m = oracle.model(3.0, 2.5)
    atoms = oracle.lj_atoms(1.0, 1.0, 800)
    x = lj_sample.astype(np.float64)
    f0, e0, w0 = oracle.naive(x, 10.0, m, atoms, oracle.CUTOFF)
    for nt in (1, 3):
        f1, e1, w1 = oracle.nonbonded_cells(x, 10.0, m, atoms, nthreads=nt)
        np.testing.assert_allclose(f1, f0, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(e1, e0, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(w1, w0, rtol=1e-11, atol=1e-11)
    m32 = oracle.model(3.0, 2.5, np.float32)
    f2, e2, w2 = oracle.nonbonded_cells(lj_sample, 10.0, m32, atoms)
    assert np.abs(f2 - f0).max() < 1e-3 and np.abs(e2 - e0).max() < 1e-4


def test_fcc864_and_mixture_against_numpy(oracle, emdee_synthetic, golden):
    syn = emdee_synthetic
    g = golden["fcc864_expected"]
    pos, L = syn.fcc_positions(6)
    assert pos.shape == (864, 3) and L == pytest.approx(10.2599, abs=1e-4)
    assert pos.sum() == g["pos_checksum"]                           # generator is bit-reproducible
    m = oracle.model(2.5, 2.0)
    atoms = oracle.lj_atoms(1.0, 1.0, 864)
    for f, e, w in (oracle.naive(pos, L, m, atoms, oracle.CUTOFF), oracle.nonbonded_cells(pos, L, m, atoms)):
        np.testing.assert_allclose(f, g["forces"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(e, g["energies"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(w, g["virials"], rtol=1e-10, atol=1e-10)
    gm = golden["mix500_expected"]
    pos, L = syn.fcc_positions(5)
    types = syn.mixture_types(500)
    assert (types == gm["types"]).all()
    eps, sigma = syn.mixture_parameters(types)
    atoms = oracle.lj_atoms(eps, sigma)
    m = oracle.model(3.5, 3.0)
    for f, e, w in (oracle.naive(pos, L, m, atoms, oracle.CUTOFF), oracle.nonbonded_cells(pos, L, m, atoms)):
        np.testing.assert_allclose(f, gm["forces"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(e, gm["energies"], rtol=1e-11, atol=1e-12)


def test_verlet_100_steps_against_numpy(oracle, emdee_synthetic, golden):
    """BASELINE.json configs[0]: 864-atom fcc box, 100 velocity-Verlet steps, dt = 0.005."""
    syn = emdee_synthetic
    g = golden["fcc864_expected"]
    pos, L = syn.fcc_positions(6)
    vel = syn.velocities(864)
    assert np.abs(vel).sum() == g["vel_checksum"]
    m = oracle.model(2.5, 2.0)
    atoms = oracle.lj_atoms(1.0, 1.0, 864)
    for use_cells in (False, True):
        out = oracle.verlet(pos, vel, L, m, atoms, 0.005, 100, use_cells=use_cells)
        np.testing.assert_allclose(out["x"], g["x100"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(out["v"], g["v100"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(out["epot"], g["epot"], rtol=1e-10)
        np.testing.assert_allclose(out["ekin"], g["ekin"], rtol=1e-10)
        etot = out["epot"] + out["ekin"]
        assert abs(etot[-1] / etot[0] - 1.0) < 1e-4                 # NVE drift bound of SURVEY 8(c)
    assert 2.0 * g["ekin"][0] / (3 * 864 - 3) == pytest.approx(1.0, rel=1e-12)   # exactly T* = 1


def test_cells_convention(oracle):
    """src/cells.jl:36,180-181: M = floor(ndiv L / cutoff); id = 1 + vx + M vy + M^2 vz."""
    rng = np.random.default_rng(7)
    pos = rng.uniform(-3.0, 13.0, size=(2000, 3))
    M, index, pop = oracle.cells(pos, 10.0, 2.0, ndiv=2)
    assert M == 10 and pop.sum() == 2000 and index.min() >= 1 and index.max() <= 1000
    s = pos / 10.0
    v = np.floor(M * (s - np.floor(s))).astype(np.int64)
    np.testing.assert_array_equal(index, 1 + v[:, 0] + M * v[:, 1] + M * M * v[:, 2])
    np.testing.assert_array_equal(pop, np.bincount(index - 1, minlength=M ** 3))
    # the commented-out test_cells (test/runtests.jl:6-17): population is invariant under re-binning
    M2, index2, pop2 = oracle.cells(pos + 10.0, 10.0, 2.0, ndiv=2)
    np.testing.assert_array_equal(pop, pop2)
    assert oracle.cells(pos.astype(np.float32), 10.0, 2.0)[0] == 10


def test_neighbor_list_matches_brute_force(oracle, lj_sample):
    x = lj_sample.astype(np.float64)
    off, nb = oracle.neighbor_list(x, 10.0, 3.3)
    d = x[:, None, :] / 10.0 - x[None, :, :] / 10.0
    rv = 10.0 * (d - np.rint(d))
    r2 = (rv * rv).sum(-1)
    np.fill_diagonal(r2, 1e9)
    want = r2 < 3.3 ** 2
    assert off[-1] == want.sum()
    for i in (0, 17, 799):
        np.testing.assert_array_equal(nb[off[i]:off[i + 1]], np.nonzero(want[i])[0])
    with pytest.raises(ValueError):
        oracle.neighbor_list(x, 10.0, 5.5)


def test_empty_and_tiny_inputs(oracle):
    m = oracle.model(2.5, 2.0)
    f, e, w = oracle.naive(np.zeros((0, 3)), 10.0, m, oracle.lj_atoms(1, 1, 0))
    assert f.shape == (0, 3)
    f, e, w = oracle.nonbonded_cells(np.zeros((1, 3)), 10.0, m, oracle.lj_atoms(1, 1, 1))
    assert (f == 0).all() and e[0] == 0
    x = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0]])
    f, e, w = oracle.nonbonded_cells(x, 6.0, m, oracle.lj_atoms(1, 1, 2))      # M = 2: de-duplicated stencil
    assert f[0, 0] == pytest.approx(-24.0) and f[1, 0] == pytest.approx(24.0) and w.sum() == pytest.approx(24.0)


def test_langevin_generator_and_thermostatted_steps_match_independent_restatement(oracle, emdee_synthetic):
    """The thermostat is build-defined: pin the oracle's counter-based normals and its
    thermostatted integrator on the python-int / numpy restatement of tests/golden/make_golden.py."""
    import json
    import os
    from .conftest import ROOT
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_langevin.json")))
    for c in kat["normals"]:
        got = oracle.langevin_normals(c["seed"], c["step"], c["id"])
        assert np.abs(got - np.array(c["normals"])).max() < 1e-15
    z = np.array([oracle.langevin_normals(11, s, i) for s in range(40) for i in range(250)])
    assert np.abs(z.mean(axis=0)).max() < 0.05 and np.abs(z.std(axis=0) - 1.0).max() < 0.05
    syn = emdee_synthetic
    pos, L = syn.fcc_positions(6)
    vel = syn.velocities(pos.shape[0])
    atoms = oracle.lj_atoms(np.ones(pos.shape[0]), np.ones(pos.shape[0]))
    mdl = oracle.model(2.5, 2.0)
    for use_cells in (False, True):
        r = oracle.verlet_langevin(pos, vel, L, mdl, atoms, 0.005, 10, 2.0, 0.7, 0x5EED, use_cells=use_cells)
        assert float(np.sum(r["x"])) == pytest.approx(kat["x10_sum"], rel=1e-12)
        assert float(np.abs(r["v"]).sum()) == pytest.approx(kat["v10_abs_sum"], rel=1e-10)
        assert float(np.abs(r["f"]).sum()) == pytest.approx(kat["f10_abs_sum"], rel=1e-9)
        assert np.abs(r["v"][0] - np.array(kat["v10_first"])).max() < 1e-11
    # gamma = 0: c1 = 1, c2 = 0 -> the plain integrator, bit for bit
    a = oracle.verlet_langevin(pos, vel, L, mdl, atoms, 0.005, 5, 0.0, 0.7, 1)
    b = oracle.verlet(pos, vel, L, mdl, atoms, 0.005, 5)
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["v"], b["v"])


def test_oracle_under_asan():
    """The checker itself under AddressSanitizer + UBSan (`make -C oracle asan`; GPU sanitizers are not available on the pool,
    SURVEY.md section 5): this file once more in a child process that loads libemdee_oracle_asan.so -- an out-of-bounds
    read in the cell-list restatement or a signed overflow in the Langevin counters would end the child."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "asan"])
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isabs(asan_rt) and os.path.exists(asan_rt), asan_rt
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1",
               EMDEE_ORACLE_LIB=os.path.join(root, "oracle", "libemdee_oracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle.py", "-x", "-q", "-p", "no:cacheprovider", "-k", "not under_asan"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, \
        r.stdout[-800:] + r.stderr[-1200:]
