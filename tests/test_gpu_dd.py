"""GPU tests of the decomposition behind the C ABI (emdee_dd_*): migration, ghost
selection, per-step halo messages with the rebuild request riding on them, batches of queued steps with
device-side guard words.  Most tests run the whole decomposition in ONE process on cuda:0 (n_local = world: the
transport is device-to-device copies between the domains' streams; everything else is the production path) and the
trajectory must match the CPU oracle on the undivided periodic box.  The last test runs real ranks -- separate
processes, ncclSend/ncclRecv -- on the one device (RCCL's TCP transport; see profiles/rccl_ranks_one_gpu.py)."""
import os

import numpy as np
import pytest

from .conftest import ROOT

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

RC, RS, SKIN, DT = 2.5, 2.0, 0.3, 0.005
NCELL = 8
LANGEVIN = (2.0, 0.7, 0x5EED)


def _global_box(syn, uniform=False, ncell=NCELL):
    pos, gid, lengths = syn.fcc_block((ncell,) * 3, (0, 0, 0), (ncell,) * 3)
    pos = pos[np.argsort(gid)]
    N = pos.shape[0]
    vel = syn.raw_normals(np.arange(N), N)
    vel -= vel.mean(axis=0)
    vel *= np.sqrt((3 * N - 3) / np.sum(vel * vel))
    eps, sigma = syn.mixture_parameters(syn.mixture_types(N))
    if uniform:
        eps, sigma = np.ones(N), np.ones(N)
    return pos, vel, eps, sigma, float(lengths[0])


def _build(E, world, pos, vel, atoms, L, dtype=torch.float64, scatter=True):
    dev = torch.device("cuda", 0)
    N = pos.shape[0]
    dd = E.DomainDecomposition([L] * 3, E.domain.rank_grid(world), E.LennardJonesModel(RC, RS), skin=SKIN, dtype=dtype, device=dev)
    ndt = np.float64 if dtype == torch.float64 else np.float32
    for r in range(world):
        # every domain starts from an arbitrary slice of the atoms: the load must hand each to its brick
        mine = np.arange(r, N, world) if scatter else np.arange(N)[(np.arange(N) * world) // N == r]
        dd.set_atoms_(r, E.cu(pos[mine].astype(ndt), dev), E.cu(vel[mine].astype(ndt), dev), E.cu(atoms[mine], dev),
                      torch.from_numpy(mine.astype(np.int64)).to(dev))
    dd.load_()
    return dd


def _gather(dd, world, N):
    x, v, f = np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3))
    seen = np.zeros(N, dtype=int)
    for r in range(world):
        gid, xr, vr, fr = dd.state(r)
        gid = gid.cpu().numpy()
        seen[gid] += 1
        x[gid], v[gid], f[gid] = xr.cpu().numpy(), vr.cpu().numpy(), fr.cpu().numpy()
    assert (seen == 1).all(), "every atom is owned by exactly one domain"
    return x, v, f


@pytest.mark.parametrize("world,rebuild_every,langevin,uniform",
                         [(2, 0, 0, 0), (4, 0, 0, 0), (8, 0, 0, 1), (3, 0, 0, 0), (2, 4, 0, 0), (4, 7, 1, 1), (8, 0, 1, 0), (1, 0, 0, 0),
                          (1, 0, 1, 1), (1, 5, 1, 0)])   # (one domain = the lock-step halves of a production rank; with the thermostat's noise)
def test_dd_trajectory_matches_oracle(emdee, oracle, world, rebuild_every, langevin, uniform):
    E = emdee
    pos, vel, eps, sigma, L = _global_box(E.synthetic, uniform)
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    dd = _build(E, world, pos, vel, atoms, L)
    if langevin:
        dd.set_langevin_(*LANGEVIN)
    c = [dd.counts(r) for r in range(world)]
    assert sum(k["n_owned"] for k in c) == N and c[0]["n_global"] == N
    assert all(k["n_ghost"] > 0 for k in c) or world == 1
    e0 = dd.totals()
    nsteps = 25
    dd.step_(13, DT, rebuild_every)                  # two calls: the closing half kick and the re-opening must chain
    dd.step_(nsteps - 13, DT, rebuild_every)
    e1 = dd.totals()
    orc_atoms = oracle.lj_atoms(eps, sigma)
    if langevin:       # noise keyed by (seed, step number, global atom id): steps are numbered across the chained calls
        ref = oracle.verlet_langevin(pos, vel, L, oracle.model(RC, RS), orc_atoms, DT, nsteps, *LANGEVIN)
    else:
        ref = oracle.verlet(pos, vel, L, oracle.model(RC, RS), orc_atoms, DT, nsteps)
    x, v, f = _gather(dd, world, N)
    dx = x - ref["x"]
    assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-9
    assert np.abs(v - ref["v"]).max() < 1e-8
    assert np.abs(f - ref["f"]).max() < 1e-6 * np.abs(ref["f"]).max()
    assert e0[0] == pytest.approx(ref["epot"][0], rel=1e-10) and e0[1] == pytest.approx(ref["ekin"][0], rel=1e-10)
    assert e1[0] == pytest.approx(ref["epot"][-1], rel=1e-8) and e1[1] == pytest.approx(ref["ekin"][-1], rel=1e-8)
    st = dd.stats()
    assert st["rebuilds"] >= 2
    if world > 1:
        assert st["migrated"] > 0                    # the scattered initial slices had to be sorted out
    dd.close()


@pytest.mark.parametrize("grid,langevin,dtype", [((2, 2, 2), 0, "f64"), ((2, 1, 1), 1, "f64"), ((2, 2, 1), 0, "f64"), ((2, 2, 2), 0, "f32")])
def test_dd_replica_rehearsal_of_one_rank_matches_the_periodic_brick(emdee, oracle, grid, langevin, dtype):
    """One rank of a 2 x 2 x 2 (2 x 1 x 1, 2 x 2 x 1) grid on its own, every peer its own periodic image (mirror=True: the
    messages a rank with SEVEN peers packs, sends, receives and unpacks -- migrants, padded ghost rows, the per-step halo on
    its stream -- are all there, copied from its own send buffer): the box is then periodic with the brick's period, and the
    run must reproduce the undivided periodic box of ONE brick.  Corner and edge directions, a peer reached in several
    directions, atoms that leave through a face and come back in through the opposite one, rebuilds in the engine's order."""
    E = emdee
    pos, vel, eps, sigma, W = _global_box(E.synthetic, uniform=False)
    vel = 1.4 * vel
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    ndt = np.float64 if dtype == "f64" else np.float32
    dev = torch.device("cuda", 0)
    dd = E.DomainDecomposition([g * W for g in grid], grid, E.LennardJonesModel(RC, RS), skin=SKIN, dtype=tdt, device=dev, rank=0, mirror=True)
    dd.set_atoms_(0, E.cu(pos.astype(ndt), dev), E.cu(vel.astype(ndt), dev), E.cu(atoms, dev), torch.arange(N, dtype=torch.int64, device=dev))
    dd.load_()
    if langevin:
        dd.set_langevin_(*LANGEVIN)
    world = grid[0] * grid[1] * grid[2]
    c = dd.counts(0)
    assert c["n_owned"] == N and c["n_global"] == world * N and c["n_ghost"] > 0
    nsteps = 30
    dd.step_(13, DT, 0)
    ph0, st0 = dd.phase_times(), dd.stats()
    dd.step_(nsteps - 13, DT, 4)
    ph1, st1 = dd.phase_times(), dd.stats()
    # every rebuild of the second call ran in the engine's own order, with ONE blocking read-back inside it (the build's words
    # carry the counts of both exchanges), and no engine had to be loaded again for room
    n_rb = st1["rebuilds"] - st0["rebuilds"]
    assert n_rb >= 3 and ph1["rebuilds_in_engine_order"] - ph0["rebuilds_in_engine_order"] == n_rb
    assert ph1["rebuild_readbacks"] - ph0["rebuild_readbacks"] == n_rb and ph1["engines_regrown"] == 0
    e1 = dd.totals()
    orc_atoms = oracle.lj_atoms(eps, sigma)
    p0, v0 = pos.astype(ndt).astype(np.float64), (vel.astype(ndt)).astype(np.float64)
    if langevin:
        ref = oracle.verlet_langevin(p0, v0, W, oracle.model(RC, RS), orc_atoms, DT, nsteps, *LANGEVIN)
    else:
        ref = oracle.verlet(p0, v0, W, oracle.model(RC, RS), orc_atoms, DT, nsteps)
    gid, x, v, f = (t.cpu().numpy() for t in dd.state(0))
    assert np.array_equal(np.sort(gid), np.arange(N))
    xs, vs, fs = np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3))
    xs[gid], vs[gid], fs[gid] = x, v, f
    dx = xs - ref["x"]
    tol = 1e-9 if dtype == "f64" else 2e-3
    assert np.abs(dx - W * np.rint(dx / W)).max() < tol
    assert np.abs(vs - ref["v"]).max() < 10 * tol
    assert np.abs(fs - ref["f"]).max() < (1e-6 if dtype == "f64" else 2e-2) * np.abs(ref["f"]).max()
    # (the totals are those of the rehearsed grid: every rank holds the same)
    assert e1[0] == pytest.approx(world * ref["epot"][-1], rel=1e-8 if dtype == "f64" else 2e-4)
    st, rs = dd.stats(), dd.rebuild_stats()
    assert st["rebuilds"] >= 4 and st["migrated"] > 0 and rs["count_free"] >= st["rebuilds"] - 2
    dd.close()


@pytest.mark.parametrize("world", [1, 4])
def test_dd_in_order_exchange_gives_the_same_trajectory(emdee, world, monkeypatch):
    """emdee_dd_set_overlap(0): pack, exchange, unpack and ONE launch over all bricks in order (for one domain per process
    also without any event) -- the same states as the overlapped form with its interior / boundary launches, switched in
    the middle of a run and back.  The overlapped form itself runs the steps that WAIT for a rebuild request in order (round 5:
    no interior launch is started that a neighbour's request would void); EMDEE_DD_HOLD_INTERIOR=0, overlapped throughout as in
    rounds 3-4, must give the same states bit for bit as well."""
    E = emdee
    pos, vel, eps, sigma, L = _global_box(E.synthetic, uniform=True)
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    a, b = _build(E, world, pos, vel, atoms, L), _build(E, world, pos, vel, atoms, L)
    monkeypatch.setenv("EMDEE_DD_HOLD_INTERIOR", "0")
    c = _build(E, world, pos, vel, atoms, L)
    monkeypatch.delenv("EMDEE_DD_HOLD_INTERIOR")
    b.set_overlap_(False)
    a.step_(11, DT, 0); b.step_(11, DT, 0); c.step_(11, DT, 0)
    b.set_overlap_(True); a.set_overlap_(False)
    a.step_(9, DT, 0); b.step_(9, DT, 0); c.step_(9, DT, 0)
    xa, va, fa = _gather(a, world, N)
    xb, vb, fb = _gather(b, world, N)
    xc, vc, fc = _gather(c, world, N)
    assert np.array_equal(xa, xb) and np.array_equal(va, vb) and np.array_equal(fa, fb)
    assert np.array_equal(xa, xc) and np.array_equal(va, vc) and np.array_equal(fa, fc)
    assert a.stats()["rebuilds"] == b.stats()["rebuilds"] == c.stats()["rebuilds"] >= 2
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("world,dtype", [(8, "f64"), (3, "f64"), (4, "f32"), (1, "f64")])
def test_dd_count_free_rebuilds_give_the_same_trajectory(emdee, world, dtype, monkeypatch):
    """Rebuilds after the first two send migrants and ghost rows in capacity-padded messages with NO count exchange and one
    read-back (emdee_dd_rebuild_stats); the states must be bitwise those of the rebuilds that exchange their counts first
    (EMDEE_DD_COUNT_FREE=0) -- and stay so when a capacity is exceeded (one migrant row per message: every rank sees the
    overflow in the headers it receives and the rebuild is redone with counts)."""
    E = emdee
    tdt = torch.float64 if dtype == "f64" else torch.float32
    pos, vel, eps, sigma, L = _global_box(E.synthetic, uniform=(world == 8))
    vel = 1.6 * vel                                  # hot: rebuilds every few steps, atoms change owner at most of them
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    monkeypatch.setenv("EMDEE_DD_NO_SHORTCUT", "1")  # (a one-domain grid goes through the ownership path too)
    runs = {}
    for name, env in (("counts", {"EMDEE_DD_COUNT_FREE": "0"}), ("free", {}), ("overflow", {"EMDEE_DD_MIG_CAP": "1"}),
                      ("ghost_overflow", {"EMDEE_DD_GHOST_CAP": "exact"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dd = _build(E, world, pos, vel, atoms, L, dtype=tdt)
        for k in env:
            monkeypatch.delenv(k)
        dd.step_(17, DT, 0)
        dd.step_(15, DT, 3)
        runs[name] = _gather(dd, world, N) + (dd.stats(), dd.rebuild_stats(), dd.totals())
        dd.close()
    xc, vc, fc, sc, rc, ec = runs["counts"]
    assert rc["count_free"] == 0 and sc["rebuilds"] >= 6
    for name in ("free", "overflow", "ghost_overflow"):
        x, v, f, st, rs, e = runs[name]
        assert np.array_equal(x, xc) and np.array_equal(v, vc) and np.array_equal(f, fc), name
        assert st["rebuilds"] == sc["rebuilds"] and st["migrated"] == sc["migrated"] and e == ec
        # everything after the load and the first rebuild from the engines (which fixes the capacities) is count-free
        assert rs["count_free"] >= sc["rebuilds"] - 2 > 0
    assert runs["free"][4]["redone"] == 0
    if world > 1:
        assert sc["migrated"] > N // 4               # (the scattered initial slices, then the hot box)
        assert runs["overflow"][4]["redone"] > 0 and runs["overflow"][4]["migrant_rows_per_peer"] == 1
        # ghost messages without headroom: a peer's ghost count grows at some rebuild, its message overflows, everybody redoes
        assert runs["ghost_overflow"][4]["redone"] > 0


@pytest.mark.parametrize("world,ncell", [(8, 12), (3, 10), (2, 8)])
def test_dd_neighbour_rows_are_complete(emdee, oracle, world, ncell):
    """Integer check of the lists inside a decomposition: the rows of the OWNED atoms of all domains together hold exactly
    the oracle's entries for the undivided periodic box (r < rc + skin), at the load and after rebuilds from the engines
    (count-free messages; local boxes are not periodic along cut dimensions; x sub-bins; own cells of ghosts skipped) --
    a pair the build dropped in the skin would not show in a short trajectory, it shows here."""
    E = emdee
    pos, vel, eps, sigma, L = _global_box(E.synthetic, uniform=True, ncell=ncell)
    rng = np.random.default_rng(5)
    pos = pos + rng.normal(0.0, 0.1, size=pos.shape)          # off the lattice: ragged rows
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    dd = _build(E, world, pos, 1.5 * vel, atoms, L)

    def check():
        x, _, _ = _gather(dd, world, N)
        x = x - L * np.floor(x / L)
        off, _ = oracle.neighbor_list(x, L, RC + SKIN)
        listed = sum(dd.engine(r).nbr_stats()["listed"] for r in range(world))
        inside = sum(dd.engine(r).count_pairs() for r in range(world))
        assert listed == int(off[-1]), "listed entries %d, oracle %d" % (listed, int(off[-1]))
        off_c, _ = oracle.neighbor_list(x, L, RC)
        # (count_pairs halves the in-cutoff entries of a domain's owned rows: up to one half per domain is truncated)
        assert abs(2 * inside - int(off_c[-1])) <= 2 * world, (inside, int(off_c[-1]))
        return listed

    first = check()
    dd.step_(6, DT, 1)                       # a rebuild at every step, the last one at the final positions
    assert dd.rebuild_stats()["count_free"] >= 4 or world == 1
    second = check()
    assert first > 0 and second > 0
    dd.close()


def _row_sample_check(E, dd, r, tree, L, rlist, nsample, rng):
    """Rows of a sample of the atoms domain r owns: entries distinct, every one inside r_list of the atom (minimum image of
    the whole box: 232.6 sigma against 2.8), and as many as the periodic box holds within r_list of that atom (KD-tree of
    the undivided box) -- together: the row IS the set."""
    eng = dd.engine(r)
    gid, xo, _, _ = dd.state(r)
    n_own = gid.shape[0]
    x_all = eng.state(velocities=False, forces=False)["positions"]         # owned atoms first, then the ghosts
    counts, nb = eng.neighbor_lists()
    assert counts.shape[0] == n_own
    dx0 = x_all[:n_own] - xo                                               # same atoms in the same order, up to whole box lengths
    assert float((dx0 - L * torch.round(dx0 / L)).abs().max()) < 1e-6
    # the atoms nearest to a face of the local box (their rows reach into the ghosts) and a random lot
    centre = 0.5 * (xo.min(dim=0).values + xo.max(dim=0).values)
    far = torch.argsort((xo - centre).abs().max(dim=1).values, descending=True)[: nsample // 2]
    pick = torch.unique(torch.cat([far, torch.from_numpy(rng.integers(0, n_own, nsample // 2)).to(far.device)]))
    c = counts[pick].long()
    rows = nb[pick].long()
    valid = torch.arange(nb.shape[1], device=nb.device)[None, :] < c[:, None]
    d = x_all[rows.clamp(min=0)] - x_all[pick][:, None, :]
    d = d - L * torch.round(d / L)                                        # (a ghost may be reported in another image than it is used in)
    d2 = (d * d).sum(dim=2)
    assert bool((d2[valid] < rlist * rlist).all()), "an entry outside r_list"
    srt = torch.where(valid, rows, torch.full_like(rows, -1)).sort(dim=1).values
    dup = (srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] >= 0)
    assert not bool(dup.any()), "an entry listed twice"
    xs = xo[pick].cpu().numpy()
    xs = xs - L * np.floor(xs / L)
    want = tree.query_ball_point(xs, rlist, return_length=True, workers=-1) - 1          # (not the atom itself)
    got = c.cpu().numpy()
    assert np.array_equal(got, want), "rows of %d sampled atoms differ in length from the undivided box" % int((got != want).sum())
    del counts, nb, x_all
    return int(pick.shape[0])


def test_full_size_box_in_eight_domains(emdee):
    """BASELINE configs[2] at the size it is quoted on, on ONE GPU: the 10,061,824-atom box (fcc 136^3 x 4, rho* = 0.8,
    rc = 2.5 sigma, fp64) cut into 2 x 2 x 2 domains held by one process (the production path but for the transport),
    against the undivided integrator on the same GPU.  At the load: the same number of listed entries (an integer, exact),
    the same counted pairs, the same energies; rows of a sample of owned atoms of EVERY domain -- those nearest to a cut
    face first -- complete against a KD-tree of the periodic box; after 12 steps with displacement-triggered rebuilds: the
    energies per atom of the undivided run to 1e-9, the same number of rebuilds."""
    from scipy.spatial import cKDTree
    E = emdee
    syn = E.synthetic
    dev = torch.device("cuda", 0)
    pos, L = syn.fcc_positions(136)
    N = pos.shape[0]
    assert N == 10061824
    vel = syn.velocities(N)
    atoms = E.lennard_jones_atoms(1.0, 1.0, N)
    nsteps = 12
    md = E.VelocityVerlet(E.cu(pos, dev), E.cu(vel, dev), L, E.LennardJonesModel(RC, RS), E.cu(atoms, dev), skin=SKIN)
    listed_md, pairs_md, e_md0 = md.nbr_stats()["listed"], md.count_pairs(), md.totals()
    md.step_(nsteps, DT)
    e_md1, builds_md = md.totals(), md.nbr_stats()["builds"]
    md.close()
    del md
    torch.cuda.empty_cache()

    world = 8
    dd = _build(E, world, pos, vel, atoms, L, scatter=False)
    c = [dd.counts(r) for r in range(world)]
    assert sum(k["n_owned"] for k in c) == N and all(k["n_ghost"] > 0.1 * k["n_owned"] for k in c)
    assert sum(dd.engine(r).nbr_stats()["listed"] for r in range(world)) == listed_md
    assert abs(2 * sum(dd.engine(r).count_pairs() for r in range(world)) - 2 * pairs_md) <= 2 * world
    e0 = dd.totals()
    assert e0[0] / N == pytest.approx(e_md0[0] / N, rel=1e-9) and e0[1] / N == pytest.approx(e_md0[1] / N, rel=1e-9)
    xw = pos - L * np.floor(pos / L)
    xw[xw >= L] = 0.0
    tree = cKDTree(xw, boxsize=L)
    rng = np.random.default_rng(3)
    checked = sum(_row_sample_check(E, dd, r, tree, L, RC + SKIN, 1600, rng) for r in range(world))
    assert checked > 8000
    del tree, xw
    dd.step_(nsteps, DT, 0)
    e1 = dd.totals()
    assert e1[0] / N == pytest.approx(e_md1[0] / N, rel=1e-9) and e1[1] / N == pytest.approx(e_md1[1] / N, rel=1e-9)
    # (the undivided count includes the load; a request raised by the last step is served inside the call by the
    # decomposition, at the next call by the undivided integrator)
    assert builds_md - 1 >= 1 and 0 <= dd.stats()["rebuilds"] - (builds_md - 1) <= 1
    assert dd.rebuild_stats()["count_free"] >= 1
    dd.close()
    torch.cuda.empty_cache()


def test_queries_of_an_in_process_domain_are_ordered_on_the_callers_stream(emdee):
    """The engines of an in-process decomposition run on streams of the library's own; emdee_md_nbr_list / emdee_md_get_state /
    emdee_dd_get_state fill arrays the CALLER hands in.  The library orders those writes against the caller's context stream
    itself (csrc/common.hpp FenceOut: two event hops, no device synchronisation, nothing in the Python binding): here torch's
    stream is kept busy for tens of milliseconds writing a block that is then freed -- the allocator hands the same memory to
    the query's arrays while that work is still queued -- and an 8-domain box of 1,048,576 atoms must still return complete
    rows and the same state as a query on an idle device.  (Round 4 found rows still holding their -1 fill at 10^7 atoms and
    fenced in Python; a C or Julia caller had the same race.)"""
    E = emdee
    dev = torch.device("cuda", 0)
    pos, L = E.synthetic.fcc_positions(64)
    N = pos.shape[0]
    assert N == 1048576
    vel = E.synthetic.velocities(N)
    atoms = E.lennard_jones_atoms(1.0, 1.0, N)
    dd = _build(E, 8, pos, vel, atoms, L, scatter=False)
    dd.step_(3, DT, 0)
    torch.cuda.synchronize(dev)
    quiet = []
    for r in range(8):
        eng = dd.engine(r)
        c, nb = eng.neighbor_lists()
        gid, x, v, f = dd.state(r)
        torch.cuda.synchronize(dev)
        quiet.append((c.clone(), nb.clone(), gid.clone(), x.clone(), f.clone()))
        del c, nb, gid, x, v, f
    torch.cuda.empty_cache()
    m = 3072
    b = torch.randn((m, m), dtype=torch.float64, device=dev)
    for r in range(8):
        eng = dd.engine(r)
        rows = int(quiet[r][1].numel())
        a = torch.empty(max(rows, 4 * m * m), dtype=torch.int32, device=dev)          # the block the query's arrays will come from
        av = a[: 2 * m * m].view(torch.float64).view(m, m)
        for _ in range(12):                                                              # ~ tens of ms of fp64 GEMMs writing into it, queued
            torch.mm(b, b, out=av)
        a[2 * m * m:].fill_(-7)
        del av, a                                                                        # freed while that work is still queued
        c, nb = eng.neighbor_lists()
        gid, x, v, f = dd.state(r)
        torch.cuda.synchronize(dev)
        assert torch.equal(c, quiet[r][0]), "row counts of domain %d changed under a busy caller stream" % r
        assert torch.equal(nb, quiet[r][1]), "rows of domain %d changed under a busy caller stream" % r
        assert torch.equal(gid, quiet[r][2]) and torch.equal(x, quiet[r][3]) and torch.equal(f, quiet[r][4])
        valid = torch.arange(nb.shape[1], device=dev)[None, :] < c[:, None].long()
        assert bool((nb[valid] >= 0).all()), "a listed entry still holds its fill"
    dd.close()


@pytest.mark.parametrize("config", ["fp32", "mixture_rc35"])
def test_full_size_box_in_two_domains(emdee, capfd, monkeypatch, config):
    """BASELINE configs[3] and configs[4] at full size through the decomposition (two domains in one process): fp32 storage
    and pair math at rc = 2.5 sigma; the binary mixture at rc = 3.5 sigma in fp64, which must be stepped by the TYPED
    kernels (csrc/typed.hpp) in both domains.  Size-independent properties, as in tests/test_gpu_parity.py for the undivided
    boxes: every atom owned once, counted pairs against nbar(rc), Newton's third law over the owned rows, energy and
    momentum over displacement-triggered rebuilds."""
    E = emdee
    syn = E.synthetic
    dev = torch.device("cuda", 0)
    pos, L = syn.fcc_positions(136)
    N = pos.shape[0]
    vel = syn.velocities(N)
    mixture = config == "mixture_rc35"
    rc, rs = (3.5, 3.0) if mixture else (2.5, 2.0)
    dtype = torch.float64 if mixture else torch.float32
    if mixture:
        eps, sigma = syn.mixture_parameters(syn.mixture_types(N))
        atoms = E.lennard_jones_atoms(eps, sigma)
    else:
        atoms = E.lennard_jones_atoms(1.0, 1.0, N)
    monkeypatch.setenv("EMDEE_DEBUG_PLAN", "1")
    capfd.readouterr()
    world = 2
    dd = E.DomainDecomposition([L] * 3, E.domain.rank_grid(world), E.LennardJonesModel(rc, rs), skin=SKIN, dtype=dtype, device=dev)
    ndt = np.float64 if mixture else np.float32
    for r in range(world):
        mine = np.arange(N)[(np.arange(N) * world) // N == r]
        dd.set_atoms_(r, E.cu(pos[mine].astype(ndt), dev), E.cu(vel[mine].astype(ndt), dev), E.cu(atoms[mine], dev),
                      torch.from_numpy(mine.astype(np.int64)).to(dev))
    dd.load_()
    del pos, vel
    err = capfd.readouterr().err
    if mixture:
        assert err.count("typed kernels on") >= 2, err[-600:]              # both domains
    else:
        assert "typed kernels on" not in err
    c = [dd.counts(r) for r in range(world)]
    assert sum(k["n_owned"] for k in c) == N and c[0]["n_global"] == N
    nbar = (4.0 / 3.0) * np.pi * rc ** 3 * 0.8
    pairs = sum(dd.engine(r).count_pairs() for r in range(world))
    assert abs(pairs / (0.5 * N) - nbar) < (6.0 if mixture else 3.0), pairs / (0.5 * N)
    ftot, fmax, ptot0 = torch.zeros(3, dtype=torch.float64, device=dev), 0.0, torch.zeros(3, dtype=torch.float64, device=dev)
    for r in range(world):
        _, _, v, f = dd.state(r)
        ftot += f.double().sum(dim=0); fmax = max(fmax, f.abs().max().item()); ptot0 += v.double().sum(dim=0)
        del v, f
    assert ftot.abs().max().item() < (1e-6 if mixture else 1e-3) * fmax * N ** 0.5      # Newton's third law across the cut
    e0 = dd.totals()
    nsteps = 24
    dd.step_(nsteps, DT, 0)
    e1 = dd.totals()
    err = capfd.readouterr().err
    if mixture:
        assert "typed kernels off" not in err, err[-600:]
    drift = ((e1[0] + e1[1]) - (e0[0] + e0[1])) / abs(e0[0] + e0[1])
    assert abs(drift) < (3e-4 if mixture else 5e-5), drift
    ptot = torch.zeros(3, dtype=torch.float64, device=dev)
    for r in range(world):
        _, _, v, _ = dd.state(r)
        ptot += v.double().sum(dim=0)
        del v
    assert (ptot - ptot0).abs().max().item() < (1e-9 if mixture else 2e-3) * N ** 0.5 * 10
    st = dd.stats()
    assert st["rebuilds"] >= 2 and dd.rebuild_stats()["count_free"] >= 1
    assert 2.0 * e1[1] / (3 * N - 3) > 0.4
    pairs1 = sum(dd.engine(r).count_pairs() for r in range(world))
    assert abs(pairs1 / (0.5 * N) - nbar) < (5.0 if mixture else 2.0), pairs1 / (0.5 * N)
    dd.close()
    torch.cuda.empty_cache()


def test_dd_langevin_single_call_matches_oracle(emdee, oracle):
    """Noise keyed by global atom id and step number: the decomposed run draws what the undivided run draws."""
    E = emdee
    pos, vel, eps, sigma, L = _global_box(E.synthetic)
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    dd = _build(E, 4, pos, vel, atoms, L)
    dd.set_langevin_(*LANGEVIN)
    dd.step_(25, DT, 0)
    ref = oracle.verlet_langevin(pos, vel, L, oracle.model(RC, RS), oracle.lj_atoms(eps, sigma), DT, 25, *LANGEVIN)
    x, v, f = _gather(dd, 4, N)
    dx = x - ref["x"]
    assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-9
    assert np.abs(v - ref["v"]).max() < 1e-8
    e1 = dd.totals()
    assert e1[0] == pytest.approx(ref["epot"][-1], rel=1e-8) and e1[1] == pytest.approx(ref["ekin"][-1], rel=1e-8)


def test_dd_long_run_with_migration_and_batches(emdee, oracle):
    """Hot box, 120 steps: many displacement-triggered rebuilds, atoms crossing brick faces, queued steps
    cancelled by a neighbour's request -- and still the undivided trajectory (to the rounding the chaotic
    dynamics allows over this length) and its conserved energy."""
    E = emdee
    pos, vel, eps, sigma, L = _global_box(E.synthetic, uniform=True)
    vel = vel * np.sqrt(2.0)                            # T* = 2
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    dd = _build(E, 8, pos, vel, atoms, L, scatter=False)
    e0 = dd.totals()
    dd.step_(120, DT, 0)
    e1 = dd.totals()
    ref = oracle.verlet(pos, vel, L, oracle.model(RC, RS), oracle.lj_atoms(eps, sigma), DT, 120)
    x, v, f = _gather(dd, 8, N)
    dx = x - ref["x"]
    assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-7
    assert e1[0] + e1[1] == pytest.approx(ref["epot"][-1] + ref["ekin"][-1], rel=1e-9)
    assert abs((e1[0] + e1[1]) - (e0[0] + e0[1])) < 2e-4 * abs(e0[0] + e0[1])
    st = dd.stats()
    assert st["rebuilds"] >= 8 and st["batches"] >= 20
    # the engines' neighbour-list builds agree with the decomposition's count
    assert dd.engine(0).nbr_stats()["builds"] >= st["rebuilds"]


def test_dd_fp32_and_bad_arguments(emdee):
    E = emdee
    pos, vel, eps, sigma, L = _global_box(E.synthetic, uniform=True)
    atoms = E.lennard_jones_atoms(eps, sigma)
    dd = _build(E, 2, pos, vel, atoms, L, dtype=torch.float32)
    e0 = dd.totals()
    dd.step_(40, DT, 0)
    e1 = dd.totals()
    assert abs((e1[0] + e1[1]) - (e0[0] + e0[1])) < 5e-4 * abs(e0[0] + e0[1])     # NVE in fp32 storage + pair math
    with pytest.raises(E.EmDeeError):
        E.DomainDecomposition([L] * 3, (4, 1, 1), E.LennardJonesModel(RC, RS), device=torch.device("cuda", 0))   # > 3 bricks
    with pytest.raises(E.EmDeeError):
        E.DomainDecomposition([5.0] * 3, (2, 1, 1), E.LennardJonesModel(RC, RS), device=torch.device("cuda", 0))   # halo > brick
    fresh = E.DomainDecomposition([L] * 3, (2, 1, 1), E.LennardJonesModel(RC, RS), device=torch.device("cuda", 0))
    with pytest.raises(E.EmDeeError):
        fresh.step_(1, DT)                               # step before load


@pytest.mark.parametrize("world", [2, 8])
def test_dd_box_with_interior_bricks(emdee, world):
    """A box wide enough (L = 51 sigma, bricks of 25.6 sigma) for every domain to HAVE interior bricks -- the launches
    that run while the halo exchange is in flight and are guarded by the domain's own rebuild word only -- against the
    undivided integrator on the same GPU: 108,000 atoms, 30 steps, several rebuilds."""
    E = emdee
    dev = torch.device("cuda", 0)
    pos, vel, eps, sigma, L = _global_box(E.synthetic, uniform=True, ncell=30)
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    dd = _build(E, world, pos, vel, atoms, L, scatter=False)
    dd.step_(30, DT, 0)
    md = E.VelocityVerlet(E.cu(pos, dev), E.cu(vel, dev), L, E.LennardJonesModel(RC, RS), E.cu(atoms, dev), skin=SKIN)
    md.step_(30, DT, 0)
    st = md.state()
    x, v, f = _gather(dd, world, N)
    dx = x - st["positions"].cpu().numpy()
    assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-9
    assert np.abs(v - st["velocities"].cpu().numpy()).max() < 1e-8
    assert np.abs(f - st["forces"].cpu().numpy()).max() < 1e-6 * np.abs(f).max()
    e_dd, e_md = dd.totals(), md.totals()
    assert e_dd[0] == pytest.approx(e_md[0], rel=1e-10) and e_dd[1] == pytest.approx(e_md[1], rel=1e-10)
    assert dd.stats()["rebuilds"] >= 3


def _jittered_slab(rng, x0, x1, L, spacing, jitter):
    nx, nyz = int(round((x1 - x0) / spacing)), int(L // spacing)
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(nyz), np.arange(nyz), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    return np.array([x0, 0.0, 0.0]) + (g + 0.5) * spacing + rng.uniform(-jitter, jitter, size=g.shape)


@pytest.mark.parametrize("langevin", [0, 1])
def test_dd_domains_on_different_kernel_classes(emdee, oracle, langevin):
    """ADVICE r2 (dd.hpp:743): which kernels a domain steps with is data-dependent.  Here domain 0 holds a slab of small
    atoms at 4.6 per sigma^3 -- a tile of 96 cells would need ~300 KB of LDS, so its engine falls back to the direct
    (global-gather) kernels -- while domain 1 holds a dilute fluid and runs the LDS-tiled ones.  Both must follow the same
    batches of guarded steps (a domain that silently skipped its launches, or queued a different number of exchanges,
    would give a wrong trajectory or a hang), and the trajectory must be the undivided oracle's."""
    E = emdee
    rng = np.random.default_rng(5)
    L = 28.0
    dense = _jittered_slab(rng, 1.0, 13.0, L, 0.6, 0.05)
    dilute = _jittered_slab(rng, 15.0, 27.0, L, 1.12, 0.08)
    pos = np.concatenate([dense, dilute])
    N = pos.shape[0]
    eps = np.ones(N)
    sigma = np.concatenate([np.full(dense.shape[0], 0.4), np.ones(dilute.shape[0])])
    vel = 1.5 * E.synthetic.raw_normals(np.arange(N), N)
    vel -= vel.mean(axis=0)
    atoms = E.lennard_jones_atoms(eps, sigma)
    dd = _build(E, 2, pos, vel, atoms, L, scatter=False)
    if langevin:
        dd.set_langevin_(*LANGEVIN)
    for r in range(2):
        dd.engine(r).profile_(True)
    nsteps, dt = 40, 0.004                               # hot enough for several displacement-triggered rebuilds
    dd.step_(17, dt, 0)
    dd.step_(nsteps - 17, dt, 0)
    # domain 0 never launched the fused brick kernel, domain 1 did: the two kernel classes really ran side by side
    fused = [dd.engine(r).kernel_time("lj_force_nbr_fused_step")[1] for r in range(2)]
    split = [dd.engine(r).kernel_time("verlet_kick_drift")[1] for r in range(2)]
    assert fused[0] == 0 and split[0] >= nsteps - 2, (fused, split)
    assert fused[1] >= nsteps - 8, (fused, split)
    orc_atoms = oracle.lj_atoms(eps, sigma)
    if langevin:
        ref = oracle.verlet_langevin(pos, vel, L, oracle.model(RC, RS), orc_atoms, dt, nsteps, *LANGEVIN)
    else:
        ref = oracle.verlet(pos, vel, L, oracle.model(RC, RS), orc_atoms, dt, nsteps)
    x, v, f = _gather(dd, 2, N)
    dx = x - ref["x"]
    assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-9
    assert np.abs(v - ref["v"]).max() < 1e-8
    assert np.abs(f - ref["f"]).max() < 1e-6 * np.abs(ref["f"]).max()
    e1 = dd.totals()
    assert e1[0] == pytest.approx(ref["epot"][-1], rel=1e-8) and e1[1] == pytest.approx(ref["ekin"][-1], rel=1e-8)
    assert dd.stats()["rebuilds"] >= 2
    dd.close()


def test_dd_with_an_empty_domain(emdee, oracle):
    """All atoms in the left half of the box, none within a halo of the cut: domain 1 owns nothing and sees no ghost.  It
    must still take part in every exchange of every batch, and the trajectory must be the oracle's."""
    E = emdee
    rng = np.random.default_rng(9)
    L = 28.0
    pos = _jittered_slab(rng, 3.0, 10.0, L, 1.12, 0.08)
    N = pos.shape[0]
    eps, sigma = np.ones(N), np.ones(N)
    vel = E.synthetic.raw_normals(np.arange(N), N)
    vel -= vel.mean(axis=0)
    atoms = E.lennard_jones_atoms(eps, sigma)
    dd = _build(E, 2, pos, vel, atoms, L, scatter=True)
    c = [dd.counts(r) for r in range(2)]
    assert c[0]["n_owned"] == N and c[1]["n_owned"] == 0 and c[1]["n_ghost"] == 0
    dd.step_(25, DT, 0)
    ref = oracle.verlet(pos, vel, L, oracle.model(RC, RS), oracle.lj_atoms(eps, sigma), DT, 25)
    x, v, f = _gather(dd, 2, N)
    dx = x - ref["x"]
    assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-9
    assert np.abs(f - ref["f"]).max() < 1e-6 * np.abs(ref["f"]).max()
    e1 = dd.totals()
    assert e1[0] == pytest.approx(ref["epot"][-1], rel=1e-8) and e1[1] == pytest.approx(ref["ekin"][-1], rel=1e-8)
    assert dd.stats()["rebuilds"] >= 2
    dd.close()


def test_dd_two_species_long_cutoff_takes_the_typed_kernels(emdee, capfd, monkeypatch):
    """BASELINE configs[4] decomposed, at a size where every domain's tiles are those of the 10^7-atom box (rc = 3.5 sigma,
    97,556 atoms, 2 domains): the domains agree on the box's two species at the load, every engine then sorts by
    (cell, species) and steps with the typed kernels (csrc/typed.hpp), interior and boundary phases, guard words and all.
    Checked against the undivided integrator on the same GPU and, through it, against the general-species kernels."""
    E = emdee
    dev = torch.device("cuda", 0)
    rc, rs = 3.5, 3.0
    pos, vel, eps, sigma, L = _global_box(E.synthetic, ncell=29)
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    model = E.LennardJonesModel(rc, rs)
    monkeypatch.setenv("EMDEE_DEBUG_PLAN", "1")
    capfd.readouterr()
    dd = E.DomainDecomposition([L] * 3, (2, 1, 1), model, skin=SKIN, device=dev)
    for r in range(2):
        mine = np.arange(r, N, 2)
        dd.set_atoms_(r, E.cu(pos[mine], dev), E.cu(vel[mine], dev), E.cu(atoms[mine], dev), torch.from_numpy(mine).to(dev))
    dd.load_()
    dd.step_(20, DT, 0)
    e_dd = dd.totals()
    torch.cuda.synchronize()
    err = capfd.readouterr().err
    assert err.count("typed kernels on") >= 2, err[-600:]                 # both domains, at the load and after it
    monkeypatch.setenv("EMDEE_NO_TYPED", "1")                             # the witness: the general-species kernels, undivided
    md = E.VelocityVerlet(E.cu(pos, dev), E.cu(vel, dev), L, model, E.cu(atoms, dev), skin=SKIN)
    md.step_(20, DT, 0)
    torch.cuda.synchronize()
    assert "typed kernels on" not in capfd.readouterr().err
    st = md.state()
    x, v, f = _gather(dd, 2, N)
    dx = x - st["positions"].cpu().numpy()
    assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-9
    assert np.abs(v - st["velocities"].cpu().numpy()).max() < 1e-8
    assert np.abs(f - st["forces"].cpu().numpy()).max() < 1e-6 * np.abs(f).max()
    e_md = md.totals()
    assert e_dd[0] == pytest.approx(e_md[0], rel=1e-10) and e_dd[1] == pytest.approx(e_md[1], rel=1e-10)
    assert dd.stats()["rebuilds"] >= 3
    dd.close()


def test_rccl_binding_on_one_rank(emdee):
    """The RCCL transport cannot run between two ranks on a one-GPU box; what can be checked here is the run-time
    binding it rests on: librccl resolved with dlopen, ncclGetUniqueId / ncclCommInitRank (the 128-byte id by value),
    ncclSend + ncclRecv in one group on a stream of their own, ncclAllReduce, the enum values taken from rccl.h."""
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    emdee.dd.rccl_selftest(torch.device("cuda", 0), 1 << 20)
    emdee.dd.rccl_selftest(torch.device("cuda", 0), 16)
    uid = emdee.DomainDecomposition.unique_id()
    assert len(uid) == 128 and any(uid)


def test_dd_binary_mixture_long_cutoff(emdee, oracle):
    """BASELINE configs[4] in miniature, decomposed: two species (Lorentz-Berthelot through the LJAtom encoding), rc = 3.5
    sigma (halo 3.8 of a 6.84-wide brick, long neighbour rows, the general-species kernels in both phases) on 8 domains."""
    E = emdee
    rc, rs = 3.5, 3.0
    pos, vel, eps, sigma, L = _global_box(E.synthetic)
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    dev = torch.device("cuda", 0)
    dd = E.DomainDecomposition([L] * 3, (2, 2, 2), E.LennardJonesModel(rc, rs), skin=SKIN, device=dev)
    for r in range(8):
        mine = np.arange(r, N, 8)
        dd.set_atoms_(r, E.cu(pos[mine], dev), E.cu(vel[mine], dev), E.cu(atoms[mine], dev), torch.from_numpy(mine).to(dev))
    dd.load_()
    dd.step_(20, DT, 0)
    ref = oracle.verlet(pos, vel, L, oracle.model(rc, rs), oracle.lj_atoms(eps, sigma), DT, 20)
    x, v, f = _gather(dd, 8, N)
    dx = x - ref["x"]
    assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-9
    assert np.abs(f - ref["f"]).max() < 1e-6 * np.abs(ref["f"]).max()
    e1 = dd.totals()
    assert e1[0] == pytest.approx(ref["epot"][-1], rel=1e-8) and e1[1] == pytest.approx(ref["ekin"][-1], rel=1e-8)


@pytest.mark.parametrize("world,extra", [(2, []), (4, []), (3, ["--cells", "30"]), (4, ["--switch-overlap"]),
                                         (2, ["--precision", "f32", "--mixture", "--rc", "3.5", "--cells", "20"])])
def test_ranks_over_rccl_match_the_in_process_run(world, extra):
    """profiles/rccl_ranks_one_gpu.py: `world` separate processes, one communicator rank each, halo over ncclSend/ncclRecv
    (RCCL's TCP transport: a distinct NCCL_HOSTID per rank lets them share this box's one device) against the same grid
    stepped inside one process with device copies: atoms, rebuild count and all-reduced energies must agree."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "rccl_ranks_one_gpu.py"), "--world", str(world),
                        "--timeout", "240"] + extra, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0 and "MATCH" in r.stdout and "MISMATCH" not in r.stdout, r.stdout[-1500:] + r.stderr[-500:]
