"""GPU tests of the decomposed run: the HIP engine with ghost atoms, non-periodic
local boxes and the pack/unpack halo kernels, driven by DecomposedVerlet.  The 1-GPU test box cannot
host two RCCL ranks, so the 2-rank test shares cuda:0 between two processes and moves the halo through
gloo with host staging (transport="host"): every line of the product path except the RCCL transport
itself is exercised, and the trajectory must match the CPU oracle on the undivided periodic box."""
import os
import socket
import sys

import numpy as np
import pytest

from .conftest import ROOT

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.distributed as dist            # noqa: E402
import torch.multiprocessing as mp          # noqa: E402

RC, RS, SKIN, DT = 2.5, 2.0, 0.3, 0.005
NCELL = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_box(syn, uniform=False):
    pos, gid, lengths = syn.fcc_block((NCELL,) * 3, (0, 0, 0), (NCELL,) * 3)
    order = np.argsort(gid)
    pos = pos[order]
    N = pos.shape[0]
    vel = syn.raw_normals(np.arange(N), N)
    vel -= vel.mean(axis=0)
    vel *= np.sqrt((3 * N - 3) / np.sum(vel * vel))
    eps, sigma = syn.mixture_parameters(syn.mixture_types(N))
    if uniform:            # one species: the engine switches to its single-species kernels (coordinate-plane LDS tile)
        eps, sigma = np.ones(N), np.ones(N)
    return pos, vel, eps, sigma, float(lengths[0])


LANGEVIN = (2.0, 0.7, 0x5EED)     # gamma, T*, seed of the thermostatted case


def _worker(rank, world, port, out_dir, nsteps, rebuild_every, phased, langevin=0, uniform=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        pkg = load_package()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        syn, domain = pkg.synthetic, pkg.domain
        pos, vel, eps, sigma, L = _global_box(syn, uniform)
        N = pos.shape[0]
        atoms = pkg.lennard_jones_atoms(eps, sigma)
        model = pkg.LennardJonesModel(RC, RS)
        grid = domain.rank_grid(world)
        plan = domain.DomainPlan([L] * 3, RC + SKIN, world=world, rank=rank, device=dev, grid=grid, transport="host")
        # every rank starts from an arbitrary slice of the atoms: migrate() must sort them out
        mine = np.arange(rank, N, world)
        dd = domain.DecomposedVerlet(pkg, plan, pkg.cu(pos[mine], dev), pkg.cu(vel[mine], dev), pkg.cu(atoms[mine], dev),
                                     torch.from_numpy(mine).to(dev), model, skin=SKIN)
        assert plan.n_ghost > 0
        dd.overlap = bool(phased)      # interior / boundary phases (here around a synchronous, host-staged exchange)
        if langevin:
            dd.set_langevin_(*LANGEVIN)
        e0 = dd.totals()
        dd.step_(nsteps, DT, rebuild_every)
        e1 = dd.totals()
        gid, x, v, f = dd.gather_state()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), gid=gid.cpu().numpy(), x=x.cpu().numpy(), v=v.cpu().numpy(),
                 f=f.cpu().numpy(), e0=np.array(e0), e1=np.array(e1), builds=dd.md.nbr_stats()["builds"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,rebuild_every,phased,langevin,uniform",
                         [(2, 0, 0, 0, 0), (2, 4, 1, 0, 0), (4, 0, 1, 0, 0), (2, 7, 1, 0, 0), (2, 3, 1, 1, 0), (2, 0, 1, 0, 1),
                          (4, 5, 0, 1, 1)])
def test_decomposed_run_matches_oracle(emdee, oracle, tmp_path, world, rebuild_every, phased, langevin, uniform):
    nsteps = 25
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), nsteps, rebuild_every, phased, langevin, uniform),
             nprocs=world, join=True)
    pos, vel, eps, sigma, L = _global_box(emdee.synthetic, uniform)
    N = pos.shape[0]
    if langevin:       # noise keyed by global atom id: the decomposed run must draw what the undivided run draws
        ref = oracle.verlet_langevin(pos, vel, L, oracle.model(RC, RS), oracle.lj_atoms(eps, sigma), DT, nsteps, *LANGEVIN)
    else:
        ref = oracle.verlet(pos, vel, L, oracle.model(RC, RS), oracle.lj_atoms(eps, sigma), DT, nsteps)
    seen = np.zeros(N, dtype=int)
    for r in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % r))
        gid = d["gid"]
        seen[gid] += 1
        dx = d["x"] - ref["x"][gid]
        assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-9          # same trajectory, up to the periodic image
        assert np.abs(d["v"] - ref["v"][gid]).max() < 1e-8
        assert np.abs(d["f"] - ref["f"][gid]).max() < 1e-6 * np.abs(ref["f"]).max()
        assert d["e0"][0] == pytest.approx(ref["epot"][0], rel=1e-10) and d["e0"][1] == pytest.approx(ref["ekin"][0], rel=1e-10)
        assert d["e1"][0] == pytest.approx(ref["epot"][-1], rel=1e-8) and d["e1"][1] == pytest.approx(ref["ekin"][-1], rel=1e-8)
        assert d["builds"] >= 2
    assert (seen == 1).all()


def test_single_rank_plan_equals_plain_integrator(emdee, oracle):
    """world = 1: no cut dimension, no ghosts; DecomposedVerlet must reproduce VelocityVerlet."""
    E = emdee
    dev = torch.device("cuda", 0)
    pos, vel, eps, sigma, L = _global_box(E.synthetic)
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    model = E.LennardJonesModel(RC, RS)
    plan = E.domain.DomainPlan([L] * 3, RC + SKIN, world=1, rank=0, device=dev)
    dd = E.domain.DecomposedVerlet(E, plan, E.cu(pos, dev), E.cu(vel, dev), E.cu(atoms, dev),
                                   torch.arange(N, device=dev), model, skin=SKIN)
    dd.step_(20, DT)
    ref = oracle.verlet(pos, vel, L, oracle.model(RC, RS), atoms, DT, 20)
    gid, x, v, f = dd.gather_state()
    dx = x.cpu().numpy() - ref["x"][gid.cpu().numpy()]
    assert np.abs(dx - L * np.rint(dx / L)).max() < 1e-9
    ep, ek, _ = dd.totals()
    assert ep == pytest.approx(ref["epot"][-1], rel=1e-8) and ek == pytest.approx(ref["ekin"][-1], rel=1e-8)


def test_engine_with_ghosts_and_open_boundaries(emdee, oracle):
    """The C ABI pieces a decomposed run needs, in one process: a local box that is open along x (ghost
    images supplied by the caller), periodic along y and z; pack + unpack; forces only on owned atoms."""
    E = emdee
    dev = torch.device("cuda", 0)
    pos, vel, eps, sigma, L = _global_box(E.synthetic)
    pos = pos - L * np.floor(pos / L)
    N = pos.shape[0]
    atoms = E.lennard_jones_atoms(eps, sigma)
    h = RC + SKIN
    f0, e0, w0 = oracle.nonbonded_cells(pos, L, oracle.model(RC, RS), atoms)
    own = np.nonzero(pos[:, 0] < 0.5 * L)[0]                              # left half of the box
    lo_side = np.nonzero(pos[:, 0] >= L - h)[0]                            # images at x - L
    hi_side = np.nonzero((pos[:, 0] >= 0.5 * L) & (pos[:, 0] < 0.5 * L + h))[0]
    gpos = np.concatenate([pos[lo_side] - [L, 0, 0], pos[hi_side]])
    gat = np.concatenate([atoms[lo_side], atoms[hi_side]])
    x_all = np.concatenate([pos[own], gpos])
    md = E.VelocityVerlet(E.cu(x_all, dev), E.cu(vel[own], dev), None, E.LennardJonesModel(RC, RS),
                          E.cu(np.concatenate([atoms[own], gat]), dev), skin=SKIN, lo=[-h, 0.0, 0.0],
                          lengths=[0.5 * L + 2 * h, L, L], periodic=[0, 1, 1], n_ghost=gpos.shape[0])
    st = md.state(energies=True, virials=True)
    assert np.abs(st["forces"].cpu().numpy() - f0[own]).max() < 1e-6 * np.abs(f0).max()
    assert np.abs(st["energies"].cpu().numpy() - e0[own]).max() < 1e-6 * np.abs(e0).max()
    assert np.abs(st["virials"].cpu().numpy() - w0[own]).max() < 1e-6 * np.abs(w0).max()
    # interior + boundary phases together cover every brick exactly once
    md.kick_drift_(0.002)                                                  # move the atoms so that stale forces would show
    md.forces_(E.FORCES, phase=1)
    md.forces_(E.FORCES, phase=2)
    f_phases = md.state()["forces"].clone()
    md.forces_(E.FORCES, phase=0)
    assert torch.equal(md.state()["forces"], f_phases)
    assert (f_phases.cpu() - st["forces"].cpu()).abs().max() > 1e-3
    # pack with per-atom image codes, unpack into the ghost slots
    ids = torch.arange(0, 10, dtype=torch.int32, device=dev)
    codes = torch.tensor([0, 1] * 5, dtype=torch.int32, device=dev)
    buf = md.pack_positions(ids, [0.0, 0.0, 0.0, -L, 0.5, 0.0], codes=codes)
    want = md.state()["positions"].cpu().numpy()[:10] + np.where(np.arange(10)[:, None] % 2 == 1, np.array([[-L, 0.5, 0.0]]), 0.0)
    assert np.abs(buf.cpu().numpy() - want).max() < 1e-12
    new_ghosts = E.cu(gpos + 0.01, dev)
    md.unpack_ghosts_(new_ghosts, 0)
    assert np.abs(md.state()["positions"].cpu().numpy()[own.shape[0]:] - (gpos + 0.01)).max() < 1e-12
