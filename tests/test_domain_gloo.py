"""Multi-rank host logic on CPU (gloo): brick ownership, migration, ghost selection with periodic
shifts and the per-step halo exchange of emdee.jl_amd/domain.py.  DomainPlan never computes a force;
here each rank evaluates its owned atoms from its local (owned + ghost) configuration with a small numpy
pair sum and the result must equal the CPU oracle on the undivided periodic box."""
import os
import socket
import sys

import numpy as np
import pytest

from .conftest import ROOT

torch = pytest.importorskip("torch")
import torch.distributed as dist            # noqa: E402
import torch.multiprocessing as mp          # noqa: E402

RC, RS, SKIN = 2.5, 2.0, 0.3
NCELL = 8                                    # global fcc box: 8^3 x 4 = 2048 atoms, L = 13.68


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pair_sum(x_own, x_all, skip, lengths, periodic, hs, te, rc2, rs2, idl2):
    """Numpy restatement of src/lennard_jones.jl:25-42 + src/nonbonded.jl:136-145 for owned atoms against a
    local cluster; minimum image only along the listed periodic dimensions."""
    d = x_own[:, None, :] - x_all[None, :, :]
    for k in range(3):
        if periodic[k]:
            d[:, :, k] -= lengths[k] * np.rint(d[:, :, k] / lengths[k])
    r2 = np.einsum("ijk,ijk->ij", d, d)
    keep = r2 < rc2
    keep[np.arange(x_own.shape[0]), skip] = False
    r2s = np.where(keep, r2, 1.0)
    sg = hs[skip][:, None] + hs[None, :]
    s2 = sg * sg / r2s
    s6 = s2 ** 3
    e4s6 = te[skip][:, None] * te[None, :] * s6
    E = e4s6 * (s6 - 1.0)
    W = 6.0 * e4s6 * (2.0 * s6 - 1.0)
    x = np.clip((r2s - rs2) * idl2, 0.0, None)
    g = 1.0 + x ** 3 * (15.0 * x - 6.0 * x * x - 10.0)
    mgr = 60.0 * x * x * (1.0 - x) ** 2 * idl2 * r2s
    Wg = np.where(keep, W * g + E * mgr, 0.0)
    Eg = np.where(keep, E * g, 0.0)
    return np.einsum("ij,ijk->ik", Wg / r2s, d), 0.5 * Eg.sum(axis=1)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util
        def load(name):
            spec = importlib.util.spec_from_file_location("emdee_" + name, os.path.join(ROOT, "emdee.jl_amd", name + ".py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
        syn, domain = load("synthetic"), load("domain")       # numpy/torch only: no HIP library needed
        from oracle import oracle as orc

        grid = domain.rank_grid(world)
        coords = (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))
        bn = [NCELL // g for g in grid]
        pos, gid, lengths = syn.fcc_block((NCELL,) * 3, [c * b for c, b in zip(coords, bn)], bn)
        L = float(lengths[0])
        # a decomposition-independent kick that pushes atoms across brick faces and the periodic boundary
        kick = 1.2 * (syn.uniform(0xBEEF, (3 * gid)[:, None] + np.arange(3)[None, :]) - 0.5)
        pos = pos + kick
        vel = syn.raw_normals(gid, 4 * NCELL ** 3)
        types = syn.mixture_types(gid)
        eps, sigma = syn.mixture_parameters(types)
        atoms = np.stack([(0.5 * sigma).astype(np.float32), (2.0 * np.sqrt(eps)).astype(np.float32)], axis=1)

        plan = domain.DomainPlan(lengths, RC + SKIN, world=world, rank=rank, grid=grid)
        x, v, a, g = plan.migrate(torch.from_numpy(pos), torch.from_numpy(vel), torch.from_numpy(atoms),
                                  torch.from_numpy(gid))
        # ownership: inside the brick, every atom exactly once over all ranks
        for d in range(3):
            assert (x[:, d] >= plan.lo[d]).all() and (x[:, d] < plan.hi[d]).all()
        all_g = [None] * world
        dist.all_gather_object(all_g, g.tolist())
        flat = sorted(i for l in all_g for i in l)
        assert flat == list(range(4 * NCELL ** 3))

        gx, ga = plan.build_ghosts(x, a)
        assert gx.shape[0] == plan.n_ghost == sum(plan.recv_counts)
        ggid = plan._exchange_rows(g[plan.send_ids.long()].contiguous())
        # ghosts lie in the halo shell of the local box, outside the brick along at least one cut dimension
        for d in range(3):
            if plan.cut[d]:
                assert (gx[:, d] >= plan.local_lo[d] - 1e-12).all()
                assert (gx[:, d] < plan.local_lo[d] + plan.local_len[d] + 1e-12).all()

        # the undivided reference: every rank rebuilds the whole kicked box and asks the oracle
        gpos, ggl, _ = syn.fcc_block((NCELL,) * 3, (0, 0, 0), (NCELL,) * 3)
        gpos = gpos + 1.2 * (syn.uniform(0xBEEF, (3 * ggl)[:, None] + np.arange(3)[None, :]) - 0.5)
        order = np.argsort(ggl)
        gpos = gpos[order]
        geps, gsig = syn.mixture_parameters(syn.mixture_types(np.arange(4 * NCELL ** 3)))
        f0, e0, _ = orc.nonbonded_cells(gpos, L, orc.model(RC, RS), orc.lj_atoms(geps, gsig))

        def local_forces(xo, xg):
            x_all = np.concatenate([xo, xg])
            a_all = np.concatenate([a.numpy(), ga.numpy()]).astype(np.float64)
            return _pair_sum(xo, x_all, np.arange(xo.shape[0]), plan.local_len, plan.periodic, a_all[:, 0], a_all[:, 1],
                             RC * RC, RS * RS, 1.0 / (RC * RC - RS * RS))

        f, e = local_forces(x.numpy(), gx.numpy())
        idx = g.numpy()
        assert np.abs(f - f0[idx]).max() < 1e-9 * np.abs(f0).max()
        assert np.abs(e - e0[idx]).max() < 1e-9 * np.abs(e0).max()

        # per-step halo exchange: move owned atoms a little, ghosts must follow their owners (+ shift)
        x2 = x + 0.05 * torch.from_numpy(syn.uniform(0xF00D, (3 * idx)[:, None] + np.arange(3)[None, :]) - 0.5)
        gx2 = plan.exchange(plan.pack_torch(x2))
        gpos2 = gpos + 0.05 * (syn.uniform(0xF00D, (3 * np.arange(gpos.shape[0]))[:, None] + np.arange(3)[None, :]) - 0.5)
        delta = gx2.numpy() - gpos2[ggid.numpy()]
        assert np.abs(delta - L * np.rint(delta / L)).max() < 1e-12     # same atom, some periodic image
        f0b, e0b, _ = orc.nonbonded_cells(gpos2, L, orc.model(RC, RS), orc.lj_atoms(geps, gsig))
        f2, e2 = local_forces(x2.numpy(), gx2.numpy())
        assert np.abs(f2 - f0b[idx]).max() < 1e-9 * np.abs(f0b).max()

        # migration is idempotent and conserves atoms after the move
        x3, v3, a3, g3 = plan.migrate(x2, v, a, g)
        cnt = torch.tensor([x3.shape[0]])
        dist.all_reduce(cnt)
        assert cnt.item() == 4 * NCELL ** 3
        with open(os.path.join(out_dir, "ok_%d" % rank), "w") as fh:
            fh.write("%d %d\n" % (x.shape[0], plan.n_ghost))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_decomposition_matches_undivided_box(world, tmp_path, oracle):
    if world > (os.cpu_count() or 1):
        pytest.skip("not enough cores")
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    owned = 0
    for r in range(world):
        n, ng = (int(t) for t in open(tmp_path / ("ok_%d" % r)).read().split())
        owned += n
        assert ng > 0
    assert owned == 4 * NCELL ** 3


def test_rank_grid_and_plan_geometry():
    import importlib.util
    spec = importlib.util.spec_from_file_location("emdee_domain", os.path.join(ROOT, "emdee.jl_amd", "domain.py"))
    domain = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(domain)
    assert domain.rank_grid(1) == (1, 1, 1) and domain.rank_grid(2) == (2, 1, 1)
    assert domain.rank_grid(4) == (2, 2, 1) and domain.rank_grid(8) == (2, 2, 2)
    p = domain.DomainPlan([20.0, 20.0, 20.0], 2.8, world=8, rank=7)
    assert p.coords == (1, 1, 1) and p.lo == [10.0, 10.0, 10.0] and len(p.dirs) == 26
    assert p.periodic == [0, 0, 0] and p.local_lo == [7.2, 7.2, 7.2]
    k = p.dirs.index((1, 0, 0))
    assert p.dir_rank[k] == 6 and p.dir_shift[k] == [-20.0, 0.0, 0.0]     # wraps around: image shifted by -L
    p2 = domain.DomainPlan([20.0, 20.0, 20.0], 2.8, world=2, rank=0)
    assert p2.periodic == [0, 1, 1] and len(p2.dirs) == 2 and p2.local_len == [15.6, 20.0, 20.0]
    assert p2.dir_rank == [1, 1] and sorted(s[0] for s in p2.dir_shift) == [0.0, 20.0]
    with pytest.raises(ValueError):
        domain.DomainPlan([5.0, 20.0, 20.0], 2.8, world=2, rank=0)          # halo wider than the brick
