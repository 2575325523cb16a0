"""How much later could the neighbour list be rebuilt with a LOCAL displacement criterion?

The engine rebuilds when any atom has moved skin/2 from where it was when the list was built (2 d_max > skin: the two
fastest atoms of the box might be neighbours approaching head-on).  A pair that is not in the list can only come inside
the cutoff if its two atoms TOGETHER have moved more than the skin, and both sit within r_list of each other, i.e. in
one brick's tile.  This script measures, on the benchmark box, after how many steps the two criteria fire:

    global:  2 max_i d_i > skin
    local:   max over bricks of (largest + second largest d among the atoms of the brick and its 26 neighbours) > skin

(the 27-brick neighbourhood is a superset of the brick's tile, so the local figure is conservative).  No list is used:
the positions come from the engine, which rebuilds on its own as usual.

    python profiles/rebuild_criterion_study.py [cells=136] [trials=6]
"""
import sys

import torch

sys.path.insert(0, ".")
from __graft_entry__ import load_package  # noqa: E402

E = load_package()
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
RC, SKIN, DT = 2.5, 0.3, 0.005
pos, L = E.synthetic.fcc_positions(cells)
N = pos.shape[0]
md = E.VelocityVerlet(E.cu(pos, dev), E.cu(E.synthetic.velocities(N), dev), L, E.LennardJonesModel(RC, RC - 0.5),
                      E.cu(E.lennard_jones_atoms(1.0, 1.0, N), dev), skin=SKIN)
del pos
M = int(L // (RC + SKIN))                      # cells of side >= r_list, bricks of 4 x 2 x 2 cells (csrc/nbsys.hpp)
nb = [(M + 3) // 4, (M + 1) // 2, (M + 1) // 2]
side = L / M
print("N = %d, L = %.3f, %d^3 cells, bricks %d x %d x %d" % (N, L, M, *nb))


def brick_of(x):
    c = torch.clamp((x / side).floor().long(), 0, M - 1)
    return (c[:, 0] // 4) + nb[0] * ((c[:, 1] // 2) + nb[1] * (c[:, 2] // 2))


def local_bound(d, brick):
    nbr = nb[0] * nb[1] * nb[2]
    t1 = torch.zeros(nbr, device=dev, dtype=d.dtype).scatter_reduce(0, brick, d, "amax", include_self=True)
    rest = torch.where(d >= t1[brick], torch.zeros_like(d), d)       # (ties: harmless for a bound study)
    t2 = torch.zeros(nbr, device=dev, dtype=d.dtype).scatter_reduce(0, brick, rest, "amax", include_self=True)
    g1, g2 = t1.view(nb[2], nb[1], nb[0]), t2.view(nb[2], nb[1], nb[0])
    vals = []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                vals.append(torch.roll(g1, (dz, dy, dx), (0, 1, 2)))
                vals.append(torch.roll(g2, (dz, dy, dx), (0, 1, 2)))
    top = torch.stack(vals, 0).topk(2, dim=0).values
    return (top[0] + top[1]).max().item()


for phase, nsteps in (("melting lattice (steps 0-40)", 0), ("after 300 steps", 300)):
    if nsteps:
        md.step_(nsteps, DT)
    g_steps, l_steps = [], []
    for _ in range(trials):
        x0 = md.state()["positions"].clone()
        brick = brick_of(torch.remainder(x0, L))
        g_fire = l_fire = None
        for k in range(1, 40):
            md.step_(1, DT)
            dx = md.state()["positions"] - x0
            dx -= L * torch.round(dx / L)
            d = dx.norm(dim=1)
            if g_fire is None and 2.0 * d.max().item() > SKIN:
                g_fire = k
            if l_fire is None and local_bound(d, brick) > SKIN:
                l_fire = k
            if g_fire is not None and l_fire is not None:
                break
        g_steps.append(g_fire)
        l_steps.append(l_fire)
    print("%s: global criterion fires after %s steps, local after %s  (means %.1f / %.1f -> %.0f %% fewer rebuilds)" %
          (phase, g_steps, l_steps, sum(g_steps) / trials, sum(l_steps) / trials,
           100.0 * (1.0 - (sum(g_steps) / trials) / (sum(l_steps) / trials))))
