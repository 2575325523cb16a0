"""What ONE rank of an 8-rank strong-scaling run of the 10^7-atom box computes per step, timed alone on one GPU.

The rank's local box is rebuilt here without its neighbours: the atoms of the 136^3-cell box that fall into brick
(0,0,0) of the 2x2x2 grid are owned, the atoms within cutoff + skin of its faces (periodic images included) are
ghosts.  Ghosts stay frozen (no peer integrates them), so this is a timing proxy, not a simulation: the same kernels,
grids, tile populations and launch sequence as the rank would run -- pack, interior bricks, unpack, boundary bricks --
minus the RCCL exchange itself.  Usage: python profiles/dd_rank_proxy.py [cells=136] [steps=40]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from __graft_entry__ import load_package  # noqa: E402

E = load_package()
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rc, rs, skin, dt = 2.5, 2.0, 0.3, 0.005
h = rc + skin
pos, L = E.synthetic.fcc_positions(cells)
pos -= L * np.floor(pos / L)
N = pos.shape[0]
vel = E.synthetic.velocities(N)
w = 0.5 * L
own = np.all(pos < w, axis=1)
# ghost images: shift every atom by -L, 0 along each dimension and keep what lies in the shell [-h, w + h)^3 \ brick
ghosts = []
for sx in (0.0, -L):
    for sy in (0.0, -L):
        for sz in (0.0, -L):
            q = pos + np.array([sx, sy, sz])
            inside = np.all((q >= -h) & (q < w + h), axis=1)
            brick = np.all((q >= 0.0) & (q < w), axis=1)
            ghosts.append(q[inside & ~brick])
gpos = np.concatenate(ghosts)
n_own, n_ghost = int(own.sum()), gpos.shape[0]
print("rank box: %d owned + %d ghost atoms (%.1f %%), local box %.1f sigma" % (n_own, n_ghost, 100.0 * n_ghost / n_own, w + 2 * h))
x_all = np.concatenate([pos[own], gpos])
atoms = E.lennard_jones_atoms(1.0, 1.0, x_all.shape[0])
md = E.VelocityVerlet(E.cu(x_all, dev), E.cu(vel[own], dev), None, E.LennardJonesModel(rc, rs), E.cu(atoms, dev), skin=skin,
                      lo=[-h] * 3, lengths=[w + 2 * h] * 3, periodic=[0, 0, 0], n_ghost=n_ghost)
ids = torch.arange(0, n_ghost, dtype=torch.int32, device=dev) % n_own          # as many positions packed as a rank sends
gx = E.cu(gpos, dev)


def run(nsteps, split):
    """nsteps decomposed steps: pack -> [interior] -> unpack -> boundary (split) or pack -> unpack -> all bricks."""
    rebuilds = 0
    md.kick_drift_(dt, 0.5)
    for _ in range(nsteps):
        if md.needs_rebuild():                      # (one read-back per step: the library batches these, the proxy does not)
            md.rebuild_()
            md.forces_(E.FORCES, 0)
            md.kick_drift_(dt, 1.0)
            rebuilds += 1
            continue
        md.pack_positions(ids, [0.0, 0.0, 0.0])
        if split:
            md.fused_step_(dt, 1.0, phase=1)
            md.unpack_ghosts_(gx, 0)
            md.fused_step_(dt, 1.0, phase=2)
        else:
            md.unpack_ghosts_(gx, 0)
            md.fused_step_(dt, 1.0, phase=0)
    md.forces_(E.FORCES, 0)
    md.kick_(dt)
    return rebuilds


for split in (True, False):
    run(10, split)
    torch.cuda.synchronize()
    md.profile_(True)
    t0 = time.perf_counter()
    rb = run(steps, split)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    f_ms, f_n = md.kernel_time("lj_force_nbr_fused_step")
    r_ms, r_n = md.kernel_time("rebuild")
    print("%s: %.3f ms/step wall (%d rebuilds in %d steps); fused launches %.3f ms each x %d; rebuild (sort + list) %.3f ms each x %d"
          % ("interior + boundary launches" if split else "one launch per step        ", ms, rb, steps,
             f_ms / max(f_n, 1), f_n, r_ms / max(r_n, 1), r_n))
    md.profile_(False)
