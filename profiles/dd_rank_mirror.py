"""What ONE rank of a strong-scaling run costs per step, measured on one GPU through the production path: the replica
rehearsal (DomainDecomposition(..., rank=0, mirror=True)).  The rank holds brick (0,0,0) of the 2x2x2 grid of the
`cells`^3-cell fcc box and every peer is its own periodic image, so all seven messages of a step (and the padded migrant and
ghost messages of a rebuild) are packed, "sent" (a device copy of the rank's own send buffer), received and unpacked with
the production sizes, on the production streams; the physics is the periodic box of one brick.  Against it: the plain
integrator on that brick (no ghosts, no messages) -- the ideal an 8-rank run divides the undivided step by.
Usage: python profiles/dd_rank_mirror.py [cells=136] [steps=200] [world=8]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
model = E.LennardJonesModel(2.5, 2.0)
dd = E.DomainDecomposition.synthetic(cells, world, 0, dev, model, pkg=E, mirror=True, raw_velocities=True)
c = dd.counts(0)
print("rank box: %d owned + %d ghost atoms (%.1f %% ghosts), grid %s" % (c["n_owned"], c["n_ghost"], 100.0 * c["n_ghost"] / (c["n_owned"] + c["n_ghost"]), dd.grid))
for form, overlap in (("lock step (interior || halo)", True), ("in order", False)):
    dd.set_overlap_(overlap)
    dd.step_(30, 0.005); torch.cuda.synchronize()
    eng = dd.engine(0); eng.profile_(True)
    s0, p0 = dd.stats(), dd.phase_times()
    t0 = time.perf_counter(); dd.step_(steps, 0.005); torch.cuda.synchronize(); t = time.perf_counter() - t0
    s1, p1 = dd.stats(), dd.phase_times()
    rb = s1["rebuilds"] - s0["rebuilds"]
    k = {name: eng.kernel_time(i) for name, i in (("interior", 5), ("boundary", 6), ("halo", 7), ("sort+list", 2))}
    eng.profile_(False)
    print("%-30s %.4f ms/step, %d rebuilds (%d cancelled steps, %d migrated), read-backs per rebuild %.2f, rebuild wall %.3f ms each | device: %s"
          % (form, 1e3 * t / steps, rb, s1["cancelled_steps"] - s0["cancelled_steps"], s1["migrated"] - s0["migrated"],
             1.0 + (p1["rebuild_readbacks"] - p0["rebuild_readbacks"]) / max(rb, 1), (p1["rebuild_ms"] - p0["rebuild_ms"]) / max(rb, 1),
             ", ".join("%s %.4f ms x %d" % (n, ms / max(cnt, 1), cnt) for n, (ms, cnt) in k.items())))
dd.close()
# the brick alone, periodic: the ideal share of a rank
pos, gid, lengths = E.synthetic.fcc_block(tuple(cells // g for g in dd.grid), (0, 0, 0), tuple(cells // g for g in dd.grid))
N = pos.shape[0]
vel = E.synthetic.raw_normals(gid, N)
md = E.VelocityVerlet(E.cu(pos, dev), E.cu(vel, dev), None, model, E.cu(E.lennard_jones_atoms(1.0, 1.0, N), dev), lengths=[float(v) for v in lengths])
md.step_(30, 0.005); torch.cuda.synchronize()
t0 = time.perf_counter(); md.step_(steps, 0.005); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("%-30s %.4f ms/step (%d atoms, periodic brick, no ghosts)" % ("plain integrator", 1e3 * t / steps, N))
