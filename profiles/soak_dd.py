"""The decomposed box over many rebuilds: the 10^7-atom box in 2 x 2 x 2 in-process domains (all on one GPU) and, beside it, ONE rank of
that grid by the replica rehearsal (mirror), for thousands of steps while the lattice melts -- populations, ghost counts and migrant
counts drift, capacities are outgrown and re-learnt.  Printed per window: energy drift, rebuilds, how many ran in the engines' own
order, how many were redone with exact counts, engines loaded again for room, atoms that changed owner, the rate.
Usage: python profiles/soak_dd.py [cells=136] [windows=6] [steps_per_window=500] [world=8] [lj|mix35|f32]
(mix35: BASELINE configs[4], the binary mixture at rc = 3.5 sigma; f32: configs[3], Float32 storage and pair math)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 6
per = int(sys.argv[3]) if len(sys.argv) > 3 else 500
world = int(sys.argv[4]) if len(sys.argv) > 4 else 8
mode = sys.argv[5] if len(sys.argv) > 5 else "lj"
model = E.LennardJonesModel(3.5, 3.0) if mode == "mix35" else E.LennardJonesModel(2.5, 2.0)
tdt = torch.float32 if mode == "f32" else torch.float64
# the undivided box from the same start (same lattice block, same raw unit-normal velocities per global id): what the ranks' step divides
import numpy as np
pos, gid, lengths = E.synthetic.fcc_block((cells,) * 3, (0, 0, 0), (cells,) * 3)
n = pos.shape[0]
ndt = np.float32 if mode == "f32" else np.float64
if mode == "mix35":
    atoms = E.lennard_jones_atoms(*E.synthetic.mixture_parameters(E.synthetic.mixture_types(gid)))
else:
    atoms = E.lennard_jones_atoms(1.0, 1.0, n)
md = E.VelocityVerlet(E.cu(pos.astype(ndt), dev), E.cu(E.synthetic.raw_normals(gid, n).astype(ndt), dev), float(lengths[0]), model, E.cu(atoms, dev), skin=0.3)
del pos, gid
md.step_(50, 0.005)
e0 = sum(md.totals()[:2])
print("== undivided box, same start: %d atoms" % n, flush=True)
t_undivided = []
for w in range(windows):
    b0 = md.nbr_stats()["builds"]
    t0 = time.perf_counter(); md.step_(per, 0.005); torch.cuda.synchronize(); t = time.perf_counter() - t0
    ep, ek, _ = md.totals()
    t_undivided.append(1e3 * t / per)
    print("step %5d  dE/E %.2e  T %.4f  rebuilds %d  %.3f ms/step  (/%d = %.4f)" % (50 + per * (w + 1), (ep + ek) / e0 - 1.0, 2 * ek / (3 * n - 3),
          md.nbr_stats()["builds"] - b0, t_undivided[-1], world, t_undivided[-1] / world), flush=True)
md.close(); del md
torch.cuda.empty_cache()
for label, kw in (("%d in-process domains" % world, dict(rank=None)), ("one rank of %d, replica rehearsal, lock step" % world, dict(rank=0, mirror=True)), ("one rank of %d, replica rehearsal, in order" % world, dict(rank=0, mirror=True))):
    dd = E.DomainDecomposition.synthetic(cells, world, kw.pop("rank"), dev, model, precision=tdt, mixture=(mode == "mix35"), pkg=E, raw_velocities=True, **kw)
    n = dd.counts(0)["n_global"]
    if "in order" in label:
        dd.set_overlap_(False)
    dd.step_(50, 0.005)
    e0 = sum(dd.totals()[:2])
    print("== %s: %d atoms in the box" % (label, n), flush=True)
    for w in range(windows):
        s0, r0, p0 = dd.stats(), dd.rebuild_stats(), dd.phase_times()
        t0 = time.perf_counter(); dd.step_(per, 0.005); torch.cuda.synchronize(); t = time.perf_counter() - t0
        s1, r1, p1 = dd.stats(), dd.rebuild_stats(), dd.phase_times()
        ep, ek, _ = dd.totals()
        c = dd.counts(0)
        print("step %5d  dE/E %.2e  T %.4f  rebuilds %d (engine order %d, count-free %d, redone %d, engines regrown %d)  migrated %d  domain 0: %d owned + %d ghosts  %.3f ms/step" %
              (50 + per * (w + 1), (ep + ek) / e0 - 1.0, 2 * ek / (3 * n - 3), s1["rebuilds"] - s0["rebuilds"],
               p1["rebuilds_in_engine_order"] - p0["rebuilds_in_engine_order"], r1["count_free"] - r0["count_free"], r1["redone"] - r0["redone"],
               p1["engines_regrown"] - p0["engines_regrown"], s1["migrated"] - s0["migrated"], c["n_owned"], c["n_ghost"], 1e3 * t / per) +
              ("   undivided / %d / this = %.3f" % (world, t_undivided[w] / world / (1e3 * t / per)) if "rank" in label else ""), flush=True)
    dd.close()
    torch.cuda.empty_cache()
