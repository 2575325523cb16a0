"""What the decomposition machinery costs when there is nothing to decompose: emdee_dd_* with ONE domain (no peers, no
ghosts: every rebuild still goes caller order -> ownership -> partition -> engine load) against the plain integrator on the
same box.  Usage: python profiles/dd_one_domain_overhead.py [cells] [dd|plain|both]   (one of the two alone: for a kernel trace)"""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 68          # 68^3 x 4 = 1,257,728 atoms: a rank of the 8-GPU 10^7-atom run
model = E.LennardJonesModel(2.5, 2.0)
skin = float(os.environ.get("SKIN", "0.3"))
dd = E.DomainDecomposition.synthetic(cells, 1, None, dev, model, skin=skin, pkg=E)
pos, L = E.synthetic.fcc_positions(cells)
N = pos.shape[0]
vel = E.synthetic.velocities(N)
atoms = E.lennard_jones_atoms(1.0, 1.0, N)
md = E.VelocityVerlet(E.cu(pos, dev), E.cu(vel, dev), L, model, E.cu(atoms, dev), skin=skin)
only = sys.argv[2] if len(sys.argv) > 2 else "both"
for name, obj in (("dd, one domain", dd), ("plain integrator", md)):
    if only != "both" and not name.startswith(only):
        continue
    obj.step_(20, 0.005); torch.cuda.synchronize()
    b0 = (obj.engine(0) if name.startswith("dd") else obj).nbr_stats()["builds"]
    t0 = time.perf_counter(); obj.step_(100, 0.005); torch.cuda.synchronize(); t = time.perf_counter() - t0
    b1 = (obj.engine(0) if name.startswith("dd") else obj).nbr_stats()["builds"]
    print("%-18s %d atoms: %.3f ms/step, %d rebuilds in 100 steps" % (name, N, 10 * t, b1 - b0))
