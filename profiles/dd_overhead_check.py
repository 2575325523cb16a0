"""Glue overhead of the decomposed driver on one GPU: VelocityVerlet.step_ vs DecomposedVerlet.step_ (world = 1)."""
import sys, time
import torch
sys.path.insert(0, ".")
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 63
pos, L = E.synthetic.fcc_positions(cells)
N = pos.shape[0]
vel = E.synthetic.velocities(N)
atoms = E.lennard_jones_atoms(1.0, 1.0, N)
model = E.LennardJonesModel(2.5, 2.0)
md = E.VelocityVerlet(E.cu(pos, dev), E.cu(vel, dev), L, model, E.cu(atoms, dev))
plan = E.domain.DomainPlan([L] * 3, 2.8, world=1, rank=0, device=dev)
dd = E.domain.DecomposedVerlet(E, plan, E.cu(pos, dev), E.cu(vel, dev), E.cu(atoms, dev), torch.arange(N, device=dev), model)
for name, run in (("VelocityVerlet.step_", lambda k: md.step_(k, 0.005)), ("DecomposedVerlet.step_", lambda k: dd.step_(k, 0.005))):
    run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(50); torch.cuda.synchronize()
    print("%-24s %d atoms: %.3f ms/step" % (name, N, 1e3 * (time.perf_counter() - t0) / 50))
