#!/usr/bin/env python3
"""A thermostatted Lennard-Jones fluid on one MI355X through the EmDee-shaped API: equilibrate with the Langevin
thermostat, switch it off, run NVE, write an XYZ frame and a checkpoint, and restart from the checkpoint.

    python examples/lj_fluid.py [cells]        # cells^3 x 4 atoms, default 20 (32,000 atoms)

Needs the built library (python -c "import __graft_entry__ as g; g.build()") and a gfx950 device."""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

E = load_package()
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 20

pos, L = E.synthetic.fcc_positions(cells)                       # jittered fcc lattice at rho* = 0.8
N = pos.shape[0]
vel = E.synthetic.velocities(N)                                  # Maxwell-Boltzmann at T* = 1, zero total momentum
model = E.LennardJonesModel(2.5, 2.0)                            # cutoff, switch   (src/lennard_jones.jl:6-11)
atoms = E.lennard_jones_atoms(1.0, 1.0, N)                       # LennardJonesAtom(eps, sigma) for every atom

md = E.VelocityVerlet(E.cu(pos, dev), E.cu(vel, dev), L, model, E.cu(atoms, dev), skin=0.3)
dt = 0.005

md.set_langevin_(gamma=2.0, temperature=0.9, seed=2026)          # NVT: v = c1 v + c2 sqrt(T/m) xi every step
md.step_(2000, dt)
obs = md.observables()
print("after 2000 thermostatted steps: T* = %.4f  P* = %.4f  U/N = %.4f" % (obs["temperature"], obs["pressure"], obs["potential"] / N))

md.set_langevin_(0.0, 0.0)                                       # NVE from here
e0 = sum(md.totals()[:2])
md.step_(2000, dt)
e1 = sum(md.totals()[:2])
print("2000 NVE steps: relative energy change %.2e, %d list rebuilds so far" % (e1 / e0 - 1.0, md.nbr_stats()["builds"]))

st = md.state()
with tempfile.TemporaryDirectory() as tmp:
    E.ingest.write_xyz(os.path.join(tmp, "frame.xyz"), ["Ar"] * N, st["positions"].cpu().numpy(), comment="L = %.6f" % L)
    E.ingest.save_checkpoint(os.path.join(tmp, "run.npz"), st["positions"], st["velocities"], 4000, L)
    x, v, step, box = E.ingest.load_checkpoint(os.path.join(tmp, "run.npz"))
again = E.VelocityVerlet(E.cu(x, dev), E.cu(v, dev), box, model, E.cu(atoms, dev), skin=0.3)
again.step_(100, dt)
md.step_(100, dt)
dx = (again.state()["positions"] - md.state()["positions"]).abs().max().item()
print("restart from the checkpoint at step %d: positions after 100 more steps differ by %.1e" % (step, dx))

# the reference-shaped operator on caller-owned arrays (src/nonbonded.jl:109-120)
f = torch.zeros((N, 3), dtype=torch.float64, device=dev)
e = torch.zeros(N, dtype=torch.float64, device=dev)
w = torch.zeros(N, dtype=torch.float64, device=dev)
tiles = E.nonbonded_computation_tiles(N)
E.compute_nonbonded_(f, e, w, md.state()["positions"], L, tiles, model, E.cu(atoms, dev), E.Val(E.FORCES | E.ENERGIES | E.VIRIALS))
print("compute_nonbonded_: sum E = %.6f (integrator says %.6f), |sum F| = %.1e" % (e.sum().item(), md.totals()[0], f.sum(dim=0).abs().max().item()))
