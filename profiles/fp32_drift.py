"""Energy conservation of the fp32 configuration (BASELINE configs[3]: fp32 storage and pair math, fp64 reductions) over a few
hundred steps: relative error of the total energy after 40 and after `steps` steps.  EMDEE_F32_ABS=1 = absolute fp32 records
(rounds 1-4); default = cell-relative records (round 5, csrc/kernels.hpp RelGrid).
Usage: python profiles/fp32_drift.py [cells=136] [steps=400]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
pos, L = E.synthetic.fcc_positions(cells)
N = pos.shape[0]
vel = E.synthetic.velocities(N)
atoms = E.lennard_jones_atoms(1.0, 1.0, N)
f64 = bool(os.environ.get("F64"))             # the same box in fp64, from the same fp32-representable start: what the integrator itself does to the energy
ndt = np.float64 if f64 else np.float32
md = E.VelocityVerlet(E.cu(pos.astype(np.float32).astype(ndt), dev), E.cu(vel.astype(np.float32).astype(ndt), dev), float(np.float32(L)) if f64 else L,
                      E.LennardJonesModel(2.5, 2.0), E.cu(atoms, dev), skin=0.3)
del pos, vel
ep0, ek0, _ = md.totals()
md.step_(40, 0.005)
ep1, ek1, _ = md.totals()
torch.cuda.synchronize(); t0 = time.perf_counter()
md.step_(steps - 40, 0.005)
torch.cuda.synchronize(); t = time.perf_counter() - t0
ep2, ek2, _ = md.totals()
e0, e1, e2 = ep0 + ek0, ep1 + ek1, ep2 + ek2
print("  per atom: potential %.9f %.9f %.9f   kinetic %.9f %.9f %.9f" % (ep0 / N, ep1 / N, ep2 / N, ek0 / N, ek1 / N, ek2 / N))
st = md.state(positions=False, forces=False)
p = st["velocities"].double().sum(dim=0).abs().max().item()
print("%s %s records, %d atoms (L = %.1f): E0/N %.9f, relative energy error after 40 steps %.2e, after %d steps %.2e; %d rebuilds; %.1f steps/s; |sum v| %.2e"
      % ("fp64" if f64 else "fp32", "absolute" if (os.environ.get("EMDEE_F32_ABS") or f64) else "cell-relative", N, L, e0 / N, (e1 - e0) / abs(e0), steps, (e2 - e0) / abs(e0),
         md.nbr_stats()["builds"], (steps - 40) / t, p))
