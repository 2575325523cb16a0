import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
pos, L = E.synthetic.fcc_positions(136)
N = pos.shape[0]
md = E.VelocityVerlet(E.cu(pos, dev), E.cu(E.synthetic.velocities(N), dev), L, E.LennardJonesModel(2.5, 2.0), E.cu(E.lennard_jones_atoms(1.0, 1.0, N), dev))
del pos
md.step_(500, 0.005)
e0 = sum(md.totals()[:2]); t0 = time.perf_counter()
t_prev, b_prev = t0, md.nbr_stats()["builds"]
for k in range(10):
    md.step_(3000, 0.005)
    ep, ek, _ = md.totals()
    s = md.nbr_stats()
    now = time.perf_counter()
    # (the last column of round 4's file was the running mean since step 500 only: it creeps towards the steady rate for
    # as long as the run lasts; the rate of THIS window and its steps per rebuild say whether anything decays)
    print("step %6d  dE/E %.2e  T %.4f  builds %d  max row %d  capacity %d  %.1f steps/s since step 500, %.1f in this window, %.2f steps per rebuild" %
          (500 + 3000 * (k + 1), (ep + ek) / e0 - 1.0, 2 * ek / (3 * N - 3), s["builds"], s["max_count"], s["capacity"],
           3000 * (k + 1) / (now - t0), 3000 / (now - t_prev), 3000 / max(s["builds"] - b_prev, 1)), flush=True)
    t_prev, b_prev = now, s["builds"]
st = md.state(positions=False)
assert torch.isfinite(st["velocities"]).all() and torch.isfinite(st["forces"]).all()
print("momentum", st["velocities"].sum(dim=0).abs().max().item())
