"""configs[4] over thousands of steps: the 10^7-atom binary mixture at rc = 3.5 sigma on the typed kernels (2 x 2 x 2 bricks, (cell, species,
quarter) sort), fp64 or fp32: energy drift, temperature, rebuilds, longest row against its capacity, rate per window.
Usage: python profiles/soak_mixture.py [f64|f32] [windows=5] [steps_per_window=600]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f64"
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 5
per = int(sys.argv[3]) if len(sys.argv) > 3 else 600
ndt = np.float64 if prec == "f64" else np.float32
syn = E.synthetic
pos, L = syn.fcc_positions(136)
N = pos.shape[0]
eps, sigma = syn.mixture_parameters(syn.mixture_types(N))
md = E.VelocityVerlet(E.cu(pos.astype(ndt), dev), E.cu(syn.velocities(N).astype(ndt), dev), L, E.LennardJonesModel(3.5, 3.0), E.cu(E.lennard_jones_atoms(eps, sigma), dev), skin=0.3)
del pos
md.step_(200, 0.005)
e0 = sum(md.totals()[:2]); b_prev = md.nbr_stats()["builds"]
for k in range(windows):
    t0 = time.perf_counter(); md.step_(per, 0.005); torch.cuda.synchronize(); t = time.perf_counter() - t0
    ep, ek, _ = md.totals(); s = md.nbr_stats()
    print("%s step %5d  dE/E %.2e  T %.4f  rebuilds %d (%.2f steps each)  max row %d  capacity %d  %.1f steps/s" %
          (prec, 200 + per * (k + 1), (ep + ek) / e0 - 1.0, 2 * ek / (3 * N - 3), s["builds"] - b_prev, per / max(s["builds"] - b_prev, 1), s["max_count"], s["capacity"], per / t), flush=True)
    b_prev = s["builds"]
st = md.state(positions=False, forces=False)
print("momentum per atom", (st["velocities"].sum(dim=0).abs().max().item()) / N)
