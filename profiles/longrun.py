import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
for prec in (np.float64, np.float32):
    pos, L = E.synthetic.fcc_positions(63)
    N = pos.shape[0]
    md = E.VelocityVerlet(E.cu(pos.astype(prec), dev), E.cu(E.synthetic.velocities(N).astype(prec), dev), L, E.LennardJonesModel(2.5, 2.0),
                          E.cu(E.lennard_jones_atoms(1.0, 1.0, N), dev))
    md.step_(500, 0.005)                       # melt + equilibrate
    e0 = sum(md.totals()[:2]); t0 = time.perf_counter()
    out = []
    for k in range(10):
        md.step_(2000, 0.005)
        ep, ek, _ = md.totals()
        out.append("%.2e" % ((ep + ek) / e0 - 1.0))
    torch.cuda.synchronize()
    o = md.observables()
    print(prec.__name__, "20000 steps in %.1f s;" % (time.perf_counter() - t0), "dE/E:", " ".join(out), "| T %.4f P %.4f builds %d cap %d" %
          (o["temperature"], o["pressure"], md.nbr_stats()["builds"], md.nbr_stats()["capacity"]), flush=True)
    st = md.state()
    assert torch.isfinite(st["positions"]).all() and torch.isfinite(st["velocities"]).all()
    assert (st["velocities"].double().sum(dim=0).abs().max().item()) < 1e-3 * np.sqrt(N)
