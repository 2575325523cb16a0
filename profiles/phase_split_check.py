"""Cost of splitting lj_force_nbr into interior + boundary launches (open box along x, one GPU)."""
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
md = E.VelocityVerlet(E.cu(pos, dev), E.cu(vel, dev), None, E.LennardJonesModel(2.5, 2.0), E.cu(atoms, dev),
                      lo=[-3.0, 0.0, 0.0], lengths=[L + 6.0, L, L], periodic=[0, 1, 1])
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
print("phase 0        : %.3f ms" % timeit(lambda: md.forces_(1, 0)))
print("phase 1        : %.3f ms" % timeit(lambda: md.forces_(1, 1)))
print("phase 2        : %.3f ms" % timeit(lambda: md.forces_(1, 2)))
print("phase 1 + 2    : %.3f ms" % timeit(lambda: (md.forces_(1, 1), md.forces_(1, 2))))
