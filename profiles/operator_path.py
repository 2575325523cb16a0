"""Per-call cost of the reference-shaped operator compute_nonbonded_(forces, energies, virials, positions, L, tiles,
model, atoms, Val(mask)) at 10^7 atoms, driven the way a user of the reference drives it: positions owned by the
caller and moved by the caller's own integrator (here torch expressions) between calls."""
import sys, time
import torch
sys.path.insert(0, ".")
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 136
pos, L = E.synthetic.fcc_positions(cells)
N = pos.shape[0]
x = E.cu(pos, dev); v = E.cu(E.synthetic.velocities(N), dev); a = E.cu(E.lennard_jones_atoms(1.0, 1.0, N), dev)
f = torch.zeros_like(x); e = torch.zeros(N, dtype=x.dtype, device=dev); w = torch.zeros_like(e)
model = E.LennardJonesModel(2.5, 2.0)
tiles = E.nonbonded_computation_tiles(N)
dt = 0.005
def step(mask):
    v.add_(f, alpha=0.5 * dt); x.add_(v, alpha=dt)
    E.compute_nonbonded_(f, e, w, x, L, tiles, model, a, mask)
    v.add_(f, alpha=0.5 * dt)
E.compute_nonbonded_(f, e, w, x, L, tiles, model, a, 7)
for mask in (1, 7):
    for _ in range(10): step(mask)
    torch.cuda.synchronize(); t0 = time.perf_counter(); b0 = tiles.stats()["builds"]
    for _ in range(50): step(mask)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 50
    # the operator alone (positions unchanged -> no rebuild, list reused)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): E.compute_nonbonded_(f, e, w, x, L, tiles, model, a, mask)
    torch.cuda.synchronize(); op = (time.perf_counter() - t0) / 20
    print("mask %d: %.3f ms per velocity-Verlet step driven from the host side (%d rebuilds in 50 steps), operator alone %.3f ms per call"
          % (mask, 1e3 * dtm, tiles.stats()["builds"] - b0, 1e3 * op), flush=True)
