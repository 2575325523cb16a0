"""Kernel-level look at the operator alone (positions unchanged between calls): python profiles/operator_trace.py [mask=7] under rocprofv3 --kernel-trace --stats"""
import sys, torch
sys.path.insert(0, ".")
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 7
pos, L = E.synthetic.fcc_positions(136)
N = pos.shape[0]
x = E.cu(pos, dev); a = E.cu(E.lennard_jones_atoms(1.0, 1.0, N), dev)
f = torch.zeros_like(x); e = torch.zeros(N, dtype=x.dtype, device=dev); w = torch.zeros_like(e)
model = E.LennardJonesModel(2.5, 2.0)
tiles = E.nonbonded_computation_tiles(N)
for _ in range(25):
    E.compute_nonbonded_(f, e, w, x, L, tiles, model, a, mask)
torch.cuda.synchronize()
