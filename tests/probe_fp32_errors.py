"""Max absolute deviation of the fp32 kernels on the reference's fixture (test/data/lj_sample.xyz, L = 10, rc = 3,
rs = 2.5) from the fp32 oracle (same operation order as src/nonbonded.jl:122-155) and from the fp64 oracle.
The reference's own bound between its two fp32 implementations is 1e-4 absolute (test/runtests.jl:39-41)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package          # noqa: E402
from oracle import oracle as orc                  # noqa: E402

E = load_package()
orc.build()
dev = torch.device("cuda", 0)
with open(os.path.join(ROOT, "tests", "golden", "lj_sample.xyz")) as fh:
    n = int(fh.readline()); fh.readline()
    x = np.array([[float(t) for t in fh.readline().split()[1:4]] for _ in range(n)]).astype(np.float32)
N = 800
atoms = E.lennard_jones_atoms(1.0, 1.0, N)
model = E.LennardJonesModel(3.0, 2.5)
xd, ad = E.cu(x, dev), E.cu(atoms, dev)
z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
for mode, om, em in (("literal", orc.LITERAL, E.LITERAL), ("cutoff", orc.CUTOFF, E.CUTOFF)):
    f32, e32, w32 = orc.naive(x, 10.0, orc.model(3.0, 2.5, np.float32), atoms, om)
    f64, e64, w64 = orc.naive(x.astype(np.float64), 10.0, orc.model(3.0, 2.5), atoms, om)
    print("%s: fp32 oracle vs fp64 oracle  dF %.2e dE %.2e dW %.2e   (max|F| %.1f max|W| %.1f)"
          % (mode, np.abs(f32 - f64).max(), np.abs(e32 - e64).max(), np.abs(w32 - w64).max(), np.abs(f64).max(), np.abs(w64).max()))
    for which in ("tiles", "naive", "nbr"):
        if which == "nbr" and mode == "literal":
            continue
        f, e, w = z(N, 3), z(N), z(N)
        if which == "tiles":
            E.compute_nonbonded_(f, e, w, xd, 10.0, E.nonbonded_computation_tiles(N, all_pairs=True, mode=em), model, ad, 7)
        elif which == "naive":
            E.naively_compute_nonbonded_(f, e, w, xd, 10.0, model, ad, mode=em)
        else:
            E.compute_nonbonded_(f, e, w, xd, 10.0, E.nonbonded_computation_tiles(N), model, ad, 7)
        f, e, w = (t.cpu().numpy() for t in (f, e, w))
        print("  %-6s vs fp32 oracle dF %.2e dE %.2e dW %.2e | vs fp64 oracle dF %.2e dE %.2e dW %.2e"
              % (which, np.abs(f - f32).max(), np.abs(e - e32).max(), np.abs(w - w32).max(),
                 np.abs(f - f64).max(), np.abs(e - e64).max(), np.abs(w - w64).max()))
