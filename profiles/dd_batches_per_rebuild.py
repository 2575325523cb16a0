import os, sys, torch
sys.path.insert(0, ".")
from __graft_entry__ import load_package
E = load_package()
dev = torch.device("cuda", 0)
dd = E.DomainDecomposition.synthetic(136, 8, 0, dev, E.LennardJonesModel(2.5, 2.0), pkg=E, raw_velocities=True, mirror=True)
dd.step_(300, 0.005)
s0 = dd.stats(); dd.step_(1000, 0.005); s1 = dd.stats()
rb = s1["rebuilds"] - s0["rebuilds"]
print("1000 steps: rebuilds %d, batches %d (%.2f per rebuild), cancelled %d, steps per rebuild %.2f" % (rb, s1["batches"] - s0["batches"], (s1["batches"] - s0["batches"]) / rb, s1["cancelled_steps"] - s0["cancelled_steps"], 1000 / rb))
