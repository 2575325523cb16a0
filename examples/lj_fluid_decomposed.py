#!/usr/bin/env python3
"""The same Lennard-Jones fluid cut into bricks (emdee_dd_*): migration, ghosts, halo exchange and the batched
step loop run inside libemdee_hip.so.

    python examples/lj_fluid_decomposed.py [domains] [cells]                          # one process, one GPU:
                                                                                       #   every domain on cuda:0, device copies
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
           examples/lj_fluid_decomposed.py 8 136                                       # one process per GPU: RCCL send/recv

The trajectory is that of the undivided box to rounding; the thermostat's noise is keyed by global atom id."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

E = load_package()
domains = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 24
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
dist = None
if world > 1:                                                    # one rank per GPU
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    domains = world
    # a small decomposed box stepped over RCCL by a child of every rank, under a time limit: a halo exchange that
    # cannot complete waits for ever instead of failing, and only a child can be abandoned (emdee.jl_amd/dd_probe.py)
    ok, note = E.dd.probe_over_rccl(world, rank, torch.cuda.current_device(), dist, timeout=150.0)
    if not ok:
        raise SystemExit("rank %d: the decomposition cannot run over RCCL here (%s)" % (rank, note))
dev = torch.device("cuda", torch.cuda.current_device())

model = E.LennardJonesModel(2.5, 2.0)
dd = E.DomainDecomposition.synthetic(cells, domains, rank if world > 1 else None, dev, model, skin=0.3, pkg=E, dist=dist)
say = print if rank == 0 else (lambda *a: None)
c = dd.counts(0)
say("%d atoms in %s bricks; this domain owns %d and sees %d ghosts" % (c["n_global"], "x".join(map(str, dd.grid)), c["n_owned"], c["n_ghost"]))

dd.set_langevin_(gamma=2.0, temperature=0.9, seed=2026)
dd.step_(500, 0.005)
obs = dd.observables()
say("after 500 thermostatted steps: T* = %.4f  P* = %.4f  U/N = %.4f" % (obs["temperature"], obs["pressure"], obs["potential"] / c["n_global"]))
dd.set_langevin_(0.0, 0.0)
e0 = sum(dd.totals()[:2])
dd.step_(500, 0.005)
e1 = sum(dd.totals()[:2])
st = dd.stats()
say("500 NVE steps: relative energy change %.2e; %d rebuilds, %d batches of queued steps, %d steps cancelled by a rebuild request, "
    "%d atoms changed owner here" % (e1 / e0 - 1.0, st["rebuilds"], st["batches"], st["cancelled_steps"], st["migrated"]))
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
