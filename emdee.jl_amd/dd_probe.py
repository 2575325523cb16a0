"""Connectivity check of the native decomposition (emdee_dd_* over RCCL), run as a CHILD process of each rank.

Why a child: a halo exchange whose peers disagree (a communicator that never forms, a send without its receive) does
not fail, it waits -- and a process whose HIP stream waits for ever cannot fall back to anything.  bench.py (and
examples/lj_fluid_decomposed.py) therefore start this script once per rank before they create the decomposition they
time: it builds a small box cut into the same grid of bricks, steps it through several neighbour rebuilds (migration,
ghost selection, halo messages, the rebuild request riding on them) and prints one line.  The parent gives it a time
limit; a child that does not answer is killed and all ranks agree to use the torch.distributed driver instead.

Protocol (stdin/stdout, text): rank 0 prints `ID <256 hex digits>` (the RCCL unique id, which must be generated in
the process that will serve the bootstrap); the parent hands that line to every other rank's child on stdin.  Every
child ends with `OK <n_global> <owned> <potential energy> <rebuilds> <kinetic energy>` and exit code 0, or a traceback and exit code 1.

The child never outlives its purpose: it asks the kernel to kill it when the process that started it dies
(PR_SET_PDEATHSIG), and a timer of its own ends it with status 3 after --timeout seconds whatever it is waiting for -- a
child blocked in an RCCL wait whose rank has gone would otherwise hold the GPU and its memory for ever.

    python emdee.jl_amd/dd_probe.py --world 8 --rank 3 --device 3 [--cells 36] [--steps 24] [--precision f64] [--timeout 120]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--cells", type=int, default=36)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--precision", choices=("f64", "f32"), default="f64")
    ap.add_argument("--rc", type=float, default=2.5)
    ap.add_argument("--mixture", action="store_true")
    ap.add_argument("--switch-overlap", action="store_true",
                    help="second half of the steps without interior/boundary overlap (emdee_dd_set_overlap 0)")
    ap.add_argument("--in-process", action="store_true",
                    help="all --world domains in this one process (device copies instead of RCCL): the reference a "
                         "multi-process run of the same grid is compared with")
    ap.add_argument("--timeout", type=float, default=300.0, help="seconds after which this process ends itself (status 3)")
    args = ap.parse_args()

    import ctypes
    import signal
    import threading
    try:                                    # die with the rank that started me (PR_SET_PDEATHSIG = 1)
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)
    except (OSError, AttributeError):
        pass
    own_limit = threading.Timer(max(args.timeout, 1.0), lambda: os._exit(3))
    own_limit.daemon = True
    own_limit.start()

    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    torch.cuda.set_device(args.device)
    dev = torch.device("cuda", args.device)

    if args.in_process:
        uid = None
    elif args.rank == 0:
        uid = pkg.DomainDecomposition.unique_id()
        print("ID " + uid.hex(), flush=True)
    else:
        line = sys.stdin.readline().split()
        if len(line) != 2 or line[0] != "ID" or len(line[1]) != 256:
            raise SystemExit("dd_probe: expected `ID <256 hex digits>` on stdin")
        uid = bytes.fromhex(line[1])

    model = pkg.LennardJonesModel(args.rc, args.rc - 0.5)
    dd = pkg.DomainDecomposition.synthetic(args.cells, args.world, None if args.in_process else args.rank, dev, model,
                                           precision=torch.float64 if args.precision == "f64" else torch.float32,
                                           mixture=args.mixture, pkg=pkg, unique_id=uid, raw_velocities=True)
    dd.step_(args.steps, 0.005, 6)          # a rebuild (migration + new ghosts) every 6 steps, batches in between
    if args.switch_overlap:
        dd.set_overlap_(False)              # exchange and one launch over all bricks, in order on the compute stream
    dd.step_(args.steps, 0.005, 0)          # and the displacement-triggered form the benchmark runs
    ep, ek, vir = dd.totals()               # all-reduced over the ranks inside the library
    torch.cuda.synchronize(dev)
    st = dd.stats()
    if not (ep == ep and ek == ek and ek > 0.0):
        raise SystemExit("dd_probe: energies are not finite (%r, %r)" % (ep, ek))
    print("OK %d %d %.12g %d %.12g" % (dd.n_global, dd.n_owned, ep, st["rebuilds"], ek), flush=True)
    dd.close()


if __name__ == "__main__":
    main()
