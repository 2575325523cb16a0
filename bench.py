#!/usr/bin/env python3
"""bench.py -- MD steps/s and pair-interactions/s of the nonbonded hot path on MI355X.

Workload (BASELINE.json metric): Lennard-Jones fcc box, rho* = 0.8, rc = 2.5 sigma
(switch at rc - 0.5), fp64, jittered lattice + Maxwell-Boltzmann velocities at T* = 1, dt = 0.005,
skin 0.3 sigma, neighbour rebuild when any atom has moved skin/2.  A "step" is one velocity-Verlet
step of the whole box (fused kick/drift pass + LJ force pass, rebuilds included).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--cells n] [--precision f64|f32]
                  [--rc 2.5] [--mixture] [--rebuild-every 0] [--no-cpu-baseline] [--scaling strong|weak]

N = 1: fcc 136^3 x 4 = 10,061,824 atoms (the 10^7-atom config the metric is quoted on) on one GPU.
N > 1: one rank per GPU, spatial domain decomposition with ghost-atom halo exchange over RCCL.  Either the
driver starts the ranks (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) or a
plain `python bench.py --gpus N` starts them itself as a child process (before this process touches the
GPU).  Default is STRONG scaling (BASELINE configs[2], north_star): the SAME --cells^3 x 4-atom box is cut
into rank_grid(N) bricks and `value` = steps/s of that box.  `--scaling weak` gives every rank its own
--cells^3 x 4-atom brick instead (`value` is still steps/s of the -- then N times larger -- box).

N > 1 runs carry `target_box`: the same measurement on the north-star 293^3 x 4 = 100,615,028-atom box.
Nothing in a multi-rank run is allowed to wait for ever, and everything derives from ONE deadline (--deadline, 540 s from
the start of the process the driver started: below the driver's 600 s): a child of every rank first steps a small
decomposed box over RCCL under a time limit (emdee.jl_amd/dd_probe.py; failure -> all ranks take the torch.distributed
driver together), the timed native run and the target-box leg have watchdogs (the latter prints the line without the
leg), every limit is the smaller of its flag and what is left of the deadline minus a reserve for the stages behind it,
and the self-launched form repeats a failed native run on the torch driver only if no line has been printed and enough
time is left.  A line that the torch driver produced because the native decomposition failed says so: top-level
`"degraded": "<reason>"`.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (lj_force_nbr) with HIP events
recorded on its own stream inside the timed region; `cpu_baseline` times the CPU oracle on the host
cores on a bounded sample of the same box.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6.29 TB/s measured copy ceiling
RHO = 0.8


def nbar(rc, rho=RHO):
    return 4.0 / 3.0 * np.pi * rc ** 3 * rho


def algorithmic_bytes_per_atom_step(w, rc):
    """B_alg / atom-step = 21 w + 4 nbar(rc)."""
    return 21.0 * w + 4.0 * nbar(rc)


def force_kernel_bytes_per_atom(w, rc):
    """Share of B_alg moved by one lj_force_nbr launch: read x (3w) + write f (3w) + one int32 per
    in-cutoff neighbour of the full list."""
    return 6.0 * w + 4.0 * nbar(rc)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cells", type=int, default=136, help="fcc cells per side per GPU (136 -> 10,061,824 atoms)")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--rc", type=float, default=2.5)
    ap.add_argument("--skin", type=float, default=0.3)
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--mixture", action="store_true", help="binary LJ mixture (config 5)")
    ap.add_argument("--rebuild-every", type=int, default=0, help="0 = displacement trigger; k = fixed cadence")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--langevin", type=float, default=0.0,
                    help="friction of the Langevin thermostat (T* = 1); 0 = NVE, the BASELINE workload")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI (one GPU per rank); gloo = host-staged halo, for rehearsals")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on cuda:0 (1-GPU rehearsal, with --backend gloo)")
    ap.add_argument("--rccl-loopback", action="store_true",
                    help="with --share-gpu: give every rank its own NCCL_HOSTID, so that RCCL accepts several ranks on the "
                         "one device and carries their messages over its TCP transport on the loopback interface -- the "
                         "native decomposition with real ranks on a 1-GPU box (a correctness rehearsal: the rate says "
                         "nothing about xGMI)")
    ap.add_argument("--cpu-sample-cells", type=int, default=63)
    ap.add_argument("--dd", choices=["native", "torch"], default="native",
                    help="N > 1: native = emdee_dd_* (migration, ghosts, halo over RCCL and the batched step loop inside "
                         "libemdee_hip.so); torch = the host-side driver of emdee.jl_amd/domain.py over torch.distributed")
    ap.add_argument("--no-probe", action="store_true", help="N > 1, native: skip the connectivity probe (emdee.jl_amd/dd_probe.py)")
    ap.add_argument("--deadline", type=float, default=540.0,
                    help="seconds, from the start of the process the driver started, by which the JSON line must be out; every "
                         "other limit below is clipped to what is left of it (the driver stops a run after 600 s)")
    ap.add_argument("--retry-min", type=float, default=200.0,
                    help="self-launched N > 1 runs: seconds that must be left for the failed native run to be repeated on --dd torch")
    ap.add_argument("--degraded", default=None, help=argparse.SUPPRESS)   # reason handed to the torch-driver retry
    ap.add_argument("--probe-timeout", type=float, default=90.0, help="N > 1, native: seconds the probe children may take")
    ap.add_argument("--no-halo-trial", action="store_true", help="N > 1, native: keep the overlapped halo exchange without trying the in-order form")
    ap.add_argument("--halo-trial-steps", type=int, default=12, help="N > 1, native: untimed steps per form of the halo exchange before the timed ones")
    ap.add_argument("--native-timeout", type=float, default=150.0, help="N > 1, native: seconds warm-up + timed steps may take")
    ap.add_argument("--target-timeout", type=float, default=180.0, help="N > 1: seconds the second (target) box may take before the line is printed without it")
    ap.add_argument("--launch-timeout", type=float, default=0.0, help="self-launched N > 1 runs: seconds before the ranks are stopped (0: what is left of --deadline)")
    ap.add_argument("--domains", type=int, default=0,
                    help="one-GPU rehearsal of the native decomposition: cut the box into this many domains, all in this "
                         "process on cuda:0 (device-to-device halo copies instead of RCCL)")
    ap.add_argument("--target-cells", type=int, default=-1,
                    help="native decomposition, strong scaling: also time a second, larger box of this many fcc cells per side and "
                         "report it as `target_box`.  Default (-1): the north-star target box (293 -> 100,615,028 atoms) on "
                         "N > 1 runs of the default 136-cell box; 0 = skip")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the same --cells^3x4 box on N GPUs (default); weak = one such brick per GPU")
    return ap.parse_args()


T_PROCESS_START = time.time()
DEADLINE_ENV = "EMDEE_BENCH_DEADLINE_AT"


class Budget:
    """One wall-clock deadline for the whole run (absolute epoch seconds, inherited by the ranks a self-launched run
    starts); every stage asks for min(its own flag, what is left minus a reserve for the stages behind it)."""

    def __init__(self, args):
        at = os.environ.get(DEADLINE_ENV)
        self.at = float(at) if at else T_PROCESS_START + float(args.deadline)

    def left(self):
        return self.at - time.time()

    def limit(self, want, reserve=0.0, floor=1.0):
        return max(float(floor), min(float(want), self.left() - float(reserve)))


def descendants(pid):
    """PIDs of all living descendants of `pid` (children first), from the parent links in /proc."""
    parent = {}
    for name in os.listdir("/proc"):
        if not name.isdigit():
            continue
        try:
            with open("/proc/%s/stat" % name) as fh:
                fields = fh.read().rsplit(")", 1)[1].split()                     # after the command name, which may hold spaces
            parent[int(name)] = int(fields[1])
        except (OSError, IndexError, ValueError):
            continue
    out, frontier = [], [pid]
    while frontier:
        nxt = [c for c, p in parent.items() if p in frontier]
        out += nxt
        frontier = nxt
    return out


def is_result_line(text):
    return text.startswith('{"metric"')


def self_launch(args, budget=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (this process has not
    imported torch or touched the GPU; it relays the child's output and exit code).  The child may take what is left of
    the deadline.  If the native decomposition run fails WITHOUT having printed its line, and at least --retry-min
    seconds are left, the ranks are started once more on the torch.distributed driver, which marks its line as degraded."""
    import signal
    import socket
    import subprocess
    import threading

    budget = budget or Budget(args)
    printed = []

    def run(extra):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        env[DEADLINE_ENV] = repr(budget.at - 10.0)                              # the ranks finish before their launcher gives up
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.environ.get("EMDEE_BENCH_RANK_SCRIPT", os.path.abspath(__file__))] + sys.argv[1:] + extra   # (the override: CPU tests of this launcher)
        limit = budget.limit(args.launch_timeout if args.launch_timeout > 0 else 1e9, reserve=3.0)
        child = subprocess.Popen(cmd, env=env, start_new_session=True, stdout=subprocess.PIPE, text=True)   # its own process group: killable as a whole

        held = []

        def relay():
            for ln in child.stdout:
                if is_result_line(ln):
                    if printed or held:                                          # ONE line, whatever the ranks do
                        continue
                    held.append(ln)                                              # printed when the rank group has ended: it carries its status
                    continue
                sys.stdout.write(ln)
                sys.stdout.flush()
        t = threading.Thread(target=relay, daemon=True)
        t.start()
        def emit(status):
            """the held result line, once, with the rank group's status in it"""
            if not held or printed:
                return
            ln = held[0]
            try:
                d = json.loads(ln)
                d["exit_status"] = status
                ln = json.dumps(d) + "\n"
            except ValueError:
                pass
            printed.append(ln)
            sys.stdout.write(ln)
            sys.stdout.flush()

        try:
            rc = child.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            print("bench.py: the %d-rank run exceeded %.0f s and is being stopped" % (args.gpus, limit), file=sys.stderr)
            emit(124)                                                            # a finished measurement goes out BEFORE anything is killed
            # torch.distributed.run gives every rank a process group of its own: killing the launcher's group alone would
            # orphan ranks that are stuck in a wait, with the GPUs in their hands.  Collect the launcher's descendants while
            # the parent links still exist, then stop exactly those processes.
            victims = descendants(child.pid)
            for pid in [child.pid] + victims:
                try:
                    os.kill(pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
            try:
                os.killpg(child.pid, signal.SIGKILL)                             # and whatever else shares the launcher's group
            except ProcessLookupError:
                pass
            child.wait()
            rc = 124
        t.join(timeout=5.0)
        if t.is_alive():
            # a descendant that survived still holds the pipe open: stop reading (a line that arrives after this is lost, one
            # that arrived is in `held`)
            try:
                child.stdout.close()
            except (OSError, ValueError):
                pass
            t.join(timeout=2.0)
        if held and not printed:
            # The measurement is complete once its line exists, whatever happens to the ranks afterwards (target-box leg,
            # teardown, a barrier): the line is kept -- but a rank group that then ends abnormally (a crash, a GPU fault, a
            # stop at the deadline) must show in the run's records, not only on stderr: top-level "exit_status".
            if rc != 0:
                print("bench.py: the rank group ended with status %d AFTER its result line was out; the line carries exit_status"
                      % rc, file=sys.stderr)
            emit(rc)
        return rc

    rc = run([])
    if printed:
        return 0                                                                 # the line is out: nothing is repeated
    if rc != 0 and args.dd == "native":
        if budget.left() >= args.retry_min:
            reason = "native decomposition run ended with status %d" % rc
            print("bench.py: %s; once more with --dd torch" % reason, file=sys.stderr)
            rc = run(["--dd", "torch", "--degraded", reason])
            if printed:
                return 0
        else:
            print("bench.py: native decomposition run ended with status %d and %.0f s are left: no second attempt"
                  % (rc, budget.left()), file=sys.stderr)
    return rc


def make_box(pkg, cells, mixture):
    syn = pkg.synthetic
    pos, L = syn.fcc_positions(cells)
    N = pos.shape[0]
    vel = syn.velocities(N)
    if mixture:
        eps, sigma = syn.mixture_parameters(syn.mixture_types(N))
        atoms = pkg.lennard_jones_atoms(eps, sigma)
    else:
        atoms = pkg.lennard_jones_atoms(1.0, 1.0, N)
    return pos, vel, atoms, L


def load_traffic_entries(path=None):
    """profiles/traffic.json: PMC-measured HBM bytes per launch of the step kernel, one entry per measured configuration
    (profiles/pmc_traffic.sh, profiles/collect.sh).  [] if the file is missing or unreadable."""
    path = path or os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            data = json.load(fh)
    except (OSError, ValueError):
        return []
    if isinstance(data, dict) and "entries" in data:
        return list(data["entries"])
    return [dict(data, rc=2.5, mixture=False)] if isinstance(data, dict) else []


def traffic_floor(atoms, word_bytes, rc, density=0.8):
    """Bytes a fused step launch cannot do without: the 21 words per atom of SURVEY.md 8(d) and ONE 2-byte list entry per
    in-cutoff neighbour.  (The algorithmic figure of 8(d) counts 4 bytes per neighbour; this library's entries are uint16
    tile slots, so a measured launch may come in slightly under it -- fp32 at 10^7 atoms: 291 against 293 B/atom -- but never
    under this.)"""
    nbar = (4.0 / 3.0) * math.pi * rc ** 3 * density
    return atoms * (21.0 * word_bytes + 2.0 * nbar)


def load_valu_entries():
    """profiles/valu.json: the VALU issue floor of the step kernel (SQ counter passes, profiles/pmc_valu.sh), one entry per
    configuration like traffic.json"""
    try:
        with open(os.path.join(ROOT, "profiles", "valu.json")) as fh:
            data = json.load(fh)
    except (OSError, ValueError):
        return []
    return data.get("entries", []) if isinstance(data, dict) else []


def pick_valu(entries, atoms, dtype, rc, mixture):
    """The entry measured on EXACTLY this configuration (atoms per GPU, arithmetic type, cutoff, one or two species), or None:
    an instruction count of the single-species fp64 kernel says nothing about the fp32 or the two-species one."""
    for e in entries:
        try:
            same = (int(e["atoms"]) == int(atoms) and e["dtype"] == dtype and abs(float(e.get("rc", 2.5)) - float(rc)) < 1e-9
                    and bool(e.get("mixture", False)) == bool(mixture))
        except (KeyError, TypeError, ValueError):
            continue
        if same:
            return e if e.get("issue_floor_ms") and e.get("valu_insts_per_launch") else None
    return None


def pick_traffic(entries, atoms, dtype, rc, mixture, floor_bytes):
    """The entry measured on EXACTLY this configuration -- atoms per GPU, arithmetic type, cutoff, one or two species -- or
    None: a line never carries another configuration's traffic (round 3 attached the single-species figure to the mixture
    line: 4.19 GB of traffic against 7.44 GB of algorithmic bytes).  An entry below the bytes the launch cannot do without
    (traffic_floor) cannot be a measurement of this kernel and is refused as well."""
    for e in entries:
        try:
            same = (int(e["atoms"]) == int(atoms) and e["dtype"] == dtype and abs(float(e.get("rc", 2.5)) - float(rc)) < 1e-9
                    and bool(e.get("mixture", False)) == bool(mixture))
        except (KeyError, TypeError, ValueError):
            continue
        if same:
            b = e.get("lj_force_nbr_bytes_per_launch")
            if b is None or (floor_bytes is not None and b < floor_bytes):
                return None
            return e
    return None


def cpu_baseline(pkg, args):
    """Oracle (CPU restatement, OpenMP over all host cores) timed on a bounded sample: whole
    velocity-Verlet steps of a smaller box of the same density and parameters; O(N) per step, so the
    rate for the benchmark box is the sample rate scaled by the atom-count ratio."""
    from oracle import oracle as orc
    orc.build()
    if 4 * args.cells ** 3 <= 4000:
        # BASELINE configs[0] (864 atoms): the reference's own CPU path is a single-threaded all-pairs double loop
        # (src/nonbonded.jl:122-155); time its restatement the same way, whole benchmark box, no scaling
        pos, vel, atoms, L = make_box(pkg, args.cells, args.mixture)
        nsteps = 100
        t0 = time.perf_counter()
        orc.verlet(pos, vel, L, orc.model(args.rc, args.rc - 0.5), atoms, args.dt, nsteps, use_cells=False, nthreads=1)
        per_step = (time.perf_counter() - t0) / (nsteps + 1)
        return dict(value=1.0 / per_step, unit="steps/s", cores=1, kind="port",
                    sample="%d velocity-Verlet steps of the benchmark box itself (%d atoms), all-pairs double loop, one "
                           "thread, fp64 oracle" % (nsteps, pos.shape[0]), sample_steps_per_sec=1.0 / per_step,
                    sample_atoms=pos.shape[0])
    n = args.cpu_sample_cells
    pos, vel, atoms, L = make_box(pkg, n, args.mixture)
    model = orc.model(args.rc, args.rc - 0.5)
    cores = orc.max_threads()
    t0 = time.perf_counter()
    f, e, w = orc.nonbonded_cells(pos, L, model, atoms)
    t_force = time.perf_counter() - t0
    nsteps = int(max(1, min(8, round(12.0 / max(t_force, 1e-3)))))
    t0 = time.perf_counter()
    orc.verlet(pos, vel, L, model, atoms, args.dt, nsteps)
    dt_run = time.perf_counter() - t0
    per_step = dt_run / (nsteps + 1)            # nsteps force passes + the initial one, integrator passes are noise
    N_sample = pos.shape[0]
    N_bench = 4 * args.cells ** 3
    return dict(value=(1.0 / per_step) * N_sample / N_bench, unit="steps/s", cores=cores, kind="port",
                sample="%d velocity-Verlet steps of fcc %d^3x4 = %d atoms (same rho*, rc, fp64 oracle, OpenMP cell list); "
                       "per-step time scaled by %d/%d atoms" % (nsteps, n, N_sample, N_bench, N_sample),
                sample_steps_per_sec=1.0 / per_step, sample_atoms=N_sample)


def cache_path(args):
    """Where the 1-GPU line of a configuration is remembered, so that an N-GPU run of the same box on the same
    checkout can report its strong-scaling efficiency (the driver computes its own from the per-N values)."""
    key = "c%d_%s_rc%g_skin%g_dt%g%s%s" % (args.cells, args.precision, args.rc, args.skin, args.dt,
                                          "_mix" if args.mixture else "", "_re%d" % args.rebuild_every if args.rebuild_every else "")
    return os.path.join(ROOT, ".bench_cache", "onegpu_%s.json" % key)


def main():
    args = parse_args()
    budget = Budget(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, budget))
    import threading
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if args.share_gpu:
        local_rank = 0                     # rehearsal on a 1-GPU box: all ranks on cuda:0 (needs --backend gloo)
        if args.rccl_loopback:             # before anything loads librccl; the probe children inherit it
            os.environ.update(NCCL_HOSTID="emdee-share-gpu-rank-%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                              NCCL_NET_GDR_LEVEL="0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    cdev = dev                             # where the small collective tensors live
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
            cdev = torch.device("cpu")

    # ONE line, printed once, by whoever gets there first: the normal end of main(), the target-box watchdog, or the
    # last-resort timer that fires shortly before the deadline (it prints the line if the headline measurement is complete)
    line = {"out": None, "lock": threading.Lock(), "printed": False}

    def emit():
        with line["lock"]:
            if line["printed"] or line["out"] is None:
                return line["printed"]
            if rank == 0:
                sys.stdout.write(json.dumps(line["out"]) + "\n")          # one write: the line cannot be cut in two
                sys.stdout.flush()
            line["printed"] = True
            return True

    def last_resort():
        done = emit()
        print("rank %d: the deadline (%.0f s) is here; %s" % (rank, args.deadline, "line printed" if done else "no measurement to print"),
              file=sys.stderr, flush=True)
        os._exit(0 if done else 18)
    if world > 1:
        final_timer = threading.Timer(max(budget.left() - 4.0, 1.0), last_resort)
        final_timer.daemon = True
        final_timer.start()

    w = 8 if args.precision == "f64" else 4
    tdtype = torch.float64 if w == 8 else torch.float32
    ndtype = np.float64 if w == 8 else np.float32
    rc, rs = args.rc, args.rc - 0.5
    model = pkg.LennardJonesModel(rc, rs)

    domain = None
    dd_engine = None
    dd_probe = None
    degraded = args.degraded           # why the torch driver runs although the native decomposition was asked for
    if world == 1 and args.domains <= 1:
        pos, vel, atoms, L = make_box(pkg, args.cells, args.mixture)
        N_total = N_rank = pos.shape[0]
        md = pkg.VelocityVerlet(pkg.cu(pos.astype(ndtype), dev), pkg.cu(vel.astype(ndtype), dev), L, model,
                                pkg.cu(atoms, dev), skin=args.skin)
        del pos, vel
        run = lambda k: md.step_(k, args.dt, args.rebuild_every)
        engine = md
        parallelism = "single-gpu"
    else:
        ndom = world if world > 1 else args.domains
        if args.dd == "native" or world == 1:
            # the decomposition inside the library; every rank must get there, or all fall back together
            ok, err = 1, None
            if world > 1 and not args.no_probe:
                # a child of every rank steps a small box over RCCL first, under a time limit: a halo exchange that
                # cannot complete waits for ever instead of failing, and only a child can be abandoned (dd_probe.py)
                ok, note = pkg.dd.probe_over_rccl(world, rank, local_rank, dist, precision=args.precision,
                                                  timeout=budget.limit(args.probe_timeout, reserve=240.0, floor=20.0))
                dd_probe = note if ok else "failed: " + note
                if not ok:
                    err = "connectivity probe: " + note
                ok = int(ok)
            try:
                if not ok:
                    raise RuntimeError(err)
                domain = pkg.DomainDecomposition.synthetic(args.cells, ndom, rank if world > 1 else None, dev, model,
                                                           precision=tdtype, skin=args.skin, mixture=args.mixture, pkg=pkg,
                                                           scaling=args.scaling, dist=dist)
                dd_engine = "native (emdee_dd_*: %s)" % ("RCCL send/recv" if world > 1 else "%d domains in one process, device copies" % ndom)
            except Exception as e:                                  # noqa: BLE001 -- reported below, then the torch driver runs
                ok, err = 0, repr(e)
            if dist is not None:
                t_ok = torch.tensor([ok], dtype=torch.int32, device=cdev)
                dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
                ok = int(t_ok.item())
            if not ok:
                if world == 1:
                    raise SystemExit("native decomposition failed: %s" % err)
                if err is not None:
                    print("rank %d: native decomposition unavailable (%s); falling back to --dd torch" % (rank, err), file=sys.stderr)
                reasons = [err]
                if dist is not None:                               # the line is rank 0's: it names a rank that failed
                    reasons = [None] * world
                    dist.all_gather_object(reasons, err)
                degraded = next(("rank %d: %s" % (r, e) for r, e in enumerate(reasons) if e), "native decomposition unavailable")
                domain = None
        if domain is None:
            domain = pkg.domain.DecomposedVerlet.synthetic(args.cells, world, rank, dev, model, precision=tdtype,
                                                           skin=args.skin, mixture=args.mixture, pkg=pkg,
                                                           transport="device" if args.backend == "nccl" else "host",
                                                           scaling=args.scaling)
            dd_engine = "torch (emdee.jl_amd/domain.py over torch.distributed %s)" % args.backend
            engine = domain.md
            overlap = domain.overlap
        else:
            engine = domain.engine(0)
            overlap = os.environ.get("EMDEE_DD_OVERLAP", "1") != "0"
        N_rank, N_total = domain.n_owned, domain.n_global
        run = lambda k: domain.step_(k, args.dt, args.rebuild_every)
        parallelism = "dd%s" % "x".join(str(g) for g in domain.grid)
    if args.langevin > 0.0:                # not the BASELINE workload: prices the thermostat (config.thermostat says so)
        (md if domain is None else domain).set_langevin_(args.langevin, 1.0, 0x5EED)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def choose_halo_mode(obj, k):
        """N > 1, native: a few untimed steps with the halo exchange overlapped with the interior bricks (two launches,
        events between streams) and a few with everything in order on the compute stream (one launch, no events); keep the
        faster one for the timed steps.  Which wins depends on the size of a rank's domain and on the link."""
        trial = {}
        for name, on in (("overlapped", True), ("in order", False)):
            obj.set_overlap_(on)
            obj.step_(2, args.dt, args.rebuild_every)
            fence()
            t0 = time.perf_counter()
            obj.step_(k, args.dt, args.rebuild_every)
            fence()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            trial[name] = 1e3 * tt.item() / k
        best = min(trial, key=trial.get)
        obj.set_overlap_(best == "overlapped")
        return {"chosen": best, "trial_ms_per_step": trial, "trial_steps": k}

    main_watchdog = None
    halo_mode = None
    if world > 1 and dd_engine is not None and dd_engine.startswith("native"):
        # the probe has passed, so this is not expected to fire; if the timed run stalls all the same, leave with a status
        # the self-launching parent answers with the torch driver (if time is left), instead of waiting for somebody's limit

        native_limit = budget.limit(args.native_timeout, reserve=60.0, floor=10.0)

        def give_up():
            print("rank %d: the native decomposition run did not finish within %.0f s" % (rank, native_limit), file=sys.stderr, flush=True)
            os._exit(17)
        main_watchdog = threading.Timer(native_limit, give_up)
        main_watchdog.daemon = True
        main_watchdog.start()
    run(args.warmup)
    fence()
    if main_watchdog is not None and not args.no_halo_trial and os.environ.get("EMDEE_DD_OVERLAP") is None:
        halo_mode = choose_halo_mode(domain, args.halo_trial_steps)
        overlap = halo_mode["chosen"] == "overlapped"
    builds0 = engine.nbr_stats()["builds"]
    engine.profile_(True)
    native_dd = domain is not None and dd_engine is not None and dd_engine.startswith("native")
    local_engines = [engine]
    if native_dd and world == 1:                         # every domain lives in this process: time them all
        local_engines = [domain.engine(l) for l in range(args.domains)]
        for e in local_engines[1:]:
            e.profile_(True)
    phase0 = domain.phase_times() if native_dd else None
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if main_watchdog is not None:
        main_watchdog.cancel()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    # ---- where a rank's step goes (N > 1, native decomposition): means per timed step of the fused launches over interior and
    # boundary bricks, the halo (pack -> exchange -> unpack, on the stream it runs on: the wait of the boundary half), the
    # engine's sort + list, the host wall-clock inside rebuilds and inside blocking read-backs, and the ghost share;
    # every rank's figures, gathered to rank 0 -- so that a scaling curve explains itself the day it is measured
    per_rank = None
    if native_dd:
        phase1 = domain.phase_times()
        mine = []
        for l, e in enumerate(local_engines):
            per = lambda name: e.kernel_time(name)[0] / args.steps
            c = domain.counts(l)
            mine.append({"rank": rank if world > 1 else l, "atoms_owned": c["n_owned"], "ghost_fraction": c["n_ghost"] / max(c["n_owned"] + c["n_ghost"], 1),
                         "force_interior_ms": per("fused_step_interior"), "force_boundary_ms": per("fused_step_boundary"),
                         "halo_ms": per("halo"), "rebuild_device_ms": per("rebuild"),
                         "rebuild_wall_ms": (phase1["rebuild_ms"] - phase0["rebuild_ms"]) / args.steps,
                         "readback_wall_ms": (phase1["readback_ms"] - phase0["readback_ms"]) / args.steps,
                         "readbacks": phase1["readbacks"] - phase0["readbacks"], "rebuilds": phase1["rebuilds"] - phase0["rebuilds"],
                         # a rebuild = the read-back of the batch's request words that asked for it + those inside it (ONE, with
                         # the build's words, when it ran in the engines' own order)
                         "rebuilds_in_engine_order": phase1["rebuilds_in_engine_order"] - phase0["rebuilds_in_engine_order"],
                         # (the counters are the PROCESS's: with several domains in one process every domain's build has its read-back,
                         # so the figure is per local domain -- a rank of a real run is one process with one domain)
                         "readbacks_per_rebuild": 1.0 + (phase1["rebuild_readbacks"] - phase0["rebuild_readbacks"]) /
                                                  max((phase1["rebuilds"] - phase0["rebuilds"]) * max(len(local_engines), 1), 1)})
        ranks = mine
        if dist is not None:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            ranks = [r for g in gathered for r in g]
        per_rank = {"unit": "ms per timed step (means); rebuild_wall / readback_wall: host wall-clock of the process, shared by its domains",
                    "ranks": ranks}

    plain_ms, plain_launches = engine.kernel_time("lj_force_nbr")
    fused_ms, fused_launches = engine.kernel_time("lj_force_nbr_fused_step")
    force_ms, force_launches = plain_ms + fused_ms, plain_launches + fused_launches
    kd_ms, kd_launches = engine.kernel_time("verlet_kick_drift")
    rb_ms, rb_launches = engine.kernel_time("rebuild")
    stats = engine.nbr_stats()
    if domain is not None and world == 1:                # every domain lives in this process
        pairs = sum(domain.engine(l).count_pairs() for l in range(args.domains))
    else:
        pairs = engine.count_pairs()
    if dist is not None:
        tp = torch.tensor([pairs], dtype=torch.int64, device=cdev)
        dist.all_reduce(tp)
        pairs = int(tp.item())
    ep, ek, vir = (domain if domain is not None else engine).totals()        # decomposed: all-reduced over the ranks
    N_energy = N_total

    steps_per_sec = args.steps / elapsed
    b_step = algorithmic_bytes_per_atom_step(w, rc) * N_total
    # Launch mix of the dominant kernel: on one GPU every inner step of emdee_md_step is ONE launch of
    # lj_force_nbr with the velocity-Verlet kick/drift fused in (it then carries the whole step's algorithmic
    # bytes, 21 w + 4 nbar per atom); the last step of the call (and every step of a decomposed run) is a plain
    # force launch (6 w + 4 nbar per atom).
    # A decomposed run splits every pass into an interior and a boundary launch (halo exchange in between):
    # the two launches together carry one pass worth of bytes.
    split = 2 if (domain is not None and overlap) else 1
    fused, plain = fused_launches, plain_launches
    b_launch = (fused * algorithmic_bytes_per_atom_step(w, rc) + plain * force_kernel_bytes_per_atom(w, rc)) \
        * N_rank / split / max(force_launches, 1)
    force_avg_s = force_ms / max(force_launches, 1) * 1e-3
    achieved = b_launch / force_avg_s / 1e9 if force_launches else 0.0

    scaling = args.scaling if world > 1 else "strong"
    if scaling == "strong":
        shape = "%d^3x4-atom box" % args.cells + (" cut into %s bricks, one per %s" % ("x".join(str(g) for g in domain.grid), "GPU" if world > 1 else "domain (all on one GPU)") if domain is not None else "")
    else:
        shape = "one %d^3x4-atom brick per GPU (%s bricks)" % (args.cells, "x".join(str(g) for g in domain.grid))
    out = {
        "metric": "md_steps_per_sec",
        # whole-job figure: velocity-Verlet steps per second of the WHOLE box named in config (all GPUs advance it together)
        "value": steps_per_sec,
        "unit": "steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f64" if w == 8 else "f32",
        "data": "synthetic",
        "config": {"workload": "LJ fcc box rho*=0.8 rc=%gsigma rs=%gsigma%s, %d atoms (%s), velocity-Verlet dt=%g, skin %g"
                               % (rc, rs, " binary mixture" if args.mixture else "", N_total, shape, args.dt, args.skin),
                   "atoms": N_total, "atoms_per_gpu": N_total / world, "atoms_rank0": N_rank, "rc": rc, "mixture": bool(args.mixture),
                   "parallelism": parallelism,
                   "decomposition": dd_engine, "decomposition_probe": dd_probe, "halo_exchange": halo_mode,
                   "rebuild": "every %d steps" % args.rebuild_every if args.rebuild_every else "max displacement > skin/2",
                   "thermostat": "langevin gamma=%g T*=1" % args.langevin if args.langevin > 0.0 else "none (NVE)"},
        "pair_interactions_per_sec": pairs * steps_per_sec,
        "pairs_in_cutoff": pairs,
        "atom_steps_per_sec": N_total * steps_per_sec,
        "roofline": {"bound": "hbm", "kernel": "lj_force_nbr (k_brick; velocity-Verlet update fused in on %d of %d launches)"
                                                 % (fused, force_launches),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": b_launch, "avg_launch_ms": force_avg_s * 1e3,
                     "launches": force_launches, "fused_launches": fused},
        "step_roofline": {"algorithmic_bytes_per_step": b_step, "achieved": b_step * steps_per_sec / 1e9 / world,
                          "unit": "GB/s per GPU", "frac": b_step * steps_per_sec / 1e9 / world / HBM_PEAK_GBS},
        "kernels_ms": {"lj_force_nbr": [force_ms, force_launches], "verlet_kick_drift": [kd_ms, kd_launches],
                       "rebuild(bin+sort+nbr_build)": [rb_ms, rb_launches]},
        "neighbor_list": {"builds_in_timed_region": stats["builds"] - builds0, "listed": stats["listed"],
                          "max_count": stats["max_count"], "capacity": stats["capacity"]},
        "energy_per_atom": {"potential": ep / N_energy, "kinetic": ek / N_energy},
    }
    if per_rank is not None:
        out["per_rank"] = per_rank
    if degraded is not None and world > 1 and not (dd_engine or "").startswith("native"):
        out["degraded"] = degraded     # the native decomposition (north_star's) did not produce this line
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:        # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(pkg, args)
        cp = cache_path(args)
        if world == 1 and domain is None:
            try:
                os.makedirs(os.path.dirname(cp), exist_ok=True)
                with open(cp, "w") as fh:
                    json.dump({"value": steps_per_sec, "steps": args.steps, "warmup": args.warmup, "atoms": N_total}, fh)
            except OSError:
                pass
        elif world > 1 and scaling == "strong" and os.path.exists(cp):
            try:
                with open(cp) as fh:
                    one = json.load(fh)
                if one.get("atoms") == N_total and one.get("value", 0) > 0:
                    out["efficiency_vs_1gpu"] = steps_per_sec / (world * one["value"])
                    out["one_gpu_reference"] = one
            except (OSError, ValueError):
                pass
        t = pick_traffic(load_traffic_entries(), N_rank, out["dtype"], args.rc, bool(args.mixture),
                         traffic_floor(N_rank, 8 if out["dtype"] == "f64" else 4, args.rc))
        if t is not None:
            out["roofline"]["traffic"] = t.get("lj_force_nbr_bytes_per_launch")
            out["roofline"]["traffic_source"] = t.get("source")
        # the same kernel against the limit that actually binds it in fp64 ("report VALU
        # utilisation next to GB/s"): instruction counts from the committed SQ counter pass, live duration
        t = pick_valu(load_valu_entries(), N_rank, out["dtype"], args.rc, bool(args.mixture))
        if t is not None and force_avg_s > 0:
            out["valu_issue"] = {"kernel": t.get("kernel", "lj_force_nbr (fused)"), "floor_ms": t["issue_floor_ms"],
                                 "avg_launch_ms": force_avg_s * 1e3, "frac": t["issue_floor_ms"] / (force_avg_s * 1e3),
                                 "valu_insts_per_launch": t["valu_insts_per_launch"], "fp64_share": t.get("fp64_share"),
                                 "fp32_share": t.get("fp32_share"),
                                 "model": t.get("model"), "clock_ghz": t.get("clock_ghz"), "source": t.get("source"),
                                 # same launches inside the counter pass, at the SQ clock measured there
                                 "profiled_clock_ghz": t.get("profiled_clock_ghz"),
                                 "frac_at_profiled_clock": t.get("frac_at_profiled_clock"),
                                 "valu_busy_fraction": t.get("valu_busy_fraction")}
    # ---- the north-star target box (>= 10^8 atoms) on the same ranks: a second, smaller measurement riding on the N > 1
    # runs, so that the driver's 1/2/4/8-GPU sweep also yields the strong-scaling curve of the target size.  `value` above
    # stays the BASELINE metric (the 10^7-atom box); a failure here is reported inside the object and changes nothing else.
    target_cells = args.target_cells
    if target_cells < 0:
        target_cells = 293 if (world > 1 and args.cells == 136 and not args.mixture and args.langevin == 0.0) else 0
    if dd_engine is not None and dd_engine.startswith("native") and scaling == "strong" and target_cells > args.cells:
        target = {"cells": target_cells}
        # The headline measurement above is complete; this second box must not be able to lose it.  If the leg has not
        # finished within its limit (a collective that never completes cannot be interrupted from inside), rank 0 prints the
        # line with the leg marked as timed out and every rank leaves the process at once.
        line["out"] = out                                               # from here on the last-resort timer has a line to print
        target_limit = min(float(args.target_timeout), budget.left() - 15.0)
        done = threading.Lock()

        def bail():
            if not done.acquire(blocking=False):
                return
            out["target_box"] = dict(target, error="timed out after %.0f s" % max(target_limit, 0.0))
            emit()
            os._exit(0)
        watchdog = threading.Timer(max(target_limit, 0.0), bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            domain.close()
            del domain, engine
            torch.cuda.empty_cache()
            big = pkg.DomainDecomposition.synthetic(target_cells, ndom, rank if world > 1 else None, dev, model,
                                                    precision=tdtype, skin=args.skin, mixture=args.mixture, pkg=pkg,
                                                    scaling="strong", dist=dist)
            k_w, k_t = min(args.warmup, 5), min(args.steps, 20)
            big.step_(k_w, args.dt, args.rebuild_every)
            fence()
            if halo_mode is not None:
                target["halo_exchange"] = choose_halo_mode(big, max(4, args.halo_trial_steps // 2))
            t0 = time.perf_counter()
            big.step_(k_t, args.dt, args.rebuild_every)
            fence()
            dt_big = time.perf_counter() - t0
            if dist is not None:
                tt = torch.tensor([dt_big], dtype=torch.float64, device=cdev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_big = tt.item()
            ep_b, ek_b, _ = big.totals()
            nb = big.n_global
            target.update(atoms=nb, steps=k_t, warmup=k_w, steps_per_sec=k_t / dt_big, ms_per_step=1e3 * dt_big / k_t,
                          atom_steps_per_sec=nb * k_t / dt_big, rebuilds=big.stats()["rebuilds"],
                          energy_per_atom={"potential": ep_b / nb, "kinetic": ek_b / nb},
                          step_roofline_frac=algorithmic_bytes_per_atom_step(w, rc) * nb * k_t / dt_big / 1e9 / world / HBM_PEAK_GBS)
            ref = os.path.join(ROOT, "profiles", "target_box_1gpu.json")
            if os.path.exists(ref):
                with open(ref) as fh:
                    one = json.load(fh)
                if one.get("atoms") == nb and one.get("value", 0) > 0:
                    target["one_gpu_steps_per_sec"] = one["value"]
                    target["one_gpu_source"] = one.get("source")
                    target["efficiency_vs_1gpu"] = target["steps_per_sec"] / (world * one["value"])
            big.close()
        except Exception as e:                                      # noqa: BLE001
            target["error"] = repr(e)
        watchdog.cancel()
        if not done.acquire(blocking=False):                        # the watchdog is printing: it also ends the process
            time.sleep(30)
        out["target_box"] = target
        if "error" in target:
            # this rank's leg failed: the others may be waiting in a collective it will never enter.  The line goes out and
            # the process ends here, without a barrier nobody may reach (their own watchdogs end them, status 0)
            line["out"] = out
            emit()
            os._exit(0)
    line["out"] = out
    emit()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
