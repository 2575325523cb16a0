"""The native decomposition over RCCL with W ranks on a ONE-GPU box.

RCCL refuses two ranks of a communicator on one device ("Duplicate GPU detected") only when both report the same
host: with a different NCCL_HOSTID per process every rank believes it sits on a node of its own, the duplicate check
does not apply, and the ranks talk through RCCL's network transport (TCP over the loopback interface, proxy threads,
host staging buffers).  The bandwidth says nothing about xGMI; what the run does establish is that W independent
processes, each with its own communicator rank, get through communicator set-up, migration, ghost selection, the
per-step halo messages (ncclSend/ncclRecv groups on the side stream), the rebuild request riding on them and the
all-reduced energies WITHOUT deadlock and with the same answer as the same grid stepped inside one process with
device copies.  Each rank is emdee.jl_amd/dd_probe.py.

    python profiles/rccl_ranks_one_gpu.py [--world 2] [--cells 24] [--steps 24] [--timeout 420] [--precision f64]

Prints one line per rank, the in-process reference, and `MATCH` / `MISMATCH`; exit code 0 only on MATCH.
"""
import argparse
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "emdee.jl_amd", "dd_probe.py")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--cells", type=int, default=24)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--timeout", type=float, default=420.0)
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--mixture", action="store_true")
    ap.add_argument("--rc", type=float, default=2.5)
    ap.add_argument("--switch-overlap", action="store_true", help="the ranks run the second half without overlap (in-order exchange)")
    args = ap.parse_args()
    if args.world > 6:
        raise SystemExit("at most 6 processes may share the card on the GPU boxes")

    common = ["--world", str(args.world), "--device", "0", "--cells", str(args.cells), "--steps", str(args.steps),
              "--precision", args.precision, "--rc", str(args.rc)] + (["--mixture"] if args.mixture else [])
    rank_only = ["--switch-overlap"] if args.switch_overlap else []      # (the in-process reference keeps the default)
    kids, logs = [], []
    deadline = time.monotonic() + args.timeout

    def stop_all():
        for k in kids:
            if k.poll() is None:
                try:
                    os.killpg(k.pid, signal.SIGKILL)
                except OSError:
                    pass

    try:
        uid_line = None
        for r in range(args.world):
            env = dict(os.environ)
            env.update(NCCL_HOSTID="emdee-one-gpu-rank-%d" % r, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                       NCCL_NET_GDR_LEVEL="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.setdefault("NCCL_DEBUG", "WARN")
            log = open(os.path.join(ROOT, "gpurun_out", "rccl_rank%d.err" % r), "w") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else subprocess.DEVNULL
            logs.append(log)
            k = subprocess.Popen([sys.executable, PROBE, "--rank", str(r)] + common + rank_only, env=env, stdin=subprocess.PIPE,
                                 stdout=subprocess.PIPE, stderr=log, text=True, start_new_session=True)
            kids.append(k)
            if r == 0:
                uid_line = k.stdout.readline().strip()          # rank 0 answers within seconds or not at all
                if not uid_line.startswith("ID "):
                    print("rank 0 produced no communicator id: %r" % uid_line)
                    return 2
            else:
                k.stdin.write(uid_line + "\n")
                k.stdin.flush()
        results = []
        for r, k in enumerate(kids):
            try:
                out, _ = k.communicate(timeout=max(deadline - time.monotonic(), 1.0))
            except subprocess.TimeoutExpired:
                print("rank %d: no answer within %.0f s -- stopping all ranks" % (r, args.timeout))
                return 3
            ok = [l for l in out.splitlines() if l.startswith("OK ")]
            print("rank %d: exit %s  %s" % (r, k.returncode, ok[0] if ok else "(no OK line)"), flush=True)
            if k.returncode != 0 or not ok:
                return 4
            results.append(ok[0].split())
    finally:
        stop_all()
        for l in logs:
            if l is not subprocess.DEVNULL:
                l.close()

    ref = subprocess.run([sys.executable, PROBE, "--rank", "0", "--in-process"] + common, capture_output=True, text=True,
                         timeout=300)
    ok = [l for l in ref.stdout.splitlines() if l.startswith("OK ")]
    print("in-process reference: exit %s  %s" % (ref.returncode, ok[0] if ok else ref.stderr[-800:]))
    if ref.returncode != 0 or not ok:
        return 5
    want = ok[0].split()
    owned = sum(int(r[2]) for r in results)
    tol = 1e-9 if args.precision == "f64" else 2e-4
    good = owned == int(want[1])
    for r in results:
        good &= r[1] == want[1] and r[4] == want[4]                                   # atoms, rebuild count
        good &= abs(float(r[3]) - float(want[3])) <= tol * abs(float(want[3]))        # potential energy (all-reduced)
        good &= abs(float(r[5]) - float(want[5])) <= tol * abs(float(want[5]))        # kinetic energy
    print("MATCH" if good else "MISMATCH", "(owned atoms over ranks %d of %s; tolerance %g)" % (owned, want[1], tol))
    return 0 if good else 6


if __name__ == "__main__":
    sys.exit(main())
