"""Stand-in for bench.py's ranks in the CPU tests of its self-launcher (tests/test_bench_host.py): started by
torch.distributed.run like the real thing.  Behaviour is chosen by FAKE_RANKS_MODE:
  stall   without `--dd torch`: never finish (a halo exchange that waits for ever); with it: print the line
  fail    without `--dd torch`: exit status 17 at once; with it: print the line
  late    print the line, then never finish (a rank stuck in a final barrier)
  twice   every rank prints a line (a launcher must let one through)
  crash   rank 0 prints the line, then rank 1 dies with status 9 (a fault after the measurement: teardown, the target-box leg)
"""
import json
import os
import sys
import time

mode = os.environ.get("FAKE_RANKS_MODE", "fail")
rank = int(os.environ.get("RANK", "0"))
argv = sys.argv[1:]
torch_driver = "--dd" in argv and argv[argv.index("--dd") + 1] == "torch"
degraded = argv[argv.index("--degraded") + 1] if "--degraded" in argv else None


def line():
    out = {"metric": "md_steps_per_sec", "value": 1.0, "n_gpus": int(os.environ.get("WORLD_SIZE", "1")),
           "deadline_at": float(os.environ.get("EMDEE_BENCH_DEADLINE_AT", "0"))}
    if degraded:
        out["degraded"] = degraded
    os.write(1, (json.dumps(out) + "\n").encode())     # one write: the ranks share the launcher's pipe


if mode in ("stall", "fail") and not torch_driver:
    if mode == "fail":
        sys.exit(17)
    time.sleep(3600)
if mode == "twice" or rank == 0:
    line()
if mode == "late":
    time.sleep(3600)
if mode == "crash" and rank == 1:
    time.sleep(1.0)
    os._exit(9)
