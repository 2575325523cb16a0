"""Turns the raw rocprofv3 output of profiles/collect.sh into the small files committed under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.abspath(__file__))
dst = os.path.join(root, tag)
os.makedirs(dst, exist_ok=True)

bench = json.load(open(os.path.join(out, "bench_default.json")))
N = bench["config"]["atoms_per_gpu"]
shutil.copy(os.path.join(out, "bench_default.json"), os.path.join(dst, "final_bench_default.json"))
stats = glob.glob(os.path.join(out, "stats", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "final_kernel_stats.csv"))


def load(sub, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fe, wr = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
lines = ["rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 12 --warmup 3; N = %d" % N,
         "raw counter KB per launch (mean); gfx950: FETCH_SIZE counts 1/2 of streamed read bytes (x2), WRITE_SIZE is exact"]
for k in sorted(fe, key=lambda k: -sum(fe[k])):
    v, w = fe[k], wr.get(k, [0.0])
    lines.append("%-120s launches %3d  FETCH %12.1f KB (%6.1f B/atom raw)  WRITE %12.1f KB (%6.1f B/atom)"
                 % (k[:120], len(v), sum(v) / len(v), sum(v) / len(v) * 1024 / N, sum(w) / len(w), sum(w) / len(w) * 1024 / N))
open(os.path.join(dst, "final_pmc_fetch_write.txt"), "w").write("\n".join(lines) + "\n")


def mean(d, pred):
    vals = [x for k, v in d.items() if pred(k) for x in v]
    return sum(vals) / len(vals) if vals else None


fused = lambda k: "k_brick<" in k and ", 3, 1>" in k
kd = lambda k: "k_kick_drift" in k
f_kb, w_kb = mean(fe, fused), mean(wr, fused)
cal_f, cal_w = mean(fe, kd), mean(wr, kd)
if f_kb is not None and w_kb is not None:
    traffic = dict(
        source="profiles/%s/final_pmc_fetch_write.txt (profiles/collect.sh): FETCH_SIZE x2 + WRITE_SIZE, KB -> bytes; "
               "x2 calibrated in the same run on k_kick_drift: 2 x %.1f = %.1f B/atom against 108 known read bytes, "
               "WRITE %.1f against 56 known" % (tag, (cal_f or 0) * 1024 / N, 2 * (cal_f or 0) * 1024 / N, (cal_w or 0) * 1024 / N),
        atoms=N, dtype=bench["dtype"], kernel="k_brick<..., BRICK_STEP, 1> (lj_force_nbr with the fused velocity-Verlet update)",
        fetch_size_kb_per_launch=f_kb, write_size_kb_per_launch=w_kb,
        lj_force_nbr_bytes_per_launch=int((2 * f_kb + w_kb) * 1024), bytes_per_atom=(2 * f_kb + w_kb) * 1024 / N)
    json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
    print("traffic: %.1f B/atom per launch" % traffic["bytes_per_atom"])
print(json.dumps({k: bench[k] for k in ("value", "ms_per_step", "roofline", "step_roofline", "cpu_baseline") if k in bench})[:1500])
