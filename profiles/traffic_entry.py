"""HBM traffic of the dominant (fused force + integrator) kernel of ONE bench configuration, from two rocprofv3 passes
(--pmc FETCH_SIZE, --pmc WRITE_SIZE; MI355X_MICROARCH.md: separate passes, FETCH_SIZE x2 on gfx950, calibrated in the same
run on k_kick_drift, whose traffic is known), merged into profiles/traffic.json under the configuration's key
(atoms, dtype, rc, mixture).  bench.py attaches an entry only to a line of exactly that configuration.

usage: traffic_entry.py OUT_DIR TAG --atoms N --dtype f64|f32 --rc 2.5 --mixture 0|1 [--note "..."]
       (OUT_DIR/pmc_fetch, OUT_DIR/pmc_write: the rocprofv3 output directories)"""
import argparse
import collections
import csv
import glob
import json
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("out"); ap.add_argument("tag")
ap.add_argument("--atoms", type=int, required=True); ap.add_argument("--dtype", required=True)
ap.add_argument("--rc", type=float, required=True); ap.add_argument("--mixture", type=int, default=0)
ap.add_argument("--command", default="")
args = ap.parse_args()
root = os.path.dirname(os.path.abspath(__file__))


def load(sub, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(args.out, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fe, wr = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
# the step kernels: k_brick<real, Shape, THREADS, G, MODE = 3, BITMASK = 1 ...> and k_typed<real, Shape, THREADS, G, MODE = 3, ...>
fused = [k for k in fe if re.search(r"k_brick<.*?, 3, 1[,>]", k) or re.search(r"k_typed<.*?\), \d+, \d+, 3, ", k) or re.search(r"k_typed<[^>]*>, \d+, \d+, 3, ", k)]
assert fused, "no fused step kernel among: %s" % sorted(fe)[:12]
name = max(fused, key=lambda k: sum(fe[k]))
mean = lambda v: sum(v) / len(v) if v else 0.0
N = args.atoms
w = 8 if args.dtype == "f64" else 4
f_kb, w_kb = mean(fe[name]), mean(wr.get(name, []))
kd = [k for k in fe if "k_kick_drift" in k]
cal = None
if kd:
    # k_kick_drift reads x (record), v, f, xb and writes x, v: known bytes per atom
    cal = dict(fetch_raw_b_per_atom=mean(fe[kd[0]]) * 1024 / N, write_b_per_atom=mean(wr.get(kd[0], [])) * 1024 / N)
entry = dict(atoms=N, dtype=args.dtype, rc=args.rc, mixture=bool(args.mixture), kernel=name[:100], launches=len(fe[name]),
             fetch_size_kb_per_launch=f_kb, write_size_kb_per_launch=w_kb,
             lj_force_nbr_bytes_per_launch=int((2 * f_kb + w_kb) * 1024), bytes_per_atom=(2 * f_kb + w_kb) * 1024 / N,
             calibration_k_kick_drift=cal,
             source="profiles/%s/traffic_%s%s_rc%g.txt (profiles/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE in separate passes; "
                    "FETCH_SIZE x2 + WRITE_SIZE, KB -> bytes)%s" % (args.tag, args.dtype, "_mix" if args.mixture else "", args.rc,
                                                                  "; " + args.command if args.command else ""))
dst = os.path.join(root, args.tag)
os.makedirs(dst, exist_ok=True)
lines = ["rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py %s; N = %d" % (args.command, N),
         "raw counter KB per launch (mean); gfx950: FETCH_SIZE counts 1/2 of streamed read bytes (x2), WRITE_SIZE is exact"]
for k in sorted(fe, key=lambda k: -sum(fe[k]))[:12]:
    v, x = fe[k], wr.get(k, [0.0])
    lines.append("%-110s launches %3d  FETCH %12.1f KB (%6.1f B/atom raw)  WRITE %12.1f KB (%6.1f B/atom)"
                 % (k[:110], len(v), mean(v), mean(v) * 1024 / N, mean(x), mean(x) * 1024 / N))
lines.append("step kernel: %.1f B/atom per launch (2 x FETCH + WRITE)" % entry["bytes_per_atom"])
open(os.path.join(dst, "traffic_%s%s_rc%g.txt" % (args.dtype, "_mix" if args.mixture else "", args.rc)), "w").write("\n".join(lines) + "\n")
path = os.path.join(root, "traffic.json")
data = {"entries": []}
if os.path.exists(path):
    old = json.load(open(path))
    data = old if "entries" in old else {"entries": [dict(old, rc=2.5, mixture=False)]}
key = lambda e: (int(e["atoms"]), e["dtype"], float(e.get("rc", 2.5)), bool(e.get("mixture", False)))
data["entries"] = [e for e in data["entries"] if key(e) != key(entry)] + [entry]
data["entries"].sort(key=key)
json.dump(data, open(path, "w"), indent=1)
print("traffic %s: %.1f B/atom per launch of %s" % (key(entry), entry["bytes_per_atom"], name[:60]))
