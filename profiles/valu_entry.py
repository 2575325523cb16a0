"""VALU issue floor of the dominant (fused force + integrator) kernel of ONE bench configuration, from SQ counter passes
(rocprofv3 --pmc, 8 counters per pass), merged into profiles/valu.json under the configuration's key (atoms, dtype, rc,
mixture) -- as profiles/traffic_entry.py does for the HBM traffic.  bench.py attaches an entry only to a line of exactly
that configuration (`valu_issue`).

Model (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.07 quad-cycles on the fp64 kernel): one
wave64 VALU instruction holds its SIMD for 4 cycles whatever its type -- fp64, fp32, packed fp32, integer -- and a
transcendental (v_rcp_f64, v_rcp_f32, v_rsq ...) for 16; floor = (INSTS_VALU x 4 + TRANS x 12) cycles / 1024 SIMDs / clock.

usage: valu_entry.py OUT_DIR TAG --atoms N --dtype f64|f32 --rc 2.5 --mixture 0|1 [--command "..."]
       (OUT_DIR/pmc_sq1, pmc_sq2[, pmc_sq3]: the rocprofv3 output directories of profiles/pmc_valu.sh)"""
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
STEP = (r"k_brick<.*?, 3, 1[,>]", r"k_typed<.*?\), \d+, \d+, 3, ", r"k_typed<[^>]*>, \d+, \d+, 3, ")
fused = lambda k: any(re.search(p, k) for p in STEP)

sq, names = collections.defaultdict(list), collections.Counter()
for sub in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
    for f in glob.glob(os.path.join(args.out, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if fused(r["Kernel_Name"]):
                sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names[r["Kernel_Name"]] += 1
assert sq, "no fused step kernel in the counter passes under %s" % args.out
prof_ns = []
for f in glob.glob(os.path.join(args.out, "pmc_sq1", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if fused(r["Kernel_Name"]):
            prof_ns.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
avg = {k: sum(v) / len(v) for k, v in sq.items()}
simds, clock_ghz = 256 * 4, 2.4
insts = avg.get("SQ_INSTS_VALU", 0.0)
trans = avg.get("SQ_INSTS_VALU_TRANS_F64", 0.0) + avg.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
floor_ms = (insts * 4 + trans * 12) / simds / (clock_ghz * 1e6)
f64 = sum(avg.get("SQ_INSTS_VALU_%s_F64" % k, 0.0) for k in ("FMA", "MUL", "ADD", "TRANS"))
f32 = sum(avg.get("SQ_INSTS_VALU_%s_F32" % k, 0.0) for k in ("FMA", "MUL", "ADD", "TRANS"))
prof_ms = sum(prof_ns) / len(prof_ns) * 1e-6 if prof_ns else None
# SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4): busy cycles / 32 / duration = the clock the kernel held
clock = avg.get("SQ_BUSY_CYCLES", 0.0) / 32.0 / (prof_ms * 1e6) if prof_ms and avg.get("SQ_BUSY_CYCLES") else None
floor_clock_ms = floor_ms * clock_ghz / clock if clock else None
# how busy the vector units were, counted rather than modelled: quad-cycles with a VALU instruction in flight on the 1024 SIMDs
busy = avg.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (simds * avg["SQ_BUSY_CYCLES"] / 32.0) if avg.get("SQ_BUSY_CYCLES") else None
kernel = names.most_common(1)[0][0]
sfx = "%s%s_rc%g" % (args.dtype, "_mix" if args.mixture else "", args.rc)
entry = dict(atoms=args.atoms, dtype=args.dtype, rc=args.rc, mixture=bool(args.mixture), kernel=kernel[:110],
             launches=len(next(iter(sq.values()))), valu_insts_per_launch=insts, valu_trans_per_launch=trans,
             fp64_share=f64 / insts if insts else None, fp32_share=f32 / insts if insts and f32 else None,
             simds=simds, clock_ghz=clock_ghz, issue_floor_ms=floor_ms, profiled_ms=prof_ms, profiled_clock_ghz=clock,
             issue_floor_at_profiled_clock_ms=floor_clock_ms, frac_at_profiled_clock=floor_clock_ms / prof_ms if clock else None,
             valu_busy_fraction=busy, model="wave64 VALU instruction = 4 SIMD cycles, transcendental = 16",
             source="profiles/%s/valu_%s.txt (profiles/pmc_valu.sh: rocprofv3 --pmc SQ counters, passes of 8)%s"
                    % (args.tag, sfx, "; " + args.command if args.command else ""))
dst = os.path.join(root, args.tag)
os.makedirs(dst, exist_ok=True)
lines = ["rocprofv3 --kernel-trace --pmc <SQ counters, passes of 8> -- python3 bench.py %s; N = %d" % (args.command, args.atoms),
         "%s, mean per launch over %d launches" % (kernel[:110], entry["launches"])]
lines += ["%-28s %16.0f" % (k, v) for k, v in sorted(avg.items())]
lines.append("VALU issue floor = (SQ_INSTS_VALU x 4 + TRANS x 12 cycles) / %d SIMDs / %.1f GHz = %.3f ms per launch" % (simds, clock_ghz, floor_ms))
lines.append("share of VALU instructions: fp64 %.2f, fp32 %.2f" % (f64 / insts if insts else 0.0, f32 / insts if insts else 0.0))
if clock:
    lines.append("inside the counter pass the kernel took %.3f ms at SQ_BUSY_CYCLES / 32 / duration = %.3f GHz: issue floor at that "
                 "clock %.3f ms = %.2f of the measured time; SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x busy cycles) = %.2f"
                 % (prof_ms, clock, floor_clock_ms, floor_clock_ms / prof_ms, busy or 0.0))
open(os.path.join(dst, "valu_%s.txt" % sfx), "w").write("\n".join(lines) + "\n")
vpath = os.path.join(root, "valu.json")
data = {"entries": []}
if os.path.exists(vpath):
    old = json.load(open(vpath))
    data = old if "entries" in old else {"entries": [dict(old, rc=2.5, mixture=False)]}
key = lambda e: (int(e["atoms"]), e["dtype"], float(e.get("rc", 2.5)), bool(e.get("mixture", False)))
data["entries"] = sorted([e for e in data["entries"] if key(e) != key(entry)] + [entry], key=key)
json.dump(data, open(vpath, "w"), indent=1)
print("\n".join(lines[-3:]))
