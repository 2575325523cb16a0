"""Turns the raw rocprofv3 output of profiles/collect.sh into the small files committed under profiles/."""
import collections
import csv
import glob
import json
import os
import re
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


# k_brick<real, Shape, THREADS, G, MODE = 3 (force + integrator), BITMASK = 1 [, single-species fast path]>
fused = lambda k: re.search(r"k_brick<.*?, 3, 1[,>]", k) is not None
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
    traffic.update(rc=float(bench["config"].get("rc", 2.5)), mixture=bool(bench["config"].get("mixture", False)))
    tpath = os.path.join(root, "traffic.json")
    data = {"entries": []}
    if os.path.exists(tpath):
        old = json.load(open(tpath))
        data = old if "entries" in old else {"entries": [dict(old, rc=2.5, mixture=False)]}
    key = lambda e: (int(e["atoms"]), e["dtype"], float(e.get("rc", 2.5)), bool(e.get("mixture", False)))
    data["entries"] = sorted([e for e in data["entries"] if key(e) != key(traffic)] + [traffic], key=key)
    json.dump(data, open(tpath, "w"), indent=1)
    print("traffic: %.1f B/atom per launch" % traffic["bytes_per_atom"])

# ---- SQ instruction counters of the fused kernel: how close the kernel is to the VALU issue limit ----------
sq = collections.defaultdict(list)
for sub in ("pmc_sq1", "pmc_sq2"):
    for f in glob.glob(os.path.join(out, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if fused(r["Kernel_Name"]):
                sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
# duration of the same launches inside the counter pass, and the SQ clock they ran at: SQ_BUSY_CYCLES is summed over
# the 32 shader engines (8 XCDs x 4), so busy cycles / 32 / duration = the engine clock while the kernel was running
# (checked on the streaming kernels of the same pass: 2.25-2.3 GHz; the fp64 pair kernel holds ~1.9 GHz)
prof_ns = []
for f in glob.glob(os.path.join(out, "pmc_sq1", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if fused(r["Kernel_Name"]):
            prof_ns.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if sq:
    avg = {k: sum(v) / len(v) for k, v in sq.items()}
    lines = ["rocprofv3 --kernel-trace --pmc <SQ counters, two passes of 8> -- python3 bench.py --steps 12 --warmup 3; N = %d" % N,
             "fused lj_force_nbr kernel (k_brick MODE=3), mean per launch over %d launches" % len(next(iter(sq.values())))]
    lines += ["%-28s %16.0f" % (k, v) for k, v in sorted(avg.items())]
    # one wave64 VALU instruction holds its SIMD for 4 cycles (16 lanes/clk, fp64 FMA/MUL/ADD included);
    # v_rcp_f64 and friends (TRANS) run at a quarter of that rate: 16 cycles
    simds, clock_ghz = 256 * 4, 2.4
    insts, trans = avg.get("SQ_INSTS_VALU", 0.0), avg.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
    floor_ms = (insts * 4 + trans * 12) / simds / (clock_ghz * 1e6)
    f64 = sum(avg.get("SQ_INSTS_VALU_%s_F64" % k, 0.0) for k in ("FMA", "MUL", "ADD", "TRANS"))
    lines.append("VALU issue floor = (SQ_INSTS_VALU x 4 + TRANS_F64 x 12 cycles) / %d SIMDs / %.1f GHz = %.3f ms per launch" % (simds, clock_ghz, floor_ms))
    lines.append("fp64 share of VALU instructions: %.2f" % (f64 / insts if insts else 0.0))
    prof_ms = sum(prof_ns) / len(prof_ns) * 1e-6 if prof_ns else None
    clock = avg.get("SQ_BUSY_CYCLES", 0.0) / 32.0 / (prof_ms * 1e6) if prof_ms else None
    floor_clock_ms = floor_ms * clock_ghz / clock if clock else None
    if clock:
        lines.append("inside this counter pass the kernel took %.3f ms at SQ_BUSY_CYCLES / 32 / duration = %.3f GHz: issue floor "
                     "at that clock %.3f ms = %.2f of the measured time" % (prof_ms, clock, floor_clock_ms, floor_clock_ms / prof_ms))
    open(os.path.join(dst, "final_pmc_sq.txt"), "w").write("\n".join(lines) + "\n")
    ventry = dict(source="profiles/%s/final_pmc_sq.txt (profiles/collect.sh)" % tag, atoms=int(N), dtype=bench["dtype"],
                  rc=float(bench["config"].get("rc", 2.5)), mixture=bool(bench["config"].get("mixture", False)),
                  kernel="k_brick<..., BRICK_STEP, 1> (lj_force_nbr with the fused velocity-Verlet update)",
                  valu_insts_per_launch=insts, valu_trans_per_launch=trans, fp64_share=f64 / insts if insts else None,
                  simds=simds, clock_ghz=clock_ghz, issue_floor_ms=floor_ms, profiled_ms=prof_ms, profiled_clock_ghz=clock,
                  issue_floor_at_profiled_clock_ms=floor_clock_ms,
                  frac_at_profiled_clock=floor_clock_ms / prof_ms if clock else None,
                  model="wave64 VALU instruction = 4 SIMD cycles, TRANS_F64 = 16")
    vpath = os.path.join(root, "valu.json")
    vdata = {"entries": []}
    if os.path.exists(vpath):
        vold = json.load(open(vpath))
        vdata = vold if "entries" in vold else {"entries": [dict(vold, rc=2.5, mixture=False)]}
    vkey = lambda e: (int(e["atoms"]), e["dtype"], float(e.get("rc", 2.5)), bool(e.get("mixture", False)))
    vdata["entries"] = sorted([e for e in vdata["entries"] if vkey(e) != vkey(ventry)] + [ventry], key=vkey)
    json.dump(vdata, open(vpath, "w"), indent=1)
    print("valu issue floor: %.3f ms per launch" % floor_ms)
print(json.dumps({k: bench[k] for k in ("value", "ms_per_step", "roofline", "step_roofline", "cpu_baseline") if k in bench})[:1500])
