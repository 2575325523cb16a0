"""Per-launch means of the SQ counters of ONE kernel (name regular expression) from the rocprofv3 --pmc passes
profiles/pmc_valu.sh leaves under gpurun_out/valu_<tag>_...: the build kernels' side of a counter run.
usage: kernel_counters.py OUT_DIR 'k_typed_build<' """
import collections, csv, glob, os, re, sys
out, pat = sys.argv[1], re.compile(sys.argv[2])
sq, names, dur = collections.defaultdict(list), collections.Counter(), []
for sub in sorted(glob.glob(os.path.join(out, "pmc_sq*"))):
    if not os.path.isdir(sub):
        continue
    for f in glob.glob(os.path.join(sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names[r["Kernel_Name"]] += 1
    for f in glob.glob(os.path.join(sub, "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
assert sq, "no kernel matches"
for k in names:
    print(k[:140])
m = {k: sum(v) / len(v) for k, v in sq.items()}
for k in sorted(m):
    print("%-28s %16.0f" % (k, m[k]))
ms = sum(dur) / len(dur)
print("mean duration inside the counter passes: %.3f ms over %d launches" % (ms, len(dur)))
if "SQ_BUSY_CYCLES" in m:
    clk = m["SQ_BUSY_CYCLES"] / 32 / (ms * 1e-3) / 1e9
    print("clock = SQ_BUSY_CYCLES / 32 / duration = %.3f GHz" % clk)
    if "SQ_INSTS_VALU" in m:
        floor = (m["SQ_INSTS_VALU"] * 4 + 12 * (m.get("SQ_INSTS_VALU_TRANS_F32", 0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0))) / 1024 / (clk * 1e9) * 1e3
        print("VALU issue floor at that clock: %.3f ms = %.2f of the measured time" % (floor, floor / ms))
    if "SQ_ACTIVE_INST_VALU" in m:
        print("VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x busy cycles / 32 x 4 ...) : %.2f" % (m["SQ_ACTIVE_INST_VALU"] * 4 / (m["SQ_BUSY_CYCLES"] / 32 * 1024)))
    if "SQ_ACTIVE_INST_LDS" in m:
        print("LDS busy  = SQ_ACTIVE_INST_LDS x 4 / (1024 x busy cycles / 32): %.2f" % (m["SQ_ACTIVE_INST_LDS"] * 4 / (m["SQ_BUSY_CYCLES"] / 32 * 1024)))
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
        print("LDS bank-conflict cycles / LDS index-active cycles: %.2f" % (m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)))
