#!/usr/bin/env python3
"""Basic-block census of one kernel in a hipcc -S listing: for every label-delimited block the number of VALU /
SALU / LDS / global instructions, and the backward branches (loops) with the blocks they span.
usage: isa_blocks.py listing.s kernel-name-substring"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(("E:", "EE:")) or (l.startswith("_Z") and key in l and ":" in l and not l.startswith("\t")))
blocks, cur = [], {"name": "entry", "line": start, "v": 0, "s": 0, "ds": 0, "g": 0, "br": []}
for i in range(start + 1, len(lines)):
    l = lines[i]
    t = l.strip()
    if t.startswith(".Lfunc_end"): break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"name": m.group(1), "line": i, "v": 0, "s": 0, "ds": 0, "g": 0, "br": []}
        continue
    if not t or t.startswith((";", ".")): continue
    op = t.split()[0]
    if op.startswith("v_"): cur["v"] += 1
    elif op.startswith("ds_"): cur["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["g"] += 1
    elif op.startswith("s_"):
        cur["s"] += 1
        if op.startswith(("s_cbranch", "s_branch")): cur["br"].append(t.split()[1])
blocks.append(cur)
idx = {b["name"]: k for k, b in enumerate(blocks)}
tot = {k: sum(b[k] for b in blocks) for k in ("v", "s", "ds", "g")}
print("blocks %d  VALU %d  SALU %d  LDS %d  global %d" % (len(blocks), tot["v"], tot["s"], tot["ds"], tot["g"]))
loops = []
for k, b in enumerate(blocks):
    for t in b["br"]:
        if t in idx and idx[t] <= k: loops.append((idx[t], k))
loops.sort(key=lambda p: (p[0], -p[1]))
for a, z in loops:
    depth = sum(1 for (a2, z2) in loops if a2 <= a and z2 >= z and (a2, z2) != (a, z))
    v = sum(b["v"] for b in blocks[a:z + 1]); s = sum(b["s"] for b in blocks[a:z + 1]); d = sum(b["ds"] for b in blocks[a:z + 1]); g = sum(b["g"] for b in blocks[a:z + 1])
    print("%sloop %s..%s (lines %d-%d): VALU %d SALU %d LDS %d global %d" % ("  " * depth, blocks[a]["name"], blocks[z]["name"], blocks[a]["line"] + 1, blocks[z]["line"] + 1, v, s, d, g))
