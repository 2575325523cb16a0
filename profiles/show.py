"""Prints the few numbers of bench.py JSON lines that matter during A/B runs: python profiles/show.py a.json b.json"""
import json
import sys

for f in sys.argv[1:]:
    for line in open(f):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        k = d["kernels_ms"]
        per = lambda v: v[0] / max(v[1], 1)
        print("%-28s %8.1f steps/s %7.3f ms/step | force %.3f ms x%d | rebuild %.3f ms x%d | frac %.3f | list max %d"
              % (f.split("/")[-1], d.get("box_steps_per_sec", d["value"]), d["ms_per_step"], per(k["lj_force_nbr"]), k["lj_force_nbr"][1],
                 per(k["rebuild(bin+sort+nbr_build)"]), k["rebuild(bin+sort+nbr_build)"][1], d["roofline"]["frac"],
                 d["neighbor_list"]["max_count"]))
