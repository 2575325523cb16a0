#!/bin/bash
# A/B of library variants built by profiles/build_variant.sh on the same box, same process conditions:
#   bash profiles/ab_libs.sh OUTDIR "base opp abl1 ..." [bench.py arguments]      ("base" = the in-tree library)
O=$1; NAMES=$2; shift 2; ARGS=${@:---steps 40 --warmup 10}
mkdir -p $O
for V in $NAMES; do
  if [ "$V" = base ]; then unset EMDEE_HIP_LIB; else export EMDEE_HIP_LIB=$PWD/emdee.jl_amd/variants/libemdee_hip_$V.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline $ARGS > $O/ab_$V.json 2> $O/ab_$V.err; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "variant $V killed rc=$rc"; exit $rc; fi
  python - "$V" "$O" <<'PY'
import json, sys
v, o = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open("%s/ab_%s.json" % (o, v)).read().strip().splitlines()[-1])
    k = d["kernels_ms"]; rb = k["rebuild(bin+sort+nbr_build)"]; f = k["lj_force_nbr"]
    print("%-10s %8.1f steps/s  %7.3f ms/step  force %6.3f ms  frac %.3f  rebuild %6.3f ms x %d  E/N %.5f" % (
        v, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0] / max(rb[1], 1), rb[1],
        d["energy_per_atom"]["potential"]))
except Exception as e:
    print(v, "ERR", e, open("%s/ab_%s.err" % (o, v)).read()[-400:])
PY
done
