#!/bin/bash
# A/B sweep of the brick-kernel variants (shape, workgroup size, lanes per atom) against the
# direct (global-gather) kernels, same box, same process conditions. Usage: sweep_variants.sh [cells] [steps]
CELLS=${1:-136}; STEPS=${2:-30}
mkdir -p gpurun_out
for V in direct 0 1 2 3 4 5 6 7; do
  if [ "$V" = direct ]; then export EMDEE_PATH=direct; unset EMDEE_BRICK_VARIANT; else export EMDEE_PATH=brick EMDEE_BRICK_VARIANT=$V; fi
  timeout -k 10 300 python bench.py --cells $CELLS --steps $STEPS --warmup 10 --no-cpu-baseline > gpurun_out/sweep_$V.json 2> gpurun_out/sweep_$V.err || { echo "variant $V failed"; tail -3 gpurun_out/sweep_$V.err; continue; }
  python - "$V" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open("gpurun_out/sweep_%s.json" % v))
k = d["kernels_ms"]
print("variant %-6s steps/s %8.1f  ms/step %7.3f  force %7.3f ms  kick_drift %6.3f ms  rebuild %7.3f ms x%d  frac %.4f" % (
    v, d["value"], d["ms_per_step"], k["lj_force_nbr"][0] / max(k["lj_force_nbr"][1], 1),
    k["verlet_kick_drift"][0] / max(k["verlet_kick_drift"][1], 1),
    k["rebuild(bin+sort+nbr_build)"][0] / max(k["rebuild(bin+sort+nbr_build)"][1], 1), k["rebuild(bin+sort+nbr_build)"][1],
    d["roofline"]["frac"]))
PY
done
