#!/bin/bash
# round 4, GPU call B: first run of the transposed build (k_brick_build_t): neighbour-set tests, then A/B against k_brick_build
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu --timeout 300 -k "neighbour or rebuilds" > $O/pytest_sets.log 2>&1; echo "sets rc=$?"; tail -5 $O/pytest_sets.log
for V in tb old; do
  if [ $V = old ]; then export EMDEE_NO_TBUILD=1; else unset EMDEE_NO_TBUILD; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 > $O/ab_$V.json 2> $O/ab_$V.err; echo "bench $V rc=$?"
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/abd_$V.json 2> $O/abd_$V.err
done
python - <<'PY'
import json
for f in ("ab_tb", "ab_old", "abd_tb", "abd_old"):
    try:
        d = json.loads(open("gpurun_out/r04b/%s.json" % f).read().strip().splitlines()[-1])
        k = d["kernels_ms"]; rb = k["rebuild(bin+sort+nbr_build)"]
        print("%-8s %8.1f steps/s %7.3f ms/step force %6.3f frac %.3f rebuild %6.3f ms x %d  E/N %.6f pairs %s" % (f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], rb[0] / max(rb[1], 1), rb[1], d["energy_per_atom"]["potential"], d.get("pairs_in_cutoff")))
    except Exception as e:
        print(f, "ERR", e, open("gpurun_out/r04b/%s.err" % f).read()[-600:])
PY
