#!/bin/bash
# round 3 (second session), GPU call X: the worked examples, and the full 10^7-atom benchmark box on two and three RCCL ranks sharing the one GPU
# (bench.py --gpus N --share-gpu --rccl-loopback: the driver's form of an N-GPU run, RCCL over its TCP transport)
O=gpurun_out/r03x; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 300 python examples/lj_fluid.py > $O/example_lj_fluid.txt 2>&1; tail -4 $O/example_lj_fluid.txt
step timeout -k 10 300 python examples/lj_fluid_decomposed.py 8 24 > $O/example_decomposed.txt 2>&1; tail -4 $O/example_decomposed.txt
step timeout -k 10 500 python bench.py --gpus 2 --share-gpu --rccl-loopback --steps 20 --warmup 5 --target-cells 0 --no-cpu-baseline > $O/bench_2ranks.json 2> $O/bench_2ranks.err
step timeout -k 10 500 python bench.py --gpus 3 --share-gpu --rccl-loopback --no-probe --steps 20 --warmup 5 --target-cells 0 --no-cpu-baseline > $O/bench_3ranks.json 2> $O/bench_3ranks.err
step timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_1rank.json 2> $O/bench_1rank.err
python - <<'PY'
import json
for n in ("1rank", "2ranks", "3ranks"):
    try:
        d = json.loads(open("gpurun_out/r03x/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "%.1f steps/s  %.3f ms/step  E/N %.15g %.15g" % (d["value"], d["ms_per_step"], d["energy_per_atom"]["potential"], d["energy_per_atom"]["kinetic"]),
              d["config"].get("decomposition"), d["config"].get("halo_exchange"), d.get("degraded"))
    except Exception as e:
        print(n, "ERR", e, open("gpurun_out/r03x/bench_%s.err" % n).read()[-400:])
PY
