#!/bin/bash
# round 5, GPU call R: the round's evidence on the final build, part 1: profiles/collect.sh r05 (default bench line, kernel
# statistics, PMC traffic and SQ counters of the fused kernel), then the 30,000-step soak with per-window rates
O=gpurun_out/r05r; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1000 bash profiles/collect.sh r05 > $O/collect.log 2>&1
tail -4 $O/collect.log | cut -c1-400
cp -r profiles/r05/final_* $O/ 2>/dev/null; cp profiles/traffic.json profiles/valu.json $O/
step timeout -k 10 150 python profiles/soak.py > $O/soak_30000_steps.txt 2>&1
grep -v amdgpu.ids $O/soak_30000_steps.txt
