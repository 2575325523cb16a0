#!/bin/bash
# round 5, GPU call S: the round's evidence on the final build, part 2: profiles/collect_configs.sh r05 (the other BASELINE
# configurations on one GPU, the driver's form of the headline run, in-process domains)
O=gpurun_out/r05s; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1100 bash profiles/collect_configs.sh r05 > $O/collect_configs.log 2>&1
tail -18 $O/collect_configs.log
cp gpurun_out/collect_r05/configs.jsonl $O/configs.jsonl; cp profiles/target_box_1gpu.json $O/
