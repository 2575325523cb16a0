#!/bin/bash
# round 4, GPU call M: the whole -m gpu suite on the current build, then the round's evidence (profiles/collect.sh r04, collect_configs.sh r04)
O=gpurun_out/r04m; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 --durations=12 > $O/pytest.log 2>&1
tail -18 $O/pytest.log
step timeout -k 10 1100 bash profiles/collect.sh r04 > $O/collect.log 2>&1
tail -4 $O/collect.log | cut -c1-400
step timeout -k 10 1000 bash profiles/collect_configs.sh r04 > $O/collect_configs.log 2>&1
tail -18 $O/collect_configs.log
cp gpurun_out/collect_r04/configs.jsonl $O/configs.jsonl; cp profiles/target_box_1gpu.json $O/; cp -r profiles/r04/final_* $O/ 2>/dev/null; cp profiles/traffic.json profiles/valu.json $O/
