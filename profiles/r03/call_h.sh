#!/bin/bash
# round 3, GPU call H: the whole -m gpu suite on the final build, then one bench line per BASELINE configuration
O=gpurun_out/r03h; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED" $O/pytest.log | tail -8
step timeout -k 10 200 python profiles/dd_rank_proxy.py > $O/dd_rank_proxy.txt 2>&1
grep -v amdgpu $O/dd_rank_proxy.txt
step timeout -k 10 900 bash profiles/collect_configs.sh r03 > $O/collect_configs.log 2>&1
tail -22 $O/collect_configs.log
