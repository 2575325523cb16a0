#!/bin/bash
# round 3, GPU call G: DD suite after the one-domain shortcut, then the round's evidence (profiles/collect.sh r03)
O=gpurun_out/r03g; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 600 python -m pytest tests/test_gpu_dd.py -m gpu -q --timeout 600 -k "not rccl" > $O/pytest_dd.log 2>&1
grep -E "passed|failed|^FAILED" $O/pytest_dd.log | tail -5
step timeout -k 10 200 python profiles/dd_one_domain_overhead.py > $O/dd_one_domain.txt 2>&1
grep -v amdgpu $O/dd_one_domain.txt
step timeout -k 10 1000 bash profiles/collect.sh r03 > $O/collect.log 2>&1
tail -5 $O/collect.log
