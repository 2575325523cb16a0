#!/bin/bash
# round 3 (second session), GPU call W: the whole -m gpu suite on the round's last commit, then a 30,000-step soak of the 10^7-atom box
O=gpurun_out/r03w; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" $O/pytest.log | tail -12
step timeout -k 10 400 python profiles/soak.py > $O/soak.txt 2>&1
tail -12 $O/soak.txt
step timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
