#!/bin/bash
# round 5, GPU call P: the whole -m gpu suite on the round's final build (product, bounds-checked and experiments libraries in the tree)
O=gpurun_out/r05p; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -q --timeout 600 --durations=15 > $O/pytest.log 2>&1; rc=$?
tail -25 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
