#!/bin/bash
# round 3 (second session), GPU call AG: the whole -m gpu suite and smoke() on the round's last commit, the default bench line
O=gpurun_out/r03ag; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python -c "
import json
d=json.loads(open('gpurun_out/r03ag/bench_driver.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
