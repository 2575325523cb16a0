#!/bin/bash
# round 3 (second session), GPU call Z: parity suites after the read-back switch became a per-call look at the environment
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_dd.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --cells 6 --steps 200 > $O/b864.json 2>$O/b864.err; EMDEE_READBACK=copy timeout -k 10 300 python bench.py --no-cpu-baseline --cells 6 --steps 200 > $O/b864c.json 2>$O/b864c.err
python -c "
import json
for n in ('b864','b864c'):
    d=json.loads(open('gpurun_out/r03z/%s.json'%n).read().strip().splitlines()[-1]); print(n, d['value'])"
