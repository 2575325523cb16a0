#!/bin/bash
# round 3 (second session), GPU call AL: the flush loop of the build kernel left as written (EMDEE_PLAIN_LOOP) against the compiler's
# interleaved version ("noplain" = -DEMDEE_NO_PLAIN_LOOPS), same box, alternating; parity suite first
O=gpurun_out/r03al; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity2.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash profiles/ab_libs.sh $O "base noplain base noplain" --steps 60 --warmup 15
bash profiles/ab_libs.sh $O/f32 "base noplain" --precision f32 --steps 60 --warmup 15
bash profiles/ab_libs.sh $O/mix "base noplain" --mixture --rc 3.5 --steps 40 --warmup 10
bash profiles/ab_libs.sh $O/drv "base noplain" --steps 20 --warmup 5
