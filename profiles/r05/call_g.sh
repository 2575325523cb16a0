#!/bin/bash
# round 5, GPU call G: whole -m gpu suite on the product library (experiments behind make EXPERIMENTS=1), then the VALU issue
# floors of the fp32 kernel and of the mixture's typed kernel (SQ counters, keyed entries of profiles/valu.json)
O=gpurun_out/r05g; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --timeout 800 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 1000 bash profiles/pmc_valu.sh r05 f32 2.5 0 --precision f32 > $O/valu_f32.log 2>&1; tail -4 $O/valu_f32.log
timeout -k 10 1000 bash profiles/pmc_valu.sh r05 f64 3.5 1 --mixture --rc 3.5 > $O/valu_mix.log 2>&1; tail -4 $O/valu_mix.log
cp profiles/valu.json $O/valu.json; cp profiles/r05/valu_*.txt $O/ 2>/dev/null
