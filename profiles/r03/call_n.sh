#!/bin/bash
# round 3 (second session), GPU call N: lock-step overlapped halo form: DD tests (RCCL ranks included), one-domain overhead in the three forms
O=gpurun_out/r03n; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests/test_gpu_dd.py tests/test_gpu_bench.py tests/test_gpu_domain.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" $O/pytest.log | tail -8
for form in "lockstep:" "threestream:EMDEE_DD_LOCKSTEP=0" "inorder:EMDEE_DD_OVERLAP=0"; do
  name=${form%%:*}; envs=${form#*:}
  env $envs timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/one_domain_$name.txt 2>&1
  env $envs EMDEE_DD_NO_SHORTCUT=1 timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/one_domain_full_$name.txt 2>&1
done
timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 plain > $O/one_domain_plain.txt 2>&1
grep -H atoms $O/one_domain*.txt
