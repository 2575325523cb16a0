#!/bin/bash
# round 3 (second session), GPU call R: the round's evidence on the final build -- profiles/collect.sh r03 (bench line, rocprofv3 kernel
# stats, PMC traffic and SQ counters), DD probes, fp32 error probe, RCCL ranks on the one GPU
O=gpurun_out/r03r; mkdir -p $O
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step killed rc=$rc: $*" | tee -a $O/killed.txt; exit $rc; fi; return 0; }
step timeout -k 10 1100 bash profiles/collect.sh r03 > $O/collect.log 2>&1
tail -8 $O/collect.log
for form in "lockstep:A=1" "threestream:EMDEE_DD_LOCKSTEP=0" "inorder:EMDEE_DD_OVERLAP=0"; do
  name=${form%%:*}; envs=${form#*:}
  env $envs step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/one_domain_$name.txt 2>&1
  env $envs EMDEE_DD_NO_SHORTCUT=1 step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/one_domain_full_$name.txt 2>&1
done
EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_OVERLAP=0 EMDEE_DD_COUNT_FREE=0 EMDEE_READBACK=copy step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/one_domain_full_inorder_round2form.txt 2>&1
step timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 plain > $O/one_domain_plain.txt 2>&1
grep -H atoms $O/one_domain*.txt
step timeout -k 10 200 python profiles/dd_rank_proxy.py > $O/dd_rank_proxy.txt 2>&1
grep -v amdgpu $O/dd_rank_proxy.txt
step timeout -k 10 300 bash profiles/dd_rebuild_timeline.sh 68 $O/tl > $O/timeline_stdout.txt 2>&1; tail -3 $O/timeline_stdout.txt
step timeout -k 10 200 python tests/probe_fp32_errors.py > $O/fp32_error_probe.txt 2>&1; tail -12 $O/fp32_error_probe.txt
