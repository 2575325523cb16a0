#!/bin/bash
# round 5, GPU call C: whole -m gpu suite on the build with the rebuild in the engines' own order (fewer launches), then the
# one-domain overhead probe against the round-4 library on the same box, the kernel timeline of one rebuild, the rank proxy
O=gpurun_out/r05c; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
R04=$PWD/emdee.jl_amd/variants/libemdee_hip_r04.so
for rep in 1 2; do
for lib in new r04; do
  for form in "lockstep:A=1" "inorder:EMDEE_DD_OVERLAP=0"; do
    name=${form%%:*}; envs=${form#*:}
    if [ $lib = r04 ]; then export EMDEE_HIP_LIB=$R04; else unset EMDEE_HIP_LIB; fi
    env $envs EMDEE_DD_NO_SHORTCUT=1 timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/one_domain_full_${name}_${lib}_$rep.txt 2>&1 || exit 1
    echo "full_${name}_${lib}_$rep $(grep atoms $O/one_domain_full_${name}_${lib}_$rep.txt)"
  done
done
done
unset EMDEE_HIP_LIB
timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 plain > $O/one_domain_plain.txt 2>&1; grep atoms $O/one_domain_plain.txt
timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/one_domain_shortcut_lockstep.txt 2>&1; echo "shortcut_lockstep $(grep atoms $O/one_domain_shortcut_lockstep.txt)"
timeout -k 10 300 bash profiles/dd_rebuild_timeline.sh 68 $O/ddtl > $O/ddtl.log 2>&1; tail -45 $O/ddtl.log
