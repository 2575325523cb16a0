#!/bin/bash
# round 3 (second session), GPU call T: the decomposition's per-rank constants, every form on ONE box (profiles/dd_one_domain_overhead.py:
# one domain of a rank's size against the plain integrator; EMDEE_DD_NO_SHORTCUT=1 = the whole ownership path at every rebuild)
O=gpurun_out/r03t; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 dd > $O/$name.txt 2>&1; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed $name"; exit $rc; fi; printf "%-58s %s\n" "$name" "$(grep atoms $O/$name.txt)"; }
timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 plain > $O/plain.txt 2>&1; printf "%-58s %s\n" plain "$(grep atoms $O/plain.txt)"
run shortcut_lockstep A=1
run shortcut_threestream EMDEE_DD_LOCKSTEP=0
run shortcut_inorder EMDEE_DD_OVERLAP=0
run full_lockstep EMDEE_DD_NO_SHORTCUT=1
run full_threestream EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_LOCKSTEP=0
run full_inorder EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_OVERLAP=0
run full_inorder_counts EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_OVERLAP=0 EMDEE_DD_COUNT_FREE=0
run full_inorder_counts_copysync EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_OVERLAP=0 EMDEE_DD_COUNT_FREE=0 EMDEE_READBACK=copy
run full_threestream_counts_copysync EMDEE_DD_NO_SHORTCUT=1 EMDEE_DD_LOCKSTEP=0 EMDEE_DD_COUNT_FREE=0 EMDEE_READBACK=copy
timeout -k 10 200 python profiles/dd_one_domain_overhead.py 68 plain > $O/plain2.txt 2>&1; printf "%-58s %s\n" plain_again "$(grep atoms $O/plain2.txt)"
