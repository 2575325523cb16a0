#!/bin/bash
# round 3 (second session), GPU call AK: SQ counters of the build kernel on the last build, with and without the x sub-bins
mkdir -p gpurun_out/r03ak
KERNEL_RE=k_brick_build REBUILD_EVERY=2 timeout -k 10 600 bash profiles/pmc_force.sh r03ak/sub > gpurun_out/r03ak/pmc_build_sub.txt 2>&1
EMDEE_NO_SUBBINS=1 KERNEL_RE=k_brick_build REBUILD_EVERY=2 timeout -k 10 600 bash profiles/pmc_force.sh r03ak/nosub > gpurun_out/r03ak/pmc_build_nosub.txt 2>&1
tail -28 gpurun_out/r03ak/pmc_build_sub.txt; echo ----; grep -E "SQ_INSTS_VALU |SQ_INSTS_LDS|mean|SQ_INSTS_SALU|SQ_LDS_IDX_ACTIVE|SQ_BUSY" gpurun_out/r03ak/pmc_build_nosub.txt
