#!/bin/bash
# round 5, GPU call O: SQ counters of the headline configuration (fp64, rc = 2.5): the fused step kernel's entry of
# profiles/valu.json again on this build, and the neighbour build's side of the same passes
rm -rf gpurun_out/valu_r05_f64_2.5_0
bash profiles/pmc_valu.sh r05 f64 2.5 0 2>&1 | tail -4
python3 profiles/kernel_counters.py gpurun_out/valu_r05_f64_2.5_0 'k_brick_build<' | tee gpurun_out/valu_r05_f64_2.5_0/brick_build_counters.txt
find gpurun_out/valu_r05_f64_2.5_0 -name "*.csv" -size +2M -delete
