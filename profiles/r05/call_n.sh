#!/bin/bash
# round 5, GPU call N: SQ counters of configs[4] on the 2 x 2 x 2-brick kernels: the step kernel's entry of profiles/valu.json
# and the typed build's side of the same passes
bash profiles/pmc_valu.sh r05 f64 3.5 1 --mixture --rc 3.5 2>&1 | tail -5
python3 profiles/kernel_counters.py gpurun_out/valu_r05_f64_3.5_1 'k_typed_build<' | tee gpurun_out/valu_r05_f64_3.5_1/typed_build_counters.txt
find gpurun_out/valu_r05_f64_3.5_1 -name "*.csv" -size +2M -delete
