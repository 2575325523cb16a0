#!/bin/bash
# round 5, GPU call T: HBM traffic of the configs[4] step kernel on the 2 x 2 x 2-brick kernels (its profiles/traffic.json entry was
# round 4's, measured on 4 x 2 x 2 bricks), then profiles/collect_configs.sh r05 again so that the mixture line quotes it
O=gpurun_out/r05t; mkdir -p $O
bash profiles/pmc_traffic.sh r05 f64 3.5 1 --mixture --rc 3.5 2>&1 | tail -6
cp profiles/traffic.json $O/traffic.json; cp profiles/r05/traffic_f64_mix_rc3.5.txt $O/ 2>/dev/null
timeout -k 10 1000 bash profiles/collect_configs.sh r05 > $O/collect_configs.log 2>&1
tail -18 $O/collect_configs.log
cp gpurun_out/collect_r05/configs.jsonl $O/configs.jsonl; cp profiles/target_box_1gpu.json $O/
