#!/bin/bash
# round 4, GPU call H: debug of the row sample check (full-size box) + PMC traffic of the fp32 and the rc = 3.5 mixture configurations
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 300 python profiles/tools/dbg_rows.py 136 > $O/dbg_rows.log 2>&1; tail -14 $O/dbg_rows.log
timeout -k 10 500 bash profiles/pmc_traffic.sh r04 f32 2.5 0 --precision f32 > $O/traffic_f32.log 2>&1; tail -2 $O/traffic_f32.log
timeout -k 10 500 bash profiles/pmc_traffic.sh r04 f64 3.5 1 --mixture --rc 3.5 > $O/traffic_mix.log 2>&1; tail -2 $O/traffic_mix.log
cp profiles/traffic.json $O/; cp profiles/r04/traffic_*.txt $O/ 2>/dev/null; true
