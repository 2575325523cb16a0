#!/bin/bash
# round 5, GPU call AH: is the skin (0.3 sigma since round 1) still the best trade between list length and rebuild frequency?  headline box, both bench forms
O=gpurun_out/r05ah; mkdir -p $O
for skin in 0.30 0.25 0.20 0.35 0.30; do
  for form in "100 20" "20 5"; do set -- $form
    timeout -k 10 300 python bench.py --no-cpu-baseline --skin $skin --steps $1 --warmup $2 > $O/b.json 2> $O/b.err
    python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); k=d['kernels_ms']; rb=k['rebuild(bin+sort+nbr_build)']
print('skin $skin steps $1:', round(d['value'],1), 'steps/s, fused launch', round(d['roofline']['avg_launch_ms'],4), 'ms, rebuild', round(rb[0]/max(rb[1],1),3), 'ms x', rb[1], ' capacity', d['neighbor_list']['capacity'])"
  done
done
