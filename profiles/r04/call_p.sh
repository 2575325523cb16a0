#!/bin/bash
# round 4, GPU call P: is k_brick_build bound by its LDS candidate reads?  ablations: half the reads (8), every read twice (16) -- lists wrong, timing only
O=$PWD/gpurun_out/r04p; mkdir -p $O
bash profiles/ab_libs.sh $O "base babl8 babl16" --steps 12 --warmup 3 --rebuild-every 2
