#!/bin/bash
# round 4, GPU call V: where does the near/far build (ALG 23) spend what it costs over ALG 13?  ablations (lists wrong, timing only)
O=$PWD/gpurun_out/r04v; mkdir -p $O
echo "plain rows (ALG 13): full / no emission"
bash profiles/ab_libs.sh $O/p "base nfabl2" --steps 12 --warmup 3 --rebuild-every 2
echo "near/far (ALG 23): full / no emission / one class"
EMDEE_BUILD_NEARFAR=1 bash profiles/ab_libs.sh $O/n "base nfabl2 nfabl32" --steps 12 --warmup 3 --rebuild-every 2
