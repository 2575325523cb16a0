#!/bin/bash
# round 4, GPU call S: near/far rows (EMDEE_BUILD_NEARFAR=1, ALG 23 as round 3 left it) on this round's build + smoke()
O=$PWD/gpurun_out/r04s; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
bash profiles/ab_libs.sh $O/plain "base" --steps 100 --warmup 20
EMDEE_BUILD_NEARFAR=1 bash profiles/ab_libs.sh $O/nf "base" --steps 100 --warmup 20
bash profiles/ab_libs.sh $O/plaind "base" --steps 20 --warmup 5
EMDEE_BUILD_NEARFAR=1 bash profiles/ab_libs.sh $O/nfd "base" --steps 20 --warmup 5
