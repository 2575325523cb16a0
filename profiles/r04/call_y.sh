#!/bin/bash
# round 4, GPU call Y: the reference-shaped operator on its own (profiles/operator_path.py) on the last build
O=gpurun_out/r04y; mkdir -p $O
timeout -k 10 400 python profiles/operator_path.py > $O/operator_path.txt 2>&1; echo rc=$?; tail -8 $O/operator_path.txt
