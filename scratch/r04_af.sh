#!/bin/bash
# round 4, call af: pairwise interference of the three stages
O=gpurun_out/r04_af; mkdir -p $O
timeout -k 10 300 python3 scratch/pair_probe.py 2> $O/err.log | tee $O/pairs.txt
echo done
