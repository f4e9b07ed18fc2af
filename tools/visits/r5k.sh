#!/bin/bash
# round 5 visit k: accuracy of the packed-row stem on the hardware (stem alone vs fp64; reference-init stress by stem kernel, conv algorithm, seed)
mkdir -p gpurun_out/r5k
timeout 900 python tools/stem_numerics.py > gpurun_out/r5k/stem_numerics.txt 2>&1
tail -40 gpurun_out/r5k/stem_numerics.txt
