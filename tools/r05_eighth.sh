#!/bin/bash
OUT=gpurun_out/r05_eighth
mkdir -p $OUT
GPU_MAX_HW_QUEUES=8 python tools/lab/chains_threads.py --ks 1,2,3,4,6 > $OUT/threads.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/lab/chains_threads.py --workload convnet --ks 1,2,3,4,6 >> $OUT/threads.txt 2>&1
python tools/lab/chains_threads.py --ks 2,3,4 >> $OUT/threads.txt 2>&1
grep -v amdgpu.ids $OUT/threads.txt
