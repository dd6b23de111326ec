#!/bin/bash
OUT=gpurun_out/r05_third
mkdir -p $OUT
python -m pytest tests/test_priors.py tests/test_full_size.py -q -m gpu 2>&1 | tail -40 > $OUT/tests.log
python tools/stream_concurrency_probe.py > $OUT/streams.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/stream_concurrency_probe.py >> $OUT/streams.txt 2>&1
timeout 600 python tools/conditioning_probe.py --steps 0,200 > $OUT/conditioning.txt 2>/dev/null
cat $OUT/tests.log $OUT/streams.txt $OUT/conditioning.txt
