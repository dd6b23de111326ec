#!/bin/bash
OUT=gpurun_out/r05_fifth
mkdir -p $OUT
python -m pytest tests/test_priors.py tests/test_abi.py -q -m gpu -x --deselect tests/test_abi.py::test_the_alternatives_build_passes_its_own_tests_on_this_box 2>&1 | tail -60 > $OUT/priors.log
python -m pytest tests/test_abi.py -q -m gpu 2>&1 | tail -80 > $OUT/abi.log
python tools/stream_concurrency_probe.py > $OUT/streams.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/stream_concurrency_probe.py >> $OUT/streams.txt 2>&1
cat $OUT/priors.log $OUT/abi.log $OUT/streams.txt
