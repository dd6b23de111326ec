#!/bin/bash
# round 5, first GPU call: the new bench tests, the driver's bench command (compact line), the convolution stall counters
OUT=gpurun_out/r05_first
mkdir -p $OUT
python -m pytest tests/test_bench_line.py -q -m gpu -x 2>&1 | tail -5 > $OUT/bench_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_detail.json > $OUT/bench_line.json 2> $OUT/bench.err
wc -c $OUT/bench_line.json
bash tools/conv_stalls.sh $OUT/stalls > $OUT/stalls.log 2>&1
tail -5 $OUT/bench_tests.log; cat $OUT/bench_line.json; tail -40 $OUT/stalls.log
