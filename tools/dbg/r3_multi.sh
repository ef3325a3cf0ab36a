#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_bench_multi.py -x -q -m gpu > gpurun_out/r3/multi.log 2>&1; rc=$?
tail -25 gpurun_out/r3/multi.log
exit $rc
