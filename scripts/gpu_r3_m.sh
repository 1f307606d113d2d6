#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3m
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r3m/log.txt 2>&1
grep -n "Error\|error\|assert\|Traceback\|raise\|line [0-9]*, in" gpurun_out/r3m/log.txt | tail -40
tail -5 gpurun_out/r3m/log.txt
