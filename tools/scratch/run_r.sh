#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o ff --output-format csv -- python tools/scratch/time_ff.py > /tmp/ff.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
head -12 "$f" > gpurun_out/ab/ff_stats.csv
