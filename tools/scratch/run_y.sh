#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
{
for v in det1024 default det256; do
  if [ $v = default ]; then unset TACO_LIB; else export TACO_LIB=/root/repo/tools/scratch/libtaco_$v.so; fi
  python tools/bench_train.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d.get('phase_ms'))"
done
} > gpurun_out/ab/det3.txt 2>&1
