#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
{
python -m pytest tests/test_gpu_e2e.py -x -q -k "fused or pointwise" 2>&1 | tail -5
python tools/scratch/time_ff.py
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('stages'))"
} > gpurun_out/ab/head.txt 2>&1
