#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
{
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "reproduc or determin or rerun or split_bf16" 2>&1 | tail -5
python tools/bench_train.py --deterministic 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deterministic', d['ms_per_step'], d.get('phase_ms'))"
python tools/bench_train.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d.get('phase_ms'))"
} > gpurun_out/ab/det.txt 2>&1
