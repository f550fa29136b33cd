#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
{
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_two_ranks.py -x -q 2>&1 | tail -5
python tools/bench_train.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default (deterministic)', d['ms_per_step'], d.get('phase_ms'), d.get('engine'))"
python tools/bench_train.py --deterministic 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('atomics', d['ms_per_step'], d.get('phase_ms'))"
./tools/time_train_native
} > gpurun_out/ab/det2.txt 2>&1
