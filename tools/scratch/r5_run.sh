#!/bin/bash
out=gpurun_out/r05_final; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee -a $out/pytest_gpu.txt
timeout 200 python bench.py --no-cpu-baseline --no-companions --steps 20 --warmup 4 > $out/bench_C2_quick.json 2> $out/bench.err; grep -o '"ms_per_step": [0-9.]*' $out/bench_C2_quick.json | head -1; grep -o '"traffic": [0-9.a-z]*' $out/bench_C2_quick.json | head -1
