#!/bin/bash
out=gpurun_out/r05_p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "gradients_match_autograd or full_reference_widths or mid_size or C4_shard or split_bf16_training or train_step_matches or long_input" 2>&1 | tail -8 | tee $out/pytest_train.txt
if grep -q "failed\|error" $out/pytest_train.txt; then exit 0; fi
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err; grep -o '"ms_per_step": [0-9.]*' $out/train_step.json | head -1; grep -o '"phase_ms": {[^}]*}' $out/train_step.json
timeout 300 python tools/bench_train.py --exact-wgrad 2 > $out/train_step_w2.json 2>> $out/train.err; grep -o '"ms_per_step": [0-9.]*' $out/train_step_w2.json | head -1
timeout 400 rocprofv3 --kernel-trace --stats -d $out/tks -o tks --output-format csv -- python tools/bench_train.py --steps 4 --warmup 1 > $out/tks.log 2>&1
cp $out/tks/*kernel_stats.csv $out/train_kernel_stats.csv 2>/dev/null; rm -rf $out/tks
python - <<PY
import csv
for r in list(csv.DictReader(open('$out/train_kernel_stats.csv')))[:22]:
    print("  %-80s %4s %10.1f  %9.1f" % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs'])/6e3))
PY
