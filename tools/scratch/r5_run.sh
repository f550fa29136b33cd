#!/bin/bash
out=gpurun_out/r05_k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q --durations=6 2>&1 | tail -14 | tee $out/pytest_e2e.txt
B="python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 10 --warmup 2 --lanes 1"
for v in base pf2 pf8; do
  lib=$L/libtaco_hip.so; [ $v != base ] && lib=$L/libtaco_hip_$v.so
  TACO_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_$v -o ks --output-format csv -- $B > $out/ks_$v.log 2>&1
  cp $out/ks_$v/*kernel_stats.csv $out/ks_$v.csv; rm -rf $out/ks_$v
  echo "== $v $(grep -o '"ms_per_step": [0-9.]*' $out/ks_$v.log)"
  python - <<PY
import csv
for r in list(csv.DictReader(open('$out/ks_$v.csv')))[:13]:
    print("  %-70s %4s %10.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
timeout 200 python bench.py --no-cpu-baseline --no-companions --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench.err; grep -o '"ms_per_step": [0-9.]*' $out/bench_C2.json | head -1
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "full_reference_widths or C4_shard or whole_chip_bigru or sync_bn or mid_size or collective" 2>&1 | tail -4 | tee $out/pytest_train.txt
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err; grep -o '"ms_per_step": [0-9.]*' $out/train_step.json | head -3
