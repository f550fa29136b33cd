#!/bin/bash
out=gpurun_out/r05_l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc
B="python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 10 --warmup 2 --lanes 1"
for v in base c4 c6; do
  lib=$L/libtaco_hip.so; [ $v != base ] && lib=$L/libtaco_hip_$v.so
  TACO_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_$v -o ks --output-format csv -- $B > $out/ks_$v.log 2>&1
  cp $out/ks_$v/*kernel_stats.csv $out/ks_$v.csv; rm -rf $out/ks_$v
  echo "== $v $(grep -o '"ms_per_step": [0-9.]*' $out/ks_$v.log)"
  python - <<PY
import csv
for r in list(csv.DictReader(open('$out/ks_$v.csv')))[:12]:
    print("  %-70s %4s %10.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
TACO_LIB=$L/libtaco_hip_trace.so timeout 200 python tools/trace_chain.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|warnings.warn" | tee $out/chain_timeline.txt
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "mid_size or C4_shard or collective" 2>&1 | tail -3 | tee $out/pytest_train.txt
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err; cat $out/train_step.json | head -c 1500
