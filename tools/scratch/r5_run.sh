#!/bin/bash
out=gpurun_out/r05_q; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "front or C2_parity or stage_level or golden" 2>&1 | tail -5 | tee $out/pytest_e2e.txt
if grep -q "failed\|error" $out/pytest_e2e.txt; then exit 0; fi
B="python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 10 --warmup 2 --lanes 1"
for v in base fr0; do
  lib=$L/libtaco_hip.so; [ $v != base ] && lib=$L/libtaco_hip_$v.so
  TACO_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_$v -o ks --output-format csv -- $B > $out/ks_$v.log 2>&1
  cp $out/ks_$v/*kernel_stats.csv $out/ks_$v.csv; rm -rf $out/ks_$v
  echo "== $v $(grep -o '"ms_per_step": [0-9.]*' $out/ks_$v.log)"
  python - <<PY
import csv
for r in list(csv.DictReader(open('$out/ks_$v.csv')))[:10]:
    print("  %-70s %4s %10.1f" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
