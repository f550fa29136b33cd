#!/bin/bash
out=gpurun_out/r05_j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 10 --warmup 2 --lanes 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks0 -o ks --output-format csv -- $B > $out/ks0.log 2>&1
cp $out/ks0/*kernel_stats.csv $out/ks_base.csv; rm -rf $out/ks0
TACO_LIB=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_rot.so timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks1 -o ks --output-format csv -- $B > $out/ks1.log 2>&1
cp $out/ks1/*kernel_stats.csv $out/ks_rot.csv; rm -rf $out/ks1
for f in ks_base ks_rot; do echo "== $f"; grep -E "k_cbhg_front|k_pointwise_chain|k_head_sweep" $out/$f.csv | cut -d, -f1-4 | cut -c1-120; done
tail -c 400 $out/ks0.log | grep -o '"ms_per_step": [0-9.]*'; tail -c 4000 $out/ks1.log | grep -o '"ms_per_step": [0-9.]*'
timeout 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5 | tee $out/pytest_train.txt
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err; grep -o '"ms_per_step": [0-9.]*' $out/train_step.json | head -3
timeout 400 rocprofv3 --kernel-trace --stats -d $out/tks -o tks --output-format csv -- python tools/bench_train.py --steps 4 --warmup 1 > $out/tks.log 2>&1
cp $out/tks/*kernel_stats.csv $out/train_kernel_stats.csv 2>/dev/null; rm -rf $out/tks
head -40 $out/train_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
