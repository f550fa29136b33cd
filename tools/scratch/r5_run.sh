#!/bin/bash
out=gpurun_out/r05_o; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_decoder_xcd.py -x -q -k "training_forward_matches_oracle or gradients_match_autograd or C4_shard or whole_chip or engine_plan or persistent_bptt or training_forward_on_the_persistent" 2>&1 | tail -6 | tee $out/pytest.txt
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err; grep -o '"ms_per_step": [0-9.]*' $out/train_step.json | head -1; grep -o '"phase_ms": {[^}]*}' $out/train_step.json
