out=gpurun_out/r04_o; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_two_ranks.py -x -q -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; grep -E "six-product|default engine|split-bf16 vs|passed|failed|Error" $out/pytest.log | head -20
timeout 120 ./tools/time_train_native > $out/time_train_native.txt 2>&1; tail -12 $out/time_train_native.txt
