out=gpurun_out/r04_n; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -s -k "split_bf16_training or full_reference_widths" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; grep -E "six-product|default engine|split-bf16 vs|passed|failed|Error|error" $out/pytest.log | head -20
for m in 3 4 0; do timeout 300 python tools/bench_train.py --exact-gemm $m --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode $m', round(d['ms_per_step'],2), d['phase_ms'], d['loss_without_coeff_first_last'])"; done
