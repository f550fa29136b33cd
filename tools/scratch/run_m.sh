out=gpurun_out/r04_m; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_decoder_xcd.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['launch'])"
timeout 300 python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['launch'])"
