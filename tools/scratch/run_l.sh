out=gpurun_out/r04_l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q -k "bigru or stage_level or fused_cbhg or pointwise or full_size or golden or edge_lengths or deepvoice or C1 or tiny" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks -o ks --output-format csv -- python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 10 --warmup 3 --lanes 1 > $out/ks.log 2>&1; cp $out/ks/*kernel_stats.csv $out/kernel_stats.csv; rm -rf $out/ks; head -12 $out/kernel_stats.csv | cut -c1-110; grep -o '"ms_per_step": [0-9.]*' $out/ks.log
