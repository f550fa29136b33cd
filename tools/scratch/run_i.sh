out=gpurun_out/r04_i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "fused_cbhg_front or pointwise_chain or full_size_C2_parity or golden" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
timeout 120 ./tools/time_stages_native > $out/time_stages_native.txt 2>&1; cat $out/time_stages_native.txt | tail -9
timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks -o ks --output-format csv -- python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 5 --warmup 2 --lanes 1 > $out/ks.log 2>&1; cp $out/ks/*kernel_stats.csv $out/kernel_stats.csv; rm -rf $out/ks; head -13 $out/kernel_stats.csv | cut -c1-110; grep -o '"ms_per_step": [0-9.]*' $out/ks.log
