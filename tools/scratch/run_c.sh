out=gpurun_out/r04_c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -s -k "fused_cbhg_front or pointwise_chain" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
TACO_LIB=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_trace.so timeout 200 python tools/trace_front.py 2>&1 | grep -v amdgpu.ids | tee $out/front_timeline.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks -o ks --output-format csv -- python bench.py --no-cpu-baseline --no-companions --no-stage-timing --steps 5 --warmup 2 --lanes 1 > $out/ks.log 2>&1; cp $out/ks/*kernel_stats.csv $out/kernel_stats.csv; rm -rf $out/ks; head -16 $out/kernel_stats.csv | cut -c1-120
