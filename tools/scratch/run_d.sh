out=gpurun_out/r04_d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "fused_cbhg_front or pointwise_chain" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 300 python tools/time_front.py 2>&1 | grep -v amdgpu.ids | tee $out/time_front.txt
for d in 0 6000; do TACO_FRONT_DELAY=$d TACO_LIB=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_trace.so timeout 200 python tools/trace_front.py 2>&1 | grep -v amdgpu.ids | tee -a $out/front_timeline.txt; done
