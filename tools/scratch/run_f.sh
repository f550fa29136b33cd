out=gpurun_out/r04_f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/time_front.py 2>&1 | grep -v amdgpu.ids | tee $out/time_front.txt
for pr in 1 3; do TACO_FRONT_PRIO=$pr TACO_FRONT_DELAY=0 TACO_LIB=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_trace.so timeout 200 python tools/trace_front.py 2>&1 | grep -v amdgpu.ids | grep -v "pool + planes [0-9]\{6,\}" | tee -a $out/front_timeline.txt; done
