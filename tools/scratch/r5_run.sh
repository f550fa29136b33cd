#!/bin/bash
out=gpurun_out/r05_i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/scratch/trace_waves.py 2>&1 | grep -v amdgpu.ids | tee $out/trace_waves.txt
