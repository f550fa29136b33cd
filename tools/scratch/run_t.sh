#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
{
for v in B0 A0 NOMFMA; do echo "== $v"; TACO_LIB=/root/repo/tools/scratch/libtaco_abl_$v.so python tools/trace_chain.py 2>&1 | grep -v amdgpu; done
echo "== full"; TACO_LIB=/root/repo/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_trace.so python tools/trace_chain.py 2>&1 | grep -v amdgpu
} > gpurun_out/ab/ablate.txt 2>&1
