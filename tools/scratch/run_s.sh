#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
{
python -m pytest tests/test_gpu_e2e.py -x -q -k "fused or pointwise" 2>&1 | tail -3
python tools/scratch/time_ff.py
TACO_LIB=/root/repo/tools/scratch/libtaco_prev.so python tools/scratch/time_ff.py
TACO_LIB=/root/repo/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip_trace.so python tools/trace_chain.py
} > gpurun_out/ab/chain2.txt 2>&1
