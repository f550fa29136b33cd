#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
timeout 800 python -m pytest tests/test_gpu_train.py -x -q -k "test_mode or persistent_kernels or surface or bptt or C4 or horizon" > gpurun_out/ab/testmode.txt 2>&1
