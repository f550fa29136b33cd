#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
TACO_POISON=nan timeout 1400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_decoder_xcd.py tests/test_gpu_ops.py tests/test_gpu_train.py -x -q > gpurun_out/ab/poison.txt 2>&1
