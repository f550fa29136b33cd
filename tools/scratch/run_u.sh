#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
timeout 800 python -m pytest tests/test_gpu_decoder_xcd.py -x -q -k "teacher or presets or manual"  > gpurun_out/ab/presets.txt 2>&1
