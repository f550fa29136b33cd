#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/ab/full_gpu.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/ab/full_gpu.txt 2>&1
