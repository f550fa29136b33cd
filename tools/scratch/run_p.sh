#!/bin/bash
# A/B: depth of the weight-fragment prefetch ring in k_pointwise_chain (CH_PF256 / CH_PF128)
cd /root/repo; mkdir -p gpurun_out/ab
{
for v in pf2 pf3 default pf5; do
  if [ $v = default ]; then unset TACO_LIB; else export TACO_LIB=/root/repo/tools/scratch/libtaco_$v.so; fi
  python tools/scratch/time_ff.py
done
unset TACO_LIB
python -m pytest tests/test_gpu_e2e.py -x -q -k "fused or chain or e2e_matches or forward" 2>&1 | tail -3
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3
} > gpurun_out/ab/chain_pf.txt 2>&1
