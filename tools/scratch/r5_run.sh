#!/bin/bash
# round 5, GPU call D: k_bigru_dir (directions on different waves, barrier-free, direct poll) vs k_bigru_oct
out=gpurun_out/r05_d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_decoder_xcd.py::test_post_net_scan_spread_over_the_chip tests/test_gpu_train.py::test_gradients_at_full_reference_widths -s > $out/pytest_d.txt 2>&1; echo "pytest rc=$?" >> $out/pytest_d.txt
{ for p in 1 12; do python tools/trace_bigru.py 32 512 $p; python tools/trace_bigru.py 32 512 $p; done
  python tools/trace_bigru.py 16 512 12; python tools/trace_bigru.py 8 512 13; python tools/trace_bigru.py 8 4000 13; python tools/trace_bigru.py 8 4000 11; } 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|warnings.warn" > $out/scan_timeline.txt
grep -E "passed|failed|rc=|Error|assert" $out/pytest_d.txt | tail; cat $out/scan_timeline.txt
