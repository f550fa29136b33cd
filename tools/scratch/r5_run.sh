#!/bin/bash
# round 5, GPU call H: k_bigru_ks (one exchange per direction and step) vs k_bigru_oct
out=gpurun_out/r05_h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest -x -q -m gpu "tests/test_gpu_decoder_xcd.py::test_post_net_scan_spread_over_the_chip" -s > $out/pytest_h.txt 2>&1; echo "pytest rc=$?" >> $out/pytest_h.txt
{ for p in 1 14; do python tools/trace_bigru.py 32 512 $p; python tools/trace_bigru.py 32 512 $p; done; python tools/trace_bigru.py 20 512 14; } 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|warnings.warn" > $out/scan_timeline.txt
grep -E "passed|failed|rc=|Error|assert|rror" $out/pytest_h.txt | tail; cat $out/scan_timeline.txt
