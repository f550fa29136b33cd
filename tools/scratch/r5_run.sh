#!/bin/bash
# round 5, GPU call A: the new post-net scan (k_bigru_oct) -- parity, timelines, and the C2 line with the x6 companion
out=gpurun_out/r05_a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_decoder_xcd.py::test_post_net_scan_spread_over_the_chip tests/test_gpu_decoder_xcd.py::test_engine_plan_says_which_engine_a_call_gets_and_why_not \
  tests/test_gpu_train.py::test_whole_chip_bigru_scans_forward_tape_and_backward tests/test_gpu_e2e.py::test_full_size_C2_parity_and_properties \
  "tests/test_gpu_train.py::test_training_forward_on_the_persistent_kernels" -s > $out/pytest_a.txt 2>&1; echo "pytest rc=$?" >> $out/pytest_a.txt
rm -f $out/scan_timeline.json
{ for p in 1 11; do python tools/trace_bigru.py 32 512 $p --json $out/scan_timeline.json; python tools/trace_bigru.py 16 512 $p --json $out/scan_timeline.json; done
  for p in 10 11; do python tools/trace_bigru.py 8 4000 $p --json $out/scan_timeline.json; python tools/trace_bigru.py 8 512 $p --json $out/scan_timeline.json; done; } 2>&1 | grep -v amdgpu.ids > $out/scan_timeline.txt
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench_C2.err
tail -5 $out/pytest_a.txt; cat $out/scan_timeline.txt; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_a/bench_C2.json"))
print("C2", d["value"], d["ms_per_step"], d["roofline"]["frac"])
print({k:(v.get("forward_ms"), v.get("mel_frames_per_s"), v.get("ms_per_step")) for k,v in d["companions"].items()})
print(d["companions"].get("fp32_grade_x6"))
print(d["roofline"]["stages"])
PY
