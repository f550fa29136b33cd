#!/bin/bash
# round 5, GPU call C: decoder body templated on protocol / trace; remaining training tests
out=gpurun_out/r05_c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_decoder_xcd.py tests/test_gpu_train.py::test_whole_chip_bigru_scans_forward_tape_and_backward tests/test_gpu_train.py::test_gradients_match_autograd tests/test_gpu_train.py::test_gradients_at_full_reference_widths \
  tests/test_gpu_train.py::test_C4_shard_shape_forward_and_properties tests/test_gpu_train.py::test_deepvoice_multispeaker_training_gradients tests/test_gpu_train.py::test_training_forward_on_the_persistent_kernels tests/test_gpu_train.py::test_rnn_decoder_test_mode_on_the_persistent_kernel -s > $out/pytest_c.txt 2>&1; echo "pytest rc=$?" >> $out/pytest_c.txt
timeout 300 python tools/time_decoder.py C2:8 C5 C1 --json $out/decoder_timeline.json 2>&1 | grep -v amdgpu.ids > $out/decoder_timeline.txt
timeout 300 python bench.py --no-cpu-baseline --no-companions --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench_C2.err
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err
grep -E "passed|failed|rc=|conv-tap|encoder scan|other side|Error" $out/pytest_c.txt | tail -20; cat $out/decoder_timeline.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_c/train_step.json")); print("train", d["ms_per_step"], d["phase_ms"], d["launch"])
d=json.load(open("gpurun_out/r05_c/bench_C2.json")); print("C2", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v["ms_alone_eager"], v.get("feed_forward_ms"), v.get("scan_ms")) for k,v in d["roofline"]["stages"].items()})
PY
