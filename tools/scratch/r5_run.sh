#!/bin/bash
# round 5, GPU call E: the decoder's VALU diet (weights as register pairs, packed FMAs, swap-based reduction)
out=gpurun_out/r05_e; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest -x -q -m gpu tests/test_gpu_decoder_xcd.py tests/test_gpu_e2e.py::test_full_size_C2_parity_and_properties tests/test_gpu_e2e.py::test_golden_fixture_deepvoice tests/test_gpu_e2e.py::test_alignment_argmax_is_compared_on_every_step_of_a_full_horizon \
  tests/test_gpu_train.py::test_gradients_at_full_reference_widths tests/test_gpu_train.py::test_training_forward_on_the_persistent_kernels tests/test_gpu_train.py::test_rnn_decoder_test_mode_on_the_persistent_kernel tests/test_gpu_train.py::test_C4_shard_shape_forward_and_properties -s > $out/pytest_e.txt 2>&1; echo "pytest rc=$?" >> $out/pytest_e.txt
timeout 300 python tools/time_decoder.py C2:8 --json $out/decoder_timeline.json 2>&1 | grep -v amdgpu.ids > $out/decoder_timeline.txt
timeout 300 python bench.py --no-cpu-baseline --no-companions --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench_C2.err
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err
grep -E "passed|failed|rc=|conv-tap|Error|C2 max" $out/pytest_e.txt | tail -20; cat $out/decoder_timeline.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_e/train_step.json")); print("train", d["ms_per_step"], d["phase_ms"], d["launch"])
d=json.load(open("gpurun_out/r05_e/bench_C2.json")); print("C2", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v["ms_alone_eager"], v.get("feed_forward_ms"), v.get("scan_ms")) for k,v in d["roofline"]["stages"].items()})
PY
