#!/bin/bash
# round 5, GPU call B: k_bigru_oct v2 (packed FMAs, second poll, templated protocol) + A/B variants; train step eager vs captured
out=gpurun_out/r05_b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_decoder_xcd.py::test_post_net_scan_spread_over_the_chip tests/test_gpu_decoder_xcd.py::test_encoder_scan_fast_transcendentals_against_libm_and_the_oracle \
  tests/test_gpu_train.py::test_whole_chip_bigru_scans_forward_tape_and_backward tests/test_gpu_train.py::test_gradients_match_autograd tests/test_gpu_train.py::test_gradients_at_full_reference_widths \
  tests/test_gpu_train.py::test_C4_shard_shape_forward_and_properties tests/test_gpu_train.py::test_deepvoice_multispeaker_training_gradients -s > $out/pytest_b.txt 2>&1; echo "pytest rc=$?" >> $out/pytest_b.txt
{ python tools/trace_bigru.py 32 512 1; python tools/trace_bigru.py 32 512 1
  for v in noreq2 dyn; do echo "== variant $v"; TACO_LIB=$GRAFT_REPO_ROOT/tools/scratch/libtaco_$v.so python tools/trace_bigru.py 32 512 1; done
  python tools/trace_bigru.py 16 512 1; python tools/trace_bigru.py 8 512 10; python tools/trace_bigru.py 8 512 11; } 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|warnings.warn" > $out/scan_timeline.txt
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err
timeout 300 python tools/bench_train.py --graph 1 > $out/train_step_graph.json 2>> $out/train.err
timeout 300 python bench.py --no-cpu-baseline --no-companions --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench_C2.err
grep -E "passed|failed|rc=|conv-tap|encoder scan|other side" $out/pytest_b.txt | tail -20; cat $out/scan_timeline.txt
python - <<'PY'
import json
for f in ("train_step","train_step_graph"):
    d=json.load(open("gpurun_out/r05_b/%s.json"%f)); print(f, d["ms_per_step"], d["phase_ms"], d["launch"])
d=json.load(open("gpurun_out/r05_b/bench_C2.json")); print("C2", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v["ms_alone_eager"], v.get("feed_forward_ms"), v.get("scan_ms")) for k,v in d["roofline"]["stages"].items()})
PY
