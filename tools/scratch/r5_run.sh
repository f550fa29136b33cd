#!/bin/bash
# round 5, GPU call G: the whole GPU suite + smoke on the current tree; train step; C2 line with companions
out=gpurun_out/r05_g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x > $out/pytest_gpu.txt 2>&1; echo "rc=$?" >> $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" >> $out/pytest_gpu.txt 2>&1
timeout 300 python tools/bench_train.py > $out/train_step.json 2> $out/train.err
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench_C2.err
tail -12 $out/pytest_gpu.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_g/train_step.json")); print("train", d["ms_per_step"], d["phase_ms"], d["launch"])
d=json.load(open("gpurun_out/r05_g/bench_C2.json")); print("C2", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v["ms_alone_eager"], v.get("feed_forward_ms"), v.get("scan_ms")) for k,v in d["roofline"]["stages"].items()})
print({k:(v.get("forward_ms"), v.get("mel_frames_per_s"), v.get("ms_per_step")) for k,v in d["companions"].items()})
print(d["roofline"]["latency_floor_ms"])
PY
