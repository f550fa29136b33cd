#!/bin/bash
# round 5: the two bench lines whose CPU baseline ran into the collection's time limit (all host threads on small ops), re-run with the bounded arms
out=gpurun_out/r05_v1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
nproc
timeout 900 python bench.py --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench_C2.err
timeout 600 python bench.py --workload C1 --steps 12 --warmup 3 > $out/bench_C1.json 2>> $out/bench_C2.err
python - <<'PY'
import json
for f in ("bench_C2","bench_C1"):
    d=json.load(open("gpurun_out/r05_v1/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"]); print(json.dumps(d["cpu_baseline"])[:900])
d=json.load(open("gpurun_out/r05_v1/bench_C2.json")); print(d["roofline"]["latency_floor_ms"]); print({k:(v.get("forward_ms"), v.get("mel_frames_per_s"), v.get("ms_per_step")) for k,v in d["companions"].items()})
PY
