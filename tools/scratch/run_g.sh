out=gpurun_out/r04_g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -s -k "argmax_is_compared" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; grep -v "^$" $out/pytest.log | tail -14
timeout 600 python bench.py --steps 20 --warmup 4 > $out/bench_C2.json 2> $out/bench_C2.err; echo "bench rc=$?"; tail -c 300 $out/bench_C2.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_g/bench_C2.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in d["roofline"]["stages"].items(): print(k, {a:b for a,b in v.items() if a in ("ms_alone_eager","feed_forward_ms","scan_ms")}, v.get("mfma_bf16",{}).get("frac"))
print(json.dumps(d["companions"].get("train_step_c4_shard"))[:1500])
print({k: v.get("mel_frames_per_s") for k, v in d["companions"].items() if isinstance(v, dict)})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
timeout 600 python bench.py --workload C4 --steps 10 --warmup 2 > $out/bench_C4.json 2> $out/bench_C4.err; echo "bench C4 rc=$?"; tail -c 300 $out/bench_C4.err; head -c 1800 $out/bench_C4.json
