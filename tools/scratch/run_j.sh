out=gpurun_out/r04_j; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_decoder_xcd.py tests/test_gpu_e2e.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $out/bench_C2.json 2> $out/bench_C2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_j/bench_C2.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in d["roofline"]["stages"].items(): print(k, {a:round(b,4) for a,b in v.items() if a in ("ms_alone_eager","feed_forward_ms","scan_ms")}, v.get("mfma_bf16",{}).get("frac"))
print({k: v.get("mel_frames_per_s") for k, v in d["companions"].items() if isinstance(v, dict)})
PY
