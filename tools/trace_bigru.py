#!/usr/bin/env python
"""Phase timeline of the whole-chip post-net scans (row / group 0, member 0, thread 0 shader-clock stamps, steps 8-15):
python tools/trace_bigru.py [B] [T] [persist] [--json out.json]   (--json appends one record to a JSON list: bench.py's latency_floor_ms reads it)"""
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
argv = list(sys.argv[1:])
jpath = None
if "--json" in argv:
    i = argv.index("--json"); jpath = argv[i + 1]; del argv[i:i + 2]
B = int(argv[0]) if len(argv) > 0 else 32
T = int(argv[1]) if len(argv) > 1 else 512
# 1: the default (k_bigru_oct from 9 to 32 rows, else k_bigru_duo); 10: k_bigru_oct wherever it fits; 11: k_bigru_duo; 8: k_bigru_xcd, one 8-wave workgroup per CU; 9: two 4-wave workgroups per CU
PERSIST = int(argv[2]) if len(argv) > 2 else 1
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
L.taco_debug_set_persistent(m._handle, PERSIST)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
x = torch.randn(B, T, 256, device="cuda") * 0.3; out = torch.empty(B, T, 512, device="cuda")
nb = L.taco_stage_workspace_bytes(m._handle, B, T) + (64 << 20)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
fn = lambda: taco_amd._lib.check(L.taco_bigru_f32(m._handle, st(), b"post_cbhg", P(x), P(None), P(None), B, T, P(out), P(ws), nb))


def timed(reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


us = timed()
L.taco_debug_set_skip_scans(m._handle, 1)
us_gemm = timed()                     # the hoisted input projection alone
L.taco_debug_set_skip_scans(m._handle, 0)
m.decoder_trace(True)
fn(); torch.cuda.synchronize()
tr = m.decoder_trace(True, read=True, scan=True); m.decoder_trace(False)
m.check_device_errors()
if PERSIST in (1, 10, 11):
    names = ["gates F (+publish)", "collect h'(B) + barrier", "request + gates B (+publish)", "collect r*h(F) + barrier", "request + cand F (+publish, store)",
             "collect r*h(B) + barrier", "request + cand B (+publish, store)", "collect h'(F) + barrier"]
    d = np.diff(tr[:, :9], axis=1).astype(np.float64)
else:
    names = ["gates pass+reduce", "gates epilogue+publish", "poll r*h", "barrier", "cand pass+reduce", "cand epilogue+publish+store", "poll h'", "x-part fetch issue", "barrier"]
    d = np.diff(tr[:, :10], axis=1).astype(np.float64)
step = float(np.median((tr[1:, 0] - tr[:-1, 0]).astype(np.float64)))
scan_us = us - us_gemm
cpu = step * T / scan_us if scan_us > 0 else 0.0
oct_ = PERSIST in (1, 10) and B <= 32 and (B > 8 or PERSIST == 10)
kernel = ("k_bigru_oct<%d>" % (4 if B > 16 else 2 if B > 8 else 1)) if oct_ else ("k_bigru_duo<%d>" % max(1, 1 << int(np.ceil(np.log2(max(1, (B + 7) // 8)))))) if PERSIST in (1, 10, 11) else "k_bigru_xcd"
print("B=%d T=%d persist %d kernel %s: %.1f us total (input projection %.1f + scan %.1f = %.3f us per step), step = %.0f clocks, %.0f clocks per us"
      % (B, T, PERSIST, kernel, us, us_gemm, scan_us, scan_us / T, step, cpu))
med = np.median(d[1:], axis=0)
print("  " + "  ".join("%s %.0f" % (n, c) for n, c in zip(names, med)) + "   (clocks)")
if jpath:
    recs = json.load(open(jpath)) if os.path.exists(jpath) else []
    import bench
    recs.append({"kernel_source_hash": bench.source_hash(), "B": B, "T": T, "persist": PERSIST, "kernel": kernel, "total_us": us, "input_projection_us": us_gemm, "scan_us": scan_us,
                 "step_clocks": step, "clocks_per_us": cpu, "phases_clocks": {n: float(c) for n, c in zip(names, med)}})
    json.dump(recs, open(jpath, "w"), indent=1)
