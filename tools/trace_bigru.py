#!/usr/bin/env python
"""Phase timeline of k_bigru_xcd (group 0 / member 0 / thread 0 shader-clock stamps, steps 8-15): python tools/trace_bigru.py [B] [T]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
PERSIST = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # 1: k_bigru_duo (default); 8: k_bigru_xcd, one 8-wave workgroup per CU; 9: two 4-wave workgroups per CU
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
fn(); torch.cuda.synchronize()
m.decoder_trace(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
tr = m.decoder_trace(True, read=True, scan=True); m.decoder_trace(False)
if PERSIST == 1:
    names = ["gates F (+publish)", "collect h'(B) + barrier", "request + gates B (+publish)", "collect r*h(F) + barrier", "request + cand F (+publish, store)",
             "collect r*h(B) + barrier", "request + cand B (+publish, store)", "collect h'(F) + barrier"]
    d = np.diff(tr[:, :9], axis=1).astype(np.float64)
else:
    names = ["gates pass+reduce", "gates epilogue+publish", "poll r*h", "barrier", "cand pass+reduce", "cand epilogue+publish+store", "poll h'", "x-part fetch issue", "barrier"]
    d = np.diff(tr[:, :10], axis=1).astype(np.float64)
step = np.median((tr[1:, 0] - tr[:-1, 0]).astype(np.float64))
print("B=%d T=%d geometry %d: %.1f us total (GEMM + scan), step = %.0f clocks" % (B, T, PERSIST, us, step))
med = np.median(d[1:], axis=0)
print("  " + "  ".join("%s %.0f" % (n, c) for n, c in zip(names, med)) + "   (clocks; ~2.1-2.3 per ns)")
