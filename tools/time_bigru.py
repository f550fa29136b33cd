"""Per-step cost of the BiGRU scan (taco_bigru_f32 = hoisted x-projection GEMM + k_bigru_rows) vs T."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, taco_amd
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
import os
L.taco_debug_set_persistent(m._handle, int(os.environ.get("PERSIST", "1")))
for scope, I in (("post_cbhg", 256), ("encoder_cbhg", 128)):
    for B in (32, 8):
        prev = None
        for T in (64, 128, 256, 512, 1024):
            x = torch.randn(B, T, I, device="cuda") * 0.3; out = torch.empty(B, T, 2 * I, device="cuda")
            nb = L.taco_stage_workspace_bytes(m._handle, B, T) + (64 << 20)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            fn = lambda: taco_amd._lib.check(L.taco_bigru_f32(m._handle, st(), scope.encode(), P(x), P(None), P(None), B, T, P(out), P(ws), nb))
            us = timeit(fn)
            d = "" if prev is None else "  marginal %.2f us/step" % ((us - prev[1]) / (T - prev[0]))
            print("%-13s B=%2d T=%4d  %8.1f us  (%.2f us/step)%s" % (scope, B, T, us, us / T, d))
            prev = (T, us)
