"""Phase timeline of one k_gemm_bf3 workgroup (the linear head at C2 shape), in shader clocks: needs a library built with
-DTACO_TRACE (cd multi-speaker-tacotron-tensorflow_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DTACO_TRACE
-o libtaco_hip.so taco_lib.hip).  Slots: 0 entry, 1 arguments + first loads issued, 2 first weight group issued, then per
chunk (after the staging barrier, after staging), end of the MFMA loop, end of the epilogue.  Usage: trace_gemm.py [tile]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, taco_amd, numpy as np
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, T = 32, 512
x = torch.randn(B, T, 512, device="cuda"); out = torch.empty(B, T, 1025, device="cuda")
L.taco_debug_set_bf3(m._handle, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 3)
fn = lambda: taco_amd._lib.check(L.taco_dense_f32(m._handle, st(), b"linear", C.c_void_p(x.data_ptr()), B * T, 0, C.c_void_p(out.data_ptr())))
for _ in range(5): fn()
torch.cuda.synchronize()
buf = (C.c_longlong * 64)()
L.taco_debug_read_trace.restype = C.c_int
print("rc", L.taco_debug_read_trace(buf))
t = np.array(buf[:40], dtype=np.int64)
d = np.diff(t)
print("clocks since entry", t[:24] - t[0])
print("diffs", d[:23])
