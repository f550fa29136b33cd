"""Phase timeline of one k_pointwise_chain workgroup (csrc/taco_chain.h) in shader clocks, post-net and encoder at the C2 shapes.  Needs a
library built with -DTACO_TRACE:  cd multi-speaker-tacotron-tensorflow_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared
-fPIC -DTACO_TRACE -o libtaco_hip_trace.so taco_lib.hip;  then  TACO_LIB=.../libtaco_hip_trace.so python tools/trace_chain.py [B T_in T_mel]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
B, T_in, T_mel = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 128, 512)
hp = taco_amd.hparams.copy(max_iters=T_mel // 4)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
L.taco_debug_read_trace_chain.restype = C.c_int
ids = torch.randint(2, 80, (B, T_in), dtype=torch.int32, device="cuda"); ids[:, -1] = 1
lens = torch.full((B,), T_in - 1, dtype=torch.int32, device="cuda")
mel = torch.rand(B, T_mel, hp.num_mels, device="cuda")
L.taco_debug_set_skip_scans(m._handle, 1)
def show(name, fn, hidden, passes):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    assert L.taco_debug_read_trace_chain(buf) == 0
    t = np.array(buf[:64], dtype=np.int64)
    rel = t - t[0]
    print(name, "entry: partial sums -> planes %d | proj_2 products %d | epilogue %d | barrier %d" % (rel[1], rel[2] - rel[1], rel[3] - rel[2], rel[4] - rel[3]))
    k = 5
    for li in range(hidden):
        print("  layer %d: products %6d | epilogue %5d | wait for the other waves %5d | planes + barrier %5d" % (
            li, rel[k] - rel[k - 1], rel[k + 1] - rel[k], rel[k + 2] - rel[k + 1], rel[k + 3] - rel[k + 2]))
        k += 4
    npass = 0
    while k + 1 < 64 and rel[k + 1] > rel[k] > rel[k - 1] and rel[k] - rel[k - 1] < 10 ** 7 and npass < passes:      # (later slots may hold another launch's stamps)
        print("  projection pass: products %6d | stores %5d" % (rel[k] - rel[k - 1], rel[k + 1] - rel[k]))
        k += 2; npass += 1
    print("  total %d clocks" % rel[k - 1])
def show_head():
    L.taco_debug_read_trace_head.restype = C.c_int
    buf = (C.c_longlong * 16)()
    assert L.taco_debug_read_trace_head(buf) == 0
    t = np.array(buf[:16], dtype=np.int64)
    n = 1
    while n < 16 and t[n] > t[n - 1]: n += 1
    # stamps: entry, rows staged, then per pass "products done" (+ "stores issued" for a pass whose stores are not interleaved into the next
    # pass's product loop: the last one), tail columns done
    print("linear head (k_head_sweep): stamps (clocks since entry) %s; total %d" % (" ".join(str(int(v - t[0])) for v in t[1:n]), int(t[n - 1] - t[0])))
try:
    show("post-net", lambda: m.postnet(mel), 5, 2)      # 6H = 768 columns at W = 256: an odd group alone, then a pair
    show_head()
    show("encoder", lambda: m.encoder(ids, lens), 4, 3)           # 768 columns at W = 128: three pairs
finally:
    L.taco_debug_set_skip_scans(m._handle, 0)
