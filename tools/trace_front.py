"""Phase timeline of one k_cbhg_front workgroup (csrc/taco_front.h) in shader clocks, post-net and encoder at the C2 shapes.  Needs a
library built with -DTACO_TRACE:  cd multi-speaker-tacotron-tensorflow_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared
-fPIC -DTACO_TRACE -o libtaco_hip_trace.so taco_lib.hip;  then  TACO_LIB=.../libtaco_hip_trace.so python tools/trace_front.py [B T_in T_mel]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
B, T_in, T_mel = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 128, 512)
hp = taco_amd.hparams.copy(max_iters=T_mel // 4)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
L.taco_debug_read_trace_front.restype = C.c_int
if len(sys.argv) > 5: L.taco_debug_set_front(m._handle, int(sys.argv[4]), int(sys.argv[5]))      # start delay (clocks), priority of the second K half
ids = torch.randint(2, 80, (B, T_in), dtype=torch.int32, device="cuda"); ids[:, -1] = 1
lens = torch.full((B,), T_in - 1, dtype=torch.int32, device="cuda")
mel = torch.rand(B, T_mel, hp.num_mels, device="cuda")
def show(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    assert L.taco_debug_read_trace_front(buf) == 0
    t = np.array(buf[:64], dtype=np.int64)
    last = int(t[2])                       # index of the last stamp this launch wrote: slots beyond it are another launch's
    rel = t - t[0]
    print(name, "staged", rel[1], "(%d chunks in the traced workgroup)" % ((last - 4) // 4))
    k = 3
    while k + 3 < last - 1:
        print("  chunk: produce loop %6d | wait for the other waves %5d | pool + planes %5d | consume %6d" % (
            rel[k] - (rel[k - 1] if k > 3 else rel[1]), rel[k + 1] - rel[k], rel[k + 2] - rel[k + 1], rel[k + 3] - rel[k + 2]))
        k += 4
    assert k == last - 1, (k, last)
    print("  K halves met +%d, stored +%d, total %d clocks" % (rel[k] - rel[k - 1], rel[k + 1] - rel[k], rel[k + 1]))
show("post-net", lambda: m.postnet(mel))
show("encoder", lambda: m.encoder(ids, lens))
