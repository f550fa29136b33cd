#!/usr/bin/env python
"""Phase timeline of the persistent decoder BPTT (k_decoder_bwd_xcd: group 0 / member 0 / thread 0 shader-clock stamps, steps 8-15 of the
launch) at the C4 shard shapes:  python tools/trace_bptt.py [B] [T_in] [T_out]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
from taco_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T_in = int(sys.argv[2]) if len(sys.argv) > 2 else 128
T_out = int(sys.argv[3]) if len(sys.argv) > 3 else 512
hp = taco_amd.hparams.copy(max_iters=max(200, T_out // 4))
tr = taco_amd.Trainer(hp, taco_amd.weights.random_weights(hp, 1, seed=4321))
rs = np.random.RandomState(0)
ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); L = np.full(B, T_in, np.int32)
mt, lt = rs.rand(B, T_out, hp.num_mels).astype(np.float32), rs.rand(B, T_out, hp.num_freq).astype(np.float32)
tr.forward_backward(ids, L, mt, lt, None); torch.cuda.synchronize()
mh = C.c_void_p(tr._lib.taco_train_model(tr._h))
_lib.check(tr._lib.taco_debug_decoder_trace(mh, 1, None))
tr.forward_backward(ids, L, mt, lt, None); torch.cuda.synchronize()
out = (C.c_longlong * 128)()
_lib.check(tr._lib.taco_debug_decoder_trace(mh, 1 | 4, out))
_lib.check(tr._lib.taco_debug_decoder_trace(mh, 0, None))
t = np.array(out[:], np.int64).reshape(8, 16)
names = ["fetch issue + GRU 2 a + collect d c_pre", "GRU 2 b + collect gate grads", "GRU 2 c / GRU 1 a + collect d c_pre", "GRU 1 b + collect gate grads",
         "GRU 1 c + collect d o0", "concat^T + collect d ctx", "d alpha partials + collect", "normaliser backward + query", "d q partials + collect",
         "query^T / att GRU a + collect d c_pre", "att GRU b + collect gate grads", "att GRU c + collect d z2", "prenet 2^T + collect d z1",
         "prenet 1^T (context rows)"]
d = np.diff(t[:, :15], axis=1).astype(np.float64)
step = np.median((t[1:, 0] - t[:-1, 0]).astype(np.float64))
print("B=%d T_in=%d steps=%d: step = %.0f clocks; engine %s" % (B, T_in, T_out // hp.reduction_factor, step, tr.decoder_engine_info()))
med = np.median(d[1:], axis=0)
labels = ["%2d %s" % (i + 1, names[i]) for i in range(14)]
for lab, c in zip(labels, med):
    print("  %-70s %6.0f" % (lab, c))
print("  (clocks of the shader counter; ~2.1-2.3 per ns)")
