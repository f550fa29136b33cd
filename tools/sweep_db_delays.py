#!/usr/bin/env python
"""Coordinate descent over the first-poll sleeps of the persistent decoder BPTT (csrc/taco_decoder_bwd_xcd.h, sites 0..11 of DB_DLY) on the A/B
build that reads them from a constant table (-DDX_DLY_RT, TACO_DB_DLY), as tools/sweep_dx_delays.py does for the forward loop.  The measure is
the kernel's own step length: the shader-clock stamps of steps 8-15 of a C4-shard backward pass (tools/trace_bptt.py), median of three passes.
    TACO_LIB=.../libtaco_hip_dly.so python tools/sweep_db_delays.py [passes]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
from taco_amd import _lib

SITES = ["d c_pre 2", "gate grads 2", "d c_pre 1", "gate grads 1", "d o0", "d ctx", "d alpha partials", "d q partials", "d c_pre att", "gate grads att", "d z2", "d z1"]


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B, T_in, T_out = 32, 128, 512
    hp = taco_amd.hparams.copy(max_iters=max(200, T_out // 4))
    tr = taco_amd.Trainer(hp, taco_amd.weights.random_weights(hp, 1, seed=4321))
    rs = np.random.RandomState(0)
    ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); L = np.full(B, T_in, np.int32)
    mt, lt = rs.rand(B, T_out, hp.num_mels).astype(np.float32), rs.rand(B, T_out, hp.num_freq).astype(np.float32)
    tr.forward_backward(ids, L, mt, lt, None); torch.cuda.synchronize()
    mh = C.c_void_p(tr._lib.taco_train_model(tr._h))
    out = (C.c_longlong * 128)()

    def measure(d, rounds=3):
        os.environ["TACO_DB_DLY"] = ",".join(str(x) for x in d)
        vals = []
        for _ in range(rounds):
            _lib.check(tr._lib.taco_debug_decoder_trace(mh, 1, None))
            tr.forward_backward(ids, L, mt, lt, None); torch.cuda.synchronize()
            _lib.check(tr._lib.taco_debug_decoder_trace(mh, 1 | 4, out))
            t = np.array(out[:], np.int64).reshape(8, 16)
            vals.append(float(np.median((t[1:, 0] - t[:-1, 0]).astype(np.float64))))
        _lib.check(tr._lib.taco_debug_decoder_trace(mh, 0, None))
        return float(np.median(vals))

    cur = [int(x) for x in os.environ.get("SWEEP_START", ",".join(["5"] * 12)).split(",")]
    best = measure(cur)
    print("start %s: %.0f clocks per backward step" % (cur, best))
    for v in (0, 3, 4, 6, 7):
        print("  all sites at %d: %.0f" % (v, measure([v] * 12)))
    for p in range(passes):
        for site in range(12):
            row = []
            for v in range(0, 10):
                d = list(cur); d[site] = v
                row.append((measure(d), v))
            line = "  ".join("%d: %.0f" % (v, t) for t, v in row)
            row.sort()
            if row[0][0] < best - 40 and row[0][1] != cur[site]:
                cur[site] = row[0][1]; best = row[0][0]
            print("pass %d site %2d %-18s %s  -> %d" % (p, site, SITES[site], line, cur[site]))
        print("after pass %d: %s  %.0f clocks (re-measured %.0f)" % (p, cur, best, measure(cur)))
    print("check: all 5 -> %.0f clocks; tuned %s -> %.0f clocks" % (measure([5] * 12), cur, measure(cur)))


if __name__ == "__main__":
    main()
