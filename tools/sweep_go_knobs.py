#!/usr/bin/env python
"""Coordinate descent over the run-time knobs of the post-net scan k_bigru_oct on the A/B build -DGO_KNOB_RT (csrc/taco_bigru_xcd.h): per phase
(gates F, gates B, cand F, cand B) the place of the second request (0 behind the products, 1 behind the reduction, 2 none), a sleep in front
of the first request, a sleep in front of the collect's fallback poll (units of 64 clocks).  The measure is the op-level BiGRU call at the C2
post-net shape (32 x 512: input projection ~80 us + the scan), 3 x 10 launches per point, median.
    TACO_LIB=.../libtaco_hip_gok.so python tools/sweep_go_knobs.py [passes]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd

NAMES = ["second request: gates F", "second request: gates B", "second request: cand F", "second request: cand B",
         "sleep before the request of h'(B)", "sleep before the request of r*h(F)", "sleep before the request of r*h(B)", "sleep before the request of h'(F)",
         "sleep before the fallback poll h'(B)", "... r*h(F)", "... r*h(B)", "... h'(F)"]


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B, T = 32, 512
    hp = taco_amd.hparams.copy(max_iters=128)
    m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
    L = m._lib
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
    x = torch.randn(B, T, 256, device="cuda") * 0.3; out = torch.empty(B, T, 512, device="cuda")
    nb = L.taco_stage_workspace_bytes(m._handle, B, T) + (64 << 20)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    fn = lambda: taco_amd._lib.check(L.taco_bigru_f32(m._handle, st(), b"post_cbhg", P(x), P(None), P(None), B, T, P(out), P(ws), nb))

    def measure(k, reps=10, rounds=3):
        os.environ["TACO_GO_KNOB"] = ",".join(str(v) for v in k)
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        return float(np.median(ts))

    cur = [0] * 12
    best = measure(cur)
    ref = out.clone()
    print("start %s: %.1f us per call" % (cur, best))
    for p in range(passes):
        for i in range(12):
            vals = (0, 1, 2) if i < 4 else (0, 1, 2, 3, 4, 6)
            row = []
            for v in vals:
                k = list(cur); k[i] = v
                row.append((measure(k), v))
            line = "  ".join("%d: %.1f" % (v, t) for t, v in row)
            row.sort()
            if row[0][0] < best - 1.0 and row[0][1] != cur[i]:
                cur[i] = row[0][1]; best = row[0][0]
            print("pass %d knob %2d %-40s %s  -> %d" % (p, i, NAMES[i], line, cur[i]))
        print("after pass %d: %s  %.1f us (re-measured %.1f)" % (p, cur, best, measure(cur)))
    t0, t1 = measure([0] * 12), measure(cur)
    fn(); torch.cuda.synchronize()
    print("check: production knobs -> %.1f us; tuned %s -> %.1f us; outputs identical: %s" % (t0, cur, t1, bool(torch.equal(out, ref))))
    m.check_device_errors()


if __name__ == "__main__":
    main()
