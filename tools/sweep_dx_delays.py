#!/usr/bin/env python
"""Coordinate descent over the sleeps in front of the first poll of the persistent decoder's gathers (csrc/taco_decoder_xcd.h, sites 0..10 of
DX_DLY), on the A/B build that reads them from a constant table (-DDX_DLY_RT; the production build has them as immediates):
    cd multi-speaker-tacotron-tensorflow_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDX_DLY_RT -o libtaco_hip_dly.so taco_lib.hip
    TACO_LIB=$PWD/libtaco_hip_dly.so python tools/sweep_dx_delays.py [passes]
Times the decoder stage alone at C2 (stage-level C ABI, eager launches: the table is refreshed from TACO_DX_DLY at every launch), 3 x 20
launches per setting, median.  Units of 64 clocks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import taco_amd
from bench import WORKLOADS

START = [int(x) for x in os.environ.get("SWEEP_START", "5,5,5,5,5,5,5,5,5,5,5").split(",")]
HALF_B = os.environ.get("SWEEP_HALF_B", "0") == "1"      # sweep the sleeps of waves 4-7 (the second wave of every SIMD) on their own
SITES = ["p2", "p3 (3-layer prenet only)", "r*h att", "h att", "partial scores", "context", "r*h 1", "h1", "r*h 2", "h2", "prenet 1 of the next step"]


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.cuda.set_device(0)
    B, T_in, r, n, ns, mt = WORKLOADS["C2"]
    B = int(os.environ.get("SWEEP_ROWS", B))          # 64: the eight-rows-per-group instantiation of a 64-row pass
    hp = taco_amd.hparams.copy(max_iters=n, reduction_factor=r, model_type=mt)
    model = taco_amd.create_model(hp)
    model.load_weights(taco_amd.weights.random_weights(hp, ns, seed=1234))
    model.initialize(None, None, ns, None, device="cuda:0")
    rs = np.random.RandomState(7)
    ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32)
    ids[:, T_in - 1] = 1
    L = taco_amd.input_lengths_from_tokens(ids)
    enc = model.encoder(ids, L, None)
    model.set_decoder_engine(1, 0)

    def measure(d, reps=20, rounds=3):
        if HALF_B:      # the table of waves 4-7 alone moves; waves 0-3 keep SWEEP_START
            os.environ["TACO_DX_DLY"] = ",".join(str(x) for x in START)
            os.environ["TACO_DX_DLY_B"] = ",".join(str(x) for x in d)
        else:
            os.environ["TACO_DX_DLY"] = ",".join(str(x) for x in d)
        model.decoder(enc, n, None)
        torch.cuda.synchronize()
        ts = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                model.decoder(enc, n, None)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
        return float(np.median(ts))

    cur = list(START)
    best = measure(cur)
    print("start %s: %.1f us per decoder call (%.3f us per step)" % (cur, best, best / n))
    for v in (0, 3, 4, 6, 7):
        print("  all sites at %d: %.1f us" % (v, measure([v] * 11)))
    for p in range(passes):
        for site in [0, 2, 3, 4, 5, 6, 7, 8, 9, 10]:
            row = []
            for v in range(0, 10):
                d = list(cur); d[site] = v
                t = measure(d)
                row.append((t, v))
            row.sort()
            line = "  ".join("%d: %.1f" % (v, t) for t, v in sorted(row, key=lambda x: x[1]))
            if row[0][0] < best - 0.7 and row[0][1] != cur[site]:
                cur[site] = row[0][1]; best = row[0][0]
            print("pass %d site %2d %-26s %s  -> %d" % (p, site, SITES[site], line, cur[site]))
        print("after pass %d: %s  %.1f us (re-measured %.1f)" % (p, cur, best, measure(cur)))
    print("check: all 5 -> %.1f us; tuned %s -> %.1f us" % (measure([5] * 11), cur, measure(cur)))


if __name__ == "__main__":
    main()
