#!/usr/bin/env python
"""Decoder stage alone (stage-level C ABI, HIP events): us per decoder step for the launch-per-stage engine and the
persistent XCD-local engine (csrc/taco_decoder_xcd.h), with the persistent engine's per-phase timeline
(group 0 / member 0 shader-clock stamps).  Usage: python tools/time_decoder.py [C2 C5 C1 ...] [--json out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import taco_amd
from bench import WORKLOADS

PHASES = ["prenet2", "attGRU gates", "attGRU cand", "query(local)", "partial scores+sum", "normaliser+context", "GRU1 gates(+concat)", "GRU1 cand",
          "GRU2 gates", "GRU2 cand", "prenet1(next)+frame"]


def time_engine(model, enc, n, spk, reps=3):
    model.decoder(enc, n, spk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = model.decoder(enc, n, spk)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def main():
    jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    names = [a for a in sys.argv[1:] if not a.startswith("--") and a != jpath] or ["C2", "C5", "C1"]
    torch.cuda.set_device(0)
    res = {}
    for name in names:
        extra_rows = []
        if ":" in name:
            name, rr = name.split(":")
            extra_rows = [int(x) for x in rr.split(",")]
        B, T_in, r, n, ns, mt = WORKLOADS[name]
        hp = taco_amd.hparams.copy(max_iters=n, reduction_factor=r, model_type=mt)
        model = taco_amd.create_model(hp)
        model.load_weights(taco_amd.weights.random_weights(hp, ns, seed=1234))
        model.initialize(None, None, ns, None, device="cuda:0")
        rs = np.random.RandomState(7)
        ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32)
        ids[:, T_in - 1] = 1
        L = taco_amd.input_lengths_from_tokens(ids)
        spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
        enc = model.encoder(ids, L, spk)
        rec = {}
        base = None
        for label, mode, rows in [("launch-per-stage", 0, 0), ("persistent", 1, 0), ("persistent write-through", 2, 0)] + \
                                 [("persistent rows/group=%d" % x, 1, x) for x in extra_rows]:
            model.set_decoder_engine(mode, rows)
            if mode == 1 and rows == 0:
                model.decoder_trace(True)
            ms, out = time_engine(model, enc, n, spk)
            info = model.decoder_engine_info()
            model.check_device_errors()
            mel = out[0].cpu().numpy()
            if base is None:
                base = mel
            rec[label] = {"ms": ms, "us_per_step": ms * 1e3 / n, "protocol": info["protocol"], "per_xcd": info["per_xcd"],
                          "max_abs_diff_vs_launch": float(np.abs(mel - base).max())}
            print("%s %-28s %8.3f ms  %7.2f us/step  protocol %d  per-xcd %s  |diff| %.2e"
                  % (name, label, ms, ms * 1e3 / n, info["protocol"], info["per_xcd"], rec[label]["max_abs_diff_vs_launch"]), flush=True)
            if mode == 1 and rows == 0:
                tr = model.decoder_trace(True, read=True)
                model.decoder_trace(False)
                d = np.diff(tr[:, :12], axis=1).astype(np.float64)          # [8 steps][11 phases] shader clocks
                sub = np.stack([tr[:, 12] - tr[:, 6], tr[:, 13] - tr[:, 12], tr[:, 14] - tr[:, 13], tr[:, 7] - tr[:, 14]], 1).astype(np.float64)
                step_clk = (tr[1:, 0] - tr[:-1, 0]).astype(np.float64)
                us_step = ms * 1e3 / n
                clk_per_us = float(np.median(step_clk)) / us_step if us_step > 0 else 0.0
                med = np.median(d[1:], axis=0)
                rec[label]["timeline_us"] = {p: float(c / clk_per_us) for p, c in zip(PHASES, med)} if clk_per_us else {}
                rec[label]["clock_MHz_est"] = clk_per_us
                print("   timeline (median of steps 1-7, us; clock ~%.0f MHz): " % clk_per_us +
                      "  ".join("%s %.2f" % (p, c / clk_per_us) for p, c in zip(PHASES, med)), flush=True)
                sm = np.median(sub[1:], axis=0) / clk_per_us
                rec[label]["gates_stage_split_us"] = dict(zip(("pass+reduce", "epilogue+publish", "gather(poll)", "barrier"), (float(x) for x in sm)))
                print("   GRU1 gates stage split: pass+reduce %.2f  epilogue+publish %.2f  gather(poll) %.2f  barrier %.2f" % tuple(sm), flush=True)
        res[name] = rec
        model.close()
    if jpath:
        import bench
        res["kernel_source_hash"] = bench.source_hash()      # ties the timeline to the build it was taken from (bench.py floor_constants)
        json.dump(res, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
