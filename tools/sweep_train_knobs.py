#!/usr/bin/env python
"""The four second requests of the backward post-net scan k_bigru_oct_bwd (entries 12..15 of TACO_GO_KNOB on the -DGO_KNOB_RT build: 0 = as
built, 1 = none), measured on the whole C4-shard forward + backward (the kernel has no entry point of its own): every setting gets ROUNDS x 12
passes, the settings interleaved round by round so that drift cancels; median per setting.
    TACO_LIB=.../libtaco_hip_gok.so python tools/sweep_train_knobs.py"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd

ROUNDS = 5


def main():
    B, T_in, T_out = 32, 128, 512
    hp = taco_amd.hparams.copy(max_iters=max(200, T_out // 4))
    tr = taco_amd.Trainer(hp, taco_amd.weights.random_weights(hp, 1, seed=4321))
    rs = np.random.RandomState(0)
    ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); L = np.full(B, T_in, np.int32)
    mt, lt = rs.rand(B, T_out, hp.num_mels).astype(np.float32), rs.rand(B, T_out, hp.num_freq).astype(np.float32)
    ids, L, mt, lt = [torch.as_tensor(v).cuda() for v in (ids, L, mt, lt)]
    fwd = "0,2,0,0,0,0,0,0,0,0,0,0"      # the forward scan's production knobs
    settings = [k for k in itertools.product((0, 1), repeat=4)]
    times = {k: [] for k in settings}

    def run(k, reps=12):
        os.environ["TACO_GO_KNOB"] = fwd + "," + ",".join(str(v) for v in k)
        tr.forward_backward(ids, L, mt, lt, None); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            tr.forward_backward(ids, L, mt, lt, None)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    # every setting must give the SAME gradients (ordered reductions: to the bit) -- a second request that is dropped must not let a stale answer pass
    os.environ["TACO_GO_KNOB"] = fwd + ",0,0,0,0"
    tr.forward_backward(ids, L, mt, lt, None); torch.cuda.synchronize()
    ref = tr.grads.clone()
    for k in settings:
        os.environ["TACO_GO_KNOB"] = fwd + "," + ",".join(str(v) for v in k)
        tr.forward_backward(ids, L, mt, lt, None); torch.cuda.synchronize()
        assert torch.equal(tr.grads, ref), ("gradients differ with knobs", k)
    print("all %d settings give bit-identical gradients" % len(settings))
    for r in range(ROUNDS):
        for k in (settings if r % 2 == 0 else settings[::-1]):
            times[k].append(run(k))
    base = float(np.median(times[(0, 0, 0, 0)]))
    print("forward + backward at the C4 shard, us per pass (median of %d rounds x 12); knobs = second request of phase_ca F, phase_ca B, phase_b F, phase_b B (1 = none)" % ROUNDS)
    for k in sorted(settings, key=lambda k: np.median(times[k])):
        print("  %s  %.1f  (%+.1f)   rounds: %s" % (k, np.median(times[k]), np.median(times[k]) - base, " ".join("%.0f" % t for t in times[k])))


if __name__ == "__main__":
    main()
