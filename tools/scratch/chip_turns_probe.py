#!/usr/bin/env python
"""Negative control of tests/test_gpu_chip_turns.py: C2-sized plans replayed from two threads on two streams that were probed to run concurrently,
with the ordering of whole-chip kernels on (1) or off (0: taco_debug_set_chip_turns).  Prints the time, whether the results equal the plans' results
alone and whether a persistent kernel reported starvation.  Run under `timeout`:   timeout 300 python tools/scratch/chip_turns_probe.py 0"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, taco_amd
from taco_amd.tacotron import _concurrent_streams, _Plan
on = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
hp = taco_amd.hparams.copy(max_iters=128)
B, T_in, n = 32, 128, 128
m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1)); m.initialize(None, None, 1, None)
m._lib.taco_debug_set_chip_turns(on)
SS = _concurrent_streams(m.device, 2)
plans, ref = [], []
for k in range(2):
    rs = np.random.RandomState(10 + k)
    ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
    with torch.cuda.stream(SS[k]):
        p = _Plan(m, B, T_in, n, False)
        p.inputs.copy_(torch.as_tensor(ids)); p.lengths.copy_(torch.as_tensor(taco_amd.input_lengths_from_tokens(ids)))
        p.launch()
    SS[k].synchronize()
    plans.append(p); ref.append(p.linear.clone())
print("whole_chip flags:", [m._lib.taco_plan_whole_chip(p.handle) for p in plans])
bar = threading.Barrier(2)
def worker(k):
    bar.wait()
    with torch.cuda.stream(SS[k]):
        for _ in range(reps): plans[k].launch()
    SS[k].synchronize()
t0 = time.time()
ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
torch.cuda.synchronize()
dt = time.time() - t0
print("chip turns %d: %d + %d replays in %.3f s (%.3f ms per forward); equal to the plans alone: %s; stop words %s" %
      (on, reps, reps, dt, 1e3 * dt / (2 * reps), [bool(torch.equal(plans[k].linear, ref[k])) for k in range(2)], [int(p.stop.item()) for p in plans]))
try: m.check_device_errors(); print("no device error")
except Exception as e: print("device error: %r" % (e,))
