"""Whole-chip kernels of ONE process take turns on the device (csrc/taco_lib.hip: ChipTurn; include/taco_abi.h: taco_plan_whole_chip).

The persistent decoder loop and the post-net scan need all 256 workgroups resident at once; two of them dispatched together from different
streams used to starve each other until their bounded spins reported a device fault.  Every such launch -- eager or as
part of a replayed plan -- that follows one enqueued on ANOTHER stream now waits for an event recorded on that stream, so forwards issued
concurrently from several threads / streams / models compute exactly what they compute alone.  (tools/scratch/chip_turns_probe.py is the
negative control: C2-sized plans from two threads with the ordering switched off.)"""
import threading

import numpy as np
import pytest

import taco_oracle as O
from util import build_model

pytestmark = pytest.mark.gpu


def _requests(n, B, T_in, seed):
    return [O.synthetic_inputs(B, T_in, seed + i, ragged=(i % 2 == 1)) for i in range(n)]


def test_persistent_forwards_from_two_threads_and_streams_take_turns():
    import torch
    ohp = O.OracleHParams(max_iters=12)                       # reference widths: the persistent engine
    w = O.init_weights(ohp, 1, 731)
    models = [build_model(ohp, w), build_model(ohp, w)]
    B, T_in = 8, 24
    assert "persistent" in models[0].engine_plan(B, T_in), models[0].engine_plan(B, T_in)
    reqs = _requests(6, B, T_in, 900)
    alone = []
    for ids, L in reqs:
        lin, al = models[0].run(ids, L, honor_stop=False)
        alone.append((models[0].mel_outputs.clone(), lin.clone(), al.clone()))
    torch.cuda.synchronize()
    assert models[0].decoder_engine_info()["protocol"] == 1
    got = [[None] * len(reqs), [None] * len(reqs)]
    errors = []
    start = threading.Barrier(2)
    from taco_amd.tacotron import _concurrent_streams
    streams = _concurrent_streams(models[0].device, 2)         # probed to sit on different hardware queues: two arbitrary streams may share one and serialise anyway

    def worker(k):
        try:
            m = models[k]
            st = streams[k]
            start.wait()
            with torch.cuda.stream(st):
                for rep in range(3):                           # 18 forwards per thread, enqueued as fast as the host can
                    for i, (ids, L) in enumerate(reqs):
                        lin, al = m.run(ids, L, honor_stop=False)
                        got[k][i] = (m.mel_outputs.clone(), lin.clone(), al.clone())
            st.synchronize()
        except Exception as e:                                 # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for k in range(2):
        for i in range(len(reqs)):
            for a, b in zip(alone[i], got[k][i]):
                assert torch.equal(a, b), "thread %d, request %d differs from the same request served alone" % (k, i)
    for m in models:
        m.check_device_errors()


def test_a_two_lane_pool_on_the_persistent_engine_is_exact():
    import torch
    import taco_amd
    ohp = O.OracleHParams(max_iters=10)
    w = O.init_weights(ohp, 1, 733)
    m = build_model(ohp, w)
    B, T_in = 8, 20
    reqs = _requests(8, B, T_in, 950)
    alone = []
    for ids, L in reqs:
        lin, al = m.run(ids, L, honor_stop=False)
        alone.append((m.mel_outputs.clone(), lin.clone(), al.clone()))
    torch.cuda.synchronize()
    pool = taco_amd.tacotron.PlanPool(m, B, T_in, lanes=2, engine="persistent")
    assert pool.engine == "persistent"
    assert all(m._lib.taco_plan_whole_chip(p.handle) == 1 for p in pool.plans)
    for base in range(0, len(reqs), 2):
        tickets = [pool.submit(*reqs[base + k]) for k in range(2)]
        for k in (1, 0):
            r = pool.result(tickets[k])
            for a, b in zip(alone[base + k], (r["mel"], r["linear"], r["alignments"])):
                assert torch.equal(a, b), "request %d differs between the pool and the same request served alone" % (base + k)
    m.check_device_errors()
    pool.close()
    lp = taco_amd.tacotron.PlanPool(m, B, T_in, lanes=2)          # engine="auto": several lanes get the launch-per-stage engine, which holds no whole-chip kernel
    assert lp.engine == "launch-per-stage"
    assert all(m._lib.taco_plan_whole_chip(p.handle) == 0 for p in lp.plans)
    lp.close()
