"""Throughput of S independent C2 forwards in flight (one hipGraph plan + workspace + stream each)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
from taco_amd.tacotron import _Plan
import ctypes as C
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
B, T_in, n = 32, 128, 128
rs = np.random.RandomState(0)
ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); lens = np.full((B,), T_in, np.int32)
K = 48
for S in [int(v) for v in os.environ.get("SWEEP", "1,2,3,4,6,8").split(",")]:
    pool = m.plan_pool(B, T_in, n, lanes=S)
    streams, plans = pool.streams, pool.plans
    for p in plans:
        p.inputs.copy_(torch.from_numpy(ids)); p.lengths.copy_(torch.from_numpy(lens))
    torch.cuda.synchronize()
    evs = [torch.cuda.Event() for _ in range(S)]
    REC = os.environ.get("REC", "0") == "1"
    def go(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % S]):
                plans[i % S].launch()
                if REC: evs[i % S].record()
    go(2 * S); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = all(bool(torch.isfinite(p.linear).all()) for p in plans)
    same = all(bool(torch.equal(p.linear, plans[0].linear)) for p in plans)
    print("S=%d  %.2f ms/forward  %.2f M frames/s  finite=%s identical=%s" % (S, dt / K * 1e3, B * n * hp.reduction_factor / (dt / K) / 1e6, ok, same), flush=True)
    del plans, pool
