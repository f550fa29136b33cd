"""Which pairs of PlanPool lanes actually run concurrently?  (stream -> hardware-queue mapping probe)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
hp = taco_amd.hparams.copy(max_iters=32)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
B, T_in, n = 32, 128, 32
S = int(os.environ.get("S", "8"))
pool = m.plan_pool(B, T_in, n, lanes=S)
def t_plans(lanes, reps=2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for l in lanes:
            with torch.cuda.stream(pool.streams[l]): pool.plans[l].launch()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def t_sleep(lanes, cyc=2_000_000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for l in lanes:
        with torch.cuda.stream(pool.streams[l]): torch.cuda._sleep(cyc)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
t_plans([0]); base = t_plans([0]); sb = t_sleep([0]); sb = t_sleep([0])
print("alone: plan %.2f ms, sleep %.2f ms" % (base, sb))
print("streams:", [hex(s.cuda_stream) for s in pool.streams])
for name, fn, b in (("plan", t_plans, base), ("sleep", t_sleep, sb)):
    print(name, "pair ratio (1 = concurrent, 2 = serialized):")
    for i in range(S):
        print("  " + " ".join("%.1f" % (fn([i, j]) / b) if j > i else " . " for j in range(S)))
print("all %d lanes: %.2f ms per forward" % (S, t_plans(list(range(S)), reps=4) / S))
