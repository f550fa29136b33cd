import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, taco_amd
if len(sys.argv) > 1:
    taco_amd._lib.LIB_PATH = os.path.abspath(sys.argv[1])
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1)); m.initialize(None, None, 1, None)
B = 32
rs = np.random.RandomState(1)
mel = torch.from_numpy(rs.rand(B, 512, 80).astype(np.float32)).cuda()
def timeit(fn, k=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
t = {}
for flag, name in ((1, "chain"), (5, "per-layer")):
    m._lib.taco_debug_set_bf3(m._handle, flag, 0)
    t[name] = timeit(lambda: m.postnet(mel))
print("%s: postnet chain %.3f ms, per-layer %.3f ms -> chain kernel ~ %.0f us" % (sys.argv[1] if len(sys.argv) > 1 else "default", t["chain"], t["per-layer"], 216 - (t["per-layer"] - t["chain"]) * 1e3))
