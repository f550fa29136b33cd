"""A/B of the point-wise chain kernel (taco_chain.h) against one launch per layer: encoder and post-net stage outputs and times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, taco_amd
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1)); m.initialize(None, None, 1, None)
B, T_in, n = 32, 128, 128
rs = np.random.RandomState(1)
ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
L = taco_amd.input_lengths_from_tokens(ids)
L2 = L.copy(); L2[3] = 17; L2[5] = 0; L2[7] = 128
mel = torch.from_numpy(rs.rand(B, 512, 80).astype(np.float32)).cuda()
def timeit(fn, k=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
res = {}
for flag, name in ((1, "chain"), (5, "per-layer")):
    m._lib.taco_debug_set_bf3(m._handle, flag, 0)
    enc = m.encoder(ids, L2, None); post = m.postnet(mel); torch.cuda.synchronize()
    res[name] = (enc.cpu().numpy(), post.cpu().numpy())
    print("%-10s encoder %.3f ms  postnet %.3f ms" % (name, timeit(lambda: m.encoder(ids, L2, None)), timeit(lambda: m.postnet(mel))))
m._lib.taco_debug_set_bf3(m._handle, 1, 0)
for i, st in enumerate(("encoder", "postnet")):
    a, b = res["chain"][i], res["per-layer"][i]
    print("%s: max|chain - per-layer| = %.3e (max|.| %.2f)" % (st, float(np.abs(a - b).max()), float(np.abs(b).max())))
