import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, taco_amd
if len(sys.argv) > 1 and sys.argv[1]:
    taco_amd._lib.LIB_PATH = os.path.abspath(sys.argv[1])
hp = taco_amd.hparams.copy(max_iters=128)
m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1)); m.initialize(None, None, 1, None)
B, T_in, n = 32, 128, 128
rs = np.random.RandomState(1)
ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
L = taco_amd.input_lengths_from_tokens(ids)
mel = torch.from_numpy(rs.rand(B, 512, 80).astype(np.float32)).cuda()
def timeit(fn, k=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
enc = m.encoder(ids, L, None); post = m.postnet(mel); torch.cuda.synchronize()
print("%s: encoder %.3f ms  postnet %.3f ms  checksum %.6f %.6f" % (sys.argv[1] if len(sys.argv) > 1 else "default", timeit(lambda: m.encoder(ids, L, None)), timeit(lambda: m.postnet(mel)), float(enc.abs().mean()), float(post.abs().mean())))
