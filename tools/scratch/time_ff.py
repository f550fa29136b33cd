"""Feed-forward time of the two CBHG stages (scans skipped) with the library named by TACO_LIB.  python tools/scratch/time_ff.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, taco_amd
B, T_in, T_mel = 32, 128, 512
hp = taco_amd.hparams.copy(max_iters=T_mel // 4)
m = taco_amd.create_model(hp); m.initialize(None, None, 1, None)
L = m._lib
ids = torch.randint(2, 80, (B, T_in), dtype=torch.int32, device="cuda"); ids[:, -1] = 1
lens = torch.full((B,), T_in - 1, dtype=torch.int32, device="cuda")
mel = torch.rand(B, T_mel, hp.num_mels, device="cuda")
L.taco_debug_set_skip_scans(m._handle, 1)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
try:
    for rep in range(2):
        print("%s  post-net feed-forward %7.1f us   encoder feed-forward %7.1f us" % (os.environ.get("TACO_LIB", "default"), t(lambda: m.postnet(mel)), t(lambda: m.encoder(ids, lens))))
finally:
    L.taco_debug_set_skip_scans(m._handle, 0)
