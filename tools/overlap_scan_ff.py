#!/usr/bin/env python
"""Does feed-forward work of ANOTHER request run on the CUs while a post-net scan (k_bigru_duo: 2 waves per SIMD, 88 VGPRs, parked half
of the time) holds them?  Round 2's version of this experiment (tools/scratch/overlap.py, removed in round 3) used two arbitrary torch streams, which HIP
may bind to the same hardware queue (then nothing overlaps whatever the kernels are); this one takes streams that were PROBED to run
concurrently (tacotron._concurrent_streams).

    python tools/overlap_scan_ff.py        # prints ms for: post-net alone, encoder alone, feed-forward-only post-net alone, and the pairs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
from taco_amd.tacotron import _concurrent_streams
hp = taco_amd.hparams.copy(max_iters=128)
B, T_in, n = 32, 128, 128
rs = np.random.RandomState(1)
ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
L = taco_amd.input_lengths_from_tokens(ids)
models = []
for i in range(2):
    m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1)); m.initialize(None, None, 1, None)
    models.append(m)
m, m2 = models
enc = m.encoder(ids, L, None); mel = m.decoder(enc, n, None)[0]
m2._lib.taco_debug_set_skip_scans(m2._handle, 1)           # m2: feed-forward launches only
s1, s2 = _concurrent_streams(m.device, 2)
def run(a, b, reps=6):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for _ in range(reps):
        if a:
            with torch.cuda.stream(s1): a()
        if b:
            with torch.cuda.stream(s2): b()
    cur.wait_stream(s1); cur.wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
post = lambda: m.postnet(mel)
enc_ff = lambda: m2.encoder(ids, L, None)
post_ff = lambda: m2.postnet(mel)
for _ in range(2):
    run(post, enc_ff); run(post, post_ff)
print("post-net (FF + scan + linear) alone %.3f ms | encoder FF alone %.3f | post-net FF alone %.3f" % (run(post, None), run(None, enc_ff), run(None, post_ff)))
print("post-net || encoder FF   : %.3f ms per pair" % run(post, enc_ff))
print("post-net || post-net FF  : %.3f ms per pair" % run(post, post_ff))
m.check_device_errors(); m2.check_device_errors()
