#!/usr/bin/env python
"""Two C2 forwards in flight as a software pipeline: the encoder of request i + 1 is enqueued on a second stream behind the DECODER of request i (so it
runs beside request i's post-net), the whole-chip kernels stay strictly ordered by events (decoder i + 1 waits for post-net i).  Prints ms per forward
for the plain sequence and for the pipeline.   python tools/scratch/pipeline_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
dev = models[0].device
ids_d = torch.as_tensor(ids, device=dev); L_d = torch.as_tensor(L, device=dev)
S = _concurrent_streams(dev, 2)
def forward_seq(m):
    enc = m.encoder(ids_d, L_d, None); mel = m.decoder(enc, n, None)[0]; return m.postnet(mel)
def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def seq(reps):
    for _ in range(reps): forward_seq(models[0])
def pipe(reps, enc_after="decoder"):
    cur = torch.cuda.current_stream()
    for s in S: s.wait_stream(cur)
    ev_dec = [None, None]; ev_post = [None, None]; ev_p1 = None
    outs = []
    for i in range(reps):
        ln = i & 1; m = models[ln]; other = ln ^ 1
        with torch.cuda.stream(S[ln]):
            if ev_dec[other] is not None and enc_after == "decoder": S[ln].wait_event(ev_dec[other])
            enc = m.encoder(ids_d, L_d, None)
            if ev_post[other] is not None: S[ln].wait_event(ev_post[other])
            mel = m.decoder(enc, n, None)[0]
            ev_dec[ln] = torch.cuda.Event(); ev_dec[ln].record(S[ln])
            outs.append(m.postnet(mel))
            ev_post[ln] = torch.cuda.Event(); ev_post[ln].record(S[ln])
    for s in S: cur.wait_stream(s)
    return outs
ref = forward_seq(models[0]); torch.cuda.synchronize()
for _ in range(2): seq(4); pipe(4)
print("sequence, one forward in flight      %.4f ms per forward" % timed(seq, 20))
print("pipeline (encoder i+1 behind decoder i) %.4f ms per forward" % timed(lambda r: pipe(r), 20))
print("pipeline (encoder i+1 unordered)        %.4f ms per forward" % timed(lambda r: pipe(r, "none"), 20))
print("sequence again                        %.4f ms per forward" % timed(seq, 20))
outs = pipe(6); torch.cuda.synchronize()
print("max |pipeline - sequence| over 6 forwards: %.3e" % max(float((o - ref).abs().max()) for o in outs))
for m in models: m.check_device_errors()
print("no device errors")
