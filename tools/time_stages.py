#!/usr/bin/env python
"""Stage times (encoder / decoder / post-net, eager, alone) for a workload at one or more batch sizes:
python tools/time_stages.py C2 32 64"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
from bench import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
Bs = [int(x) for x in sys.argv[2:]] or [WORKLOADS[name][0]]
_, T_in, r, n, ns, mt = WORKLOADS[name]
hp = taco_amd.hparams.copy(max_iters=n, reduction_factor=r, model_type=mt)
m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, ns, seed=1)); m.initialize(None, None, ns, None)
for B in Bs:
    rs = np.random.RandomState(B)
    ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
    L = taco_amd.input_lengths_from_tokens(ids)
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    enc = m.encoder(ids, L, spk); mel = m.decoder(enc, n, spk)[0]
    out = {}
    for nm, fn in (("encoder", lambda: m.encoder(ids, L, spk)), ("decoder", lambda: m.decoder(enc, n, spk)), ("postnet", lambda: m.postnet(mel, speaker_id=spk))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        out[nm] = e0.elapsed_time(e1) / 3
    tot = sum(out.values())
    print("%s B=%d: %s  sum %.3f ms -> %.2f M mel-frames/s if run back to back" % (name, B, {k: round(v, 3) for k, v in out.items()}, tot, B * n * r / tot / 1e3))
    m.check_device_errors()
