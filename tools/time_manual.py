#!/usr/bin/env python
"""Second pass of synthesize(manual_attention_mode=1) (synthesizer.py:171-205: the forward is run again with one-hot manual
alignments) against the first pass, at C2 shapes; and model type 'simple' against 'single'.  Both used to fall back to the
launch-per-stage decoder (43 us per step); they are modes of the persistent decoder now.

    python tools/time_manual.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, taco_amd
B, T_in, n = 32, 128, 128


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for mt, ns in (("single", 1), ("simple", 4)):
    hp = taco_amd.hparams.copy(max_iters=n, model_type=mt)
    m = taco_amd.create_model(hp); m.load_weights(taco_amd.weights.random_weights(hp, ns, seed=1)); m.initialize(None, None, ns, None)
    rs = np.random.RandomState(1)
    ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
    L = taco_amd.input_lengths_from_tokens(ids)
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    lin, al = m.run(inputs=ids, input_lengths=L, speaker_id=spk, honor_stop=False)
    onehot = torch.zeros((B, n, T_in), device=al.device)
    onehot.scatter_(2, al.argmax(1).unsqueeze(2), 1.0)                     # [B, n, T_in] one-hot of the first pass's argmax
    t1 = timed(lambda: m.run(inputs=ids, input_lengths=L, speaker_id=spk, honor_stop=False))
    t2 = timed(lambda: m.run(inputs=ids, input_lengths=L, speaker_id=spk, manual_alignments=onehot, is_manual_attention=True, honor_stop=False))
    enc = m.encoder(ids, L, spk)
    d1 = timed(lambda: m.decoder(enc, n, spk))
    d2 = timed(lambda: m.decoder(enc, n, spk, manual_alignments=onehot))
    m.check_device_errors()
    print("%-7s forward %.3f ms, with manual alignments %.3f ms (x%.2f); decoder loop alone %.3f -> %.3f ms (%.2f -> %.2f us per step); engine %s"
          % (mt, t1, t2, t2 / t1, d1, d2, d1 * 1e3 / n, d2 * 1e3 / n, m.decoder_engine_info()))
    m.close()
