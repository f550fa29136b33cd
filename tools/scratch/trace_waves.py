"""Which wave of a decoder group is late?  The stage split of the GRU 1 gates stage and the stage times of one step, stamped by different
(member, thread) pairs of group 0 (TACO_TRACE_MEMBER / TACO_TRACE_TID pick who stamps; the traced instantiation of k_decoder_xcd<4>).
python tools/scratch/trace_waves.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, taco_amd
B, T_in, n = 32, 128, 128
hp = taco_amd.hparams.copy(max_iters=n)
model = taco_amd.create_model(hp)
model.load_weights(taco_amd.weights.random_weights(hp, 1, seed=1234))
model.initialize(None, None, 1, None, device="cuda:0")
rs = np.random.RandomState(7)
ids = rs.randint(2, 80, size=(B, T_in)).astype(np.int32); ids[:, -1] = 1
L = taco_amd.input_lengths_from_tokens(ids)
enc = model.encoder(ids, L, None)
PH = ["prenet2", "attGRU gates", "attGRU cand", "query", "scores+sum", "norm+ctx", "GRU1 gates", "GRU1 cand", "GRU2 gates", "GRU2 cand", "p1+frame"]
for member, tid in [(0, 0), (0, 64), (0, 256), (0, 448), (13, 320), (31, 0), (31, 448), (16, 192)]:
    os.environ["TACO_TRACE_MEMBER"], os.environ["TACO_TRACE_TID"] = str(member), str(tid)
    model.decoder_trace(True)
    model.decoder(enc, n, None); torch.cuda.synchronize()
    tr = model.decoder_trace(True, read=True); model.decoder_trace(False)
    d = np.diff(tr[:, :12], axis=1).astype(np.float64)
    sub = np.stack([tr[:, 12] - tr[:, 6], tr[:, 13] - tr[:, 12], tr[:, 14] - tr[:, 13], tr[:, 7] - tr[:, 14]], 1).astype(np.float64)
    step = np.median((tr[1:, 0] - tr[:-1, 0]).astype(np.float64))
    print("member %2d wave %d: step %5.0f clocks | " % (member, tid >> 6, step) + " ".join("%s %4.0f" % (p[:6], c) for p, c in zip(PH, np.median(d[1:], axis=0))) +
          " | GRU1 gates: pass+reduce %4.0f epi+publish %4.0f gather %4.0f barrier %4.0f" % tuple(np.median(sub[1:], axis=0)), flush=True)
model.check_device_errors()
