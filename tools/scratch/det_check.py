"""Two deterministic training steps at the C4 shard from the same parameters: gradients must be equal to the bit.  python tools/scratch/det_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch, taco_amd
import taco_oracle as O
hp = O.OracleHParams(max_iters=128)
w = O.init_weights(hp, 1, 5)
B, T_in, T_out = 32, 128, 512
ids, L = O.synthetic_inputs(B, T_in, 6, ragged=True)
rs = np.random.RandomState(7)
mt, lt = rs.rand(B, T_out, hp.num_mels).astype(np.float32), rs.rand(B, T_out, hp.num_freq).astype(np.float32)
php = taco_amd.hparams.copy(max_iters=128)
tr = taco_amd.Trainer(php, w)
out = []
for det in (True, False):
    tr.set_deterministic(det)
    gs = []
    for _ in range(3):
        tr.forward_backward(ids, L, mt, lt, freeze_moving_averages=True)
        torch.cuda.synchronize()
        gs.append(tr.grads.detach().clone())
    d = max(float((gs[0] - g).abs().max()) for g in gs[1:])
    out.append((det, d, float(gs[0].abs().max())))
    print("deterministic" if det else "atomics      ", "max |g_run0 - g_runk| over 2 reruns = %.3e   (|g|max %.3e)" % (d, float(gs[0].abs().max())))
assert out[0][1] == 0.0, "the deterministic step is not bit-reproducible"
tr.check_device_errors()
