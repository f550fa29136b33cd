import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/oracle', '/root/repo/tests']
import numpy as np, torch
import taco_oracle as O, torch_formulation as TF, taco_amd
from util import to_product_hp, maxabs
from test_gpu_train import _grad_report
for model_type, atype, B in [("single", "bah_mon", 9), ("deepvoice", "bah", 3), ("simple", "bah_norm", 5), ("simple", "bah_mon", 5), ("single", "bah_norm", 5)]:
    ns = 1 if model_type == "single" else 3
    hp = O.OracleHParams(max_iters=8, model_type=model_type, attention_type=atype)
    w = O.init_weights(hp, ns, 81)
    T_in, T_out = 14, 8 * hp.reduction_factor
    ids, L = O.synthetic_inputs(B, T_in, 82, ragged=True)
    rs = np.random.RandomState(83)
    mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
    co = rs.uniform(0.5, 1.5, size=B)
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    loss, g, out = TF.train_grads(w, hp, ids, L, mt, lt, co, speaker_id=spk, num_speakers=ns)
    tr = taco_amd.Trainer(to_product_hp(hp), w, num_speakers=ns)
    for eng, exact in ((1, False), (0, False), (1, True), (0, True)):
        tr.set_decoder_engine(eng); tr.set_exact_wgrad(exact)
        tr.forward_backward(ids, L, mt, lt, co, keep_outputs=True, speaker_id=spk)
        torch.cuda.synchronize()
        worst, gn = _grad_report(tr.grad_dict(), g)
        print(model_type, atype, B, "engine", eng, "exact_wgrad", exact, ["%.4f %s %.1e" % (x[0], x[1], x[2]) for x in worst[:3]], "gn %.3f" % gn)
    tr.set_exact_wgrad(False)
    tr.close()
