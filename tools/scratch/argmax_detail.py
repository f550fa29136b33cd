import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import taco_oracle as O
from util import build_model, maxabs
B, T_in, r, n, ns, mt = O.CONFIGS["C2"]
ohp = O.OracleHParams(max_iters=128, reduction_factor=r, model_type=mt)
w = O.init_weights(ohp, ns, 1234 + 2 + ns)
ids, L = O.synthetic_inputs(B, T_in, 99 + ns, ragged=False)
ref = O.forward(w, ohp, ids, L)
ar = ref["alignments"]
for bf3 in (1, 0):
    m = build_model(ohp, w)
    m._lib.taco_debug_set_bf3(m._handle, bf3, 0); m._plans.clear()
    lin, al = m.run(inputs=ids, input_lengths=L); torch.cuda.synchronize()
    a = al.cpu().numpy()
    peak = ar.max(axis=1); sel = peak > 1e-6
    mism = (a.argmax(axis=1) != ar.argmax(axis=1)) & sel
    print("bf3", bf3, "mismatches", int(mism.sum()), "of", int(sel.sum()), "max|align diff|", maxabs(a, ar))
    for b, t in zip(*np.nonzero(mism)):
        jr, jh = ar[b, :, t].argmax(), a[b, :, t].argmax()
        print("  row %d step %d: oracle argmax %d (%.9f, runner-up at %d: %.9f)  hip argmax %d (%.9f vs %.9f)" % (
            b, t, jr, ar[b, jr, t], jh, ar[b, jh, t], jh, a[b, jh, t], a[b, jr, t]))
    m.close()
