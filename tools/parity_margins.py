"""Prints max-abs errors of the HIP forward vs the float64 oracle at the BASELINE.json configs (C1, C2, C3)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import taco_oracle as O
from util import build_model, maxabs, argmax_match
for name, steps in (("C1", 200), ("C2", 128), ("C3", 128)):
    B, T_in, r, n, ns, mt = O.CONFIGS[name]
    ohp = O.OracleHParams(max_iters=steps, reduction_factor=r, model_type=mt)
    w = O.init_weights(ohp, ns, 1234 + len(name) + ns)
    ids, L = O.synthetic_inputs(B, T_in, 99 + ns, ragged=(name == "C3"))
    spk = (np.arange(B) % ns).astype(np.int32) if ns > 1 else None
    m = build_model(ohp, w, num_speakers=ns)
    lin, al = m.run(inputs=ids, input_lengths=L, speaker_id=spk); torch.cuda.synchronize()
    ref = O.forward(w, ohp, ids, L, speaker_id=spk, num_speakers=ns)
    nchk, bad = argmax_match(al.cpu().numpy(), ref["alignments"])
    print("%s: max|mel| %.2e  max|linear| %.2e  max|align| %.2e  argmax mismatches %d/%d  (|mel|max %.2f)" % (
        name, maxabs(m.mel_outputs.cpu().numpy(), ref["mel"]), maxabs(lin.cpu().numpy(), ref["linear"]),
        maxabs(al.cpu().numpy(), ref["alignments"]), bad, nchk, np.abs(ref["mel"]).max()))
