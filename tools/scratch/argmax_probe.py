"""Where the HIP path's alignment arg-max differs from the oracle's, and how accurately the HIP path tracks the oracle's peak as the peak decays
(tests/util.py's floor and tie criterion rest on these numbers).  python tools/scratch/argmax_probe.py [C3|C5]   (needs tests/golden/full_*.npz)"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "oracle", os.path.join("tests", "golden"), "."):
    sys.path.insert(0, os.path.join(ROOT, p))
import taco_oracle as O, make_full_size_golden as G
from util import build_model
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "C3"
if which == "C3":
    g = np.load(os.path.join(ROOT, "tests/golden/full_C3.npz"))
    hp, ns, seed, ids, L, spk = G.c3_case()
    m = build_model(hp, O.init_weights(hp, ns, seed), num_speakers=ns)
    lin, al = m.run(inputs=ids, input_lengths=L, speaker_id=spk, honor_stop=False)
    torch.cuda.synchronize()
    al = al.cpu().numpy(); ref = g["alignments"].astype(np.float64)
    peak = ref.max(1); refmax = ref.argmax(1); second = np.sort(ref, axis=1)[:, -2, :]
    hip_at_ref = np.take_along_axis(al, refmax[:, None, :], 1)[:, 0, :]
else:
    g = np.load(os.path.join(ROOT, "tests/golden/full_C5.npz"))
    hp, seed, ids, L = G.c5_case()
    m = build_model(hp, O.init_weights(hp, 1, seed))
    lin, al = m.run(inputs=ids, input_lengths=L, honor_stop=False)
    torch.cuda.synchronize()
    al = al.cpu().numpy()
    peak, second, refmax = g["align_peak"], g["align_second"], g["align_argmax"].astype(np.int64)
    hip_at_ref = np.take_along_axis(al, refmax[:, None, :], 1)[:, 0, :]
hipmax = al.argmax(1)
bad = np.argwhere((hipmax != refmax) & (peak > 1e-30))
print(which, "steps with peak > 1e-30: %d of %d; arg-max differs at %d" % (int((peak > 1e-30).sum()), peak.size, len(bad)))
for b, t in bad[:20]:
    print("  row %d step %d: oracle position %d (peak %.6e, runner-up %.6e: gap %.2e of the peak), HIP position %d (HIP values there / at the oracle's position %.6e / %.6e)"
          % (b, t, refmax[b, t], peak[b, t], second[b, t], (peak[b, t] - second[b, t]) / peak[b, t], hipmax[b, t], al[b, hipmax[b, t], t], al[b, refmax[b, t], t]))
rel = np.abs(hip_at_ref - peak) / np.maximum(peak, 1e-300)
for lo, hi in ((1e-6, 2), (1e-12, 1e-6), (1e-20, 1e-12), (1e-30, 1e-20), (1e-37, 1e-30)):
    s = (peak > lo) & (peak <= hi)
    if s.any():
        print("  peak in (%g, %g]: %d steps, relative error of the HIP value at the oracle's peak position: max %.2e, median %.2e" % (lo, hi, s.sum(), rel[s].max(), np.median(rel[s])))
for t0 in range(0, peak.shape[1], max(1, peak.shape[1] // 8)):
    s = np.zeros_like(peak, bool); s[:, t0:t0 + max(1, peak.shape[1] // 8)] = True; s &= peak > 1e-30
    if s.any():
        print("  steps %4d..: %d compared, max relative error %.2e" % (t0, s.sum(), rel[s].max()))
