import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import taco_oracle as O
from util import tiny_hp, to_product_hp, maxabs
import taco_amd
hp = tiny_hp(attention_type="bah_mon")
w = O.init_weights(hp, 1, 31)
ids, L = O.synthetic_inputs(3, 9, 37, ragged=True)
rs = np.random.RandomState(32)
mt, lt, co = rs.rand(3, 12, hp.num_mels), rs.rand(3, 12, hp.num_freq), rs.uniform(0.5, 1.5, size=3)
ids2, L2 = O.synthetic_inputs(6, 18, 33, ragged=True)
mt2, lt2 = rs.rand(6, 18, hp.num_mels), rs.rand(6, 18, hp.num_freq)
a, b = taco_amd.Trainer(to_product_hp(hp), w), taco_amd.Trainer(to_product_hp(hp), w)
a.capture(ids, L, mt, lt, co)
for tr in (a, b):
    tr.train_step(ids, L, mt, lt, co); tr.train_step(ids2, L2, mt2, lt2)
    tr._eager_since_replay = False          # no re-capture: replay the graph recorded before the eager step
    tr.train_step(ids, L, mt, lt, co)
torch.cuda.synchronize()
wa, wb = a.get_weights(), b.get_weights()
bad = [k for k in wa if maxabs(wa[k], wb[k]) > 1e-3]
print("replay without re-capture, mismatching tensors:", len(bad), bad[:3], flush=True)
