import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import taco_oracle as O
from util import tiny_hp, to_product_hp, maxabs
import taco_amd
hp = tiny_hp(attention_type="bah_mon")
w = O.init_weights(hp, 1, 31)
rs = np.random.RandomState(32)
r = hp.reduction_factor
for (B, T, To, seed) in ((3, 9, 12, 37), (6, 18, 18, 33)):
    ids, L = O.synthetic_inputs(B, T, seed, ragged=True)
    mt, lt = rs.rand(B, To, hp.num_mels), rs.rand(B, To, hp.num_freq)
    res = {}
    for fill in (0x00, 0xFF, 0x7F):
        tr = taco_amd.Trainer(to_product_hp(hp), w)
        nb = int(tr._lib.taco_train_workspace_bytes(tr._h, B, T, To))
        tr._ws = torch.full((nb,), fill, dtype=torch.uint8, device="cuda")
        tr.forward_backward(ids, L, mt, lt)
        torch.cuda.synchronize()
        res[fill] = tr.grad_dict()
    for fill in (0xFF, 0x7F):
        bad = [(k, maxabs(res[fill][k], res[0][k])) for k in res[0] if not np.array_equal(res[fill][k], res[0][k])]
        print("B=%d T=%d lengths %s: fill 0x%02X -> %d tensors differ from the zero-filled run: %s" % (B, T, list(L), fill, len(bad), bad[:8]), flush=True)
