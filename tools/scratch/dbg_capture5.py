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
def rng(t): return (t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
for mode in ("plain", "prealloc-eager-ws", "device-inputs"):
    a, b = taco_amd.Trainer(to_product_hp(hp), w), taco_amd.Trainer(to_product_hp(hp), w)
    if mode == "prealloc-eager-ws":
        a._ws_eager = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
    a.capture(ids, L, mt, lt, co)
    a.get_weights()
    dv = lambda x, dt: torch.as_tensor(np.asarray(x)).to("cuda", dt)
    s1 = (ids, L, mt, lt, co); s2 = (ids2, L2, mt2, lt2)
    if mode == "device-inputs":
        s1 = (dv(ids, torch.int32), dv(L, torch.int32), dv(mt, torch.float32), dv(lt, torch.float32), dv(co, torch.float32))
        s2 = (dv(ids2, torch.int32), dv(L2, torch.int32), dv(mt2, torch.float32), dv(lt2, torch.float32))
    for tr in (a, b):
        tr.train_step(*s1); tr.train_step(*s2); tr.train_step(*s1)
    torch.cuda.synchronize()
    wa, wb = a.get_weights(), b.get_weights()
    bad = [k for k in wa if maxabs(wa[k], wb[k]) > 1e-3]
    named = {"ws": a._ws, "ws_eager": a._ws_eager, "params": a.params, "grads": a.grads, "losses": a.losses, "adam_m": a.adam.m, "adam_v": a.adam.v}
    for i, t in enumerate(a._g_in):
        if t is not None: named["g_in%d" % i] = t
    r = {k: rng(v) for k, v in named.items()}
    ov = [(x, y) for x in r for y in r if x < y and r[x][0] < r[y][1] and r[y][0] < r[x][1]]
    print(mode, "bad:", bad[:2], "overlaps:", ov, flush=True)
