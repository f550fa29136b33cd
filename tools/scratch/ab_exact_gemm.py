"""A/B of the training GEMM engines (split-bf16 vs exact fp32) tensor by tensor: python tools/scratch/ab_exact_gemm.py [tiny|ref]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import taco_amd
import taco_oracle as O
from util import tiny_hp, to_product_hp

mode = sys.argv[1] if len(sys.argv) > 1 else "tiny"
hp = tiny_hp(attention_type="bah_mon") if mode == "tiny" else O.OracleHParams(max_iters=8)
B, T_in, T_out = (3, 9, 12) if mode == "tiny" else (9, 14, 8 * hp.reduction_factor)
w = O.init_weights(hp, 1, 5)
ids, L = O.synthetic_inputs(B, T_in, 11, ragged=True)
rs = np.random.RandomState(6)
mt, lt = rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq)
tr = taco_amd.Trainer(to_product_hp(hp), w)
tr.set_exact_gemm(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
tr.forward_backward(ids, L, mt, lt, None)
torch.cuda.synchronize()
got = tr.grad_dict()
tr.set_exact_gemm(1)
tr.forward_backward(ids, L, mt, lt, None)
torch.cuda.synchronize()
ref = tr.grad_dict()
rows = sorted(((float(np.abs(got[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-12)), k) for k in ref), reverse=True)
for r, k in rows[:40]:
    print("%-55s rel %.3e" % (k, r))
gn = np.sqrt(sum(float((ref[k] ** 2).sum()) for k in ref)); e2 = np.sqrt(sum(float(((got[k] - ref[k]) ** 2).sum()) for k in ref))
print("whole gradient: |diff| / |g| = %.3e" % (e2 / gn))
print("tensors over 1e-3:", sum(1 for r, _ in rows if r > 1e-3), "of", len(rows))
