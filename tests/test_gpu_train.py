"""Training path through the C ABI vs the float64 checkers: the NumPy oracle's training-mode
forward and tests/torch_formulation.py's reverse-mode autograd of the same graph."""
import numpy as np
import pytest

import taco_oracle as O
import torch_formulation as TF
from util import tiny_hp, to_product_hp, maxabs

pytestmark = pytest.mark.gpu


def _setup(atype="bah_mon", B=3, T_in=9, T_out=12, seed=5, ragged=True, **kw):
    hp = tiny_hp(attention_type=atype, **kw)
    w = O.init_weights(hp, 1, seed)
    ids, L = O.synthetic_inputs(B, T_in, seed + 6, ragged=ragged)
    rs = np.random.RandomState(seed + 1)
    mt = rs.rand(B, T_out, hp.num_mels)
    lt = rs.rand(B, T_out, hp.num_freq)
    co = rs.uniform(0.5, 1.5, size=B)
    return hp, w, ids, L, mt, lt, co


def _trainer(hp, w):
    import taco_amd
    return taco_amd.Trainer(to_product_hp(hp), w)


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("atype", ["bah_mon", "bah"])
def test_training_forward_matches_oracle(atype):
    import torch
    hp, w, ids, L, mt, lt, co = _setup(atype)
    r = hp.reduction_factor
    upd = {}
    ref = O.forward(w, hp, ids, L, n_steps=mt.shape[1] // r, honor_stop=False, teacher_frames=mt[:, r - 1::r], training=True, bn_updates=upd)
    tr = _trainer(hp, w)
    losses = tr.forward_backward(ids, L, mt, lt, co, backward=False, keep_outputs=True)
    torch.cuda.synchronize()
    assert maxabs(tr.mel_outputs.cpu().numpy(), ref["mel"]) < 1e-4
    assert maxabs(tr.linear_outputs.cpu().numpy(), ref["linear"]) < 1e-4
    assert maxabs(tr.alignments.cpu().numpy(), ref["alignments"]) < 1e-4
    want = O.add_loss(ref["mel"], mt, ref["linear"], lt, co)
    got = losses.cpu().numpy()
    for i, k in enumerate(("loss", "mel_loss", "linear_loss", "loss_without_coeff")):
        assert abs(got[i] - want[k]) < 1e-5 * max(1.0, abs(want[k])), k
    # BatchNorm moving averages were updated in the flat parameter buffer (UPDATE_OPS, tacotron.py:334)
    now = tr.get_weights()
    assert len(upd) == 2 * (hp.enc_bank_size + hp.post_bank_size + len(hp.enc_proj_sizes) + len(hp.post_proj_sizes))
    for k, v in upd.items():
        assert maxabs(now[k], v) < 1e-5, k
    for k in w:
        if k not in upd:
            assert np.array_equal(now[k], np.asarray(w[k], np.float32)), k


@pytest.mark.parametrize("atype,ragged", [("bah_mon", True), ("bah", False)])
def test_gradients_match_autograd(atype, ragged):
    import torch
    hp, w, ids, L, mt, lt, co = _setup(atype, ragged=ragged)
    loss, g, _ = TF.train_grads(w, hp, ids, L, mt, lt, co)
    tr = _trainer(hp, w)
    losses = tr.forward_backward(ids, L, mt, lt, co)
    torch.cuda.synchronize()
    assert abs(float(losses[0]) - loss) < 1e-5
    got = tr.grad_dict()
    gn = np.sqrt(sum(float((v ** 2).sum()) for v in g.values()))
    bad = []
    for k, v in g.items():
        err = float(np.abs(got[k] - v).max())
        scale = max(float(np.abs(v).max()), 1e-3 * gn)
        if err > 2e-3 * scale:
            bad.append((k, err, float(np.abs(v).max())))
    assert not bad, bad[:8]
    for k in w:
        if k.endswith(("/moving_mean", "/moving_variance")):
            assert not got[k].any(), k
    tr.close()
