"""CPU checks of the TRAINING checkers themselves (no GPU): the NumPy oracle's training-mode forward against the
torch formulation, and the autograd gradients against central finite differences of the NumPy oracle's loss."""
import numpy as np
import pytest

import taco_oracle as O
import torch_formulation as TF
from util import tiny_hp


def _case(atype="bah_mon", seed=5, B=3, T_in=9, T_out=12):
    hp = tiny_hp(attention_type=atype)
    w = O.init_weights(hp, 1, seed)
    ids, L = O.synthetic_inputs(B, T_in, seed + 6, ragged=True)
    rs = np.random.RandomState(seed + 1)
    return hp, w, ids, L, rs.rand(B, T_out, hp.num_mels), rs.rand(B, T_out, hp.num_freq), rs.uniform(0.5, 1.5, size=B)


def _loss(w, hp, ids, L, mt, lt, co):
    r = hp.reduction_factor
    o = O.forward(w, hp, ids, L, n_steps=mt.shape[1] // r, honor_stop=False, teacher_frames=mt[:, r - 1::r], training=True)
    return O.add_loss(o["mel"], mt, o["linear"], lt, co)["loss"], o


@pytest.mark.parametrize("atype", ["bah_mon", "bah"])
def test_training_forward_two_formulations_agree(atype):
    hp, w, ids, L, mt, lt, co = _case(atype)
    loss_np, o = _loss(w, hp, ids, L, mt, lt, co)
    loss_t, _, out = TF.train_grads(w, hp, ids, L, mt, lt, co)
    assert abs(loss_np - loss_t) < 1e-12
    for k in ("mel", "linear", "alignments"):
        assert np.abs(o[k] - out[k]).max() < 1e-12


def test_training_mode_differs_from_inference_only_through_batchnorm():
    hp, w, ids, L, mt, lt, co = _case()
    r = hp.reduction_factor
    kw = dict(n_steps=mt.shape[1] // r, honor_stop=False, teacher_frames=mt[:, r - 1::r])
    upd = {}
    a = O.forward(w, hp, ids, L, training=True, bn_updates=upd, **kw)
    b = O.forward(w, hp, ids, L, training=False, **kw)
    assert np.abs(a["mel"] - b["mel"]).max() > 1e-3          # batch statistics are in use
    # feed the batch statistics back as moving statistics: inference must now reproduce the training forward
    w2 = dict(w)
    for k, v in upd.items():
        w2[k] = (v - 0.99 * np.asarray(w[k], np.float64)) / 0.01
    c = O.forward(w2, hp, ids, L, training=False, **kw)
    assert np.abs(a["mel"] - c["mel"]).max() < 1e-9 and np.abs(a["linear"] - c["linear"]).max() < 1e-9


@pytest.mark.parametrize("atype", ["bah_mon", "bah"])
def test_autograd_gradients_match_finite_differences_of_the_numpy_oracle(atype):
    hp, w, ids, L, mt, lt, co = _case(atype, seed=7)
    _, g, _ = TF.train_grads(w, hp, ids, L, mt, lt, co)
    rs = np.random.RandomState(0)
    names = ["embedding", "prenet/dense_1/kernel", "encoder_cbhg/conv_bank/conv1d_3/kernel", "encoder_cbhg/conv_bank/conv1d_2/gamma",
             "encoder_cbhg/proj_1/beta", "encoder_cbhg/highway_2/T/kernel", "encoder_cbhg/bigru/bw/gates/kernel",
             "attention/memory_layer/kernel", "attention/query_layer/kernel", "attention/attention_v", "decoder/prenet/dense_1/kernel",
             "decoder/attention_gru/candidate/kernel", "decoder/concat_projection/kernel", "decoder/gru_2/gates/bias",
             "decoder/frame_projection/kernel", "post_cbhg/conv_bank/conv1d_4/kernel", "post_cbhg/proj_2/gamma", "post_cbhg/dense/kernel",
             "post_cbhg/bigru/fw/candidate/kernel", "linear/kernel"]
    if atype == "bah_mon":
        names.append("attention/attention_score_bias")
    eps = 1e-6
    for nm in names:
        gv = np.asarray(g[nm])
        flat = np.abs(gv).reshape(-1)
        idx = int(np.argmax(flat)) if rs.rand() < 0.5 else int(rs.randint(flat.size))
        wp, wm = dict(w), dict(w)
        a = np.array(w[nm], np.float64); a.reshape(-1)[idx] += eps; wp[nm] = a
        b = np.array(w[nm], np.float64); b.reshape(-1)[idx] -= eps; wm[nm] = b
        fd = (_loss(wp, hp, ids, L, mt, lt, co)[0] - _loss(wm, hp, ids, L, mt, lt, co)[0]) / (2 * eps)
        an = float(gv.reshape(-1)[idx])
        assert abs(fd - an) < 1e-6 + 2e-4 * max(abs(an), abs(fd)), (nm, idx, fd, an)
