"""One hand-computable known-answer test per TF-sem item of SURVEY Appendix A that tests/test_oracle.py does not already hold: every
expected value below is derived in the test from the DEFINITION (loops / closed forms written out here), not from the oracle's own
vectorised code -- so that a future run with a real TensorFlow 1.x has one small case per assumption to confirm or refute, and so that a
slip in the oracle's vectorisation cannot hide.  CPU only; test infrastructure."""
import math

import numpy as np
import pytest

import taco_oracle as O


def test_a1_dense_acts_on_the_last_axis_bias_then_activation():
    x = np.array([[[1.0, -2.0], [0.5, 3.0]]])                       # [1,2,2]
    w = {"d/kernel": np.array([[1.0, 2.0, -1.0], [0.0, 1.0, 1.0]]), "d/bias": np.array([0.5, -10.0, 0.0])}
    y = O.dense(x, w, "d", O.relu)
    exp = np.zeros((1, 2, 3))
    for t in range(2):
        for o in range(3):
            exp[0, t, o] = max(0.0, sum(x[0, t, i] * w["d/kernel"][i, o] for i in range(2)) + w["d/bias"][o])
    assert np.array_equal(y, exp)
    assert np.array_equal(O.dense(x, w, "d", None, bias=False)[0, 0], [1.0, 0.0, -3.0])


def test_a9_scores_plain_and_normalised():
    keys = np.array([[[0.1, -0.2], [0.3, 0.4], [0.0, 0.0]]])         # [1,3,2]
    q = np.array([[0.5, -0.5]])
    w = {"attention/attention_v": np.array([2.0, -1.0]), "attention/attention_g": np.array(0.7), "attention/attention_b": np.array([0.05, -0.05])}
    e = O.attention_score(q, keys, w, "bah")
    exp = [2.0 * math.tanh(k0 + 0.5) - 1.0 * math.tanh(k1 - 0.5) for k0, k1 in keys[0]]
    assert np.allclose(e[0], exp, rtol=0, atol=1e-15)
    assert np.array_equal(e, O.attention_score(q, keys, w, "bah_mon"))          # the monotonic mechanism scores like the plain one
    en = O.attention_score(q, keys, w, "bah_norm")
    nv = [0.7 * 2.0 / math.sqrt(5.0), 0.7 * -1.0 / math.sqrt(5.0)]
    exp = [nv[0] * math.tanh(k0 + 0.5 + 0.05) + nv[1] * math.tanh(k1 - 0.5 - 0.05) for k0, k1 in keys[0]]
    assert np.allclose(en[0], exp, rtol=0, atol=1e-15)


def _monotonic_recursive(p, prev):
    """Raffel et al. 2017 eq. (8)-(9), the definition TF's mode='recursive' implements: q_j = (1 - p_{j-1}) q_{j-1} + prev_j, alpha_j = p_j q_j"""
    B, T = p.shape
    out = np.zeros_like(p)
    for b in range(B):
        qj = 0.0
        for j in range(T):
            qj = (0.0 if j == 0 else (1.0 - p[b, j - 1]) * qj) + prev[b, j]
            out[b, j] = p[b, j] * qj
    return out


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_a10_the_parallel_monotonic_form_equals_the_recursive_definition(seed):
    rs = np.random.RandomState(seed)
    p = rs.uniform(0.02, 0.9, size=(3, 17))
    prev = rs.dirichlet(np.ones(17), size=3)
    assert np.abs(O.monotonic_attention_parallel(p, prev) - _monotonic_recursive(p, prev)).max() < 1e-13


def test_a10_first_step_from_the_one_hot_and_mass_that_leaves_the_end():
    p = np.array([[0.25, 0.5, 0.5, 1.0, 0.3]])
    a0 = O.initial_alignments(1, 5, "bah_mon", np.float64)
    assert np.array_equal(a0, [[1, 0, 0, 0, 0]])
    a1 = O.monotonic_attention_parallel(p, a0)
    # alpha_1[j] = p_j * prod_{k<j}(1 - p_k): stop at j with probability p_j after passing the ones before
    exp = [0.25, 0.75 * 0.5, 0.75 * 0.5 * 0.5, 0.75 * 0.5 * 0.5 * 1.0, 0.0]
    assert np.allclose(a1[0], exp, rtol=0, atol=1e-12)                 # p = 1 is a hard stop (clip of 1-p at float32 tiny inside)
    # rows need not sum to one: with small p everywhere mass walks off the end
    a = O.monotonic_attention_parallel(np.full((1, 4), 0.1), np.array([[0.0, 0.0, 0.0, 1.0]]))
    assert abs(a.sum() - 0.1) < 1e-15
    assert np.array_equal(O.initial_alignments(2, 3, "bah", np.float64), np.zeros((2, 3)))


def test_a10_the_score_bias_enters_before_the_sigmoid():
    keys, q = np.zeros((1, 2, 1)), np.zeros((1, 1))
    w = {"attention/attention_v": np.array([1.0]), "attention/attention_score_bias": np.array(-1.5)}
    a = O.attention_alignments(q, keys, np.array([[1.0, 0.0]]), w, "bah_mon")
    s = 1.0 / (1.0 + math.exp(1.5))
    assert np.allclose(a[0], [s, (1 - s) * s], rtol=0, atol=1e-15)


def test_a2_batch_norm_training_statistics_and_moving_averages():
    y = np.array([[[1.0, 10.0], [3.0, 10.0]], [[5.0, 14.0], [7.0, 14.0]]])     # [2,2,2]: every (b,t) counts, nothing is masked
    w = {"bn/gamma": np.array([2.0, 1.0]), "bn/beta": np.array([0.5, 0.0]), "bn/moving_mean": np.array([1.0, 0.0]), "bn/moving_variance": np.array([1.0, 2.0])}
    upd = {}
    out = O.batch_norm_train(y, w, "bn", upd)
    mu = [4.0, 12.0]
    var = [5.0, 4.0]                                                           # BIASED: mean of squared deviations
    for b in range(2):
        for t in range(2):
            for c in range(2):
                assert abs(out[b, t, c] - (w["bn/gamma"][c] * (y[b, t, c] - mu[c]) / math.sqrt(var[c] + 1e-3) + w["bn/beta"][c])) < 1e-14
    assert np.allclose(upd["bn/moving_mean"], [1.0 * 0.99 + 4.0 * 0.01, 12.0 * 0.01], rtol=0, atol=1e-15)
    assert np.allclose(upd["bn/moving_variance"], [0.99 + 0.05, 1.98 + 0.04], rtol=0, atol=1e-15)
    inf = O.batch_norm_infer(y, w, "bn")
    assert abs(inf[0, 0, 1] - 10.0 / math.sqrt(2.0 + 1e-3)) < 1e-14


def test_a13_softsign_and_the_deepvoice_vectors():
    assert np.array_equal(O.softsign(np.array([-3.0, 0.0, 1.0])), [-0.75, 0.0, 0.5])


def test_a14_adam_two_updates_by_hand_epsilon_outside_the_bias_correction():
    th, m, v = np.array([1.0, -1.0]), np.zeros(2), np.zeros(2)
    g1, g2 = np.array([0.3, -0.4]), np.array([0.1, 0.2])                       # norms 0.5 and 0.224 -- under the clip
    lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-8
    th1, m1, v1, gn1 = O.adam_clip_step(th, g1, m, v, 1, lr)
    th2, m2, v2, gn2 = O.adam_clip_step(th1, g2, m1, v1, 2, lr)
    e_th = th.copy()
    e_m, e_v = [0.0, 0.0], [0.0, 0.0]
    for t, g in ((1, g1), (2, g2)):
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        for i in range(2):
            e_m[i] = b1 * e_m[i] + (1 - b1) * g[i]
            e_v[i] = b2 * e_v[i] + (1 - b2) * g[i] * g[i]
            e_th[i] -= lr_t * e_m[i] / (math.sqrt(e_v[i]) + eps)
    assert np.allclose(th2, e_th, rtol=0, atol=1e-15) and abs(gn1 - 0.5) < 1e-15
    # the first update of TF's form is lr * sign(g) up to eps / sqrt(1 - beta2) -- and it is NOT the textbook / torch form, whose epsilon
    # is added to the bias-corrected sqrt(v_hat): the two differ in the 7th digit at the very first step
    torch_form = th - lr * (m1 / (1 - b1)) / (np.sqrt(v1 / (1 - b2)) + eps)
    assert np.abs(th1 - torch_form).max() > 1e-10 and np.abs(th1 - torch_form).max() < 1e-6


def test_a14_clip_by_global_norm_scales_every_gradient_by_the_same_factor():
    g = np.array([3.0, 4.0, 12.0])                                             # norm 13
    th, m, v, gn = O.adam_clip_step(np.zeros(3), g, np.zeros(3), np.zeros(3), 1, 1.0, beta1=0.0, beta2=0.0, eps=1.0)
    # beta1 = beta2 = 0, eps = 1: update = g_c / (|g_c| + 1) with g_c = g / 13
    gc = g / 13.0
    assert abs(gn - 13.0) < 1e-15 and np.allclose(th, -gc / (np.abs(gc) + 1.0), rtol=0, atol=1e-15)
    _, _, _, gn = O.adam_clip_step(np.zeros(2), np.array([0.3, 0.4]), np.zeros(2), np.zeros(2), 1, 1.0)
    assert abs(gn - 0.5) < 1e-15                                               # reported norm is the unclipped one (tacotron.py:329-330)


def test_a14_learning_rate_schedules_at_known_steps():
    # mode 0, pretrained start (warm-up 40000): init * w^0.5 * min(step * w^-1.5, step^-0.5), step = global_step + 1
    assert abs(O.learning_rate(0, 0.002, 0, False) - 0.002 * 200.0 * 40000.0 ** -1.5) < 1e-18
    assert abs(O.learning_rate(39999, 0.002, 0, False) - 0.002) < 1e-15         # the peak, at step == warm-up
    assert abs(O.learning_rate(159999, 0.002, 0, False) - 0.001) < 1e-15        # 4 x warm-up: half the peak
    assert abs(O.learning_rate(3999, 0.002, 0, True) - 0.002) < 1e-15           # random init: warm-up 4000
    assert abs(O.learning_rate(2999, 0.002, 1, True) - 0.002 * 0.95) < 1e-15    # mode 1: 0.95 per 3000 steps, continuous
    assert abs(O.learning_rate(5999, 0.002, 1, True) - 0.002 * 0.95 ** 2) < 1e-15


def test_a20_loss_by_hand_with_coefficients_and_the_priority_band():
    mel_o, mel_t = np.zeros((2, 1, 2)), np.array([[[1.0, 3.0]], [[2.0, 2.0]]])
    F = 24
    lin_o = np.zeros((2, 1, F))
    lin_t = np.tile(np.arange(F, dtype=np.float64), (2, 1, 1))
    c = np.array([1.0, 0.5])
    L = O.add_loss(mel_o, mel_t, lin_o, lin_t, c)
    assert abs(L["mel_loss"] - 2.0) < 1e-15 and abs(L["linear_loss"] - 11.5) < 1e-15
    assert abs(L["loss"] - ((1 + 3 + 0.5 * 2 + 0.5 * 2) / 4 + (11.5 + 0.5 * 11.5) / 2)) < 1e-14
    assert abs(L["loss_without_coeff"] - 13.5) < 1e-15
    P = O.add_loss(mel_o, mel_t, lin_o, lin_t, c, prioritize_loss=True, sample_rate=24000)
    lo, up = int(165 / 12000 * F), int(5000 / 12000 * F)                        # bins [0, 10)
    band = np.arange(lo, up).mean()
    assert (lo, up) == (0, 10)
    assert abs(P["linear_loss"] - 0.5 * (11.5 + band)) < 1e-14
    assert abs(P["loss"] - (1.5 + 0.5 * 0.75 * 11.5 + 0.5 * 0.75 * band)) < 1e-14
