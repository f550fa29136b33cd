"""CPU tests of the oracle itself: hand-derivable known answers for every TF-sem assumption of
SURVEY App. A, and a second independent formulation (torch CPU functional ops)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import taco_oracle as O
from util import tiny_hp

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "tiny_forward.npz")


def test_param_count_and_bytes_match_survey():
    hp = O.OracleHParams()
    assert sum(int(np.prod(s)) for s in O.weight_shapes(hp).values()) == 9316050
    total, per_step = O.algorithmic_bytes(hp, 32, 128, 128)
    assert abs(total / 1e9 - 1.992) < 0.001 and abs(per_step / 1e6 - 14.69) < 0.01


def test_conv1d_same_even_kernel_pads_right():
    # k=2: pad_left=0, pad_right=1 -> y[t] = x[t]*w0 + x[t+1]*w1, last uses zero
    x = np.arange(1, 6, dtype=np.float64).reshape(1, 5, 1)
    k = np.array([10.0, 1.0]).reshape(2, 1, 1)
    y = O.conv1d_same(x, k, np.zeros(1))
    assert y[0, :, 0].tolist() == [12, 23, 34, 45, 50]
    # k=3 symmetric; k=4: pad_left=1, pad_right=2
    k4 = np.array([1000.0, 100.0, 10.0, 1.0]).reshape(4, 1, 1)
    y4 = O.conv1d_same(x, k4, np.zeros(1))
    assert y4[0, :, 0].tolist() == [123, 1234, 2345, 3450, 4500]


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 16])
def test_conv1d_matches_torch(k):
    rs = np.random.RandomState(k)
    x = rs.randn(2, 11, 6); w = rs.randn(k, 6, 5); b = rs.randn(5)
    y = O.conv1d_same(x, w, b)
    pl = (k - 1) // 2
    xt = F.pad(torch.from_numpy(x).permute(0, 2, 1), (pl, k - 1 - pl))
    yt = F.conv1d(xt, torch.from_numpy(w).permute(2, 1, 0), torch.from_numpy(b)).permute(0, 2, 1).numpy()
    assert np.abs(y - yt).max() < 1e-12


def test_maxpool_last_element_and_torch():
    x = np.array([3.0, 1.0, 4.0, 1.0, 5.0]).reshape(1, 5, 1)
    assert O.maxpool_same_stride1(x, 2)[0, :, 0].tolist() == [3, 4, 4, 5, 5]
    rs = np.random.RandomState(0)
    x = rs.randn(2, 9, 4)
    yt = F.max_pool1d(F.pad(torch.from_numpy(x).permute(0, 2, 1), (0, 1), value=-np.inf), 2, 1).permute(0, 2, 1).numpy()
    assert np.array_equal(O.maxpool_same_stride1(x, 2), yt)


def test_batchnorm_affine_and_order():
    w = {"c/kernel": np.ones((1, 1, 1), np.float32), "c/bias": np.array([-2.0], np.float32),
         "c/gamma": np.array([2.0], np.float32), "c/beta": np.array([0.5], np.float32),
         "c/moving_mean": np.array([1.0], np.float32), "c/moving_variance": np.array([4.0 - 1e-3], np.float32)}
    x = np.array([1.0, 5.0]).reshape(1, 2, 1)
    # conv -> relu -> BN : relu(x-2) = [0,3]; BN = 2*(v-1)/2+0.5 = [-0.5, 2.5]   (BN after activation)
    y = O.conv1d_bn(x, w, "c", O.relu)
    assert np.allclose(y[0, :, 0], [-0.5, 2.5], atol=1e-6)


def test_highway_zero_weights():
    w = {"h/H/kernel": np.zeros((3, 3), np.float32), "h/H/bias": np.zeros(3, np.float32),
         "h/T/kernel": np.zeros((3, 3), np.float32), "h/T/bias": -np.ones(3, np.float32)}
    x = np.array([[1.0, -2.0, 3.0]])
    T = 1 / (1 + np.e)
    assert np.allclose(O.highwaynet(x, w, "h"), x * (1 - T))


def test_gru_zero_kernels_and_gate_order():
    n = 2
    w = {"g/gates/kernel": np.zeros((1 + n, 2 * n), np.float32), "g/gates/bias": np.ones(2 * n, np.float32),
         "g/candidate/kernel": np.zeros((1 + n, n), np.float32), "g/candidate/bias": np.zeros(n, np.float32)}
    h = np.array([[0.5, -1.0]]); x = np.array([[2.0]])
    u = 1 / (1 + np.exp(-1.0))
    assert np.allclose(O.gru_cell(x, h, w, "g"), u * h)            # c = tanh(0) = 0
    # reset gate multiplies the state BEFORE the candidate matmul; columns are (r | u)
    w["g/candidate/kernel"][1, 0] = 1.0                            # c0 = tanh(r0*h0)
    w["g/gates/bias"][:] = [100.0, 100.0, -100.0, -100.0]          # r=1, u=0
    assert np.allclose(O.gru_cell(x, h, w, "g")[0, 0], np.tanh(0.5))
    w["g/gates/bias"][:] = [-100.0, -100.0, -100.0, -100.0]        # r=0
    assert np.allclose(O.gru_cell(x, h, w, "g")[0, 0], 0.0)


def test_dynamic_rnn_masking_and_reverse_sequence():
    rs = np.random.RandomState(3)
    hp = tiny_hp()
    w = O.init_weights(hp, 1, 0)
    x = rs.randn(2, 6, hp.enc_rnn_size)
    L = np.array([4, 6])
    out = O.bidirectional_gru(x, L, w, "encoder_cbhg/bigru")
    assert np.all(out[0, 4:] == 0) and np.any(out[0, 3] != 0)
    # row 0 with length 4 must equal the unpadded 4-step sequence
    out4 = O.bidirectional_gru(x[:1, :4], np.array([4]), w, "encoder_cbhg/bigru")
    assert np.allclose(out[0, :4], out4[0])
    r = O.reverse_sequence(np.arange(6).reshape(1, 6, 1).astype(float), np.array([4]))
    assert r[0, :, 0].tolist() == [3, 2, 1, 0, 4, 5]


def test_softmax_rows_sum_to_one_and_monotonic_hard_stop():
    e = np.random.RandomState(0).randn(3, 7)
    assert np.allclose(O.softmax_rows(e).sum(1), 1)
    # p = 1 at j=2 -> all mass arriving at or before j=2 stops there
    p = np.array([[0.0, 0.0, 1.0, 0.5, 0.5]])
    prev = np.array([[1.0, 0, 0, 0, 0]])
    a = O.monotonic_attention_parallel(p, prev)
    assert np.allclose(a, [[0, 0, 1, 0, 0]], atol=1e-7)
    # recurrence check against the sequential definition q_j = (1-p_{j-1}) q_{j-1} + prev_j ; alpha_j = p_j q_j
    rs = np.random.RandomState(1)
    p = rs.uniform(0.05, 0.95, (2, 9)); prev = rs.dirichlet(np.ones(9), 2)
    q = np.zeros_like(p); q[:, 0] = prev[:, 0]
    for j in range(1, 9):
        q[:, j] = (1 - p[:, j - 1]) * q[:, j - 1] + prev[:, j]
    assert np.allclose(O.monotonic_attention_parallel(p, prev), p * q, atol=1e-9)


def test_initial_alignments():
    assert O.initial_alignments(2, 4, "bah", np.float64).sum() == 0
    assert O.initial_alignments(2, 4, "bah_mon", np.float64)[:, 0].tolist() == [1, 1]


def test_stop_rule_ends_loop_when_all_rows_emit_zero():
    hp = tiny_hp()
    w = O.init_weights(hp, 1, 0)
    w["decoder/frame_projection/kernel"][:] = 0
    w["decoder/frame_projection/bias"][:] = 0
    ids, L = O.synthetic_inputs(2, 6, 0)
    out = O.forward(w, hp, ids, L)
    assert out["stop_step"] == 1 and out["mel"].shape[1] == hp.reduction_factor and out["alignments"].shape[2] == 1


def test_batch_rows_independent_and_pad_invariance():
    hp = tiny_hp()
    w = O.init_weights(hp, 1, 1)
    ids, L = O.synthetic_inputs(3, 9, 5, ragged=True)
    a = O.forward(w, hp, ids, L)
    perm = np.array([2, 0, 1])
    b = O.forward(w, hp, ids[perm], L[perm])
    assert np.allclose(a["mel"][perm], b["mel"], atol=1e-12)
    # encoder BiGRU output is zero past input_lengths
    taps = {}
    O.forward(w, hp, ids, L, taps=taps)
    for r in range(3):
        assert np.all(taps["encoder"][r, L[r]:] == 0)


def test_reduction_reshape_and_feedback():
    hp = tiny_hp()
    w = O.init_weights(hp, 1, 2)
    ids, L = O.synthetic_inputs(1, 5, 2)
    taps = {}
    out = O.forward(w, hp, ids, L, taps=taps)
    r, M = hp.reduction_factor, hp.num_mels
    y0 = taps["steps"][0]["y"]
    assert np.allclose(out["mel"][0, :r].reshape(-1), y0[0])          # [B,n,r*M] -> [B,n*r,M] is a view


def test_golden_fixture_pins_the_oracle():
    g = np.load(GOLDEN, allow_pickle=False)
    from golden.make_golden import fixture_config
    hp, w, ids, L, spk, ns = fixture_config()
    out = O.forward(w, hp, ids, L, speaker_id=spk, num_speakers=ns)
    assert np.array_equal(g["inputs"], ids) and np.array_equal(g["input_lengths"], L)
    assert np.abs(out["mel"] - g["mel"]).max() < 1e-9
    assert np.abs(out["linear"] - g["linear"]).max() < 1e-9
    assert np.abs(out["alignments"] - g["alignments"]).max() < 1e-9


@pytest.mark.parametrize("mt,ns,at", [("single", 1, "bah_mon"), ("deepvoice", 3, "bah"), ("simple", 2, "bah_norm")])
def test_torch_cpu_restatement_equals_the_numpy_oracle(mt, ns, at):
    """oracle/taco_torch_cpu.py (the fp32 torch-CPU op-for-op restatement SURVEY 8(d) names as the CPU baseline; timed by bench.py's
    cpu_baseline leg) is the same function as the NumPy oracle: 1e-12 in float64, 1e-5 in float32, ragged lengths, every model type."""
    import taco_torch_cpu as TT
    hp = O.OracleHParams(max_iters=5, model_type=mt, attention_type=at)
    w = O.init_weights(hp, ns, 7)
    ids, L = O.synthetic_inputs(3, 11, 8, ragged=True)
    spk = (np.arange(3) % ns).astype(np.int32) if ns > 1 else None
    ref = O.forward(w, hp, ids, L, speaker_id=spk, num_speakers=ns, honor_stop=False)
    got = TT.TorchCpuTacotron(w, hp, ns, dtype=torch.float64).forward(ids, L, spk)
    g32 = TT.TorchCpuTacotron(w, hp, ns).forward(ids, L, spk)
    for k in ("mel", "linear", "alignments"):
        assert np.abs(got[k] - ref[k]).max() < 1e-12, k
        assert np.abs(g32[k] - ref[k]).max() < 1e-5, k
