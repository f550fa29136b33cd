"""CPU: the NumPy oracle against the independently written torch formulation (tests/torch_formulation.py)
over the whole forward, every speaker mode x attention type, ragged lengths, manual attention, plus
hypothesis-driven invariants of the oracle (SURVEY section 4 'property tests')."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import taco_oracle as O
import torch_formulation as TF2
from util import tiny_hp


@pytest.mark.parametrize("atype", ["bah", "bah_norm", "bah_mon"])
@pytest.mark.parametrize("mt,ns", [("single", 1), ("simple", 3), ("deepvoice", 3)])
def test_whole_forward_two_formulations_agree(atype, mt, ns):
    hp = tiny_hp(attention_type=atype, model_type=mt)
    w = O.init_weights(hp, ns, 50)
    ids, L = O.synthetic_inputs(3, 9, 51, ragged=True)
    spk = np.array([1, 0, 2], np.int32) if ns > 1 else None
    a = O.forward(w, hp, ids, L, speaker_id=spk, num_speakers=ns)
    b = TF2.forward(w, hp, ids, L, speaker_id=spk, num_speakers=ns)
    for k in ("mel", "linear", "alignments"):
        assert a[k].shape == b[k].shape
        assert np.abs(a[k] - b[k]).max() < 1e-9, k


def test_deepvoice_tables_and_manual_attention_agree():
    hp = tiny_hp(model_type="deepvoice", speaker_embedding_size=1)
    w = O.init_weights(hp, 2, 52)
    ids, L = O.synthetic_inputs(2, 7, 53)
    spk = np.array([1, 0], np.int32)
    man = np.random.RandomState(0).dirichlet(np.ones(7), (2, hp.max_iters))
    a = O.forward(w, hp, ids, L, speaker_id=spk, num_speakers=2, manual_alignments=man)
    b = TF2.forward(w, hp, ids, L, speaker_id=spk, num_speakers=2, manual=man)
    assert np.abs(a["linear"] - b["linear"]).max() < 1e-9 and np.abs(a["alignments"] - b["alignments"]).max() < 1e-12


def test_medium_width_agreement():
    hp = O.OracleHParams.scaled(4, num_mels=20, num_freq=65, max_iters=5, reduction_factor=4)
    w = O.init_weights(hp, 1, 54)
    ids, L = O.synthetic_inputs(2, 20, 55, ragged=True)
    a, b = O.forward(w, hp, ids, L), TF2.forward(w, hp, ids, L)
    assert np.abs(a["mel"] - b["mel"]).max() < 1e-9 and np.abs(a["linear"] - b["linear"]).max() < 1e-9


@settings(max_examples=12, deadline=None)
@given(seed=st.integers(0, 10_000), B=st.integers(1, 4), T=st.integers(2, 12))
def test_property_rows_independent_and_padding_inert(seed, B, T):
    hp = tiny_hp(max_iters=3)
    w = O.init_weights(hp, 1, 3)
    ids, L = O.synthetic_inputs(B, T, seed, ragged=T > 2)
    out = O.forward(w, hp, ids, L)
    # each row alone reproduces its row of the batch
    b = seed % B
    one = O.forward(w, hp, ids[b:b + 1], L[b:b + 1])
    assert np.allclose(one["mel"][0], out["mel"][b], atol=1e-12)
    # ids strictly after EOS+conv halo never matter for OTHER rows; encoder output is zero past the length
    taps = {}
    O.forward(w, hp, ids, L, taps=taps)
    for r in range(B):
        assert np.all(taps["encoder"][r, L[r]:] == 0)
    assert np.isfinite(out["linear"]).all()
    # monotonic alignments: non-negative, mass never grows
    assert out["alignments"].min() >= 0 and out["alignments"].sum(1).max() <= 1 + 1e-9


@settings(max_examples=10, deadline=None)
@given(seed=st.integers(0, 1000), k=st.integers(1, 9), T=st.integers(1, 14))
def test_property_conv_same_shape_and_shift(seed, k, T):
    rs = np.random.RandomState(seed)
    x = rs.randn(1, T, 3); kern = rs.randn(k, 3, 2)
    y = O.conv1d_same(x, kern, np.zeros(2))
    assert y.shape == (1, T, 2)
    # linearity
    y2 = O.conv1d_same(2 * x, kern, np.zeros(2))
    assert np.allclose(y2, 2 * y)
