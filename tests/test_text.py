"""Text front-end and batch contract, CPU."""
import numpy as np
import pytest

from taco_amd import text as X
from taco_amd.hparams import NUM_SYMBOLS, PAD_ID, EOS_ID


def test_symbol_table_is_the_80_symbols_of_the_embedding():
    assert len(X.symbols) == NUM_SYMBOLS == 80 and len(set(X.symbols)) == 80
    assert X.symbols[PAD_ID] == "_" and X.symbols[EOS_ID] == "~"
    assert X.symbols[2] == "ᄀ" and X.symbols[2 + 19] == "ᅡ" and X.symbols[2 + 19 + 21] == "ᆨ" and X.symbols[-1] == " "


def test_hangul_decomposition_known_answers():
    assert X.hangul_to_jamo("가") == "가"                       # ga  = lead 0, vowel 0, no tail
    assert X.hangul_to_jamo("힣") == "힣"                 # hih = last lead, last vowel, last tail
    assert X.hangul_to_jamo("안녕") == "안녕"   # annyeong
    assert X.hangul_to_jamo("a1 !") == "a1 !"


def test_text_to_sequence_and_back():
    s = "안녕하세요, 반갑습니다!"
    ids = X.text_to_sequence(s)
    assert ids.dtype == np.int32 and ids[-1] == EOS_ID and (ids[:-1] >= 2).all() and (ids < 80).all()
    assert X.sequence_to_text(ids, skip_eos_and_pad=True, combine_jamo=True) == s
    assert X.text_to_sequence(s, as_token=True) == s + "~"
    # symbols the table does not have are dropped; so are literal PAD / EOS characters and {ARPAbet} spans
    assert list(X.text_to_sequence("a가_~{HH AW1}가")) == [2, 2 + 19, 2, 2 + 19, EOS_ID]
    assert list(X.text_to_sequence("")) == [EOS_ID]
    up = X.text_to_sequence("x", normalizer=lambda t: "가" * len(t))
    assert list(up) == [2, 21, EOS_ID]


def test_every_syllable_round_trips():
    syl = "".join(chr(c) for c in range(0xAC00, 0xD7A4, 37))
    assert X.jamo_to_korean(X.hangul_to_jamo(syl)) == syl


def test_prepare_batch_contract():
    rs = np.random.RandomState(0)
    r = 4
    batch = []
    for n_tok, n_frames in ((5, 9), (8, 12), (3, 7)):
        batch.append((np.arange(2, 2 + n_tok, dtype=np.int32), 1.0 + n_tok, rs.rand(n_frames, 80), rs.rand(n_frames, 1025)))
    inputs, lens, coeff, mel, lin = X.prepare_batch(batch, r)
    assert inputs.shape == (3, 8) and inputs.dtype == np.int32 and list(lens) == [5, 8, 3]
    assert (inputs[0, 5:] == 0).all() and (inputs[2, 3:] == 0).all()
    assert mel.shape == (3, 16, 80) and lin.shape == (3, 16, 1025)        # round_up(12 + 1, 4) = 16: at least one padded frame
    assert (mel[0, 9:] == 0).all() and np.array_equal(mel[1, :12], batch[1][2].astype(np.float32))
    assert list(coeff) == [6.0, 9.0, 4.0]
    with_spk = [b + (i, len(b[3])) for i, b in enumerate(batch)]
    out = X.prepare_batch(with_spk, r, rng=np.random.RandomState(1), data_type="train")
    assert len(out) == 6 and sorted(out[5]) == [0, 1, 2] and out[3].shape == (3, 16, 80)
    assert X._round_up(12, 4) == 12 and X._round_up(13, 4) == 16
