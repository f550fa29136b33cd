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


def test_collate_contract():
    """datasets/datafeeder.py:289-328: zero padding, targets padded to a multiple of r strictly beyond the longest, lengths
    include the EOS, optional speaker ids."""
    from taco_amd import feeder as F
    rs = np.random.RandomState(0)
    r = 4
    ex = []
    for n_tok, n_frames in ((5, 9), (8, 12), (3, 7)):
        ex.append(F.Example(np.arange(2, 2 + n_tok, dtype=np.int32), 1.0 + n_tok, rs.rand(n_frames, 80), rs.rand(n_frames, 1025)))
    b = F.collate(ex, r)
    assert b.inputs.shape == (3, 8) and b.inputs.dtype == np.int32 and list(b.input_lengths) == [5, 8, 3]
    assert (b.inputs[0, 5:] == 0).all() and (b.inputs[2, 3:] == 0).all()
    assert b.mel_targets.shape == (3, 16, 80) and b.linear_targets.shape == (3, 16, 1025)     # 12 frames -> 16: at least one padded frame
    assert (b.mel_targets[0, 9:] == 0).all() and np.array_equal(b.mel_targets[1, :12], ex[1].mel.astype(np.float32))
    assert list(b.loss_coeff) == [6.0, 9.0, 4.0] and b.speaker_id is None
    assert [F.padded_length(n, 4) for n in (11, 12, 13, 15, 16)] == [12, 16, 16, 16, 20]
    with_spk = [e._replace(speaker_id=i) for i, e in enumerate(ex)]
    assert list(F.collate(with_spk, r).speaker_id) == [0, 1, 2]
    assert list(X.pad_token_rows([[2, 3], [4, 5, 6, 7]])[0]) == [2, 3, 0, 0]


def test_length_bucketing_of_a_group():
    """datasets/datafeeder.py:210-243: a group of batch_size x batches_per_group examples is sorted by target length, cut into
    consecutive batches and the batches are shuffled: every batch spans a narrow band of lengths and nothing is lost."""
    from taco_amd import feeder as F
    rs = np.random.RandomState(1)
    lens = rs.randint(20, 400, size=64)
    ex = [F.Example(np.full(3 + i % 5, 2, np.int32), 1.0, np.zeros((n, 8), np.float32) + i, np.zeros((n, 4), np.float32)) for i, n in enumerate(lens)]
    batches = F.bucket(ex, 8, np.random.RandomState(2))
    assert len(batches) == 8 and all(len(b) == 8 for b in batches)
    seen = sorted(int(e.mel[0, 0]) for b in batches for e in b)
    assert seen == list(range(64))
    spans = sorted((min(len(e.mel) for e in b), max(len(e.mel) for e in b)) for b in batches)
    for (lo0, hi0), (lo1, hi1) in zip(spans, spans[1:]):
        assert hi0 <= lo1                                       # the bands do not overlap: this is the sort + cut
    assert [s for s in spans] != [(min(len(e.mel) for e in b), max(len(e.mel) for e in b)) for b in batches]   # and the batch order is shuffled
    # the iterator: draws groups from two sources by ratio, collates with the reduction factor
    it = iter(lens)
    draw_a = lambda: F.Example(np.array([2, 3, 1], np.int32), 1.0, np.zeros((next(it), 8), np.float32), np.zeros((1, 4), np.float32), 0)
    draw_b = lambda: F.Example(np.array([4, 1], np.int32), 0.5, np.zeros((33, 8), np.float32), np.zeros((33, 4), np.float32), 1)
    fd = F.GroupFeeder({"a": draw_a, "b": draw_b}, batch_size=4, reduction_factor=5, batches_per_group=2, ratios={"a": 0.75, "b": 0.25})
    b0 = next(fd)
    assert b0.inputs.shape[0] == 4 and b0.mel_targets.shape[1] % 5 == 0 and fd.step == 1
    b1 = next(fd)
    assert sorted(list(b0.speaker_id) + list(b1.speaker_id)) == [0, 0, 0, 0, 0, 0, 1, 1]


def test_korean_number_reading_known_answers():
    """Hand-written readings (behaviour of text/korean.py:166-306: Sino-Korean numerals, native numerals before counters, the
    silent leading one, decimals after 쩜, signs)."""
    from taco_amd import korean as K
    assert K.read_integer("7") == "칠" and K.read_integer("10") == "십" and K.read_integer("12") == "십이"
    assert K.read_integer("100") == "백" and K.read_integer("10000") == "만" and K.read_integer("305") == "삼백오"
    assert K.read_integer("2017") == "이천일십칠"                       # the reference keeps 일 in front of an inner 십
    assert K.read_integer("1234567") == "백이십삼만사천오백육십칠"
    assert K.read_integer("5", native=True) == "다섯" and K.read_integer("12", native=True) == "열두"
    assert K.read_integer("19", native=True) == "열아홉" and K.read_integer("20", native=True) == "스물"
    assert K.read_integer("24", native=True) == "스물네" and K.read_integer("55", native=True) == "쉰다섯"
    assert K.read_integer("99", native=True) == "아흔아홉" and K.read_integer("101", native=True) == "백한"
    assert K.read_number("-12.35") == "마이너스 십이쩜 삼오" and K.read_number("+3") == "플러스 삼"
    assert K.read_number("0") == "영" and K.read_number("0", "개") == "영" and K.read_number("1,000") == "천"
    assert K.read_number("24", "살") == "스물네살" and K.read_number("12", "시") == "열두시"
    with pytest.raises(K.NumberFormatError):
        K.read_number("1.2.3")


def test_korean_normaliser_sentences():
    """The sentences the reference's own __main__ exercises (text/korean.py:308-319), with readings worked out by hand."""
    from taco_amd import korean as K
    n = K.KoreanNormalizer(english={"track": "트랙"}, phrases={"1+1": "원플러스원"})
    assert n("JTBC는 JTBCs를 DY는 A가 Absolute") == "제이티비씨는 JTBCs를 디와이는 에이가 Absolute"
    assert n("오늘(13일) 101마리 강아지가") == "오늘 백한마리 강아지가"
    assert n('"저돌"(猪突) 입니다.') == "'저돌' 입니다."
    assert n("지금은 -12.35%였고 종류는 5가지와 19가지, 그리고 55가지였다") == \
        "지금은 마이너스 십이쩜 삼오퍼센트였고 종류는 다섯가지와 열아홉가지, 그리고 쉰다섯가지였다"
    assert n("JTBC는 TH와 K 양이 2017년 9월 12일 오후 12시에 24살이 된다") == \
        "제이티비씨는 티에이치와 케이 양이 이천일십칠년 구월 십이일 오후 열두시에 스물네살이 된다"
    assert n("“첫 문장이다. 둘째 문장이다!”") == "'첫 문장이다.' '둘째 문장이다!'"
    assert n("  1+1 행사의 track 5km, 100m  ") == "원플러스원 행사의 트랙 오킬로미터, 백미터"
    # through the tokeniser: ids of the 80-symbol table, EOS last
    ids = X.text_to_sequence("24살", normalizer=n)
    assert X.sequence_to_text(ids, skip_eos_and_pad=True, combine_jamo=True) == "스물네살"
    assert K.tokenize("A", as_id=True)[-1] == EOS_ID and K.tokenize("12시")[-1] == "~"
    import json, tempfile, os
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "dict.json")
        json.dump({"english": {"idol": "아이돌"}, "phrases": {}}, open(p, "w", encoding="utf-8"))
        assert K.load_dictionaries(p)("idol LG") == "아이돌 엘지"


def test_braced_spans_contribute_nothing_and_pieces_are_cleaned_separately():
    """text/__init__.py:40-58: `{...}` is ARPAbet, which the Korean symbol table does not hold; unmatched or empty braces are ordinary
    (unknown) characters and disappear with them."""
    from taco_amd import text as T
    plain = T.text_to_sequence("안녕하세요")
    assert np.array_equal(T.text_to_sequence("안녕{HH AW1 S}하세요"), plain)
    assert np.array_equal(T.text_to_sequence("안녕{}하세요"), plain) and np.array_equal(T.text_to_sequence("안녕{하세요"), plain)
    assert T.text_to_sequence("").tolist() == [T._symbol_to_id[T.EOS]]
    seen = []
    T.text_to_sequence("가{X}나{Y}다", normalizer=lambda s: (seen.append(s), s)[1])
    assert seen == ["가", "나", "다"]


def test_hangul_decomposition_is_the_unicode_one_for_every_syllable():
    """The `jamo` package the reference imports (text/korean.py:5, absent here) decomposes precomposed syllables by the Unicode
    algorithm; Python's own unicodedata implements the same algorithm as canonical decomposition (NFD) / composition (NFC): all
    11 172 syllables, both directions, against text.py's arithmetic."""
    import unicodedata
    from taco_amd import text as T
    syll = "".join(chr(c) for c in range(0xAC00, 0xD7A4))
    assert len(syll) == 11172
    assert T.hangul_to_jamo(syll) == unicodedata.normalize("NFD", syll)
    assert T.jamo_to_korean(unicodedata.normalize("NFD", syll)) == syll == unicodedata.normalize("NFC", T.hangul_to_jamo(syll))
    mixed = "가나 ABC 12, 힣!"
    assert T.hangul_to_jamo(mixed) == unicodedata.normalize("NFD", mixed)
    # every jamo a syllable can produce has a symbol: 19 + 21 + 27 of the 80
    assert set(T.hangul_to_jamo(syll)) == set(T.JAMO_LEADS + T.JAMO_VOWELS + T.JAMO_TAILS)
