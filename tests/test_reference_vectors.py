"""Replay of the vectors the REFERENCE produced (tools/make_reference_vectors.py, run in the build container against
/root/reference/text/korean.py and datasets/datafeeder.py loaded by path) through korean.py / feeder.py: bit-exact.

These are the two parts of SURVEY 8f rank 4 that can be pinned on the reference itself (they need no TensorFlow); everything that
touches tf.* stays pinned on the oracle only (oracle/taco_oracle.py header: parity unpinned)."""
import json
import os

import numpy as np
import pytest

from taco_amd import korean as K
from taco_amd import feeder as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def kv():
    with open(os.path.join(GOLD, "korean_vectors.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def norm(kv):
    return K.KoreanNormalizer(english=kv["english"], phrases=kv["phrases"])


def test_normalize_equals_the_reference_on_2500_sentences(kv, norm):
    assert len(kv["sentences"]) == len(kv["normalize"]) >= 2500
    bad = [(s, e, norm.normalize(s)) for s, e in zip(kv["sentences"], kv["normalize"]) if norm.normalize(s) != e]
    assert not bad, bad[:3]


def test_every_stage_equals_the_reference(kv, norm):
    """text/korean.py:151-164 stage by stage, each stage fed the previous stage's REFERENCE output (English words and upper-case
    spelling are two passes there and one here: compared after the second)."""
    raw, st = kv["stage_inputs"], kv["stages"]
    assert [norm.apply_phrases(t.strip()) for t in raw] == st["dictionary"]
    assert [norm.apply_latin(t) for t in st["dictionary"]] == st["upper"]
    assert [norm.apply_latin(t) for t in st["english"]] == st["upper"]
    assert [norm.apply_numbers(t) for t in st["upper"]] == st["number"]


def test_number_readings_equal_the_reference(kv, norm):
    assert [norm.apply_numbers(s) for s in kv["number_sweep"]] == kv["number_sweep_expected"]
    assert [norm.apply_numbers(s) for s in kv["counted_sweep"]] == kv["counted_sweep_expected"]
    # a few of them by hand, so that the file cannot drift unnoticed
    table = dict(zip(kv["number_sweep"] + kv["counted_sweep"], kv["number_sweep_expected"] + kv["counted_sweep_expected"]))
    assert table["2017"] == "이천일십칠" and table["10000"] == "만" and table["-12.35"] == "마이너스 십이쩜 삼오" and table["0"] == "영"


def test_inputs_where_the_reference_does_not_give_a_reading(kv, norm):
    """Recorded, not 'fixed': on these inputs the reference raises (text/korean.py:247 ast.literal_eval on a leading zero; :270
    int('+')) or misreads (digit places taken from the un-stripped string).  The product returns a reading instead; the divergence
    is part of the committed vectors so that it cannot change silently in either direction."""
    div = {d["input"]: d for d in kv["divergences"]}
    for s, d in div.items():
        try:
            got = {"returns": norm.normalize(s)}
        except Exception as e:       # noqa: BLE001
            got = {"raises": type(e).__name__}
        assert got == d["product"], (s, got, d)
    assert div["+5"]["reference"] == {"raises": "ValueError"} and div["+5"]["product"] == {"returns": "플러스 오"}
    assert div["007"]["reference"] == {"raises": "SyntaxError"} and div["007"]["product"] == {"returns": "칠"}
    assert div["0012.5"]["reference"] == {"returns": "쩜 오"} and div["0012.5"]["product"] == {"returns": "십이쩜 오"}
    assert sum(not d["same"] for d in kv["divergences"]) == 5


# ---- datasets/datafeeder.py ----
@pytest.fixture(scope="module")
def fv():
    return np.load(os.path.join(GOLD, "feeder_vectors.npz"))


def test_round_up_table(fv):
    for mult in (1, 2, 3, 4, 5, 6):
        want = fv["round_up_m%d" % mult]
        # _prepare_targets pads to _round_up(longest + 1, r)  (datafeeder.py:313-316, :326-328)
        got = [F.padded_length(int(x) - 1, mult) for x in fv["round_up_x"]]
        assert got == want.tolist()


def _examples(fv, prefix, n, spk):
    out = []
    for j in range(n):
        k = "%s_in%d_" % (prefix, j)
        out.append(F.Example(fv[k + "tokens"], float(fv[k + "coeff"]), fv[k + "mel"], fv[k + "linear"], int(fv[k + "speaker"]) if spk else None))
    return out


def _same(batch, fv, prefix, spk):
    names = ["inputs", "input_lengths", "loss_coeff", "mel_targets", "linear_targets"] + (["speaker_id"] if spk else [])
    for name in names:
        want, got = fv[prefix + name], getattr(batch, name)
        assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), (prefix, name)


def test_prepare_batch_cases(fv):
    """_prepare_batch (datafeeder.py:289-306) on six seeded batches: reduction factors 1-5, 5- and 6-tuples, and the row shuffle of
    a training batch with the feeder's generator."""
    for ci, (r, dt, spk, nb, seed) in enumerate(fv["cases"].tolist()):
        ex = _examples(fv, "case%d" % ci, nb, spk)
        if dt == 1:                               # 'train': rng.shuffle(batch) first
            np.random.RandomState(seed).shuffle(ex)
        _same(F.collate(ex, r), fv, "case%d_out_" % ci, spk)


def test_group_logic_equals_enqueue_next_group(fv):
    """DataFeeder._enqueue_next_group (datafeeder.py:210-243) was driven twice per configuration on recorded example streams; the
    product's GroupFeeder must hand out the same batches in the same order from the same streams and generator seed."""
    for gi, row in enumerate(fv["groups"].tolist()):
        bs, bpg, r, ndirs, step, phase, seed = row[:7]
        ratios = [x / 1000.0 for x in row[7:7 + ndirs]]             # the CONFIGURED data_ratio; which phase applies is the feeder's business
        greedy = bool(row[9])
        dirs = [str(d) for d in fv["group%d_dirs" % gi]]
        streams = {d: iter(_examples(fv, "group%d_%s" % (gi, d), int(fv["group%d_%s_count" % (gi, d)]), ndirs > 1)) for d in dirs}
        feeder = F.GroupFeeder({d: (lambda d=d: next(streams[d])) for d in dirs}, bs, r, batches_per_group=bpg,
                               ratios=dict(zip(dirs, ratios)), seed=seed, training=True, initial_phase_step=phase,
                               initial_data_greedy=greedy, step=step)
        for bi in range(int(fv["group%d_nbatches" % gi])):
            _same(next(feeder), fv, "group%d_batch%d_" % (gi, bi), ndirs > 1)
        for d in dirs:                            # and it drew exactly the examples the reference drew
            assert next(streams[d], None) is None


# ---- audio/__init__.py:118-165 (the NumPy / SciPy part of the spectrogram -> waveform step) ----
@pytest.fixture(scope="module")
def av():
    return np.load(os.path.join(GOLD, "audio_vectors.npz"))


def test_audio_hparams_equal_the_reference_effective_values(av):
    """hparams.py:16-23,27-29,144-145 after the reference's own override chain (its hparams.py was executed to write the vector)."""
    from taco_amd import hparams as product
    ref = dict(zip(av["hparams_keys"].tolist(), av["hparams_values"].tolist()))
    assert ref["sample_rate"] == 24000.0                      # the override of hparams.py:28, not the 20000 of :18
    for k, v in ref.items():
        assert float(getattr(product, k)) == v, k


def test_audio_oracle_pieces_equal_the_reference_bit_for_bit(av):
    """oracle/audio_oracle.py restates inv_spectrogram (audio/__init__.py:54-56); its librosa-free pieces are pinned here on what the
    reference's own functions returned: _stft_parameters, _denormalize, _db_to_amp, the `S ** power` handed to Griffin-Lim, and
    inv_preemphasis (scipy.signal.lfilter).  The STFT / ISTFT stay restatements of librosa's documented algorithm (unpinned)."""
    import audio_oracle as AO
    ref = dict(zip(av["hparams_keys"].tolist(), av["hparams_values"].tolist()))
    hp = AO.AudioHParams(num_freq=int(ref["num_freq"]), sample_rate=int(ref["sample_rate"]), frame_length_ms=ref["frame_length_ms"],
                         frame_shift_ms=ref["frame_shift_ms"], preemphasis=ref["preemphasis"], min_level_db=ref["min_level_db"],
                         ref_level_db=ref["ref_level_db"], power=ref["power"], griffin_lim_iters=int(ref["griffin_lim_iters"]))
    assert list(hp.stft_parameters()) == av["stft_parameters"].tolist() == [2048, 300, 1200]
    spec = av["spec"]
    assert np.array_equal(AO.denormalize(spec, hp), av["denormalize"])
    S = AO.db_to_amp(AO.denormalize(spec, hp) + hp.ref_level_db)
    assert np.array_equal(S, av["db_to_amp"])
    assert np.array_equal(S ** hp.power, av["griffin_lim_input"])
    # the oracle's inverse pre-emphasis is the recurrence written out; scipy's lfilter is the same recurrence in another order of operations
    got = AO.inv_preemphasis(av["wave"], hp)
    assert np.abs(got - av["inv_preemphasis"]).max() < 1e-13
    # and the forward direction round-trips through the reference's own pair
    assert np.abs(AO.inv_preemphasis(av["preemphasis"], hp) - av["wave"]).max() < 1e-12


# ---- synthesizer.py:242-262 (attention_trim and end_of_sentence) ----
def test_attention_trim_walk_equals_what_the_reference_kept():
    """The reference's own plot_graph_and_save_audio ran on 120 alignments (tools/make_reference_vectors.py: its walk executed as it
    stands, the array it then hands to inv_spectrogram was recorded).  The oracle's restatement of the walk -- the checker of the device
    kernel k_attention_trim in tests/test_gpu_ops.py -- returns the same number of frames for every one of them."""
    import taco_oracle as O
    tv = np.load(os.path.join(GOLD, "trim_vectors.npz"))
    r = int(tv["reduction_factor"][0])
    assert r == 4 and len(tv["spec_end_idx"]) == 120
    for al, (T_in, n), L, want in zip(tv["alignments"], tv["dims"], tv["sequence_len"], tv["spec_end_idx"]):
        assert O.attention_trim_end(al[:T_in, :n], int(L), r) == int(want)


# ---- synthesizer.py:171-200 (manual_attention_mode 1 and 3: the alignments of the second pass) ----
def test_manual_alignments_equal_what_the_reference_feeds_its_second_pass():
    """The reference's own Synthesizer.synthesize ran on a recording session (tools/make_reference_vectors.py); what it fed as
    `manual_alignments` to the second pass, for 24 first passes and both working modes, is what manual_alignments_of returns -- including
    the reference's choice of axis: one hot per ENCODER position at its most-attending decoder step (`alignments[idx].argmax(1)`), so a
    decoder step may carry no one at all."""
    from taco_amd.synthesizer import manual_alignments_of
    mv = np.load(os.path.join(GOLD, "manual_vectors.npz"))
    some_empty_step = False
    for fp, m1, m3, (N, E, D) in zip(mv["first_pass"], mv["mode1"], mv["mode3"], mv["dims"]):
        al = fp[:N, :E, :D]
        got1, got3 = manual_alignments_of(al, 1), manual_alignments_of(al, 3)
        assert got1.shape == (N, D, E) and np.array_equal(got1, m1[:N, :D, :E]) and np.array_equal(got3, m3[:N, :D, :E])
        assert np.all(got1.sum(1) == 1)                       # exactly one decoder step per encoder position ...
        some_empty_step = some_empty_step or bool(np.any(got1.sum(2) == 0))
    assert some_empty_step                                    # ... and decoder steps with no position at all do occur
    with pytest.raises(Exception):
        manual_alignments_of(mv["first_pass"][0][:1, :2, :2], 2)     # np.pow: the reference raises there too


# ---- hparams.py (effective defaults) and utils/__init__.py:110-126 (load_hparams) ----
def test_every_product_hparam_equals_the_reference_effective_default_and_load_hparams_agrees(tmp_path):
    """The reference's hparams.py was executed as it stands (override chain :27-29,83-94 included) and its load_hparams applied to four
    params.json files (unknown keys, a skip list, list-valued keys, an empty file): the product's defaults and its load_hparams give the
    same values for every key the product carries."""
    import taco_amd
    with open(os.path.join(GOLD, "hparams_vectors.json")) as f:
        hv = json.load(f)
    prod = taco_amd.hparams.values()
    assert len(prod) >= 40 and all(k in hv["effective"] for k in prod)
    assert {k: hv["effective"][k] for k in prod} == prod
    assert hv["effective"]["reduction_factor"] == 4 and hv["effective"]["post_rnn_size"] == 256 and hv["effective"]["sample_rate"] == 24000
    for i, case in enumerate(hv["load_cases"]):
        d = tmp_path / ("case%d" % i)
        d.mkdir()
        (d / "params.json").write_text(json.dumps(case["params_json"]))
        hp = taco_amd.load_hparams(taco_amd.hparams.copy(), str(d), skip_list=case["skip_list"])
        got = hp.values()
        assert {k: case["after"][k] for k in got} == got, i


def test_symbol_table_equals_the_reference(kv):
    """text/korean.py:11-21 (= text/symbols.py; id = position, text/__init__.py:11-12): the ids a trained checkpoint's embedding rows mean."""
    from taco_amd import text as T
    assert T.symbols == kv["all_symbols"] and len(T.symbols) == 80
    assert T.PAD == kv["pad"] and T.EOS == kv["eos"] and T.symbols.index(T.PAD) == 0 and T.symbols.index(T.EOS) == 1


def test_input_lengths_rule_equals_what_the_reference_feeds():
    """synthesizer.py:120 read from the first feed of the reference's own synthesize(): the position of the FIRST EOS, 0 when there is none."""
    from taco_amd.hparams import EOS_ID
    mv = np.load(os.path.join(GOLD, "manual_vectors.npz"))
    tok = mv["tokens"]
    assert EOS_ID == 1 and np.array_equal(np.argmax(tok == EOS_ID, 1), mv["input_lengths"]) and mv["input_lengths"].tolist() == [2, 5, 0, 1, 0, 3]
    assert mv["manual_alignments_fed_when_off"].shape == (1, 1, 1)            # the dummy the placeholder gets when is_manual_attention is False (:128-132)


def test_most_recent_checkpoint_choice_equals_the_reference(tmp_path):
    """synthesizer.py:289-299 run on four directory listings (numeric, not lexicographic, maximum; an explicit step): the product picks the
    same checkpoint (it returns the `.index` file of it, which tf_checkpoint.py reads, where the reference returns the prefix)."""
    from taco_amd.synthesizer import get_most_recent_checkpoint
    with open(os.path.join(GOLD, "hparams_vectors.json")) as f:
        cases = json.load(f)["checkpoint_choice"]
    assert len(cases) == 4
    for i, c in enumerate(cases):
        d = tmp_path / ("d%d" % i)
        d.mkdir()
        for fn in c["files"]:
            (d / fn).write_bytes(b"")
        got = get_most_recent_checkpoint(str(d), checkpoint_step=c["checkpoint_step"])
        assert os.path.basename(got) == c["chosen"] + ".index", (c, got)


def test_npz_source_draws_the_files_the_reference_feeder_draws(tmp_path, fv):
    """DataFeeder._get_next_example (datafeeder.py:245-287) was driven over real .npz files (cursor starting at the third path, wrap and
    reshuffle with the feeder's generator, missing paths, the inline frame / token filter under skip_path_filter, the loss_coeff default):
    NpzSource over the same files, the same path list and the same generator seed returns the same examples in the same order."""
    nfiles = int(fv["npz_nfiles"])
    lo, hi, mt = (int(x) for x in fv["npz_limits"])
    for ci, row in enumerate(fv["npz_cases"]):
        dtype, skip, seed, ndraw = int(row[0]), bool(row[1]), int(row[2]), int(row[3])
        missing = [int(x) for x in row[4:] if x >= 0]
        d = tmp_path / ("case%d" % ci)
        d.mkdir()
        paths = []
        for i in range(nfiles):
            p = str(d / ("ex%d.npz" % i))
            if i not in missing:
                content = {k: fv["npz_file%d_%s" % (i, k)] for k in ("tokens", "mel", "linear")}
                if "npz_file%d_loss_coeff" % i in fv:
                    content["loss_coeff"] = fv["npz_file%d_loss_coeff" % i]
                np.savez(p, **content)
            paths.append(p)
        src = F.NpzSource(paths, speaker_id=3, rng=np.random.RandomState(seed), training=dtype == 1, skip_path_filter=skip,
                          min_n_frame=lo, max_n_frame=hi, min_tokens=mt)
        want = fv["npz_case%d_draws" % ci]
        assert len(want) == ndraw
        for k in range(ndraw):
            ex = src()
            got = [int(ex.tokens[0]) - 100, int(round(float(ex.loss_coeff) * 1000)), len(ex.linear), len(ex.mel)]
            assert got == [int(x) for x in want[k]], (ci, k, got, want[k])
            assert ex.speaker_id == 3
        assert not src.skipped
    # the bookkeeping around it (get_path_dict :41-48,66-71; __init__ :104-121)
    assert F.frame_limits(4, 30, 200) == (120, 796)
    items = [("a", 119, 60), ("b", 120, 50), ("c", 796, 49), ("d", 797, 80), ("e", 500, 50)]
    assert F.filter_items(items, 120, 796, 50) == ["b", "e"]
    assert F.split_paths(list("abcdef"), "train", 2) == list("abcd") and F.split_paths(list("abcdef"), "test", 2) == list("ef")
    with pytest.raises(Exception, match="Unkown data_type"):
        F.split_paths([], "valid", 1)
    assert F.data_ratios(["x/son", "x/park"]) == {"x/son": 0.5, "x/park": 0.5}
    r = F.data_ratios(["x/son", "x/park", "x/moon"], main_data=["son"], main_data_greedy_factor=2)
    assert r == {"x/son": 0.6, "x/park": 0.2, "x/moon": 0.2}
