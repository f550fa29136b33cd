#!/usr/bin/env python
"""Golden vectors produced BY THE REFERENCE for the parts of SURVEY 8f that can run without TensorFlow / librosa:

  * text/korean.py:151-306   normalize() and its stages (dictionary phrases, English words, upper-case spelling, units + numbers)
  * datasets/datafeeder.py:210-243,289-328   _round_up / _prepare_inputs / _prepare_targets / _prepare_batch and the group logic of
    DataFeeder._enqueue_next_group (sort by target length, cut into batches, shuffle the batches, shuffle the rows of a training batch)
  * audio/__init__.py:118-165 + hparams.py   the spectrogram -> waveform step around Griffin-Lim that is plain NumPy / SciPy: _stft_parameters,
    _denormalize, _db_to_amp, the `S ** power` the phase reconstruction starts from (inv_spectrogram, :54-56), inv_preemphasis /
    _preemphasis (scipy.signal.lfilter), _amp_to_db, _normalize -- with the reference's OWN effective hparams (its hparams.py is executed;
    the stand-in for tf.contrib.training.HParams only stores the values it is given).  _stft / _istft / the mel basis are librosa calls
    and stay unpinned (oracle/audio_oracle.py restates them).
  * synthesizer.py:242-262   the `attention_trim and end_of_sentence` walk of plot_graph_and_save_audio, observed through the array it hands
    to inv_spectrogram (a recording stand-in).
  * synthesizer.py:171-200   the manual alignments Synthesizer.synthesize builds for its second pass (manual_attention_mode 1 and 3), observed
    through the feed of that pass (a recording stand-in for the session).
  * hparams.py (every effective default after its override chain) and utils/__init__.py:110-126 load_hparams.

Run in the BUILD container only (it reads /root/reference; the GPU box has no reference):

    python tools/make_reference_vectors.py            # writes tests/golden/korean_vectors.json, feeder_vectors.npz, audio_vectors.npz, trim_vectors.npz, manual_vectors.npz, hparams_vectors.json

The reference modules are loaded BY PATH from where they lie; nothing of their source is copied.  Their import lines name packages
this image lacks (`jamo`, `tensorflow`, `nltk`, the reference's own `audio` / `utils` / `text` packages, which pull in TensorFlow and
librosa).  Those names are satisfied by EMPTY stand-in modules whose functions raise if called (`log` is a no-op): no stand-in takes
part in computing a vector -- any stage that would need one (jamo decomposition, nltk sentence splitting inside quotations) is left
out, and the vectors say so.  tests/test_reference_vectors.py replays the files bit-exactly through korean.py / feeder.py."""
import collections
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("TACO_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")


def _refuse(name):
    def fn(*a, **k):
        raise RuntimeError("stand-in '%s' was called: it must not take part in computing a vector" % name)
    return fn


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference_korean():
    _stub("jamo", hangul_to_jamo=_refuse("jamo.hangul_to_jamo"), h2j=_refuse("jamo.h2j"), j2h=_refuse("jamo.j2h"))
    pkg = _stub("reftext")
    pkg.__path__ = [os.path.join(REF, "text")]          # a namespace for relative imports; text/__init__.py is NOT executed
    return importlib.import_module("reftext.korean")


def load_reference_datafeeder():
    _stub("tensorflow")
    _stub("text")
    u = _stub("utils", parallel_run=_refuse("utils.parallel_run"), remove_file=_refuse("utils.remove_file"))
    u.__path__ = []
    _stub("utils.infolog", log=lambda *a, **k: None)
    a = _stub("audio", frames_to_hours=_refuse("audio.frames_to_hours"))
    a.__path__ = []
    _stub("audio.get_duration", get_durations=_refuse("audio.get_durations"))
    pkg = _stub("refdatasets")
    pkg.__path__ = [os.path.join(REF, "datasets")]
    return importlib.import_module("refdatasets.datafeeder")


def load_reference_audio():
    """audio/__init__.py by path, with the reference's own hparams.py executed for the values.  Stand-ins: `tensorflow` (only
    tf.contrib.training.HParams, a value holder), `librosa` / `librosa.filters` (refuse to be called)."""
    class _HP(object):                                   # tf.contrib.training.HParams as far as hparams.py uses it: holds values
        def __init__(self, **kw):
            self._v = dict(kw)
            self.__dict__.update(kw)

        def values(self):
            return dict(self._v)
    tf = _stub("tensorflow")
    tf.contrib = types.SimpleNamespace(training=types.SimpleNamespace(HParams=_HP))
    lib = _stub("librosa", stft=_refuse("librosa.stft"), istft=_refuse("librosa.istft"))
    lib.__path__ = []
    lib.core = types.SimpleNamespace(load=_refuse("librosa.core.load"))
    _stub("librosa.filters", mel=_refuse("librosa.filters.mel"))
    sys.modules.pop("hparams", None)
    spec = importlib.util.spec_from_file_location("hparams", os.path.join(REF, "hparams.py"))
    hp = importlib.util.module_from_spec(spec); sys.modules["hparams"] = hp; spec.loader.exec_module(hp)
    sys.modules.pop("audio", None); sys.modules.pop("audio.get_duration", None)
    spec = importlib.util.spec_from_file_location("refaudio", os.path.join(REF, "audio", "__init__.py"))
    A = importlib.util.module_from_spec(spec); spec.loader.exec_module(A)
    return A, hp.hparams


def audio_vectors(A, hp):
    rs = np.random.RandomState(20260927)
    out = {}
    keys = ["num_freq", "sample_rate", "frame_length_ms", "frame_shift_ms", "preemphasis", "min_level_db", "ref_level_db", "power",
            "griffin_lim_iters", "num_mels"]
    out["hparams_keys"] = np.array(keys)
    out["hparams_values"] = np.array([float(getattr(hp, k)) for k in keys], np.float64)
    out["stft_parameters"] = np.array(A._stft_parameters(), np.int64)                       # n_fft, hop_length, win_length
    spec = rs.uniform(-0.25, 1.25, size=(hp.num_freq, 9))                                   # outside [0, 1] too: _denormalize clips
    spec[3, :] = [0.0, 1.0, -1.0, 2.0, 0.5, 1e-9, 1 - 1e-9, 0.25, 0.75]
    out["spec"] = spec
    out["denormalize"] = A._denormalize(spec)
    S = A._db_to_amp(A._denormalize(spec) + hp.ref_level_db)
    out["db_to_amp"] = S
    out["griffin_lim_input"] = S ** hp.power                                                # what inv_spectrogram hands to _griffin_lim (:55-56)
    y = rs.randn(777) * 0.1
    out["wave"] = y
    out["inv_preemphasis"] = A.inv_preemphasis(y)
    out["preemphasis"] = A._preemphasis(y)
    mag = np.abs(rs.randn(hp.num_freq, 5)) * 3.0
    mag[0, :] = [0.0, 1e-6, 1e-5, 1.0, 1e3]
    out["mag"] = mag
    out["amp_to_db"] = A._amp_to_db(mag)
    out["normalize"] = A._normalize(A._amp_to_db(mag) - hp.ref_level_db)                    # spectrogram() without the STFT (:48-51)
    out["frames_to_hours"] = np.array([A.frames_to_hours([100, 250, 4000])], np.float64)
    return out


def load_reference_synthesizer(record):
    """synthesizer.py by path, for plot_graph_and_save_audio's `attention_trim and end_of_sentence` walk (:242-262).  Everything the module
    imports is a stand-in that refuses to be called, except the two calls that FOLLOW the walk: audio.inv_spectrogram, here a recorder
    of the array the reference hands it (its second dimension is the number of frames the walk kept), and audio.save_audio, a no-op.
    hparams is the reference's own (reduction_factor)."""
    load_reference_audio()                                   # leaves the reference's real `hparams` module and the tf / librosa stand-ins in sys.modules
    def inv_spectrogram(x):
        record.append(tuple(x.shape))
        return np.zeros(4)
    a = _stub("audio", save_audio=lambda *a_, **k: None, inv_spectrogram=inv_spectrogram, inv_preemphasis=_refuse("audio.inv_preemphasis"),
              inv_spectrogram_tensorflow=_refuse("audio.inv_spectrogram_tensorflow"))
    a.__path__ = []
    _stub("models", create_model=_refuse("models.create_model"), get_most_recent_checkpoint=_refuse("models.get_most_recent_checkpoint"))
    u = _stub("utils", plot=types.SimpleNamespace(plot_alignment=_refuse("plot.plot_alignment")), PARAMS_NAME="params.json",
              load_json=_refuse("utils.load_json"), load_hparams=_refuse("utils.load_hparams"), add_prefix=_refuse("utils.add_prefix"),
              add_postfix=_refuse("utils.add_postfix"), get_time=_refuse("utils.get_time"), parallel_run=_refuse("utils.parallel_run"),
              makedirs=_refuse("utils.makedirs"))
    u.__path__ = []
    t = _stub("text", text_to_sequence=_refuse("text.text_to_sequence"), sequence_to_text=_refuse("text.sequence_to_text"))
    t.__path__ = []
    _stub("text.korean", tokenize=_refuse("text.korean.tokenize"))
    spec = importlib.util.spec_from_file_location("refsynthesizer", os.path.join(REF, "synthesizer.py"))
    S = importlib.util.module_from_spec(spec); spec.loader.exec_module(S)
    return S


def trim_vectors():
    """Alignments [T_in, n] of single utterances (sharp monotone ramps that reach the end early / late / never, plateaus at the last
    position, random ones) -> the number of spectrogram frames the reference keeps (spec_end_idx = r * jdx + 3)."""
    record = []
    S = load_reference_synthesizer(record)
    r = int(sys.modules["hparams"].hparams.reduction_factor)
    rs = np.random.RandomState(77)
    cases = []
    for ci in range(120):
        T_in, n = int(rs.randint(2, 24)), int(rs.randint(1, 40))
        seq_len = int(rs.randint(1, T_in + 1)) if ci % 3 else T_in
        kind = ci % 4
        if kind == 0:                                        # a ramp that reaches the last position and stays
            speed = rs.uniform(0.3, 2.0)
            pos = np.minimum((np.arange(n) * speed).astype(int), T_in - 1)
        elif kind == 1:                                      # reaches the end, then wanders past seq_len - 1 (padding positions)
            pos = np.minimum(np.arange(n), T_in - 1)
            pos[n // 2:] = rs.randint(0, T_in, size=n - n // 2)
        elif kind == 2:                                      # random walk
            pos = np.clip(np.cumsum(rs.randint(-1, 3, size=n)), 0, T_in - 1)
        else:
            pos = rs.randint(0, T_in, size=n)
        al = rs.uniform(0, 0.05, size=(T_in, n))
        al[pos, np.arange(n)] += 1.0
        al /= al.sum(0, keepdims=True)
        wav = np.zeros((n * r + 16, 8))                      # longer than any spec_end_idx: the slice never clamps
        del record[:]
        S.plot_graph_and_save_audio((0, (wav, al, None, "", list(range(seq_len)))), end_of_sentence=True, attention_trim=True)
        assert len(record) == 1 and record[0][0] == 8
        cases.append((al.astype(np.float64), seq_len, record[0][1]))
    Tm, nm = max(c[0].shape[0] for c in cases), max(c[0].shape[1] for c in cases)
    pad = np.zeros((len(cases), Tm, nm))
    for i, (al, _, _) in enumerate(cases):
        pad[i, :al.shape[0], :al.shape[1]] = al
    return {"alignments": pad, "dims": np.array([c[0].shape for c in cases], np.int64), "sequence_len": np.array([c[1] for c in cases], np.int64),
            "spec_end_idx": np.array([c[2] for c in cases], np.int64), "reduction_factor": np.array([r], np.int64)}


def manual_vectors():
    """Synthesizer.synthesize(manual_attention_mode = 1 | 3) of the reference (synthesizer.py:69-205) driven on a plain namespace: `self.model`
    holds placeholder NAMES, `self.sess.run` is a stand-in that returns the seeded first-pass outputs the vector starts from and RECORDS the
    feed of the second pass -- `feed_dict[manual_alignments]` there is what the reference's own code (:171-200) built from them.  utils.get_time
    returns a constant and utils.parallel_run (plot + save of the results, after the fact) is a no-op; neither touches the alignments."""
    record = []
    S = load_reference_synthesizer(record)
    S.get_time = lambda: "t"
    S.parallel_run = lambda fn, items, **k: []
    rs = np.random.RandomState(4242)
    out = {"first_pass": [], "dims": [], "mode1": [], "mode3": []}
    cases = []
    for ci in range(24):
        N, E, D = int(rs.randint(1, 4)), int(rs.randint(2, 14)), int(rs.randint(2, 18))
        al = rs.uniform(0, 1, size=(N, E, D))
        if ci % 2:                                           # a mostly monotone path on top
            for b in range(N):
                pos = np.minimum((np.arange(D) * (E + 1)) // D, E - 1)
                al[b, pos, np.arange(D)] += 1.5
        al = (al / al.sum(1, keepdims=True)).astype(np.float32)       # the model's alignments are float32 [N, T_in, T_dec]
        tokens = np.ones((N, E), np.int64)                   # EOS (= 1) first: input_lengths = argmax(tokens == 1) = 0, unused by modes 1 / 3
        got = {}
        for mode in (1, 3):
            feeds = []
            class Sess(object):
                def run(self, fetches, feed_dict=None):
                    feeds.append(feed_dict)
                    return np.zeros((N, 4 * D, 8), np.float32), al.copy()
            me = types.SimpleNamespace(sess=Sess(), model=types.SimpleNamespace(
                linear_outputs="linear_outputs", alignments="alignments", inputs="inputs", input_lengths="input_lengths",
                manual_alignments="manual_alignments", is_manual_attention="is_manual_attention", speaker_id="speaker_id"))
            S.Synthesizer.synthesize(me, tokens=tokens, manual_attention_mode=mode, attention_trim=False)
            assert len(feeds) == 2 and feeds[1]["is_manual_attention"] is True
            got[mode] = np.array(feeds[1]["manual_alignments"])
            assert got[mode].shape == (N, D, E)
        cases.append((al, got[1], got[3]))
    Nm = max(c[0].shape[0] for c in cases); Em = max(c[0].shape[1] for c in cases); Dm = max(c[0].shape[2] for c in cases)
    fp = np.zeros((len(cases), Nm, Em, Dm), np.float32); m1 = np.zeros((len(cases), Nm, Dm, Em), np.float32); m3 = np.zeros_like(m1)
    for i, (al, a1, a3) in enumerate(cases):
        N, E, D = al.shape
        fp[i, :N, :E, :D] = al; m1[i, :N, :D, :E] = a1; m3[i, :N, :D, :E] = a3
    # synthesizer.py:120: input_lengths = argmax(tokens == 1) -- read from the FIRST feed of a plain call (EOS in the middle, at the end, at the
    # start, twice, absent: then 0)
    tok = np.array([[5, 9, 1, 0, 0, 0], [7, 7, 7, 7, 7, 1], [1, 4, 4, 4, 4, 4], [3, 1, 6, 1, 0, 0], [2, 3, 4, 5, 6, 7], [0, 0, 0, 1, 0, 0]], np.int64)
    feeds = []
    class Sess0(object):
        def run(self, fetches, feed_dict=None):
            feeds.append(feed_dict)
            return np.zeros((len(tok), 8, 8), np.float32), np.full((len(tok), tok.shape[1], 2), 0.5, np.float32)
    me = types.SimpleNamespace(sess=Sess0(), model=types.SimpleNamespace(
        linear_outputs="linear_outputs", alignments="alignments", inputs="inputs", input_lengths="input_lengths",
        manual_alignments="manual_alignments", is_manual_attention="is_manual_attention", speaker_id="speaker_id"))
    S.Synthesizer.synthesize(me, tokens=tok, speaker_ids=[0, 1, 2, 0, 1, 2], attention_trim=False)
    assert len(feeds) == 1 and feeds[0]["is_manual_attention"] is False
    return {"first_pass": fp, "mode1": m1, "mode3": m3, "dims": np.array([c[0].shape for c in cases], np.int64),
            "tokens": tok, "input_lengths": np.asarray(feeds[0]["input_lengths"], np.int64), "speaker_id_fed": np.asarray(feeds[0]["speaker_id"], np.int64),
            "manual_alignments_fed_when_off": np.asarray(feeds[0]["manual_alignments"], np.float64)}


def hparams_vectors():
    """hparams.py executed as it stands (the override chain :27-29,83-94 included) -> every effective default; and utils/__init__.py's
    load_hparams (:110-126; the module needs no stand-in at all) applied to it for a few params.json files -> the values afterwards."""
    import tempfile
    _, hp = load_reference_audio()                           # (re)executes the reference's hparams.py; hp is its `hparams` object
    sys.modules.pop("utils", None); sys.modules.pop("utils.infolog", None)
    spec = importlib.util.spec_from_file_location("refutils", os.path.join(REF, "utils", "__init__.py"))
    U = importlib.util.module_from_spec(spec); spec.loader.exec_module(U)
    plain = lambda v: v if isinstance(v, (int, float, str, bool, list)) or v is None else str(v)
    out = {"effective": {k: plain(v) for k, v in sorted(hp.values().items())}, "load_cases": []}
    cases = [({"reduction_factor": 5, "attention_type": "bah_norm", "not_a_key": 1, "enc_bank_size": 8}, []),
             ({"model_type": "deepvoice", "speaker_embedding_size": 32, "max_iters": 1000, "dropout_prob": 0.5}, ["dropout_prob"]),
             ({"post_proj_sizes": [128, 80], "dec_prenet_sizes": [64, 32], "sample_rate": 22050}, ["sample_rate"]),
             ({}, [])]
    for js, skip in cases:
        _, hp_i = load_reference_audio()                     # a fresh copy of the defaults
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "params.json"), "w") as f:
                json.dump(js, f)
            U.load_hparams(hp_i, d, skip_list=skip)
        out["load_cases"].append({"params_json": js, "skip_list": skip,
                                  "after": {k: plain(getattr(hp_i, k)) for k in sorted(hp.values())}})
    # synthesizer.py:289-299 get_most_recent_checkpoint: which step a directory listing selects (the function only globs and parses names)
    S = load_reference_synthesizer([])
    out["checkpoint_choice"] = []
    listings = [(["model.ckpt-100", "model.ckpt-20", "model.ckpt-3000"], None), (["model.ckpt-7", "model.ckpt-70", "model.ckpt-8"], None),
                (["model.ckpt-5000"], None), (["model.ckpt-100", "model.ckpt-200"], 100)]
    for stems, step in listings:
        with tempfile.TemporaryDirectory() as d:
            files = []
            for st in stems:
                for suffix in (".index", ".meta", ".data-00000-of-00001"):
                    files.append(st + suffix)
                    open(os.path.join(d, st + suffix), "w").close()
            open(os.path.join(d, "checkpoint"), "w").close()
            got = S.get_most_recent_checkpoint(d, checkpoint_step=step)
            out["checkpoint_choice"].append({"files": sorted(files + ["checkpoint"]), "checkpoint_step": step, "chosen": os.path.basename(got)})
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# text/korean.py
# ------------------------------------------------------------------------------------------------------------------------------
HANGUL_WORDS = ["오늘", "서울", "사람", "학교", "날씨", "그리고", "우리는", "약", "모두", "가격은", "에서", "까지", "이상", "정도", "합니다", "였다", "라고"]
MIXED_LATIN = ["Hello", "iPhone", "kTx", "mVp", "Seoul", "abc", "zzz"]     # not all upper-case; checked below not to be dictionary words


def korean_vectors(K):
    rs = np.random.RandomState(20260927)
    eng_all, etc_all = dict(K.english_dictionary), dict(K.etc_dictionary)
    eng_keys = sorted(eng_all)
    eng_sub = [eng_keys[i] for i in sorted(rs.choice(len(eng_keys), size=min(10, len(eng_keys)), replace=False))]
    etc_sub = sorted(etc_all)                      # the phrase table is short: all of it travels with the vectors it is used by
    counters = K.count_checker.strip("()").split("|")
    units = list(K.unit_to_kor1) + list(K.unit_to_kor2)
    mixed = [w for w in MIXED_LATIN if w not in eng_all]
    caps = ["LG", "KBS", "IT", "A", "CEO", "USB", "XYZ", "UN", "MBC", "SKT"]
    caps = [w for w in caps if w not in eng_all]

    def number(lead_zero_ok=False):
        kind = rs.randint(0, 8)
        if kind == 0:
            s = str(rs.randint(0, 10))
        elif kind == 1:
            s = str(rs.randint(10, 100))
        elif kind == 2:
            s = str(rs.randint(100, 10000))
        elif kind == 3:
            s = str(rs.randint(10000, 10 ** 9))
        elif kind == 4:
            s = str(int(rs.randint(1, 10 ** 9)) * int(rs.randint(1, 10 ** 9)))     # up to 10^18: 경 / 해 groups
        elif kind == 5:
            s = "{:,}".format(int(rs.randint(1000, 10 ** 8)))
        elif kind == 6:
            s = str(int(10 ** rs.randint(1, 17)) * int(rs.randint(1, 10)))          # round numbers: silent leading one, empty groups
        else:
            s = str(rs.randint(1, 1000)) + "0" * rs.randint(0, 5)
        return s

    def decorated_number():
        s = number()
        if rs.rand() < 0.25:
            s = s.replace(",", "") + "." + "".join(str(rs.randint(0, 10)) for _ in range(rs.randint(1, 4)))
        if rs.rand() < 0.15:
            s = "-" + s                              # '+' is a reference crash (recorded under `divergences`)
        return s

    def sentence():
        parts = []
        for _ in range(rs.randint(2, 7)):
            k = rs.randint(0, 9)
            if k == 0:
                parts.append(HANGUL_WORDS[rs.randint(len(HANGUL_WORDS))])
            elif k == 1:
                parts.append(decorated_number())
            elif k == 2:
                parts.append(number() + counters[rs.randint(len(counters))])
            elif k == 3:
                parts.append(decorated_number() + units[rs.randint(len(units))])
            elif k == 4:
                parts.append(eng_sub[rs.randint(len(eng_sub))])
            elif k == 5:
                parts.append(caps[rs.randint(len(caps))])
            elif k == 6:
                parts.append(mixed[rs.randint(len(mixed))])
            elif k == 7:
                parts.append(etc_sub[rs.randint(len(etc_sub))])
            else:
                parts.append(HANGUL_WORDS[rs.randint(len(HANGUL_WORDS))] + [",", ".", "!", "?"][rs.randint(4)])
        text = " ".join(parts)
        if rs.rand() < 0.1:
            text = "  " + text + " "
        if rs.rand() < 0.1:
            text += "(%d일)" % rs.randint(1, 32)
        if rs.rand() < 0.1:
            text += "(漢字)"
        return text

    import re
    quote = re.compile(K.quote_checker)
    sentences, expected = [], []
    while len(sentences) < 2500:
        t = sentence()
        if quote.search(t):
            continue                                  # the quotation stage needs nltk: not pinned here
        # the product gets only the dictionary subset: the sentence must not touch any other entry
        latin = re.findall("[A-Za-z]+", t)
        if any((w in eng_all) and (w not in eng_sub) for w in latin):
            continue
        sentences.append(t)
        expected.append(K.normalize(t))

    # stage by stage, chained the way the reference composes them (text/korean.py:151-164): each stage sees the previous one's output
    stage_in = sentences[:400]
    s1 = [K.normalize_with_dictionary(t.strip(), K.etc_dictionary) for t in stage_in]
    s2 = [K.normalize_english(t) for t in s1]
    s3 = [re.sub("[a-zA-Z]+", K.normalize_upper, t) for t in s2]
    s4 = [K.normalize_number(t) for t in s3]
    stages = {"dictionary": s1, "english": s2, "upper": s3, "number": s4}

    # number_to_korean through the two patterns normalize_number applies, on a dense sweep
    sweep = [str(i) for i in list(range(0, 130)) + [200, 1000, 1001, 1100, 2017, 9999, 10000, 10001, 10010, 11000, 100000, 1000000,
                                                     10000000, 100000000, 100010000, 1000000000000, 10000000000000000,
                                                     123456789012345678, 99999999, 20000, 30303, 400040004]]
    sweep += ["1,000", "12,345,678", "3.14", "0.5", "10.05", "-7", "-12.35", "-0.25", "1,234.5", "0", "0.0", "00", "1.", "100."]
    plain = [K.normalize_number(s) for s in sweep]
    counted_in = [str(i) + c for i in list(range(0, 100)) + [100, 101, 110, 120, 199, 200, 999, 1000, 1234, 10000] for c in (counters[0], counters[3], counters[11])]
    counted = [K.normalize_number(s) for s in counted_in]

    # inputs on which the reference does not return a reading; recorded with what it does, next to what the product does
    div_in = ["+5", "+3.5", "007", "0012.5", "012", "1.2.3", "1.5명", "-0명", "00.5"]
    divergences = []
    sys.path.insert(0, ROOT)
    from taco_amd import korean as P
    prod = P.KoreanNormalizer(english={k: eng_all[k] for k in eng_sub}, phrases={k: etc_all[k] for k in etc_sub})
    for s in div_in:
        try:
            ref = {"returns": K.normalize_number(s)}
        except Exception as e:           # noqa: BLE001 -- the exception type is the datum
            ref = {"raises": type(e).__name__}
        try:
            got = {"returns": prod.normalize(s)}
        except Exception as e:           # noqa: BLE001
            got = {"raises": type(e).__name__}
        divergences.append({"input": s, "reference": ref, "product": got, "same": ref == got})

    return {
        "generated_by": "tools/make_reference_vectors.py from /root/reference/text/korean.py (loaded by path; jamo / nltk stand-ins never called)",
        "all_symbols": K.ALL_SYMBOLS, "pad": K.PAD, "eos": K.EOS,      # text/korean.py:11-21 = text/symbols.py: symbol i of the table has id i (text/__init__.py:11-12)
        "reference_functions": ["normalize :151-164", "normalize_with_dictionary :166-171", "normalize_english :173-182",
                                "normalize_upper :184-190", "normalize_number :207-214", "number_to_korean :237-306"],
        "not_covered": ["normalize_quote :192-205 (needs nltk.sent_tokenize)", "tokenize :139-147 (needs the jamo package)"],
        "english": {k: eng_all[k] for k in eng_sub},
        "phrases": {k: etc_all[k] for k in etc_sub},
        "sentences": sentences, "normalize": expected,
        "stage_inputs": stage_in, "stages": stages,
        "number_sweep": sweep, "number_sweep_expected": plain,
        "counted_sweep": counted_in, "counted_sweep_expected": counted,
        "divergences": divergences,
    }


# ------------------------------------------------------------------------------------------------------------------------------
# datasets/datafeeder.py
# ------------------------------------------------------------------------------------------------------------------------------
def make_example(rs, num_mels, num_freq, with_speaker, t_lo=3, t_hi=40):
    n_tok = int(rs.randint(2, 25))
    T = int(rs.randint(t_lo, t_hi))
    tokens = rs.randint(2, 80, size=n_tok).astype(np.int32)
    tokens[-1] = 1
    mel = rs.rand(T, num_mels).astype(np.float32)
    lin = rs.rand(T, num_freq).astype(np.float32)
    coeff = float(rs.choice([1.0, 0.5, 2.0]))
    if with_speaker:
        return (tokens, coeff, mel, lin, int(rs.randint(0, 4)), T)
    return (tokens, coeff, mel, lin, T)


def feeder_vectors(F):
    out = {}
    xs = np.arange(0, 41)
    out["round_up_x"] = xs
    for mult in (1, 2, 3, 4, 5, 6):
        out["round_up_m%d" % mult] = np.array([F._round_up(int(x), mult) for x in xs], np.int64)
    rs = np.random.RandomState(4242)
    cases = []
    for ci, (r, data_type, spk, nb) in enumerate([(4, "test", False, 5), (5, "train", False, 6), (4, "train", True, 7), (1, "test", True, 3),
                                                  (2, "train", False, 1), (3, None, True, 4)]):
        batch = [make_example(rs, 3, 5, spk) for _ in range(nb)]
        for j, ex in enumerate(batch):
            out["case%d_in%d_tokens" % (ci, j)] = ex[0]
            out["case%d_in%d_coeff" % (ci, j)] = np.float32(ex[1])
            out["case%d_in%d_mel" % (ci, j)] = ex[2]
            out["case%d_in%d_linear" % (ci, j)] = ex[3]
            if spk:
                out["case%d_in%d_speaker" % (ci, j)] = np.int32(ex[4])
        rng = np.random.RandomState(100 + ci)
        res = F._prepare_batch(list(batch), r, rng, data_type)       # rng.shuffle(batch) when data_type == 'train' (:290-291)
        for name, arr in zip(["inputs", "input_lengths", "loss_coeff", "mel_targets", "linear_targets", "speaker_id"], res):
            out["case%d_out_%s" % (ci, name)] = np.asarray(arr)
        cases.append((r, str(data_type), int(spk), nb, 100 + ci))
    out["cases"] = np.array([[c[0], {"train": 1, "test": 2, "None": 0}[c[1]], c[2], c[3], c[4]] for c in cases], np.int64)

    # the group logic: DataFeeder._enqueue_next_group (:210-243) driven on a plain namespace -- the method touches only the attributes
    # set here; the "session" records what would have been enqueued
    class Recorder(object):
        def __init__(self):
            self.feeds = []

        def run(self, op, feed_dict=None):
            self.feeds.append(feed_dict)

    groups = []
    for gi, (bs, bpg, r, dirs, data_type, step, phase, greedy) in enumerate([(4, 3, 4, ["a"], "train", 0, 100, False), (3, 4, 5, ["a", "b"], "train", 0, 100, False),
                                                                             (4, 2, 4, ["a", "b"], "train", 500, 100, False),
                                                                             (4, 2, 4, ["a", "b"], "train", 0, 2, False),         # the phase ends between the two groups
                                                                             (2, 3, 4, ["a", "krbook_b"], "train", 0, 100, True)]):   # initial_data_greedy: all from "krbook"
        srs = {d: np.random.RandomState(900 + 10 * gi + k) for k, d in enumerate(dirs)}
        drawn = {d: [] for d in dirs}

        def next_example(data_dir, srs=srs, drawn=drawn, dirs=dirs):
            ex = make_example(srs[data_dir], 2, 3, len(dirs) > 1, 3, 30)
            drawn[data_dir].append(ex)
            return ex
        hp = types.SimpleNamespace(reduction_factor=r, initial_data_greedy=greedy, initial_phase_step=phase)
        ratio = {d: w for d, w in zip(dirs, [0.75, 0.25] if len(dirs) == 2 else [1.0])}
        names = ["inputs", "input_lengths", "loss_coeff", "mel_targets", "linear_targets"] + (["speaker_id"] if len(dirs) > 1 else [])
        me = types.SimpleNamespace(batch_size=bs, _hp=hp, static_batches=None, data_dirs=dirs, _step=step, _batches_per_group=bpg,
                                   data_ratio=ratio, _get_next_example=next_example, rng=np.random.RandomState(321 + gi),
                                   _placeholders=names, _session=Recorder(), _enqueue_op=None, data_type=data_type)
        F.DataFeeder._enqueue_next_group(me)
        F.DataFeeder._enqueue_next_group(me)          # a second group from the same generator state
        for d in dirs:
            for j, ex in enumerate(drawn[d]):
                out["group%d_%s_in%d_tokens" % (gi, d, j)] = ex[0]
                out["group%d_%s_in%d_coeff" % (gi, d, j)] = np.float32(ex[1])
                out["group%d_%s_in%d_mel" % (gi, d, j)] = ex[2]
                out["group%d_%s_in%d_linear" % (gi, d, j)] = ex[3]
                if len(dirs) > 1:
                    out["group%d_%s_in%d_speaker" % (gi, d, j)] = np.int32(ex[4])
            out["group%d_%s_count" % (gi, d)] = np.int64(len(drawn[d]))
        for bi, fd in enumerate(me._session.feeds):
            for name in names:
                out["group%d_batch%d_%s" % (gi, bi, name)] = np.asarray(fd[name])
        out["group%d_nbatches" % gi] = np.int64(len(me._session.feeds))
        groups.append([bs, bpg, r, len(dirs), step, phase, 321 + gi] + [int(round(ratio[d] * 1000)) for d in dirs] + [0] * (2 - len(dirs)) + [int(greedy)])
        out["group%d_dirs" % gi] = np.array(dirs)
    out["groups"] = np.array(groups, np.int64)
    return out


def npz_source_vectors(F):
    """DataFeeder._get_next_example (datafeeder.py:245-287) driven on a plain namespace over real .npz files in a temporary directory:
    which file each draw returns (cursor starting at 2, wrap + reshuffle with the feeder's generator, missing paths skipped, the inline
    filter under skip_path_filter), the loss_coeff default, the tuple layout.  The files' contents are stored so that the test can
    rebuild the directory."""
    import tempfile
    out = {}
    rs = np.random.RandomState(777)
    files = []
    for i in range(9):
        n_tok, T = int(rs.randint(3, 12)), int(rs.randint(4, 30))
        tokens = rs.randint(2, 80, size=n_tok).astype(np.int32)
        tokens[0] = 100 + i                      # marks the file in what a draw returns
        d = {"tokens": tokens, "mel": rs.rand(T, 2).astype(np.float32), "linear": rs.rand(T, 3).astype(np.float32)}
        if i % 3 == 1:
            d["loss_coeff"] = np.float32(0.5)
        files.append(d)
    for i, d in enumerate(files):
        for k, v in d.items():
            out["npz_file%d_%s" % (i, k)] = v
    out["npz_nfiles"] = np.int64(len(files))
    cases = []
    for ci, (data_type, skip_filter, seed, ndraw, missing) in enumerate([("train", False, 5, 25, [4]), ("test", False, 6, 14, []),
                                                                          ("train", True, 7, 30, [0, 7])]):
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for i, d in enumerate(files):
                pth = os.path.join(td, "ex%d.npz" % i)
                if i not in missing:
                    np.savez(pth, **d)
                paths.append(pth)
            me = types.SimpleNamespace(path_dict={"dirA": list(paths)}, _offset=collections.defaultdict(lambda: 2), data_type=data_type,
                                       rng=np.random.RandomState(seed), skip_path_filter=skip_filter, min_n_frame=8, max_n_frame=24, min_tokens=5,
                                       data_dir_to_id={"dirA": 3})
            seq = []
            for _ in range(ndraw):
                tokens, coeff, mel, lin, did, n = F.DataFeeder._get_next_example(me, "dirA")
                assert did == 3 and n == len(lin)
                seq.append([int(tokens[0]) - 100, int(round(float(coeff) * 1000)), int(n), int(mel.shape[0])])
            out["npz_case%d_draws" % ci] = np.array(seq, np.int64)
        cases.append([{"train": 1, "test": 2}[data_type], int(skip_filter), seed, ndraw] + missing + [-1] * (2 - len(missing)))
    out["npz_cases"] = np.array(cases, np.int64)
    out["npz_limits"] = np.array([8, 24, 5], np.int64)
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("no reference checkout at %s (this script runs in the build container only)" % REF)
    os.makedirs(GOLD, exist_ok=True)
    K = load_reference_korean()
    kv = korean_vectors(K)
    with open(os.path.join(GOLD, "korean_vectors.json"), "w", encoding="utf-8") as f:
        json.dump(kv, f, ensure_ascii=False, indent=0, sort_keys=True)
    F = load_reference_datafeeder()
    fv = feeder_vectors(F)
    fv.update(npz_source_vectors(F))
    np.savez_compressed(os.path.join(GOLD, "feeder_vectors.npz"), **fv)
    A, ahp = load_reference_audio()
    av = audio_vectors(A, ahp)
    np.savez_compressed(os.path.join(GOLD, "audio_vectors.npz"), **av)
    print("audio: %d arrays, hparams %s" % (len(av), dict(zip(av["hparams_keys"].tolist(), av["hparams_values"].tolist()))))
    tv = trim_vectors()
    np.savez_compressed(os.path.join(GOLD, "trim_vectors.npz"), **tv)
    print("trim: %d alignments, reduction_factor %d, kept frames %d .. %d" % (len(tv["spec_end_idx"]), tv["reduction_factor"][0], tv["spec_end_idx"].min(), tv["spec_end_idx"].max()))
    mv = manual_vectors()
    np.savez_compressed(os.path.join(GOLD, "manual_vectors.npz"), **mv)
    print("manual attention: %d first passes, modes 1 and 3" % len(mv["dims"]))
    hv = hparams_vectors()
    with open(os.path.join(GOLD, "hparams_vectors.json"), "w") as f:
        json.dump(hv, f, indent=0, sort_keys=True)
    print("hparams: %d effective defaults, %d load_hparams cases" % (len(hv["effective"]), len(hv["load_cases"])))
    print("korean: %d sentences, %d + %d sweep numbers, %d divergences (%d identical); feeder: %d arrays"
          % (len(kv["sentences"]), len(kv["number_sweep"]), len(kv["counted_sweep"]), len(kv["divergences"]),
             sum(d["same"] for d in kv["divergences"]), len(fv)))


if __name__ == "__main__":
    main()
