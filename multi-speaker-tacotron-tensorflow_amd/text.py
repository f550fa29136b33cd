"""Input side of the path: text -> ids and the feeder's batch contract, host code.

* symbols (text/korean.py:11-21, text/symbols.py): PAD '_' (0), EOS '~' (1), 19 lead + 21 vowel + 27 tail Hangul jamo
  (U+1100.., U+1161.., U+11A8..), punctuation !'(),-.:;? and space = 80 symbols -- the `num_symbols` the embedding table has.
* text_to_sequence (text/__init__.py:23-58): clean -> jamo decomposition -> ids of the symbols that exist -> EOS.  The Hangul
  decomposition is the Unicode algorithm (syllable = 0xAC00 + (lead*21 + vowel)*28 + tail), which is what the `jamo` package
  the reference imports does for precomposed syllables.
* The Korean normaliser (numbers, units, Latin letters, dictionaries: text/korean.py:139-306) lives in korean.py; pass
  `normalizer=KoreanNormalizer(...)` to run it in front of the decomposition.  Characters that have no symbol are dropped,
  exactly as `_should_keep_symbol` does.
* The training batcher (padding contract, length bucketing) lives in feeder.py."""
import re

import numpy as np

PAD, EOS = "_", "~"
PUNC, SPACE = "!'(),-.:;?", " "
JAMO_LEADS = "".join(chr(c) for c in range(0x1100, 0x1113))
JAMO_VOWELS = "".join(chr(c) for c in range(0x1161, 0x1176))
JAMO_TAILS = "".join(chr(c) for c in range(0x11A8, 0x11C3))
symbols = PAD + EOS + JAMO_LEADS + JAMO_VOWELS + JAMO_TAILS + PUNC + SPACE
_symbol_to_id = {s: i for i, s in enumerate(symbols)}
_id_to_symbol = {i: s for i, s in enumerate(symbols)}
_BRACED_SPAN = re.compile(r"\{.+?\}", re.S)
_HANGUL0, _HANGUL1 = 0xAC00, 0xD7A3


def hangul_to_jamo(text):
    """Precomposed Hangul syllables -> (lead, vowel[, tail]) conjoining jamo; everything else unchanged."""
    out = []
    for ch in text:
        c = ord(ch)
        if _HANGUL0 <= c <= _HANGUL1:
            i = c - _HANGUL0
            out.append(chr(0x1100 + i // (21 * 28)))
            out.append(chr(0x1161 + (i % (21 * 28)) // 28))
            if i % 28:
                out.append(chr(0x11A7 + i % 28))
        else:
            out.append(ch)
    return "".join(out)


def jamo_to_korean(text):
    """Recombine runs of lead + vowel (+ tail) into syllables (text/korean.py:54-88); anything else passes through."""
    text = hangul_to_jamo(text)
    out, i, n = [], 0, len(text)
    while i < n:
        ch = text[i]
        if ch in JAMO_LEADS and i + 1 < n and text[i + 1] in JAMO_VOWELS:
            lead, vowel, tail = ord(ch) - 0x1100, ord(text[i + 1]) - 0x1161, 0
            i += 2
            if i < n and text[i] in JAMO_TAILS:
                tail = ord(text[i]) - 0x11A7
                i += 1
            out.append(chr(_HANGUL0 + (lead * 21 + vowel) * 28 + tail))
        else:
            out.append(ch)
            i += 1
    return "".join(out)


def text_to_sequence(text, normalizer=None, as_token=False):
    """ids (int32) of `text`, EOS appended (text/__init__.py:23-58 with the korean cleaner).  A `{...}` span is ARPAbet in the
    reference; its Korean symbol table has no ARPAbet entries, so a span contributes nothing -- the text is cut at the spans and
    every piece outside them is normalised and decomposed on its own, as the reference cleans piece by piece."""
    norm = (lambda s: s.strip()) if normalizer is None else normalizer
    ids = [_symbol_to_id[ch] for piece in _BRACED_SPAN.split(text) if piece
           for ch in hangul_to_jamo(norm(piece)) if ch in _symbol_to_id and ch not in (PAD, EOS)]
    ids.append(_symbol_to_id[EOS])
    if as_token:
        return sequence_to_text(ids, combine_jamo=True)
    return np.array(ids, dtype=np.int32)


def sequence_to_text(sequence, skip_eos_and_pad=False, combine_jamo=False):
    s = "".join(_id_to_symbol[int(i)] for i in sequence
                if int(i) in _id_to_symbol and not (skip_eos_and_pad and _id_to_symbol[int(i)] in (EOS, PAD)))
    return jamo_to_korean(s) if combine_jamo else s


def pad_token_rows(rows):
    """Token rows of different lengths -> one int32 array, zero (PAD) filled on the right (what synthesize() feeds)."""
    n = max(len(x) for x in rows)
    out = np.zeros((len(rows), n), np.int32)
    for i, x in enumerate(rows):
        out[i, :len(x)] = np.asarray(x)
    return out
