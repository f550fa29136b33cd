"""Korean text normalisation in front of the jamo tokeniser (behaviour of text/korean.py:139-306 of the
reference, re-authored: nothing here is taken from its source or from its ko_dictionary tables).

What `KoreanNormalizer.normalize` does to a sentence, in this order (the reference's order, text/korean.py:150-166):

1. strip; drop "(13일)"-style day notes and parentheses that hold only CJK ideographs (hanja glosses);
2. replace whole phrases from a user dictionary (`phrases`: e.g. "1+1" -> its reading);
3. replace Latin words found in a user dictionary (`english`: exact, case-sensitive word match);
4. spell the remaining ALL-CAPS Latin words letter by letter (LG -> 엘지); mixed-case words are left alone;
5. re-quote: every sentence inside a pair of quotation marks becomes its own '...' span;
6. read units (%, cm, mm, km, kg, then bare m) and numbers: a number directly followed by a native counter (시, 명, 살, 마리,
   가지, 개 ...) is read with native numerals (열두시, 스물네살, 백한마리), every other number with Sino-Korean numerals
   (이천일십칠년), a decimal part digit by digit after "쩜 ", a sign as 플러스 / 마이너스.

The dictionaries are data the caller supplies (`load_dictionaries(path)` reads {"english": {...}, "phrases": {...}} from JSON);
the built-in tables below are only facts of the language: letter names, digits, the native numerals and the counters."""
import json
import re

LETTER_NAMES = dict(zip("ABCDEFGHIJKLMNOPQRSTUVWXYZ",
                        ["에이", "비", "씨", "디", "이", "에프", "지", "에이치", "아이", "제이", "케이", "엘", "엠", "엔", "오", "피", "큐", "알",
                         "에스", "티", "유", "브이", "더블유", "엑스", "와이", "지"]))
SINO_DIGITS = ["영", "일", "이", "삼", "사", "오", "육", "칠", "팔", "구"]
NATIVE_ONES = ["", "한", "두", "세", "네", "다섯", "여섯", "일곱", "여덟", "아홉"]          # attributive forms (before a counter)
NATIVE_TENS = ["", "열", "스물", "서른", "마흔", "쉰", "예순", "일흔", "여든", "아흔"]
SMALL_UNITS = ["", "십", "백", "천"]                     # within a group of four digits
BIG_UNITS = ["", "만", "억", "조", "경", "해"]            # per group of four digits
UNIT_WORDS = [("%", "퍼센트"), ("cm", "센치미터"), ("mm", "밀리미터"), ("km", "킬로미터"), ("kg", "킬로그람")]   # longer symbols first ...
UNIT_WORDS_LATE = [("m", "미터")]                                                                                # ... bare "m" last
COUNTERS = ["시", "명", "가지", "살", "마리", "포기", "송이", "수", "톨", "통", "점", "개", "벌", "척", "채", "다발", "그루", "자루", "줄", "켤레",
            "그릇", "잔", "마디", "상자", "사람", "곡", "병", "판"]

_DAY_NOTE = re.compile(r"\(\d+일\)")
_HANJA_NOTE = re.compile("\\([\u2e80-\u2e99\u2e9b-\u2ef3\u2f00-\u2fd5\u3005\u3007\u3021-\u3029\u3038-\u303b\u3400-\u4db5\u4e00-\u9fc3"
                         "\uf900-\ufa2d\ufa30-\ufa6a\ufa70-\ufad9]+\\)")
_LATIN_WORD = re.compile(r"[A-Za-z]+")
_QUOTED = re.compile("([`\"'＂“‘])(.+?)([`\"'＂”’])")
_NUMBER = r"([+-]?\d[\d,]*)[\.]?\d*"
_COUNTED = re.compile(_NUMBER + "(" + "|".join(COUNTERS) + ")")
_PLAIN_NUMBER = re.compile(_NUMBER)
_SENTENCE_END = re.compile(r"(?<=[.!?])\s+")


class NumberFormatError(Exception):
    pass


def _positional_reading(digits, ones):
    """Digit string -> numeral words: every non-zero digit gives ones[d] + 십/백/천 by its place inside its group of four, every
    non-empty group is closed with 만/억/조/...  (2017 -> 이천일십칠; the silent leading one is dropped by the caller)."""
    size, out, group = len(digits), "", ""
    for i, ch in enumerate(digits):
        place = size - 1 - i
        v = ord(ch) - 48
        if v:
            group += ones[v] + SMALL_UNITS[place % 4]
        if place % 4 == 0 and group:
            if place // 4 >= len(BIG_UNITS):
                raise NumberFormatError("number too large to read")
            out += group + BIG_UNITS[place // 4]
            group = ""
    return out


def _native_tens(words):
    """두십 -> 스물, 세십 -> 서른, ... and a 십 that no such digit word precedes -> 열 (left to right, first match wins)."""
    out, i = [], 0
    while i < len(words):
        for d in range(2, 10):
            if words.startswith(NATIVE_ONES[d] + "십", i):
                out.append(NATIVE_TENS[d])
                i += len(NATIVE_ONES[d]) + 1
                break
        else:
            out.append(NATIVE_TENS[1] if words[i] == "십" else words[i])
            i += 1
    return "".join(out)


def read_integer(digits, native=False):
    """Reading of a non-negative decimal digit string.  Sino-Korean: 2017 -> 이천일십칠 (the reference keeps 일 before an inner 십), 10000 -> 만 (a leading 일 is
    silent).  native=True is the counting form used before counters: 12 -> 열두, 24 -> 스물네, 101 -> 백한, 20 -> 스물; like the
    reference, the digit words of every place are the native ones (200 -> 두백)."""
    digits = digits.lstrip("0")
    if not digits:
        return SINO_DIGITS[0]
    if native:
        text = _positional_reading(digits, NATIVE_ONES)
        if text.startswith("한") and len(text) > 1:
            text = text[1:]
        return _native_tens(text)
    text = _positional_reading(digits, [""] + SINO_DIGITS[1:])
    if text.startswith("일") and len(text) > 1:
        text = text[1:]
    return text


def read_number(token, counter=""):
    """'-12.35' -> '마이너스 십이쩜 삼오'; ('24', counter='살') -> '스물네살'.  Commas are thousands separators.  A value of zero
    reads '영' and, as in the reference, swallows the counter."""
    s = token.replace(",", "")
    sign = ""
    if s[:1] in ("+", "-"):
        sign, s = ("플러스 " if s[0] == "+" else "마이너스 "), s[1:]
    parts = s.split(".")
    if len(parts) > 2 or not parts[0].isdigit() or (len(parts) == 2 and parts[1] and not parts[1].isdigit()):
        raise NumberFormatError(" [!] Wrong number format")
    frac = parts[1] if len(parts) == 2 else None
    if counter and frac is not None:
        raise NumberFormatError(" [!] `is_count` and float number does not fit each other")
    if int(parts[0]) == 0 and not (frac or "").strip("0"):
        return SINO_DIGITS[0]
    body = read_integer(parts[0], native=bool(counter)) if int(parts[0]) else ""
    if frac is not None:
        body += "쩜 " + "".join(SINO_DIGITS[ord(c) - 48] for c in frac)
    return sign + body + counter


def split_sentences(text):
    """Sentence boundaries inside a quotation: after . ! ? followed by white space."""
    return [s for s in _SENTENCE_END.split(text.strip()) if s]


class KoreanNormalizer(object):
    def __init__(self, english=None, phrases=None):
        self.english = dict(english or {})
        self.phrases = dict(phrases or {})
        self._phrase_re = self._alternation(self.phrases)
        self._unit_re = self._alternation(dict(UNIT_WORDS))
        self._unit_late_re = self._alternation(dict(UNIT_WORDS_LATE))

    @staticmethod
    def _alternation(table):
        if not table:
            return None
        return re.compile("|".join(re.escape(k) for k in sorted(table, key=len, reverse=True)))

    def _latin(self, m):
        word = m.group()
        if word in self.english:
            return self.english[word]
        if word.isupper():
            return "".join(LETTER_NAMES[c] for c in word)
        return word

    @staticmethod
    def _requote(m):
        return " ".join("'%s'" % s for s in split_sentences(m.group(2))) or "''"

    # the stages of normalize(), callable one by one (tests/test_reference_vectors.py replays the reference's stage outputs)
    def apply_phrases(self, text):
        return text if self._phrase_re is None else self._phrase_re.sub(lambda m: self.phrases[m.group()], text)

    def apply_latin(self, text):
        """dictionary words, then letter-by-letter spelling of what is left in capitals (one pass: a dictionary reading holds no Latin letters)"""
        return _LATIN_WORD.sub(self._latin, text)

    def apply_quotes(self, text):
        return _QUOTED.sub(self._requote, text)

    def apply_numbers(self, text):
        text = self._unit_re.sub(lambda m: dict(UNIT_WORDS)[m.group()], text)
        text = self._unit_late_re.sub(lambda m: dict(UNIT_WORDS_LATE)[m.group()], text)
        text = _COUNTED.sub(lambda m: read_number(m.group(1), m.group(2)), text)
        return _PLAIN_NUMBER.sub(lambda m: read_number(m.group()), text)

    def normalize(self, text):
        text = text.strip()
        text = _DAY_NOTE.sub("", text)
        text = _HANJA_NOTE.sub("", text)
        return self.apply_numbers(self.apply_quotes(self.apply_latin(self.apply_phrases(text))))

    __call__ = normalize


def load_dictionaries(path):
    """{"english": {word: reading}, "phrases": {phrase: reading}} from a JSON file -> KoreanNormalizer."""
    with open(path, encoding="utf-8") as f:
        d = json.load(f)
    return KoreanNormalizer(d.get("english"), d.get("phrases"))


def tokenize(text, normalizer=None, as_id=False):
    """text/korean.py:139-147: normalise, decompose into conjoining jamo, append EOS; ids with as_id."""
    from . import text as T
    norm = normalizer if normalizer is not None else KoreanNormalizer()
    jamo = T.hangul_to_jamo(norm(text))
    if as_id:
        return [T._symbol_to_id[c] for c in jamo] + [T._symbol_to_id[T.EOS]]
    return list(jamo) + [T.EOS]
