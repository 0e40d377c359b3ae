"""String functions of the read path: the native implementation (handwritten_ocr_amd.text -> libhwocr_text.so) and the
oracle (oracle/text_ref.py, oracle/text_ref.c) against the known answers produced by the reference's own functions
(tests/golden/text_kats.json), plus size-independent properties at full page length.  Bit-exact throughout."""
import ctypes as C
import random

import numpy as np
import pytest

from handwritten_ocr_amd import text
from oracle import text_ref
from tests._golden import load_json

KATS = load_json("text_kats.json")


@pytest.fixture(scope="module")
def cref():
    from oracle import build_c

    lib = C.CDLL(build_c.build())
    lib.ref_levenshtein_u32.restype = C.c_int64
    lib.ref_levenshtein_u32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    lib.ref_lcs_align_u32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    return lib


@pytest.mark.parametrize("impl", [text, text_ref], ids=["native", "oracle"])
def test_pair_kats(impl):
    lev = impl.levenshtein if impl is text else text_ref.edit_distance
    levw = impl._levenshtein_words if impl is text else text_ref.edit_distance
    for k in KATS["pairs"]:
        a, b = k["a"], k["b"]
        na, nb = impl.normalize_text(a), impl.normalize_text(b)
        assert na == k["normalize_a"]
        assert impl.normalize_text(a, True) == k["normalize_a_lower"]
        assert lev(na, nb) == k["levenshtein"]
        assert levw(na.split(), nb.split()) == k["levenshtein_words"]
        assert impl.cer(a, b) == k["cer"] and impl.wer(a, b) == k["wer"] and impl.cer(a, b, True) == k["cer_lower"]
        assert impl.tier1_metrics(a, b) == k["tier1"]
        assert impl.compare_versions(a, b) == k["compare"]


@pytest.mark.parametrize("impl", [text, text_ref], ids=["native", "oracle"])
def test_merge_and_align_kats(impl):
    for k in KATS["merges"]:
        assert impl.merge_versions(k["versions"]) == k["merged"], k["versions"]
    align = impl._align_to_backbone if impl is text else text_ref.align_to_backbone
    for k in KATS["aligns"]:
        assert align(k["backbone"], k["words"]) == k["aligned"]


def test_long_page_kat_native():
    k = KATS["long"]
    assert text.compare_versions(k["a"], k["b"]) == k["compare"]
    assert text.merge_versions([k["a"], k["b"], k["third"]]) == k["merged3"]


def test_misc_tools_kats(tmp_path):
    for k in KATS["ground_truth"]:
        f = tmp_path / "gt.md"
        f.write_text(k["file_text"], encoding="utf-8")
        assert text.parse_ground_truth(f) == k["parsed"]
    assert text.parse_ground_truth(tmp_path / "missing.md") is None
    for k in KATS["json"]:
        assert text.parse_json_response(k["raw"]) == k["parsed"]


def _u32(xs):
    return np.asarray(xs, dtype=np.uint32)


def test_native_distance_vs_c_oracle_random(cref):
    rng = random.Random(7)
    from handwritten_ocr_amd import _lib

    lib = _lib.text()
    for trial in range(300):
        n = rng.choice([0, 1, 2, 63, 64, 65, 127, 128, 129, 200, 700])
        m = rng.choice([0, 1, 5, 63, 64, 65, 130, 500])
        alpha = rng.choice([2, 4, 30, 5000])
        a, b = _u32([rng.randrange(alpha) for _ in range(n)]), _u32([rng.randrange(alpha) for _ in range(m)])
        want = cref.ref_levenshtein_u32(a.ctypes.data, n, b.ctypes.data, m)
        got = lib.hwocr_levenshtein_u32(a.ctypes.data, n, b.ctypes.data, m)
        assert got == want, (n, m, alpha)


def test_native_alignment_vs_c_oracle_random(cref):
    rng = random.Random(8)
    from handwritten_ocr_amd import _lib

    lib = _lib.text()
    for trial in range(200):
        n, m = rng.choice([1, 3, 40, 260]), rng.choice([1, 2, 37, 300])
        alpha = rng.choice([2, 5, 50])
        a, b = _u32([rng.randrange(alpha) for _ in range(n)]), _u32([rng.randrange(alpha) for _ in range(m)])
        want, got = np.empty(n, np.int32), np.empty(n, np.int32)
        cref.ref_lcs_align_u32(a.ctypes.data, n, b.ctypes.data, m, want.ctypes.data)
        assert lib.hwocr_lcs_align_u32(a.ctypes.data, n, b.ctypes.data, m, got.ctypes.data) == 0
        assert np.array_equal(got, want)


def test_full_page_properties(cref):
    """A full handwritten page is ~1.7k characters: check distance axioms there."""
    rng = random.Random(9)
    chars = "abcdefghijklmnopqrstuvwxyz     "
    a = "".join(rng.choice(chars) for _ in range(2100))
    b = list(a)
    for _ in range(150):
        b[rng.randrange(len(b))] = rng.choice(chars)
    b = "".join(b)[7:]
    c = "".join(rng.choice(chars) for _ in range(1900))
    d = text.levenshtein
    assert d(a, a) == 0 and d(a, "") == len(a) and d("", b) == len(b)
    assert d(a, b) == d(b, a)
    assert abs(len(a) - len(b)) <= d(a, b) <= max(len(a), len(b))
    assert d(a, c) <= d(a, b) + d(b, c)
    ua, ub = text._codepoints(a), text._codepoints(b)
    assert d(a, b) == cref.ref_levenshtein_u32(ua.ctypes.data, len(ua), ub.ctypes.data, len(ub))
    # appending the same suffix to both never increases the distance
    assert d(a + "tail", b + "tail") <= d(a, b)


def test_unicode_and_degenerate_inputs():
    assert text.levenshtein("", "") == 0
    assert text.levenshtein("\U0001f600a", "a\U0001f600") == 2
    assert text.compare_versions("", "") == {"agreement_rate": 100.0, "char_edit_distance": 0, "word_edit_distance": 0,
                                             "differing_segments": []}
    assert text.merge_versions([]) == "" and text.merge_versions(["  raw  text "]) == "  raw  text "
    assert text.merge_versions(["a b", ""]) == "a b"
