"""Candidate compare / merge / error-rate functions of the page-read path, on native integer kernels.

Same names, arguments, return shapes and rounding as the reference's string tools (ocr_agent/tools.py:28-139,
:305-320, :326-493), which run them as pure-Python O(n*m) loops inside node_initial_ocr (nodes.py:95, :114).
Here the two quadratic parts — edit distance and the LCS alignment table — run in libhwocr_text.so
(csrc/text.cpp: bit-parallel Myers/Hyyro distance; int32 LCS table with the reference's backtrack tie-break);
Python only normalises text and maps characters / words to integer ids.
"""
from __future__ import annotations

import ctypes as C
import json
import re
from pathlib import Path

import numpy as np

from . import _lib

_WS = re.compile(r"\s+")
_PUNCT = str.maketrans({"‘": "'", "’": "'", "“": '"', "”": '"', "–": "-", "—": "-"})


def parse_ground_truth(file_path) -> str | None:
    """Text after a `## Ground Truth` header, or the whole file when there is none (tools.py:28-45)."""
    p = Path(file_path)
    if not p.exists():
        return None
    raw = p.read_text(encoding="utf-8")
    head = "## Ground Truth"
    at = raw.find(head)
    body = raw if at < 0 else raw[at + len(head):]
    return body.strip() or None


def normalize_text(text: str, lower: bool = False) -> str:
    """Straight quotes, hyphen dashes, single spaces, trimmed (tools.py:51-63)."""
    t = _WS.sub(" ", text.translate(_PUNCT)).strip()
    return t.lower() if lower else t


def _codepoints(s: str) -> np.ndarray:
    return np.frombuffer(s.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)


def _lev_ids(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    d = _lib.text().hwocr_levenshtein_u32(a.ctypes.data_as(C.c_void_p), len(a), b.ctypes.data_as(C.c_void_p), len(b))
    if d < 0:
        raise _lib.HwocrError("hwocr_levenshtein_u32 rejected its arguments")
    return int(d)


def levenshtein(a: str, b: str) -> int:
    """Unit-cost edit distance over code points (tools.py:69-83)."""
    return _lev_ids(_codepoints(a), _codepoints(b))


def _word_ids(lists: list[list[str]], fold: bool = False) -> list[np.ndarray]:
    table: dict[str, int] = {}
    out = []
    for words in lists:
        ids = np.empty(len(words), dtype=np.uint32)
        for i, w in enumerate(words):
            key = w.lower() if fold else w
            ids[i] = table.setdefault(key, len(table))
        out.append(ids)
    return out


def _levenshtein_words(a: list[str], b: list[str]) -> int:
    """Edit distance over word tokens (tools.py:86-100)."""
    ia, ib = _word_ids([a, b])
    return _lev_ids(ia, ib)


def cer(ground_truth: str, ocr_output: str, lower: bool = False) -> float:
    gt = normalize_text(ground_truth, lower)
    return levenshtein(gt, normalize_text(ocr_output, lower)) / max(len(gt), 1)


def wer(ground_truth: str, ocr_output: str, lower: bool = False) -> float:
    gw = normalize_text(ground_truth, lower).split()
    return _levenshtein_words(gw, normalize_text(ocr_output, lower).split()) / max(len(gw), 1)


def tier1_metrics(ground_truth: str, ocr_output: str, lower: bool = False) -> dict:
    """CER / WER (char-joined and token) / exact match, rounded to 4 dp (tools.py:119-139)."""
    gt, ocr = normalize_text(ground_truth, lower), normalize_text(ocr_output, lower)
    gw, ow = gt.split(), ocr.split()
    gj, oj = " ".join(gw), " ".join(ow)
    return {
        "input": ocr_output,
        "cer": round(levenshtein(gt, ocr) / max(len(gt), 1), 4),
        "wer": round(levenshtein(gj, oj) / max(len(gj), 1), 4),
        "wer_token": round(_levenshtein_words(gw, ow) / max(len(gw), 1), 4),
        "exact_match": gt == ocr,
        "gt_chars": len(gt),
        "ocr_chars": len(ocr),
    }


def evaluate(transcription: str, ground_truth: str | None = None, lower: bool = False) -> dict:
    result = {}
    if ground_truth is not None:
        print("  [eval] Computing CER/WER against ground truth...")
        result["tier1_raw_vs_gt"] = tier1_metrics(ground_truth, transcription, lower)
    return result


def parse_json_response(raw: str):
    """JSON out of an LLM reply: fences stripped, else the first balanced {...} / [...] (tools.py:211-243)."""
    body = re.sub(r"\s*```$", "", re.sub(r"^```(?:json)?\s*", "", raw.strip()))
    try:
        return json.loads(body)
    except json.JSONDecodeError:
        pass
    for opener, closer in (("{", "}"), ("[", "]")):
        start = body.find(opener)
        if start < 0:
            continue
        depth = 0
        for k in range(start, len(body)):
            ch = body[k]
            if ch == opener:
                depth += 1
            elif ch == closer:
                depth -= 1
                if depth == 0:
                    try:
                        return json.loads(body[start: k + 1])
                    except json.JSONDecodeError:
                        break
    return None


def _find_differing_segments(w1: list[str], w2: list[str]) -> list[dict]:
    """Greedy resynchronising walk with a 9-word look-ahead (tools.py:353-405); `position` indexes w1."""
    segs = []
    i = j = 0
    n1, n2 = len(w1), len(w2)
    while i < n1 and j < n2:
        if w1[i] == w2[j]:
            i += 1
            j += 1
            continue
        reach = min(10, max(n1 - i, n2 - j) + 1)
        moved = False
        for k in range(1, reach):
            if i + k < n1 and w1[i + k] == w2[j]:       # w1 has k extra words
                segs.append({"position": i, "v1_text": " ".join(w1[i: i + k]), "v2_text": ""})
                i += k
                moved = True
                break
            if j + k < n2 and w2[j + k] == w1[i]:       # w2 has k extra words
                segs.append({"position": i, "v1_text": "", "v2_text": " ".join(w2[j: j + k])})
                j += k
                moved = True
                break
        if not moved:
            segs.append({"position": i, "v1_text": w1[i], "v2_text": w2[j]})
            i += 1
            j += 1
    if i < n1 or j < n2:
        segs.append({"position": i, "v1_text": " ".join(w1[i:]), "v2_text": " ".join(w2[j:])})
    return segs


def compare_versions(v1: str, v2: str) -> dict:
    """Agreement %, char / word edit distances and differing word segments of two candidates (tools.py:326-350)."""
    n1, n2 = normalize_text(v1), normalize_text(v2)
    cd = levenshtein(n1, n2)
    w1, w2 = n1.split(), n2.split()
    return {
        "agreement_rate": round((1 - cd / max(len(n1), len(n2), 1)) * 100, 1),
        "char_edit_distance": cd,
        "word_edit_distance": _levenshtein_words(w1, w2),
        "differing_segments": _find_differing_segments(w1, w2),
    }


def _align_ids(backbone: np.ndarray, words: np.ndarray) -> np.ndarray:
    out = np.empty(len(backbone), dtype=np.int32)
    rc = _lib.text().hwocr_lcs_align_u32(backbone.ctypes.data_as(C.c_void_p), len(backbone),
                                         words.ctypes.data_as(C.c_void_p), len(words), out.ctypes.data_as(C.c_void_p))
    _lib.check(rc, "hwocr_lcs_align_u32")
    return out


def _align_to_backbone(backbone: list[str], words: list[str]) -> list[str | None]:
    """Words placed on the backbone positions of a case-insensitive LCS (tools.py:465-493)."""
    ib, iw = _word_ids([backbone, words], fold=True)
    idx = _align_ids(ib, iw)
    return [None if k < 0 else words[k] for k in idx.tolist()]


def merge_versions(versions: list[str]) -> str:
    """Word-level plurality vote on the longest candidate's word positions; unresolved ties stay as [a|b]
    (tools.py:411-462)."""
    if not versions:
        return ""
    if len(versions) == 1:
        return versions[0]
    lists = [normalize_text(v).split() for v in versions]
    spine = max(range(len(lists)), key=lambda k: len(lists[k]))
    ids = _word_ids(lists, fold=True)
    placed = []
    for words, wid in zip(lists, ids):
        idx = _align_ids(ids[spine], wid).tolist()
        placed.append([None if k < 0 else words[k] for k in idx])
    merged = []
    for pos, spine_word in enumerate(lists[spine]):
        seen = [col[pos] for col in placed if col[pos] is not None]
        if not seen:
            merged.append(spine_word)
            continue
        tally: dict[str, int] = {}
        for w in seen:
            tally[w] = tally.get(w, 0) + 1
        top = max(tally.values())
        best = [w for w, n in tally.items() if n == top]
        if len(best) == 1:
            merged.append(best[0])
        else:
            distinct = list(dict.fromkeys(seen))
            merged.append(distinct[0] if len(distinct) == 1 else "[" + "|".join(distinct) + "]")
    return " ".join(merged)
