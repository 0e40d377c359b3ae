"""ORACLE — test infrastructure, not product code.

Plain-Python restatement of the reference's string functions on the page-read path, deliberately as slow and as
literal as the original so that it can serve as the checker (and as the "port" CPU baseline in bench.py):

    normalize_text        ocr_agent/tools.py:51-63
    levenshtein           ocr_agent/tools.py:69-83    (also the word-list form, :86-100)
    compare_versions      ocr_agent/tools.py:326-350  + segments :353-405
    merge_versions        ocr_agent/tools.py:411-462  + LCS alignment :465-493
    cer / wer / tier1     ocr_agent/tools.py:103-139

Pinned by tests/golden/text_kats.json (outputs of the reference's own functions, tools/make_goldens.py).
"""
from __future__ import annotations

import re


def normalize_text(text, lower=False):
    for src, dst in (("‘", "'"), ("’", "'"), ("“", '"'), ("”", '"'), ("–", "-"), ("—", "-")):
        text = text.replace(src, dst)
    text = re.sub(r"\s+", " ", text).strip()
    return text.lower() if lower else text


def edit_distance(a, b):
    """Two-row Wagner-Fischer over any pair of sequences (str or list of words)."""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        row = [i]
        for j, y in enumerate(b, 1):
            row.append(min(prev[j] + 1, row[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = row
    return prev[-1]


def cer(gt, out, lower=False):
    g = normalize_text(gt, lower)
    return edit_distance(g, normalize_text(out, lower)) / max(len(g), 1)


def wer(gt, out, lower=False):
    g = normalize_text(gt, lower).split()
    return edit_distance(g, normalize_text(out, lower).split()) / max(len(g), 1)


def tier1_metrics(gt_raw, out_raw, lower=False):
    gt, out = normalize_text(gt_raw, lower), normalize_text(out_raw, lower)
    gw, ow = gt.split(), out.split()
    return {"input": out_raw,
            "cer": round(edit_distance(gt, out) / max(len(gt), 1), 4),
            "wer": round(edit_distance(" ".join(gw), " ".join(ow)) / max(len(" ".join(gw)), 1), 4),
            "wer_token": round(edit_distance(gw, ow) / max(len(gw), 1), 4),
            "exact_match": gt == out, "gt_chars": len(gt), "ocr_chars": len(out)}


def differing_segments(w1, w2):
    out, i, j = [], 0, 0
    while i < len(w1) and j < len(w2):
        if w1[i] == w2[j]:
            i, j = i + 1, j + 1
            continue
        hit = None
        for k in range(1, min(10, max(len(w1) - i, len(w2) - j) + 1)):
            if i + k < len(w1) and w1[i + k] == w2[j]:
                hit = ("a", k)
                break
            if j + k < len(w2) and w2[j + k] == w1[i]:
                hit = ("b", k)
                break
        if hit is None:
            out.append({"position": i, "v1_text": w1[i], "v2_text": w2[j]})
            i, j = i + 1, j + 1
        elif hit[0] == "a":
            out.append({"position": i, "v1_text": " ".join(w1[i:i + hit[1]]), "v2_text": ""})
            i += hit[1]
        else:
            out.append({"position": i, "v1_text": "", "v2_text": " ".join(w2[j:j + hit[1]])})
            j += hit[1]
    if i < len(w1) or j < len(w2):
        out.append({"position": i, "v1_text": " ".join(w1[i:]), "v2_text": " ".join(w2[j:])})
    return out


def compare_versions(v1, v2):
    a, b = normalize_text(v1), normalize_text(v2)
    d = edit_distance(a, b)
    return {"agreement_rate": round((1 - d / max(len(a), len(b), 1)) * 100, 1), "char_edit_distance": d,
            "word_edit_distance": edit_distance(a.split(), b.split()),
            "differing_segments": differing_segments(a.split(), b.split())}


def align_to_backbone(backbone, words):
    n, m = len(backbone), len(words)
    bl, wl = [x.lower() for x in backbone], [x.lower() for x in words]
    table = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(n):
        for j in range(m):
            table[i + 1][j + 1] = table[i][j] + 1 if bl[i] == wl[j] else max(table[i][j + 1], table[i + 1][j])
    placed = [None] * n
    i, j = n, m
    while i and j:
        if bl[i - 1] == wl[j - 1]:
            placed[i - 1] = words[j - 1]
            i, j = i - 1, j - 1
        elif table[i - 1][j] >= table[i][j - 1]:
            i -= 1
        else:
            j -= 1
    return placed


def merge_versions(versions):
    if len(versions) < 2:
        return versions[0] if versions else ""
    lists = [normalize_text(v).split() for v in versions]
    spine = lists[max(range(len(lists)), key=lambda k: len(lists[k]))]
    cols = [align_to_backbone(spine, l) for l in lists]
    words = []
    for pos in range(len(spine)):
        seen = [c[pos] for c in cols if c[pos] is not None]
        if not seen:
            words.append(spine[pos])
            continue
        counts = {}
        for w in seen:
            counts[w] = counts.get(w, 0) + 1
        top = max(counts.values())
        best = [w for w in counts if counts[w] == top]
        uniq = list(dict.fromkeys(seen))
        words.append(best[0] if len(best) == 1 else (uniq[0] if len(uniq) == 1 else "[" + "|".join(uniq) + "]"))
    return " ".join(words)
