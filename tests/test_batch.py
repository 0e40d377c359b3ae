"""Batch-folder driver: one batched engine pass for all pages, then the reference's per-page node logic and output files.
The engine is replaced by a scripted `run_ocr_batch_tokens` (CPU test: tokens = the UTF-8 bytes of a scripted text); the node
logic itself is pinned by tests/test_nodes.py."""
import contextlib
import io
import json

import numpy as np
import pytest
from PIL import Image

from handwritten_ocr_amd import batch, tools
from handwritten_ocr_amd.compat import config, nodes
from handwritten_ocr_amd.compat.state import new_state
from handwritten_ocr_amd.synth import make_page


def _script_engine(monkeypatch, fn):
    """fn(images, params) -> list of texts; installed at the token seam the batch driver uses."""
    def tokens(images, params=None, on_done=None):
        out = [list(t.encode("utf-8")) for t in fn(images, params)]
        if on_done is not None:            # the engine reports reads as they stop, in any order
            for i in reversed(range(len(out))):
                on_done(i, out[i])
        return out

    monkeypatch.setattr(tools, "run_ocr_batch_tokens", tokens)
    monkeypatch.setattr(tools, "decode_tokens", lambda streams: [bytes(t).decode("utf-8") for t in streams])


def _pages(tmp_path, n):
    paths = []
    for i in range(n):
        p = tmp_path / f"page{i:02d}.png"
        Image.fromarray(make_page(i, 64, 64), "RGB").save(p)
        paths.append(p)
    (tmp_path / "notes.txt").write_text("not an image")
    return paths


def test_batched_initial_ocr_equals_serial_node(tmp_path, monkeypatch):
    paths = _pages(tmp_path, 3)
    # page 0: reads agree -> the speculative third read must not be consumed; pages 1, 2: disagreement -> tie-breaker
    script = {0: ["same text here", "same text here", "UNUSED"], 1: ["alpha beta gamma delta", "completely different words", "alpha beta gamma"],
              2: ["x y z", "p q r s t", "x y z w"]}
    calls = []

    def fake_batch(images, params=None):
        calls.append(len(images))
        assert all(isinstance(im, Image.Image) for im in images)
        return [script[i // 3][i % 3] for i in range(len(images))]

    _script_engine(monkeypatch, fake_batch)
    with contextlib.redirect_stdout(io.StringIO()):
        states = batch.initial_ocr_batched([str(p) for p in paths])
    assert calls == [9]  # ONE engine pass for 3 pages x 3 strategies
    for i, st in enumerate(states):
        # serial reference path: the same node with per-read fakes
        it = iter(script[i])
        monkeypatch.setattr(nodes, "run_ocr", lambda path, params=None, _it=it: next(_it))
        monkeypatch.setattr(nodes, "preprocess_image", lambda path, s: path)
        monkeypatch.setattr(nodes, "unload_ocr_model", lambda: None)
        serial = new_state(str(paths[i]), config)
        with contextlib.redirect_stdout(io.StringIO()):
            serial.update(nodes.node_initial_ocr(serial))
        assert st["current_best"] == serial["current_best"]
        assert [c["text"] for c in st["candidates"]] == [c["text"] for c in serial["candidates"]]
        assert st["strategies_used"] == serial["strategies_used"]
        strip = lambda ev: {k: v for k, v in ev.items() if k not in ("timestamp", "elapsed_seconds", "input_summary")}
        assert [strip(e) for e in st["trace_events"]] == [strip(e) for e in serial["trace_events"]]
    assert len(states[0]["candidates"]) == 2 and len(states[1]["candidates"]) == 3


def test_folder_outputs(tmp_path, monkeypatch):
    paths = _pages(tmp_path, 2)
    gtd = tmp_path / "gt"
    gtd.mkdir()
    (gtd / "page00.md").write_text("# doc\n\n## Ground Truth\n\nhello world\n")
    _script_engine(monkeypatch, lambda images, params=None: ["hello world"] * len(images))
    assert [p.name for p in batch.list_images(tmp_path)] == ["page00.png", "page01.png"]
    out_dir = tmp_path / "results"
    outs = batch.transcribe_folder(batch.list_images(tmp_path), out_dir, gtd, quiet=True)
    assert [o.name for o in outs] == ["page00_transcription.txt", "page01_transcription.txt"]
    assert outs[0].read_text() == "hello world"
    ev = json.loads((out_dir / "page00_trace.json").read_text())
    assert [e["action"] for e in ev] == ["preprocess", "ocr", "preprocess", "ocr", "compare", "merge"]
    summary = (out_dir / "page00_trace_summary.txt").read_text().splitlines()
    assert len(summary) == len(ev) and summary[-1].endswith("Merged → 11 chars")
    e0 = json.loads((out_dir / "page00_eval.json").read_text())
    assert e0["tier1_raw_vs_gt"]["cer"] == 0.0 and e0["pipeline_status"] == "initial_ocr"
    assert "tier1_raw_vs_gt" not in json.loads((out_dir / "page01_eval.json").read_text())


def test_cli_rejects_missing_input(tmp_path):
    with pytest.raises(SystemExit):
        batch.main([str(tmp_path / "nope")])


def test_reocr_rounds_are_answered_from_the_batched_pass(tmp_path, monkeypatch):
    """With agents, every distinct strategy is read in the one batched pass and the `reocr` node — the same
    node code — gets its read from there: the engine is entered exactly once, and the result equals the serial graph."""
    paths = _pages(tmp_path, 2)
    distinct = batch._speculative_strategies(list(config.PREPROCESSING_STRATEGIES), every=True)
    labels = [nodes._strategy_label(s) for s in distinct]
    assert len(labels) > 3, "the reference config has more distinct strategies than initial_ocr uses"
    calls = []

    def fake_batch(images, params=None):
        calls.append(len(images))
        return [f"page {i // len(labels)} read with {labels[i % len(labels)]} words more words" for i in range(len(images))]

    class Arb:
        def __init__(self, versions):
            self.final_text, self.confidence = versions[-1]["text"].upper(), 77
            self.decisions, self.uncertain_segments = [], []

        def model_dump(self):
            return {"final_text": self.final_text, "confidence": self.confidence}

    def critic(text, previous_critique=None):
        # first look: ask for a re-read; after it: accept
        if text.isupper():
            return {"overall_confidence": 95, "verdict": "accept", "issues": []}
        return {"overall_confidence": 40, "verdict": "needs_reocr", "issues": []}

    agents = {"critic": critic, "editor": lambda t, c: {"corrected_text": t, "changes": []}, "arbitrator": Arb}
    _script_engine(monkeypatch, fake_batch)
    monkeypatch.setattr(nodes, "run_ocr", lambda *a, **k: pytest.fail("re-OCR must not enter the engine again"))
    monkeypatch.setattr(nodes, "unload_ocr_model", lambda: None)
    outs = batch.transcribe_folder(batch.list_images(tmp_path), tmp_path / "out", agents=agents, quiet=True)
    assert calls == [2 * len(labels)]
    for i, o in enumerate(outs):
        text = o.read_text()
        # initial_ocr consumed strategies 0..2 at most; the re-read is the next unused one, upper-cased by the scripted arbitrator
        ev = json.loads((tmp_path / "out" / f"page{i:02d}_trace.json").read_text())
        used = [e["metrics"]["strategy"] for e in ev if e["action"] == "ocr"]
        assert used == labels[: len(used)] and len(used) >= 3
        assert text == f"page {i} read with {used[-1]} words more words".upper()
        assert [e["action"] for e in ev][-3:] == ["arbitrate", "critique", "accept"]


def test_lossy_input_formats_see_the_reference_reencode(tmp_path, monkeypatch):
    """the serial path hands every transformed page to the model through a temp file with the INPUT's suffix
    (tools.py:668-672), which for .jpg re-quantises the pixels.  The batched driver must feed the engine the same pixels:
    compare what it hands over with what `preprocess_image` + `Image.open` (the serial path) produce, page by page."""
    from handwritten_ocr_amd import preprocess

    src = tmp_path / "photo.jpg"
    Image.fromarray(make_page(3, 120, 160), "RGB").save(src, quality=92)
    png = tmp_path / "scan.png"
    Image.fromarray(make_page(4, 120, 160), "RGB").save(png)
    seen = []

    def fake(images, params=None):
        seen.extend(images)
        return ["t"] * len(images)

    _script_engine(monkeypatch, fake)
    with contextlib.redirect_stdout(io.StringIO()):
        strategies, _ = batch.read_pages([str(src), str(png)])
    k = len(strategies)
    assert len(seen) == 2 * k
    for p, path in enumerate((src, png)):
        for j, s in enumerate(strategies):
            with contextlib.redirect_stdout(io.StringIO()):
                serial = Image.open(preprocess.preprocess_image(str(path), s))
            got = seen[p * k + j]
            assert got.mode == serial.mode and got.size == serial.size
            assert np.array_equal(np.asarray(got), np.asarray(serial)), (path.name, s)
    # and the re-encode is not a no-op for the JPEG page (the round-1 driver skipped it)
    direct = preprocess.apply_strategy(Image.open(src), strategies[0], quiet=True)
    assert not np.array_equal(np.asarray(direct.convert("RGB")), np.asarray(seen[0].convert("RGB")))
