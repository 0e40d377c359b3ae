"""Hot-path graph nodes of the compat layer against the reference's own nodes (tests/golden/nodes_kats.json: the
reference's node_initial_ocr / node_reocr run with scripted run_ocr / preprocess_image / run_arbitrator).
Compared: returned partial-state dicts, candidate dicts, trace-event skeletons (timestamps stripped), call order."""
import contextlib
import io

import pytest

from handwritten_ocr_amd.compat import config, nodes
from handwritten_ocr_amd.compat.state import new_state
from tests._golden import load_json

K = load_json("nodes_kats.json")["cases"]


def _strip(ev):
    ev = dict(ev)
    ev.pop("timestamp"), ev.pop("elapsed_seconds")
    return ev


class _Arb:
    def __init__(self, text):
        self.final_text, self.confidence, self.decisions, self.uncertain_segments = text, 77, [], []

    def model_dump(self):
        return {"final_text": self.final_text, "confidence": 77, "decisions": [], "uncertain_segments": []}


@pytest.mark.parametrize("name", sorted(K))
def test_nodes_match_reference(name, monkeypatch):
    k = K[name]
    calls = {"n": 0, "pre": []}

    def fake_ocr(path, params=None):
        t = k["texts"][min(calls["n"], len(k["texts"]) - 1)]
        calls["n"] += 1
        return t

    def fake_pre(path, strategy):
        calls["pre"].append(strategy if isinstance(strategy, str) else list(strategy))
        return path + "#" + ("+".join(strategy) if isinstance(strategy, list) else strategy)

    monkeypatch.setattr(nodes, "run_ocr", fake_ocr)
    monkeypatch.setattr(nodes, "preprocess_image", fake_pre)
    monkeypatch.setattr(nodes, "unload_ocr_model", lambda: None)
    monkeypatch.setattr(nodes, "run_arbitrator", lambda versions: _Arb(max((v["text"] for v in versions), key=len)))
    state = new_state("/pages/p1.png", config, max_iterations=3)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        upd = nodes.node_initial_ocr(state)
    got = {kk: ([_strip(e) for e in v] if kk == "trace_events" else v) for kk, v in upd.items()}
    assert got == k["update"]
    n_initial = len(k["update"]["candidates"])
    assert calls["n"] == n_initial and calls["pre"] == k["preprocess_calls"][:n_initial]
    lines = [ln.split("] ", 1)[-1] if ln.startswith("[") else ln for ln in buf.getvalue().splitlines()]
    assert lines == k["stdout_lines"]
    state.update(upd)
    for want in k["reocr_rounds"]:
        with contextlib.redirect_stdout(io.StringIO()):
            u = nodes.node_reocr(state)
        got = {kk: ([_strip(e) for e in v] if kk == "trace_events" else v) for kk, v in u.items()}
        assert got == want
        state.update(u)
    assert state.get("reason") == "exhausted"
    assert calls["pre"] == k["preprocess_calls"]  # the golden's call log spans the initial reads and every re-OCR round


def test_state_machine_runs_to_a_terminal(monkeypatch):
    texts = iter(["alpha beta gamma", "alpha beta gamma", "alpha beta gamma delta", "y", "z"])
    monkeypatch.setattr(nodes, "run_ocr", lambda p, params=None: next(texts))
    monkeypatch.setattr(nodes, "preprocess_image", lambda p, s: p)
    monkeypatch.setattr(nodes, "unload_ocr_model", lambda: None)
    monkeypatch.setattr(nodes, "run_arbitrator", lambda v: _Arb(v[-1]["text"]))
    script = iter([{"overall_confidence": 40, "verdict": "needs_reocr"}, {"overall_confidence": 60, "verdict": "needs_editing"},
                   {"overall_confidence": 95, "verdict": "accept"}])
    monkeypatch.setattr(nodes, "run_critic", lambda text, previous_critique=None: dict(next(script), segments=[], reasoning=""))
    monkeypatch.setattr(nodes, "run_editor", lambda text, crit: {"corrected_text": text + "!", "changes": [], "unresolved": []})
    with contextlib.redirect_stdout(io.StringIO()):
        final = nodes.run_graph(new_state("/p.png", config))
    assert final["status"] == "completed" and final["reason"] == "accept" and final["iteration"] == 3
    assert final["current_best"] == "alpha beta gamma delta!"
    actions = [e["action"] for e in final["trace_events"]]
    assert actions[:5] == ["preprocess", "ocr", "preprocess", "ocr", "compare"] and actions[-1] == "accept"
