"""The graph nodes that drive page reads, restated for hosts without the reference package.

  node_initial_ocr / node_reocr / _do_ocr_pass   ocr_agent/nodes.py:27-134, :239-302  (hot path, pinned by
                                                 tests/golden/nodes_kats.json: same partial-state dicts, candidate
                                                 keys, trace-event sequence, dedup / tie-break / exhaustion rules)
  critic / editor / terminal nodes, routers      ocr_agent/nodes.py:137-236, :305-382, graph.py:22-43 — only as far as
                                                 the state machine needs; the LLM agents themselves are out of scope
                                                 and arrive as callables (scripted stand-ins on the GPU box).
"""
from __future__ import annotations

from .. import tools as _tools
from .state import trace_log

# bound at import like the reference's `from ocr_agent.tools import ...` (nodes.py:8-14); tests rebind these names
compare_versions = _tools.compare_versions
merge_versions = _tools.merge_versions
preprocess_image = _tools.preprocess_image
run_ocr = _tools.run_ocr
unload_ocr_model = _tools.unload_ocr_model
run_arbitrator = None   # callable(versions) -> object with final_text / confidence / decisions / uncertain_segments
run_critic = None       # callable(text, previous_critique=None) -> dict (CriticResult fields)
run_editor = None       # callable(text, critique_dict) -> dict (EditorResult fields)


def _strategy_label(strategy) -> str:
    return "+".join(strategy) if isinstance(strategy, list) else strategy


def _do_ocr_pass(state, strategy, candidates, strategies_used, trace_events, iteration=0) -> None:
    label = _strategy_label(strategy)
    if label in strategies_used:
        return
    strategies_used.append(label)
    processed = preprocess_image(state["image_path"], strategy)
    trace_events.append(trace_log(state, iteration=iteration, agent="reader", action="preprocess",
                                  input_summary=f"Image: {state['image_path']}",
                                  output_summary=f"Preprocessed with '{label}'", metrics={"strategy": label}))
    text = run_ocr(processed)
    candidates.append({"text": text, "source": f"ocr_{label}", "ocr_params": {"strategy": label}, "score": None})
    trace_events.append(trace_log(state, iteration=iteration, agent="reader", action="ocr",
                                  input_summary=f"Preprocessed image ({label})",
                                  output_summary=f"OCR pass ({label}) → {len(text)} chars",
                                  full_output={"text_preview": text[:200]},
                                  metrics={"chars": len(text), "strategy": label}))


def node_initial_ocr(state) -> dict:
    print("\n=== PHASE 1: Initial OCR Reads ===")
    plan = state["config"]["strategies"]
    candidates, used, events = list(state["candidates"]), list(state["strategies_used"]), []
    _do_ocr_pass(state, plan[0] if plan else "original", candidates, used, events)
    if len(plan) > 1:
        _do_ocr_pass(state, plan[1], candidates, used, events)
    if len(candidates) >= 2:
        cmp = compare_versions(candidates[0]["text"], candidates[1]["text"])
        rate = cmp["agreement_rate"]
        low = rate < state["config"]["agreement_threshold"]
        events.append(trace_log(state, iteration=0, agent="orchestrator", action="compare",
                                input_summary="Comparing candidate 1 vs 2", output_summary=f"Versions agree {rate}%",
                                full_output=cmp, metrics={"agreement_rate": rate},
                                decision="tiebreaker" if low else "merge"))
        if low and len(plan) > 2:
            _do_ocr_pass(state, plan[2], candidates, used, events)
    texts = [c["text"] for c in candidates]
    best = merge_versions(texts)
    events.append(trace_log(state, iteration=0, agent="orchestrator", action="merge",
                            input_summary=f"Merging {len(texts)} candidates", output_summary=f"Merged → {len(best)} chars",
                            metrics={"merged_chars": len(best)}))
    print("\n--- Unloading OCR model to free memory for LLM agents ---")
    unload_ocr_model()
    return {"candidates": candidates, "current_best": best, "strategies_used": used,
            "trace_events": state["trace_events"] + events}


def node_reocr(state) -> dict:
    used, candidates, events = list(state["strategies_used"]), list(state["candidates"]), []
    nxt = next((s for s in state["config"]["strategies"] if _strategy_label(s) not in used), None)
    if nxt is None:
        return {"reason": "exhausted", "trace_events": state["trace_events"]}
    print(f"\n--- Re-OCR with strategy: {_strategy_label(nxt)} ---")
    _do_ocr_pass(state, nxt, candidates, used, events, iteration=state["iteration"])
    unload_ocr_model()
    new = candidates[-1]
    arb = run_arbitrator([{"text": state["current_best"], "source": "current_best", "score": state["current_score"]},
                          {"text": new["text"], "source": new["source"]}])
    events.append(trace_log(state, iteration=state["iteration"], agent="arbitrator", action="arbitrate",
                            input_summary=f"Current best vs {new['source']}",
                            output_summary=(f"Arbitrator: merged with confidence {arb.confidence}, "
                                            f"{len(arb.uncertain_segments)} uncertain segments"),
                            full_output=arb.model_dump(),
                            metrics={"confidence": arb.confidence, "n_decisions": len(arb.decisions),
                                     "n_uncertain": len(arb.uncertain_segments)}))
    return {"current_best": arb.final_text, "candidates": candidates, "strategies_used": used,
            "prev_critique": state["critiques"][-1] if state["critiques"] else None,
            "trace_events": state["trace_events"] + events}


# ---- the rest of the loop, as far as the state machine needs it -------------------------------------------------
def node_critic(state) -> dict:
    it = state["iteration"] + 1
    crit = run_critic(state["current_best"], previous_critique=state["prev_critique"])
    conf, verdict = crit["overall_confidence"], crit["verdict"]
    ev = trace_log(state, iteration=it, agent="critic", action="critique",
                   input_summary=f"Transcription ({len(state['current_best'])} chars)",
                   output_summary=f"Critic: confidence {conf}, verdict={verdict}", full_output=crit,
                   metrics={"confidence": conf}, decision=verdict)
    plateau = state["plateau_count"] + 1 if conf <= state["prev_score"] else 0
    return {"iteration": it, "critiques": list(state["critiques"]) + [crit], "current_score": conf,
            "plateau_count": plateau, "prev_score": conf, "trace_events": state["trace_events"] + [ev]}


def node_editor(state) -> dict:
    crit = state["critiques"][-1]
    ed = run_editor(state["current_best"], crit)
    ev = trace_log(state, iteration=state["iteration"], agent="editor", action="edit",
                   input_summary="Transcription + critic issues",
                   output_summary=f"Editor: fixed {len(ed.get('changes', []))} issues", full_output=ed)
    return {"current_best": ed["corrected_text"], "edits": list(state["edits"]) + [ed], "prev_critique": crit,
            "trace_events": state["trace_events"] + [ev]}


def _terminal(state, action, decision, status, reason, summary) -> dict:
    ev = trace_log(state, iteration=state["iteration"], agent="orchestrator", action=action, input_summary=summary,
                   output_summary=f"DONE ({reason}) — {state['iteration']} iterations, final confidence "
                                  f"{state['current_score']}", decision=decision)
    return {"status": status, "reason": reason, "trace_events": state["trace_events"] + [ev]}


def route_after_critic(state) -> str:
    """graph.py:22-36."""
    last = state["critiques"][-1]
    if last["verdict"] == "accept" or last["overall_confidence"] >= state["config"]["accept_threshold"]:
        return "accept"
    if state["plateau_count"] >= state["config"]["plateau_patience"]:
        return "plateau"
    if state["iteration"] >= state["max_iterations"]:
        return "max_iterations"
    return "reocr" if last["verdict"] == "needs_reocr" else "edit"


def run_graph(state: dict) -> dict:
    """START -> initial_ocr -> critic -> {accept | plateau | max_iterations | reocr | edit} ... -> END with LangGraph's
    last-write-wins merge of partial updates (graph.py:49-79: TypedDict state, no reducers)."""
    state = dict(state)
    state.update(node_initial_ocr(state))
    return run_graph_after_initial(state)


def run_graph_after_initial(state: dict) -> dict:
    """critic -> {accept | plateau | max_iterations | reocr | edit} ... -> END, for a state that already went through
    `initial_ocr` (the batched driver computes that node for many pages at once)."""
    state = dict(state)
    while True:
        state.update(node_critic(state))
        step = route_after_critic(state)
        if step == "accept":
            state.update(_terminal(state, "accept", "accept", "completed", "accept", "accepted"))
            return state
        if step == "plateau":
            state.update(_terminal(state, "plateau", "plateau_stop", "completed", "plateau", "no improvement"))
            return state
        if step == "max_iterations":
            state.update(_terminal(state, "max_iterations", "max_iterations_stop", "max_iterations", "max_iterations",
                                   f"Reached {state['max_iterations']} iterations"))
            return state
        if step == "reocr":
            state.update(node_reocr(state))
            if state.get("reason") == "exhausted":  # graph.py:39-43
                state.update(_terminal(state, "strategies_exhausted", "exhausted_stop", "completed", "exhausted",
                                       "All preprocessing strategies tried"))
                return state
        else:
            state.update(node_editor(state))
