"""The state contract between graph nodes (keys and value types of ocr_agent/state.py:10-29) and the trace-event record
(ocr_agent/state.py:32-63), for hosts without the reference package.  Pinned by tests/golden/nodes_kats.json."""
from __future__ import annotations

import time
from datetime import datetime, timezone
from typing import Optional, TypedDict

# key -> type, grouped by who writes it
_READS = {"image_path": str, "candidates": "list[dict]", "strategies_used": "list[str]", "current_best": str}
_LOOP = {"critiques": "list[dict]", "edits": "list[dict]", "current_score": float, "prev_score": float,
         "prev_critique": Optional[dict], "plateau_count": int, "iteration": int, "max_iterations": int}
_RUN = {"status": str, "reason": str, "config": dict, "trace_events": "list[dict]", "start_time": float}
OCRState = TypedDict("OCRState", {**_READS, **_LOOP, **_RUN})

_EVENT_KEYS = ("iteration", "agent", "action", "input_summary", "output_summary")
_EVENT_OPTIONAL = ("full_input", "full_output", "metrics")


def new_state(image_path: str, cfg, max_iterations: int | None = None, accept_threshold: int | None = None) -> dict:
    """Initial state as the CLI builds it (ocr_agent/transcribe.py:44-67)."""
    knobs = {"accept_threshold": accept_threshold or cfg.ACCEPT_THRESHOLD, "plateau_patience": cfg.PLATEAU_PATIENCE,
             "strategies": list(cfg.PREPROCESSING_STRATEGIES), "agreement_threshold": cfg.AGREEMENT_THRESHOLD}
    state = {k: [] for k in ("candidates", "critiques", "edits", "strategies_used", "trace_events")}
    state.update(image_path=str(image_path), current_best="", current_score=0.0, prev_score=0.0, prev_critique=None,
                 plateau_count=0, iteration=0, max_iterations=max_iterations or cfg.MAX_ITERATIONS, status="running",
                 reason="", config=knobs, start_time=time.monotonic())
    return state


def trace_log(state, *, iteration, agent, action, input_summary, output_summary, full_input=None, full_output=None,
              metrics=None, decision=None) -> dict:
    """One trace event; prints the live `[mm:ss] summary` line the reference prints."""
    required = (iteration, agent, action, input_summary, output_summary)
    optional = (full_input, full_output, metrics)
    elapsed = round(time.monotonic() - state["start_time"], 1)
    event = {"timestamp": datetime.now(timezone.utc).isoformat(), "elapsed_seconds": elapsed}
    event.update(zip(_EVENT_KEYS, required))
    event.update((k, v or {}) for k, v in zip(_EVENT_OPTIONAL, optional))
    event["decision"] = decision
    print("[{:02d}:{:02d}] {}".format(*divmod(int(elapsed), 60), output_summary))
    return event
