"""OCRState contract and trace events (same keys as ocr_agent/state.py:10-63)."""
from __future__ import annotations

import time
from datetime import datetime, timezone
from typing import TypedDict


class OCRState(TypedDict):
    image_path: str
    candidates: list[dict]   # {text, source, ocr_params, score}
    critiques: list[dict]
    edits: list[dict]
    current_best: str
    current_score: float
    iteration: int
    max_iterations: int
    status: str              # running | completed | max_iterations
    reason: str              # accept | plateau | exhausted | max_iterations | ""
    strategies_used: list[str]
    plateau_count: int
    prev_score: float
    prev_critique: dict | None
    config: dict             # accept_threshold, plateau_patience, strategies, agreement_threshold
    trace_events: list[dict]
    start_time: float


def new_state(image_path: str, cfg, max_iterations: int | None = None, accept_threshold: int | None = None) -> dict:
    """Initial state exactly as the CLI builds it (ocr_agent/transcribe.py:44-67)."""
    return {
        "image_path": str(image_path), "candidates": [], "critiques": [], "edits": [], "current_best": "",
        "current_score": 0.0, "iteration": 0, "max_iterations": max_iterations or cfg.MAX_ITERATIONS,
        "status": "running", "reason": "", "strategies_used": [], "plateau_count": 0, "prev_score": 0.0,
        "prev_critique": None,
        "config": {"accept_threshold": accept_threshold or cfg.ACCEPT_THRESHOLD, "plateau_patience": cfg.PLATEAU_PATIENCE,
                   "strategies": list(cfg.PREPROCESSING_STRATEGIES), "agreement_threshold": cfg.AGREEMENT_THRESHOLD},
        "trace_events": [], "start_time": time.monotonic(),
    }


def trace_log(state, *, iteration, agent, action, input_summary, output_summary, full_input=None, full_output=None,
              metrics=None, decision=None) -> dict:
    elapsed = round(time.monotonic() - state["start_time"], 1)
    minutes, seconds = divmod(int(elapsed), 60)
    print(f"[{minutes:02d}:{seconds:02d}] {output_summary}")
    return {"timestamp": datetime.now(timezone.utc).isoformat(), "elapsed_seconds": elapsed, "iteration": iteration,
            "agent": agent, "action": action, "input_summary": input_summary, "output_summary": output_summary,
            "full_input": full_input or {}, "full_output": full_output or {}, "metrics": metrics or {},
            "decision": decision}
