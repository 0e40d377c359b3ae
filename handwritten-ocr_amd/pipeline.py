"""Two batches in flight on one GPU: batch k's decode (HBM / latency bound, 37 % of a step) beside batch k+1's vision tower + prefill
(matrix-pipe bound, 63 %).

The reference runs one read at a time (ocr_agent/nodes.py:86-110) and `ReadEngine.generate` one BATCH at a time: tower, prefill,
then 511 decode steps, on one stream.  The two halves stress different parts of the chip, and run on two HIP streams they take less
than their sum (tools/bench_overlap.py on the MI355X: 14.33 -> 15.35 pages/s, identical tokens).  `LanePipeline` is that schedule
made explicit:

  * `lanes` ReadEngines over the same weights (`ReadEngine.lane()`: own KV cache / state / workspaces), each driven by its own host
    thread on its own stream; job k runs on lane k % lanes;
  * device-side ordering by events, so that at most ONE tower + prefill phase and ONE decode phase are in flight and they belong to
    consecutive batches:  tower(k) waits for prefill(k-1) to finish, decode(k) for decode(k-1)  (`ReadEngine.generate(hooks=…)`);
  * everything after a batch's decode (token gather, detokenise, compare / merge on the host) runs in that batch's thread while
    the other lane's kernels keep the GPU busy.

Results are those of the sequential schedule bit for bit: a lane is an ordinary engine and batches never share state.
"""
from __future__ import annotations

import threading

import torch


class _JobHooks:
    """What `ReadEngine.generate` calls at its phase boundaries for job k (see the module docstring)."""

    def __init__(self, pipe: "LanePipeline", k: int):
        self.pipe, self.k = pipe, k

    def _after(self, recorded: dict, events: dict) -> None:
        if self.k == 0:
            return
        recorded[self.k - 1].wait()                       # host: the previous job has put its event into its stream
        torch.cuda.current_stream().wait_event(events[self.k - 1])

    def tower_begin(self) -> None:
        self._after(self.pipe._prefill_recorded, self.pipe._prefill_done)

    def prefill_end(self) -> None:
        ev = torch.cuda.Event()
        ev.record()
        self.pipe._prefill_done[self.k] = ev
        self.pipe._prefill_recorded[self.k].set()
        self._after(self.pipe._decode_recorded, self.pipe._decode_done)

    def decode_end(self) -> None:
        ev = torch.cuda.Event()
        ev.record()
        self.pipe._decode_done[self.k] = ev
        self.pipe._decode_recorded[self.k].set()

    def ordered(self, fn):
        """Run fn() when every earlier job's ordered() call has returned (job order): for what all ranks of a multi-GPU run must
        issue in the SAME order although two host threads drive this GPU - the collective that gathers a batch's token streams."""
        if self.k > 0:
            self.pipe._ordered_done[self.k - 1].wait()
        try:
            return fn()
        finally:
            self.pipe._ordered_done[self.k].set()

    def abandon(self) -> None:
        """The job raised before reaching a phase boundary: release whoever waits on it (their wait_event is then a no-op)."""
        for rec, evs in ((self.pipe._prefill_recorded, self.pipe._prefill_done), (self.pipe._decode_recorded, self.pipe._decode_done)):
            if not rec[self.k].is_set():
                ev = torch.cuda.Event()
                ev.record()
                evs[self.k] = ev
                rec[self.k].set()
        self.pipe._ordered_done[self.k].set()


class LanePipeline:
    def __init__(self, engine, lanes: int = 2):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        self.engines = [engine] + [engine.lane() for _ in range(lanes - 1)]
        self.device = engine.dev
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.engines]

    def close(self) -> None:
        for e in self.engines[1:]:
            e.close()

    def run(self, jobs: list) -> list:
        """jobs[k](engine, hooks) -> result: called on lane k % lanes, in that lane's thread, with that lane's stream current; it must
        pass `hooks` to its `engine.generate` call (exactly one per job).  Returns the results in job order; the first exception of
        any job is re-raised after every thread has finished."""
        n = len(jobs)
        if len(self.engines) == 1 or n <= 1:  # nothing to overlap: the caller's stream, no threads
            return [job(self.engines[0], None) for job in jobs]
        self._prefill_recorded = {k: threading.Event() for k in range(n)}
        self._decode_recorded = {k: threading.Event() for k in range(n)}
        self._ordered_done = {k: threading.Event() for k in range(n)}
        self._prefill_done, self._decode_done = {}, {}
        results, errors = [None] * n, []
        caller = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(caller)

        def worker(lane: int) -> None:
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.streams[lane]):
                self.streams[lane].wait_event(start)      # what the caller queued before run() (uploads, preprocessing) comes first
                for k in range(lane, n, len(self.engines)):
                    hooks = _JobHooks(self, k)
                    try:
                        if errors:
                            raise RuntimeError("an earlier batch of the pipeline failed")
                        results[k] = jobs[k](self.engines[lane], hooks)
                    except BaseException as e:  # noqa: BLE001  (re-raised by run())
                        errors.append(e)
                    finally:
                        hooks.abandon()
                self.streams[lane].synchronize()

        threads = [threading.Thread(target=worker, args=(i,), name=f"hwocr-lane{i}") for i in range(len(self.engines))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for s in self.streams:  # later work on the caller's stream sees the lanes' results
            caller.wait_stream(s)
        if errors:
            raise errors[0]
        return results
