"""Several batches in flight on one GPU: one `ReadEngine` lane per batch, each driven by its own host thread on its own HIP stream.

The reference runs one read at a time (ocr_agent/nodes.py:86-110) and `ReadEngine.generate` one BATCH at a time: tower, prefill,
then 511 decode steps, ~200 dependent launches each, on one stream.  A dependent chain leaves the chip partly idle (a decode step
alternates HBM-bound attention launches with latency-bound GEMM launches; every launch has a ramp and a tail), and a second,
independent chain fills those holes.  Measured on the MI355X (tools/bench_overlap.py, Qwen2-VL-2B shape, 84 pages x 3 reads x 512
tokens per batch, tokens identical to the serial schedule in every arm; profiles/r03f_overlap_*.json):

    one batch at a time                                   14.39 pages/s
    2 lanes started together (phases in lockstep)         15.24   (+5.9 %: decode || decode 3764 ms for two batches against 2 x 2160,
                                                                   tower || tower 5593 against 2 x 2835, prefill 1655 against 2 x 843)
    2 lanes, second half a step late (decode || tower)    15.37   (+6.8 %) - but +0.2 % at a third of a step: phase-sensitive
    2 lanes, strict alternation enforced by events        14.69   (+2.9 %: a decode chain beside a tower takes 2.8 x as long - its short
                                                                   launches queue behind 0.5-ms persistent GEMMs - and paces the pipeline)
    3 lanes                                               15.27-15.43

`LanePipeline(order="lockstep")` (the default) is the second row made the rule: the lanes start together and, doing equal work, stay
in step; nothing orders them on the device.  `order="alternate"` is the fourth row (kept selectable: it is the schedule VERDICT r2
asked to be measured): tower(k) waits for prefill(k-1), decode(k) for decode(k-1), by HIP events (`ReadEngine.generate(hooks=...)`).
Either way everything after a batch's decode (detokenise, compare / merge on the host) runs in that batch's thread; `hooks.ordered()`
serialises per-batch side effects in batch order.  Collectives are NOT issued from lane threads (RCCL wants one issuing order on every
rank): with several ranks the caller gathers the batches' token streams from ONE thread of its own, in batch order, as the lanes
hand them over (bench.py: run_steps).

Results are those of the serial schedule bit for bit: a lane is an ordinary engine over the same weights (`ReadEngine.lane()`: own
KV cache / state / workspaces) and batches never share state.
"""
from __future__ import annotations

import threading

import torch


class _JobHooks:
    """What `ReadEngine.generate` calls at its phase boundaries for job k (see the module docstring)."""

    def __init__(self, pipe: "LanePipeline", k: int):
        self.pipe, self.k = pipe, k

    def _after(self, recorded: dict, events: dict) -> None:
        if self.k == 0 or self.pipe.order != "alternate":
            return
        recorded[self.k - 1].wait()                       # host: the previous job has put its event into its stream
        torch.cuda.current_stream().wait_event(events[self.k - 1])

    def tower_begin(self) -> None:
        self._after(self.pipe._prefill_recorded, self.pipe._prefill_done)

    def prefill_end(self) -> None:
        if self.pipe.order != "alternate":
            return
        ev = torch.cuda.Event()
        ev.record()
        self.pipe._prefill_done[self.k] = ev
        self.pipe._prefill_recorded[self.k].set()
        self._after(self.pipe._decode_recorded, self.pipe._decode_done)

    def decode_end(self) -> None:
        if self.pipe.order != "alternate":
            return
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
                if self.pipe.streams is not None:
                    ev = torch.cuda.Event()
                    ev.record()
                    evs[self.k] = ev
                rec[self.k].set()
        self.pipe._ordered_done[self.k].set()


class LanePipeline:
    def __init__(self, engine, lanes: int = 2, order: str = "lockstep", reuse: list | None = None):
        """reuse: lanes of `engine` made earlier (another pipeline's engines[1:], which this one takes over) - a pipeline that grows
        from two lanes to three allocates ONE more KV cache, not two."""
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        if order not in ("lockstep", "alternate"):
            raise ValueError("order must be 'lockstep' or 'alternate'")
        self.order = order
        have = list(reuse or [])[: lanes - 1]
        self.engines = [engine] + have + [engine.lane() for _ in range(lanes - 1 - len(have))]
        self.device = engine.dev
        # (no device: the thread / ordering logic alone, for the CPU test of it — a ReadEngine always has one)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.engines] if self.device is not None else None

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
        gpu = self.streams is not None
        if gpu:
            caller = torch.cuda.current_stream(self.device)
            start = torch.cuda.Event()
            start.record(caller)

        def lane_jobs(lane: int) -> None:
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

        def worker(lane: int) -> None:
            if not gpu:
                return lane_jobs(lane)
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.streams[lane]):
                self.streams[lane].wait_event(start)      # what the caller queued before run() (uploads, preprocessing) comes first
                lane_jobs(lane)
                self.streams[lane].synchronize()

        threads = [threading.Thread(target=worker, args=(i,), name=f"hwocr-lane{i}") for i in range(len(self.engines))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if gpu:
            for s in self.streams:  # later work on the caller's stream sees the lanes' results
                caller.wait_stream(s)
        if errors:
            raise errors[0]
        return results
