"""The host side of pipeline.LanePipeline without a GPU: jobs are dealt to lanes round-robin and run concurrently, results come back
in job order, `hooks.ordered()` serialises in job order, an exception in one job is re-raised once every thread has finished and
later jobs are not started.  (The device side — streams, events, tokens equal to one batch at a time — is tests/test_pipeline_gpu.py.)"""
import threading
import time

import pytest

from handwritten_ocr_amd import pipeline


class FakeEngine:
    dev = None

    def __init__(self, name="lane0"):
        self.name, self.closed = name, False
        self._n = 0

    def lane(self):
        self._n += 1
        return FakeEngine(f"lane{self._n}")

    def close(self):
        self.closed = True


def test_jobs_are_dealt_round_robin_and_run_side_by_side():
    pipe = pipeline.LanePipeline(FakeEngine(), lanes=2)
    assert [e.name for e in pipe.engines] == ["lane0", "lane1"]
    both_inside = threading.Barrier(2, timeout=20)

    def job(e, hooks, k):
        if k < 2:
            both_inside.wait()      # jobs 0 and 1 must be running at the same time (two threads), or this times out
        return (k, e.name, threading.current_thread().name)

    out = pipe.run([(lambda e, h, k=k: job(e, h, k)) for k in range(5)])
    assert [o[0] for o in out] == list(range(5))
    assert [o[1] for o in out] == ["lane0", "lane1", "lane0", "lane1", "lane0"]
    assert {o[2] for o in out} == {"hwocr-lane0", "hwocr-lane1"}
    pipe.close()
    assert pipe.engines[1].closed and not pipe.engines[0].closed


def test_ordered_runs_in_job_order_whatever_the_lanes_do():
    pipe = pipeline.LanePipeline(FakeEngine(), lanes=3)
    order = []

    def job(e, hooks, k):
        time.sleep(0.02 * (5 - k))          # later jobs reach their ordered() call first
        hooks.ordered(lambda: order.append(k))
        return k

    assert pipe.run([(lambda e, h, k=k: job(e, h, k)) for k in range(6)]) == list(range(6))
    assert order == list(range(6))


@pytest.mark.parametrize("order", ["lockstep", "alternate"])
def test_a_failing_job_is_reraised_and_nobody_waits_for_it(order):
    pipe = pipeline.LanePipeline(FakeEngine(), lanes=2, order=order)
    started = []

    def job(e, hooks, k):
        started.append(k)
        if k == 1:
            raise ValueError("unreadable page")
        hooks.ordered(lambda: None)          # job 2 waits for job 1's turn: the failed job must release it
        return k

    t0 = time.time()
    with pytest.raises(ValueError, match="unreadable page"):
        pipe.run([(lambda e, h, k=k: job(e, h, k)) for k in range(6)])
    assert time.time() - t0 < 10
    assert 1 in started and len(started) < 6   # jobs after the failure are not started on either lane
    assert pipe.run([(lambda e, h, k=k: k) for k in range(3)]) == [0, 1, 2]   # usable afterwards


def test_one_lane_or_one_job_runs_on_the_callers_thread():
    pipe = pipeline.LanePipeline(FakeEngine(), lanes=1)
    me = threading.current_thread().name
    assert pipe.run([lambda e, h: (threading.current_thread().name, h)] * 3) == [(me, None)] * 3
    two = pipeline.LanePipeline(FakeEngine(), lanes=2)
    assert two.run([lambda e, h: (threading.current_thread().name, h)]) == [(me, None)]
    with pytest.raises(ValueError):
        pipeline.LanePipeline(FakeEngine(), lanes=0)
    with pytest.raises(ValueError):
        pipeline.LanePipeline(FakeEngine(), lanes=2, order="sideways")


def test_plan_lanes():
    """tools.plan_lanes: how a job's reads are spread over engine lanes (host logic)."""
    from handwritten_ocr_amd import tools

    assert tools.plan_lanes(100, 256, 2) == (1, 256)            # fits one lane
    assert tools.plan_lanes(768, 256, 2) == (3, 256)            # a 256-page folder: three fills, three lanes, ONE round
    assert tools.plan_lanes(768, 256, 1) == (1, 256)            # HWOCR_LANES=1
    assert tools.plan_lanes(768, 252, 2) == (2, 192)            # four fills of 252: two balanced rounds of 192 per lane
    assert tools.plan_lanes(512, 256, 2) == (2, 256)
    assert tools.plan_lanes(300, 256, 2) == (2, 150)            # one round, equal shares instead of 256 + 44
    assert tools.plan_lanes(2304, 256, 2) == (3, 256)           # nine fills: three lanes, three rounds
    lanes, per = tools.plan_lanes(100000, 256, 2)               # long jobs: every slot busy, continuous batching refills
    assert (lanes, per) == (2, 256)
    for n in (257, 600, 1000, 5000):
        lanes, per = tools.plan_lanes(n, 256, 2)
        assert 1 <= per <= 256 and lanes in (2, 3)
