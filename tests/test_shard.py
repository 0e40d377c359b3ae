"""Multi-GPU path on CPU: two ranks over gloo exercise the page sharding and the token-stream gather that bench.py
and the batch driver use over RCCL (same code, backend "nccl")."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from handwritten_ocr_amd import shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, _, w = shard.init_from_env(device_backend=False)
    pages = [f"page{i:02d}" for i in range(7)]            # 7 pages over 2 ranks: ragged shards (4 and 3)
    mine = shard.shard(pages, r, w)
    reads = 3 * len(mine)
    toks = torch.arange(reads * 5, dtype=torch.int32).view(reads, 5) + 1000 * r
    counts = torch.full((reads,), 5, dtype=torch.int32)
    got = shard.gather_token_streams(toks, counts, dst=0)
    if r == 0:
        q.put([(t.tolist(), c.tolist()) for t, c in got])
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert len(got) == 2
    (t0, c0), (t1, c1) = got
    assert len(t0) == 12 and len(t1) == 9                 # 4 pages x 3 reads, 3 pages x 3 reads
    assert t0[0] == [0, 1, 2, 3, 4] and t1[0] == [1000, 1001, 1002, 1003, 1004]
    assert c0 == [5] * 12 and c1 == [5] * 9


def test_shard_helpers_single_process():
    pages = list(range(10))
    parts = [shard.shard(pages, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == pages and [len(p) for p in parts] == [3, 3, 2, 2]
    own = shard.owner_index(10, 4)
    for i, (r, j) in enumerate(own):
        assert parts[r][j] == i
    t = torch.zeros(2, 3, dtype=torch.int32)
    assert shard.gather_token_streams(t, torch.tensor([3, 3], dtype=torch.int32))[0][0] is t
