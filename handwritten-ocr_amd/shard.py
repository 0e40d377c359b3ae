"""One process per GPU: page sharding and the gather of token streams.

The reference is strictly serial (one page, one read at a time: ocr_agent/transcribe.py:194-210, nodes.py:86-110).
Reads are independent given the page, so a batch folder shards embarrassingly: the sorted page list
(transcribe.py:185-187) is dealt round-robin to ranks, all reads of a page stay on one GPU, and the only collective
is a gather of the generated token streams (<= 8 KB per read: latency-bound, any xGMI topology) to rank 0, which
detokenises, merges and writes the per-page outputs.  Backend "nccl" is RCCL on ROCm; "gloo" serves CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def local_device_index(local_rank: int) -> int:
    """CUDA device of a rank: its LOCAL_RANK.  More ranks than GPUs is an error — unless HWOCR_ALLOW_SHARED_GPU=1 says the
    ranks may share devices (a rehearsal of the multi-rank path on a one-GPU box; then the collective runs over gloo)."""
    n = torch.cuda.device_count()
    if n and local_rank >= n:
        if os.environ.get("HWOCR_ALLOW_SHARED_GPU", "0") in ("", "0"):
            raise RuntimeError(f"LOCAL_RANK {local_rank} but only {n} GPU(s) are visible: one process per GPU")
        return local_rank % n
    return local_rank


def init_from_env(device_backend: bool = True) -> tuple[int, int, int]:
    """(rank, local_rank, world).  Initialises torch.distributed when WORLD_SIZE > 1: "nccl" (= RCCL on ROCm, over xGMI)
    when the ranks own their GPUs, "gloo" on CPU and when ranks share a GPU (RCCL refuses two ranks on one device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        use_nccl = device_backend and torch.cuda.is_available()
        if use_nccl:
            torch.cuda.set_device(local_device_index(local))
            # ranks sharing a GPU (rehearsal, HWOCR_ALLOW_SHARED_GPU=1): EVERY rank must take gloo, not only the surplus ones
            if int(os.environ.get("LOCAL_WORLD_SIZE", world)) > torch.cuda.device_count():
                use_nccl = False
        # the collectives' timeout covers the longest stretch a rank legitimately waits for another: rank 0's host work on a whole
        # folder (detokenise, node replay, agents, files) while the others sit in the closing barrier (RCCL's default: 10 minutes)
        import datetime

        timeout = datetime.timedelta(seconds=float(os.environ.get("HWOCR_DIST_TIMEOUT_S", "7200")))
        dist.init_process_group(backend="nccl" if use_nccl else "gloo", rank=rank, world_size=world, timeout=timeout)
    return rank, local, world


def failed_ranks(ok: bool, device=None) -> list[int]:
    """Every rank reports whether its share of the work succeeded; returns the ranks that did not (same list on all ranks).  Called
    BEFORE a data collective so that a rank whose reads raised still takes part in one — the others would otherwise wait in the
    gather until the backend's timeout."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [] if ok else [0]
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device or "cpu")
    flags = [torch.zeros_like(flag) for _ in range(dist.get_world_size())]
    dist.all_gather(flags, flag)
    return [r for r, f in enumerate(flags) if int(f) == 0]


def shard(items: list, rank: int, world: int) -> list:
    """Round-robin deal: rank r gets items r, r + world, ..."""
    return items[rank::world]


def owner_index(n_items: int, world: int) -> list[tuple[int, int]]:
    """For every global item: (rank that reads it, its index within that rank's shard)."""
    return [(i % world, i // world) for i in range(n_items)]


def gather_token_streams(tokens: torch.Tensor, counts: torch.Tensor, dst: int = 0):
    """tokens int32 [reads_local, N], counts int32 [reads_local] -> on `dst`: list over ranks of (tokens, counts)
    trimmed to each rank's own read count and width; elsewhere None.  Ranks may hold different numbers of reads and
    different widths N (reads stop at different lengths): everything is padded to the maxima so that ONE fixed-shape
    gather moves it (<= 8 KB per read: latency-bound, any xGMI topology)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [(tokens, counts)]
    world, rank = dist.get_world_size(), dist.get_rank()
    shape = torch.tensor([tokens.shape[0], tokens.shape[1]], dtype=torch.int32, device=tokens.device)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    shapes = [s.cpu().tolist() for s in shapes]
    n_max, w_max = max(s[0] for s in shapes), max(s[1] for s in shapes)
    buf = torch.zeros(n_max, w_max + 1, dtype=torch.int32, device=tokens.device)
    buf[: tokens.shape[0], : tokens.shape[1]] = tokens
    buf[: tokens.shape[0], w_max] = counts
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return [(out[r][: shapes[r][0], : shapes[r][1]], out[r][: shapes[r][0], w_max]) for r in range(world)]
