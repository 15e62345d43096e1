"""Prompt sharding across the GPUs of one node.

The path shards over prompts: images are independent, a CFG pair (rows 2k, 2k+1) stays on
one rank, weights are replicated.  There is no collective inside the 576-step loop: rank 0
broadcasts the collated ids/mask once per batch (RCCL over xGMI; ~0.5 MB), every rank
generates its contiguous slice, tokens / images are gathered to rank 0 at the end (2.3 KB/img
of tokens).  The reference's equivalent is accelerate's dataloader sharding with no gather
(plangen_base.py:994).  Backend: ``nccl`` (= RCCL) on GPUs, ``gloo`` in the CPU tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_images: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of ``n_images`` over ranks; the first ``n % world`` ranks get one more."""
    q, r = divmod(n_images, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _skip(ws: int, force_collectives: bool) -> bool:
    """World 1 needs no collective; ``force_collectives`` runs them anyway when a process group exists (a 1-rank RCCL
    communicator: the single-GPU self-test of the multi-GPU path, tests/test_gpu_rccl.py / bench.py ``rccl_selftest``)."""
    return ws == 1 and not (force_collectives and dist.is_available() and dist.is_initialized())


def broadcast_prompts(ids: Optional[torch.Tensor], mask: Optional[torch.Tensor], device, src: int = 0,
                      force_collectives: bool = False):
    """Rank ``src`` holds cfg_inputs_ids int32 [2B, L] and cfg_attention_mask int32 [2B, L+T];
    every rank returns its own slice (rows 2*lo .. 2*hi) plus (lo, hi, B)."""
    rank, ws = world()
    if _skip(ws, force_collectives):
        B = ids.shape[0] // 2
        return ids.to(device), mask.to(device), 0, B, B
    hdr = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == src:
        hdr = torch.tensor([ids.shape[0], ids.shape[1], mask.shape[1]], dtype=torch.int64, device=device)
    dist.broadcast(hdr, src)
    R, L, LM = (int(v) for v in hdr.tolist())
    if rank == src:
        ids_d = ids.to(device=device, dtype=torch.int32).contiguous()
        mask_d = mask.to(device=device, dtype=torch.int32).contiguous()
    else:
        ids_d = torch.empty((R, L), dtype=torch.int32, device=device)
        mask_d = torch.empty((R, LM), dtype=torch.int32, device=device)
    dist.broadcast(ids_d, src)
    dist.broadcast(mask_d, src)
    B = R // 2
    lo, hi = shard_range(B, ws, rank)
    return ids_d[2 * lo:2 * hi].contiguous(), mask_d[2 * lo:2 * hi].contiguous(), lo, hi, B


def _padded(local: torch.Tensor, n_total: int, ws: int) -> torch.Tensor:
    per = (n_total + ws - 1) // ws
    buf = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    return buf


def _unpad(outs, n_total: int, ws: int) -> torch.Tensor:
    parts = []
    for r in range(ws):
        lo, hi = shard_range(n_total, ws, r)
        parts.append(outs[r][: hi - lo])
    return torch.cat(parts, dim=0)


def gather_rows(local: torch.Tensor, n_total: int, dst: int = 0, force_collectives: bool = False) -> Optional[torch.Tensor]:
    """Gather per-image rows (tokens [b, T] or images [b, 3, S, S]) to rank ``dst`` in batch order; other ranks
    get None.  A gather, not an all-gather: only the rank that writes the results needs them (2.3 KB of tokens per
    image, but 1.7 MB per image for pixels)."""
    rank, ws = world()
    if _skip(ws, force_collectives):
        return local
    buf = _padded(local, n_total, ws)
    outs = [torch.empty_like(buf) for _ in range(ws)] if rank == dst else None
    dist.gather(buf, outs, dst=dst)
    return _unpad(outs, n_total, ws) if rank == dst else None


def all_gather_rows(local: torch.Tensor, n_total: int, force_collectives: bool = False) -> torch.Tensor:
    """The same rows on EVERY rank (only for callers that really need them everywhere)."""
    rank, ws = world()
    if _skip(ws, force_collectives):
        return local
    buf = _padded(local, n_total, ws)
    outs = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(outs, buf)
    return _unpad(outs, n_total, ws)
