"""CPU: the N>1 prompt-sharding path with world_size 2 over gloo (no GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plangen_amd.dist import broadcast_prompts, gather_rows, shard_range


def test_shard_range_partitions():
    for n in (1, 2, 7, 64, 255):
        for ws in (1, 2, 4, 8):
            spans = [shard_range(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, ws, port, B, L, T, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 1000, (2 * B, L), generator=g).int()
    mask = torch.ones((2 * B, L + T), dtype=torch.int32)
    my_ids, my_mask, lo, hi, nB = broadcast_prompts(ids if rank == 0 else None, mask if rank == 0 else None, "cpu")
    assert nB == B and my_ids.shape == (2 * (hi - lo), L) and my_mask.shape == (2 * (hi - lo), L + T)
    assert torch.equal(my_ids, ids[2 * lo:2 * hi])            # CFG pairs stay together
    # stand-in for generation: tokens derived from the cond row only
    toks = my_ids[0::2, :T].clone()
    allt = gather_rows(toks, B)
    assert torch.equal(allt, ids[0::2, :T])
    imgs = gather_rows(my_ids[0::2, :3].float().view(-1, 3, 1, 1), B)
    assert torch.equal(imgs.view(B, 3), ids[0::2, :3].float())
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 5, 6, 4, None), nprocs=2, join=True)
