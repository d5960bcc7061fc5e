"""N > 1 path on CPU: world_size-2 gloo processes exercise the sample all-gather and the game sharding."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from alphazero_amd.dist import all_gather_samples, rank_game_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, cnt = rank_game_range(rank, world, 5, wave=3)
    S = 7 + 3 * rank  # ragged per-rank sample counts
    smp = {"state": torch.full((S, 8, 8), rank, dtype=torch.int8), "pi": torch.full((S, 65), float(rank)),
           "z": torch.full((S,), rank, dtype=torch.int8),
           "meta": torch.stack([torch.arange(first, first + S, dtype=torch.int32)] * 4, dim=1)}
    out = all_gather_samples(smp)
    ok = out["z"].shape[0] == sum(7 + 3 * r for r in range(world))
    off = 0
    for r in range(world):
        s = 7 + 3 * r
        ok &= bool((out["z"][off:off + s] == r).all()) and bool((out["state"][off:off + s] == r).all())
        ok &= int(out["meta"][off, 0]) == (3 * world + r) * 5
        off += s
    ret[rank] = (ok, first, cnt)
    dist.destroy_process_group()


def test_all_gather_samples_gloo_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29533, ret), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world))
    ranges = [range(ret[r][1], ret[r][1] + ret[r][2]) for r in range(world)]
    assert not set(ranges[0]) & set(ranges[1])  # disjoint game ids -> disjoint Philox streams


def test_single_process_passthrough():
    sys.path.insert(0, ROOT)
    from alphazero_amd.dist import all_gather_samples
    d = {"z": torch.zeros(3)}
    assert all_gather_samples(d) is d
