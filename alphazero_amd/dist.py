"""Multi-GPU glue: self-play shards by game with no data-path collective; the only exchange is
the all-gather of the (state, pi, z, meta) samples into every rank's replay memory after a wave.

One process per GPU; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU (tests).
"""
import torch
import torch.distributed as dist


def rank_game_range(rank, world_size, games_per_rank, wave=0):
    """global game ids [first, first+n) played by `rank` in self-play wave `wave` (disjoint Philox streams)"""
    first = (wave * world_size + rank) * games_per_rank
    return first, games_per_rank


def all_gather_samples(samples, group=None):
    """samples: dict of tensors with equal leading dim S_rank. Returns the concatenation over ranks (rank order).

    Variable-length: one all_gather of the counts, then one padded all_gather per field (ring all-gather is
    per-link bound on xGMI: ~81 MB/rank at 4096 games, a few ms)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return samples
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":  # CPU rehearsal of the N > 1 path: gloo gathers host copies
        samples = {k: v.cpu() for k, v in samples.items()}
    any_t = next(iter(samples.values()))
    n_local = torch.tensor([any_t.shape[0]], dtype=torch.int64, device=any_t.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    out = {}
    for k, t in samples.items():
        pad = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        out[k] = torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    return out


def broadcast_state_dict(module, src=0, group=None):
    """weight hand-off after optimize_network (trainer.py:383-387): 4.46 MB for OthelloNet 8x8"""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in module.state_dict().values():
        dist.broadcast(t, src=src, group=group)
