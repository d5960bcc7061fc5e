"""Multi-GPU glue: self-play shards by game with no data-path collective; the only exchanges are
  * the all-gather of the (state, pi, z, meta) samples into every rank's replay memory after a wave,
  * the weight broadcast after optimize_network,
  * the all-gather of the per-game results of a sharded batched arena (a handful of bytes per game).

One process per GPU; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU (tests).
"""
import torch
import torch.distributed as dist

_ALIGN = 16  # every field starts on a 16-byte boundary of the packed buffer (dtype views need 4, RCCL likes 16)


def initialized():
    return dist.is_available() and dist.is_initialized()


def rank_world(group=None):
    """(rank, world size) of the torch.distributed job, (0, 1) outside one"""
    if initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def rank_game_range(rank, world_size, games_per_rank, wave=0):
    """global game ids [first, first+n) played by `rank` in self-play wave `wave` (disjoint Philox streams)"""
    first = (wave * world_size + rank) * games_per_rank
    return first, games_per_rank


def shard_range(n, rank, world):
    """(first, count, per): the contiguous block of n units (episodes of an iteration, rounds of an arena) that `rank` works on;
    `per` = ceil(n / world) is every rank's padded block length.  The last ranks of a job with more ranks than units get count 0."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(per, n - lo), per


def gather_sharded_rows(local, n, group=None, force=False):
    """`local`: this rank's [per, ...] block of per-unit results (the first `count` rows of shard_range are real, the rest padding).
    -> the [n, ...] results of all units in unit order, on every rank.  One fixed-size all-gather (a few bytes per unit)."""
    if not initialized() or (dist.get_world_size(group) == 1 and not force):
        return local[:n]
    world = dist.get_world_size(group)
    per = local.shape[0]
    assert per == (n + world - 1) // world, (per, n, world)
    rows = all_gather_rows(local, group=group, force=True).view((world, per) + tuple(local.shape[1:]))
    return torch.cat([rows[r, :shard_range(n, r, world)[1]] for r in range(world)])


def _pack_layout(samples, n_rows):
    """byte offsets of every field in the packed per-rank buffer (fields in sorted key order, n_rows rows each)"""
    off, layout = 0, {}
    for k in sorted(samples):
        t = samples[k]
        row = t.element_size()
        for d in t.shape[1:]:
            row *= d
        layout[k] = (off, row)
        off += (n_rows * row + _ALIGN - 1) // _ALIGN * _ALIGN
    return layout, max(off, _ALIGN)


def all_gather_samples(samples, group=None, force=False):
    """samples: dict of tensors with equal leading dim S_rank.  Returns the concatenation over ranks (rank order).

    Variable-length: one all_gather of the counts, then ONE all_gather_into_tensor of a packed byte buffer (every
    field padded to the largest count) -- a single large collective is the shape RCCL's ring over xGMI is tuned for
    (per-link bound: ~81 MB per rank at 4096 Othello games).  `force` runs the collectives even at world size 1
    (single-GPU rehearsal of the RCCL path)."""
    if not initialized():
        return samples
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return samples
    if dist.get_backend(group) == "gloo":  # CPU rehearsal of the N > 1 path: gloo gathers host copies
        samples = {k: v.cpu() for k, v in samples.items()}
    samples = {k: v.contiguous() for k, v in samples.items()}
    any_t = next(iter(samples.values()))
    dev = any_t.device
    n_local = torch.tensor([any_t.shape[0]], dtype=torch.int64, device=dev)
    counts_t = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts_t, n_local, group=group)
    counts = [int(c) for c in counts_t.tolist()]
    n_max = max(counts)
    if n_max == 0:
        return samples
    layout, nbytes = _pack_layout(samples, n_max)
    send = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    for k, t in samples.items():
        off, row = layout[k]
        send[off: off + t.shape[0] * row] = t.view(-1).view(torch.uint8) if t.numel() else t.new_zeros(0, dtype=torch.uint8)
    recv = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, nbytes)
    out = {}
    for k, t in samples.items():
        off, row = layout[k]
        parts = [recv[r, off: off + c * row] for r, c in enumerate(counts) if c]
        flat = torch.cat(parts) if len(parts) != 1 else parts[0].clone()
        out[k] = flat.view(t.dtype).view((sum(counts),) + tuple(t.shape[1:]))
    return out


def broadcast_state_dict(module, src=0, group=None, force=False):
    """weight hand-off after optimize_network (trainer.py:383-387): 4.46 MB for OthelloNet 8x8, one flat broadcast"""
    if not initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    tensors = [t for t in module.state_dict().values()]
    if dist.get_backend(group) != "gloo" and not all(t.is_cuda for t in tensors):
        raise ValueError("distributed training over RCCL needs config.device = 'cuda'")
    floats = [t for t in tensors if t.dtype == torch.float32]
    others = [t for t in tensors if t.dtype != torch.float32]  # BatchNorm's num_batches_tracked (int64)
    if floats:
        flat = torch.cat([t.reshape(-1) for t in floats])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in floats:
            t.copy_(flat[off: off + t.numel()].view_as(t))
            off += t.numel()
    for t in others:
        dist.broadcast(t, src=src, group=group)


def all_gather_rows(t, group=None, force=False):
    """equal-length tensors of every rank stacked in rank order ([world * n, ...]); identity outside a job"""
    if not initialized() or (dist.get_world_size(group) == 1 and not force):
        return t
    world = dist.get_world_size(group)
    src = t.contiguous()
    if dist.get_backend(group) == "gloo":
        src = src.cpu()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out
