"""N > 1 path on CPU: world_size-2 gloo processes exercise the sample all-gather and the game sharding."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, free_port


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
    # a rank without samples (fewer episodes than ranks) and the flat weight broadcast
    S2 = 0 if rank == 1 else 4
    out2 = all_gather_samples({"z": torch.full((S2,), rank, dtype=torch.int8), "pi": torch.full((S2, 9), float(rank))})
    ok &= out2["z"].shape[0] == 4 and out2["pi"].shape == (4, 9) and bool((out2["z"] == 0).all())
    from alphazero_amd.dist import all_gather_rows, broadcast_state_dict
    torch.manual_seed(rank)
    bn = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3))
    bn[1].num_batches_tracked += 7 * (rank + 1)
    broadcast_state_dict(bn, src=0)
    torch.manual_seed(0)
    want = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3))
    ok &= all(torch.equal(a, b) for a, b in zip(bn.state_dict().values(), want.state_dict().values()) if a.dtype == torch.float32)
    ok &= int(bn[1].num_batches_tracked) == 7
    rows = all_gather_rows(torch.full((3, 2), rank, dtype=torch.int32))
    ok &= rows.shape == (3 * world, 2) and bool((rows[3:] == 1).all()) and bool((rows[:3] == 0).all())
    ret[rank] = (ok, first, cnt)
    dist.destroy_process_group()


def test_all_gather_samples_gloo_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, free_port(), ret), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world))
    ranges = [range(ret[r][1], ret[r][1] + ret[r][2]) for r in range(world)]
    assert not set(ranges[0]) & set(ranges[1])  # disjoint game ids -> disjoint Philox streams


def test_single_process_passthrough():
    sys.path.insert(0, ROOT)
    from alphazero_amd.dist import all_gather_samples
    d = {"z": torch.zeros(3)}
    assert all_gather_samples(d) is d


def _worker_one(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    sys.path.insert(0, ROOT)
    from alphazero_amd.dist import all_gather_rows, all_gather_samples
    dist.init_process_group("gloo", rank=0, world_size=1)
    smp = {"state": torch.arange(5 * 9, dtype=torch.int8).view(5, 3, 3), "pi": torch.rand(5, 9), "z": torch.ones(5, dtype=torch.int8),
           "meta": torch.arange(20, dtype=torch.int32).view(5, 4)}
    same = all_gather_samples(smp)
    forced = all_gather_samples(smp, force=True)  # the collective path with one rank: what a 1-GPU box can rehearse
    ret[0] = (same is smp) and forced is not smp and all(torch.equal(forced[k], smp[k]) and forced[k].dtype == smp[k].dtype for k in smp) \
        and torch.equal(all_gather_rows(smp["meta"], force=True), smp["meta"])
    dist.destroy_process_group()


def test_forced_collectives_world1_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_one, args=(1, free_port(), ret), nprocs=1, join=True)
    assert ret[0]


def _worker_setup(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from alphazero_amd.games.othello import OthelloConfig
    from alphazero_amd.trainer import AlphaZeroTrainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = AlphaZeroTrainer(verbose=False)
    tr.game = "othello"
    tr.config = OthelloConfig(board_size=6, simulations=4, episodes=4, do_eval=True, eval_opponent="previous", eval_episodes=2, device="cpu")
    torch.manual_seed(1000 + rank)  # every rank draws its own initialisation
    tr.setup()
    ret[rank] = {k: v.clone() for k, v in tr.nn.state_dict().items()}
    dist.destroy_process_group()


def test_trainer_setup_hands_rank0_weights_to_every_rank():
    """without the broadcast in setup() wave 0 of a multi-GPU job plays with a different network per rank"""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker_setup, args=(world, free_port(), ret), nprocs=world, join=True)
    assert all(torch.equal(ret[0][k], ret[1][k]) for k in ret[0])
    from alphazero_amd.games.othello import OthelloConfig
    from alphazero_amd.games.registers import NETWORKS_REGISTER
    torch.manual_seed(1000)
    want = NETWORKS_REGISTER["othello"](config=OthelloConfig(board_size=6)).state_dict()
    assert all(torch.equal(ret[0][k], want[k]) for k in want)


def _worker8(rank, world, port, ret):
    """the N = 8 control flow of a job with FEWER units than ranks: 5 self-play episodes and 5 arena rounds over 8 ranks -> ranks
    5, 6, 7 play nothing, hold no samples and still take part in every collective (trainer.self_play, BatchedArena.play_games,
    bench.py's per-rank rows)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from alphazero_amd.dist import all_gather_samples, broadcast_state_dict, gather_sharded_rows, rank_game_range, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    n = 5
    lo, cnt, per = shard_range(n, rank, world)
    ok &= per == 1 and cnt == (1 if rank < n else 0) and lo == min(rank, n)
    # self-play: episode e yields 3 + e samples, every sample tagged with its episode
    S = sum(3 + e for e in range(lo, lo + cnt))
    tag = torch.cat([torch.full((3 + e,), e, dtype=torch.int32) for e in range(lo, lo + cnt)]) if cnt else torch.zeros(0, dtype=torch.int32)
    smp = {"state": tag.to(torch.int8).view(-1, 1, 1).expand(S, 8, 8).contiguous(), "pi": tag.float().view(-1, 1).expand(S, 65).contiguous(),
           "z": tag.to(torch.int8), "meta": torch.stack([tag] * 4, dim=1) if S else torch.zeros((0, 4), dtype=torch.int32)}
    out = all_gather_samples(smp)
    want = torch.cat([torch.full((3 + e,), e, dtype=torch.int32) for e in range(n)])
    ok &= torch.equal(out["meta"][:, 0], want) and torch.equal(out["z"], want.to(torch.int8)) and out["state"].shape == (len(want), 8, 8)
    ok &= bool((out["pi"][:, 7] == want.float()).all())
    # arena: (winner, score) rows of this rank's rounds, padded to `per`
    res = torch.full((per, 2), -2, dtype=torch.int32)
    if cnt:
        res[:cnt, 0] = torch.arange(lo, lo + cnt, dtype=torch.int32) % 3 - 1
        res[:cnt, 1] = 10 * torch.arange(lo, lo + cnt, dtype=torch.int32)
    allr = gather_sharded_rows(res, n, force=True)
    ok &= allr.shape == (n, 2) and torch.equal(allr[:, 1], 10 * torch.arange(n, dtype=torch.int32)) and torch.equal(allr[:, 0], torch.arange(n, dtype=torch.int32) % 3 - 1)
    # 17 units over 8 ranks: blocks of 3, the last rank is empty, rank 5 holds two
    lo2, cnt2, per2 = shard_range(17, rank, world)
    blk = torch.full((per2, 1), -1, dtype=torch.int64)
    blk[:cnt2, 0] = torch.arange(lo2, lo2 + cnt2)
    ok &= per2 == 3 and torch.equal(gather_sharded_rows(blk, 17, force=True)[:, 0], torch.arange(17))
    # weights: every rank starts elsewhere, rank 0's arrive
    torch.manual_seed(100 + rank)
    lin = torch.nn.Linear(4, 4)
    broadcast_state_dict(lin, src=0)
    torch.manual_seed(100)
    ok &= torch.equal(lin.weight, torch.nn.Linear(4, 4).weight)
    first, g = rank_game_range(rank, world, 4096, wave=2)
    ret[rank] = (bool(ok), first, g)
    dist.destroy_process_group()


def test_world8_with_fewer_units_than_ranks():
    """VERDICT r3 item 3 (CPU half): the sharding arithmetic and every collective of the multi-GPU path at world size 8, including
    ranks with zero episodes / zero arena rounds / zero samples"""
    world = 8
    ret = mp.Manager().dict()
    mp.spawn(_worker8, args=(world, free_port(), ret), nprocs=world, join=True)  # a free port: the test may run beside another copy of itself
    assert all(ret[r][0] for r in range(world)), {r: ret[r][0] for r in range(world)}
    ids = sorted(ret[r][1] for r in range(world))
    assert ids == [(2 * 8 + r) * 4096 for r in range(8)]  # config 3: 8 x 4096 disjoint game-id blocks per wave


def _run_bench(env_extra, n):
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=env, capture_output=True, text=True, timeout=300)


def test_bench_launcher_spawns_ranks_and_relays_rank0():
    """`python bench.py --gpus N` started plainly is its own launcher: N child ranks with the torch.distributed environment, rank 0's
    line on stdout (rehearsed here without a GPU: AZ_BENCH_DRYRUN makes the ranks rendezvous over gloo and stop)"""
    import json
    p = _run_bench({"AZ_BENCH_DRYRUN": "1"}, 4)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["sum_of_ranks_plus_one"] == 10.0


def test_bench_launcher_fails_when_a_rank_fails():
    p = _run_bench({"AZ_BENCH_DRYRUN": "fail:2"}, 3)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "rank 2 exited with 7" in p.stderr


def test_bench_launcher_with_eight_ranks():
    import json
    p = _run_bench({"AZ_BENCH_DRYRUN": "1"}, 8)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["sum_of_ranks_plus_one"] == 36.0


def test_bench_launcher_kills_ranks_that_ignore_sigterm():
    """after a rank fails the others get terminate() and, past the grace period, kill(): a rank stuck in a device wait
    (here: one that ignores SIGTERM and sleeps) must not keep the launcher -- or its GPU -- forever"""
    import time
    t0 = time.time()
    p = _run_bench({"AZ_BENCH_DRYRUN": "hang:1", "AZ_BENCH_KILL_GRACE_S": "2"}, 3)
    assert p.returncode != 0 and time.time() - t0 < 120
    assert "rank 1 exited with 7" in p.stderr


def test_bench_launcher_stops_its_ranks_when_it_is_terminated():
    """SIGTERM to the launcher (a harness time limit): the ranks it started are gone afterwards"""
    import signal
    import subprocess
    import time
    env = dict(os.environ, AZ_BENCH_DRYRUN="hang:99", AZ_BENCH_KILL_GRACE_S="2")  # no rank exits by itself: all sleep
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    kids = []
    for _ in range(300):
        try:
            kids = [int(x) for x in open(f"/proc/{p.pid}/task/{p.pid}/children").read().split()]
        except OSError:
            kids = []
        if len(kids) == 2:
            break
        time.sleep(0.1)
    assert len(kids) == 2
    time.sleep(5)  # let the ranks reach their sleep (SIGTERM ignored from then on)
    p.send_signal(signal.SIGTERM)
    p.communicate(timeout=120)
    assert p.returncode == 130
    time.sleep(0.5)
    assert not any(os.path.exists(f"/proc/{k}") and "bench.py" in open(f"/proc/{k}/cmdline").read() for k in kids)


def test_bench_under_torch_distributed_run():
    """the driver's way of starting N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- bench.py is then ONE rank (WORLD_SIZE is set: no self-launch); dry run over gloo"""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, AZ_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["sum_of_ranks_plus_one"] == 6.0
