"""RCCL on the one GPU a lease has: a world-size-1 `nccl` process group (backend "nccl" IS RCCL on ROCm), created with
device_id as bench.py and a multi-GPU trainer job create it, with every collective of alphazero_amd/dist.py FORCED through
RCCL on real engine output -- the int8 / int32 / float32 sample fields as zero-copy views of the engine's buffers
(__cuda_array_interface__), the packed all_gather_into_tensor, the flat weight broadcast, the arena result gather and
the bench's MAX all-reduce + barrier.  No scaling claim: it proves that the code path RCCL sees on 8 GPUs initialises
and moves the right bytes on this ROCm.  Runs in a child process: the process group
does not leak into the test session."""
import os
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from alphazero_amd import engine as E
    from alphazero_amd.dist import all_gather_rows, all_gather_samples, broadcast_state_dict, rank_world
    from alphazero_amd.games.othello import OthelloNet
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
    res = {"backend": dist.get_backend(), "rank_world": rank_world()}
    eng = E.SelfPlayEngine(0, 6, 6, n_slots=32, n_sim=10, evaluator=E.EVAL_FAKE, seed=1)
    eng.run(32)
    views = eng.samples(copy=False)  # zero-copy views of the engine's HBM buffers
    local = {k: v.clone() for k, v in views.items()}
    got = all_gather_samples(views, force=True)
    res["samples_equal"] = all(torch.equal(got[k], local[k]) and got[k].dtype == local[k].dtype and got[k].is_cuda for k in local)
    res["n_samples"] = int(got["z"].shape[0])
    res["passthrough"] = all_gather_samples(views) is views  # world 1 without force: no collective
    torch.manual_seed(5)
    net = OthelloNet(n=6).cuda()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    broadcast_state_dict(net, src=0, force=True)
    res["weights_equal"] = all(torch.equal(v, before[k]) for k, v in net.state_dict().items())
    rows = torch.arange(24, dtype=torch.int32, device="cuda").view(12, 2)
    res["rows_equal"] = bool(torch.equal(all_gather_rows(rows, force=True), rows))
    t = torch.tensor([3.25], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    res["allreduce"] = float(t.item())
    # the sharded batched arena inside a job: one rank plays all rounds and the result gather still runs
    from alphazero_amd.arena import BatchedArena
    stats = BatchedArena("othello", net.eval(), opponent="random", n_sim=4, seed=2, board_size=6).play_games(6)
    res["arena_games"] = len(stats["player1"]) + len(stats["player2"]) + stats["draw"]
    out.update(res)
    dist.destroy_process_group()


def test_rccl_world1_forced_collectives_on_engine_output():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    p = ctx.Process(target=_worker, args=(29611, out))
    p.start()
    p.join(420)
    assert p.exitcode == 0, f"RCCL worker exit code {p.exitcode}"
    assert out["backend"] == "nccl" and out["rank_world"] == (0, 1)
    assert out["samples_equal"] and out["n_samples"] > 32 * 20 and out["passthrough"]
    assert out["weights_equal"] and out["rows_equal"] and out["allreduce"] == 3.25 and out["arena_games"] == 6
