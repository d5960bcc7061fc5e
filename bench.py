#!/usr/bin/env python3
"""Headline benchmark: self-play games/s, Othello 8x8 @ 100 sims/move (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is `--waves` (1) self-play wave(s): every rank keeps `--games` (32768) Othello 8x8 games resident and
plays waves x games of them from the start position to the end at 100 MCTS simulations per move through the
HIP engine (finished slots are refilled at once)
(random-init OthelloNet(n=8) under torch.manual_seed(0), Dirichlet noise 0.03/0.25, tau linear(4,4),
tree reuse) and, for N > 1, all-gathers the samples over RCCL.  value = games of all ranks / time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver (before any HIP call)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
STAGE_NAMES = ["k_trunk2 (conv1-4)", "k_gemm fc1", "k_gemm fc2", "k_heads"]


def stage_flops(n):
    conv = 2 * (n * n * 9 * 32 + n * n * 9 * 32 * 32 + (n - 2) ** 2 * 9 * 32 * 32 + (n - 4) ** 2 * 9 * 32 * 32)
    fin = 32 * (n - 4) ** 2
    return [conv, 2 * fin * 1024, 2 * 1024 * 512, 2 * 512 * (n * n + 2)]


def cpu_baseline(state_dict, n_games=6, n_sim=100):
    """the CPU oracle (C port of the reference's self-play loop) on one host core, same weights/config"""
    from oracle import oracle as O
    net = O.ConvNet(O.OTHELLO, 8, 8, {k: v.cpu().numpy() for k, v in state_dict.items() if not k.endswith("num_batches_tracked")})
    t0 = time.perf_counter()
    r = O.selfplay(O.OTHELLO, 8, 8, n_games, n_sim, ("conv", net), seed=0)
    dt = time.perf_counter() - t0
    return {"value": n_games / dt, "unit": "games/s", "cores": 1, "kind": "port",
            "sample": f"{n_games} full Othello 8x8 self-play games at {n_sim} sims/move ({len(r['z'])} plies, "
                      f"{r['n_evals']} net evals) on 1 host core, oracle/liboracle.so",
            "examples_per_sec": len(r["z"]) / dt, "seconds": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=32768, help="concurrent games (engine slots) per GPU")
    ap.add_argument("--waves", type=int, default=1, help="games per step per GPU = waves x games; finished slots are refilled, "
                                                           "so the ragged end of a wave (games last 60-65 plies) overlaps the next")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # AZ_BENCH_BACKEND=gloo + AZ_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 control flow on a one-GPU box
    backend = os.environ.get("AZ_BENCH_BACKEND", "nccl")
    if os.environ.get("AZ_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from alphazero_amd import engine as E
    from alphazero_amd.dist import all_gather_samples, rank_game_range
    from alphazero_amd.games.othello import OthelloNet

    n, G = 8, args.games
    torch.manual_seed(0)
    model = OthelloNet(n=n).eval()
    hnet = model.to_hip(max_batch=G)
    eng = E.SelfPlayEngine(0, n, n, n_slots=G, n_sim=args.sims, net=hnet, dirichlet_alpha=0.03, dirichlet_epsilon=0.25,
                           temp_max_step=4, temp_min_step=4, tie_mode=E.TIE_RANDOM, noise_mode=E.NOISE_PHILOX,
                           seed=0, max_plies=128, sample_capacity=args.waves * G * 72)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def step(wave):
        first, cnt = rank_game_range(rank, world, args.waves * G, wave)
        eng.run(cnt, first_game_id=first)
        smp = eng.samples(copy=False)
        if world > 1:
            smp = all_gather_samples({k: smp[k] for k in ("state", "pi", "z", "meta")})
        return smp["z"].shape[0]

    for w in range(args.warmup):
        step(w)
    if rank == 0:
        hnet.profile(True)  # HIP events around every network kernel launch of the timed region (engine's stream)
    sync()
    t0 = time.perf_counter()
    n_samples_total, evals_total = 0, 0
    for k in range(args.steps):
        n_samples_total += step(args.warmup + k)
        evals_total += eng.stats()["net_evals"]
    sync()
    dt = time.perf_counter() - t0
    prof = hnet.profile_read() if rank == 0 else None
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    st = eng.stats()

    if rank == 0:
        games = args.steps * args.waves * G * world
        samples = n_samples_total  # after the all-gather every rank holds all ranks' samples
        out = {
            "metric": "self-play games/sec (whole node), Othello 8x8 @100 sims/move", "value": games / dt, "unit": "games/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Othello 8x8, {G} concurrent self-play games per GPU, {args.sims} sims/move, "
                                   f"random-init OthelloNet(n=8) seed 0, Dirichlet 0.03/0.25, tau linear(4,4), tree reuse",
                       "games_per_gpu_per_step": args.waves * G, "concurrent_games_per_gpu": G, "sims_per_move": args.sims,
                       "parallelism": f"game-sharded x{world}"},
            "examples_per_sec": samples / dt, "sims_per_sec": samples * args.sims / dt,
            "plies_per_game": samples / games, "net_evals_last_step": st["net_evals"], "lockstep_iters_last_step": st["lockstep_iters"],
            "max_tree_nodes_per_game": st["max_nodes_used"],
        }
        # roofline of the dominant kernel, measured live over the timed region: algorithmic FLOPs of the boards the
        # network evaluated there / the time its launches took (HIP events on the engine's stream, az_net_profile)
        fl = stage_flops(n)
        names = ["k_trunk2", "k_gemm fc1", "k_gemm fc2", "k_heads"]
        tot_ms = [prof[k][0] for k in names]
        tot_ms[0] += prof["k_trunk"][0]  # the few small-batch launches of the one-board-per-wave trunk kernel
        dom = int(np.argmax(tot_ms))
        ach = fl[dom] * evals_total / (tot_ms[dom] * 1e-3) / 1e12
        kname = names[dom]
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes per full-batch launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        if os.path.exists(tfile) and json.load(open(tfile)).get("batch") == G:
            traffic = json.load(open(tfile)).get(["k_trunk", "k_gemm_fc1", "k_gemm_fc2", "k_heads"][dom])
        full = [hnet.time_stage(s, G, iters=20) for s in range(4)]  # context: one launch at the full batch
        out["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                           "traffic_is_for": f"one launch at the full batch of {G} boards (PMC passes run tools/prof_net.py); "
                                             f"algorithmic bytes of that launch: {G * (n * n * 4 + 32 * (n - 4) ** 2 * 4)}",
                           "launches": prof[kname][1], "avg_launch_ms": prof[kname][0] / max(1, prof[kname][1]),
                           "avg_boards_per_launch": evals_total / max(1, prof[kname][1] + (prof["k_trunk"][1] if dom == 0 else 0)),
                           "algorithmic_flops_per_board": fl[dom], "boards_evaluated": evals_total,
                           "timed_region_ms": {k: prof[k][0] for k in prof}, "timed_region_launches": {k: prof[k][1] for k in prof},
                           "full_batch_launch_ms": dict(zip(STAGE_NAMES, full)),
                           "full_batch_tflops": {STAGE_NAMES[i]: fl[i] * G / (full[i] * 1e-3) / 1e12 for i in range(4)},
                           "forward_tflops": sum(fl) * evals_total / (sum(tot_ms) * 1e-3) / 1e12}
        # the tree kernels' side of SURVEY 8d: algorithmic HBM bytes of select / expand / backup per simulation
        sims_per_gpu = samples * args.sims / dt / world
        out["tree_hbm"] = {"algorithmic_bytes_per_sim": 919, "achieved": sims_per_gpu * 919 / 1e9, "peak": 8000.0, "unit": "GB/s",
                           "frac": sims_per_gpu * 919 / 8e12,
                           "note": "per GPU; k_step takes 4-5 % of a step, the path is bound by the network's MFMA work"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model.state_dict())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
