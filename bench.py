#!/usr/bin/env python3
"""Headline benchmark: self-play games/s, Othello 8x8 @ 100 sims/move (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one self-play wave: every rank keeps `--games` Othello 8x8 games resident and plays them from the start
position to the end at 100 MCTS simulations per move through the HIP engine (random-init OthelloNet(n=8) under
torch.manual_seed(0), Dirichlet noise 0.03/0.25, tau linear(4,4), tree reuse) and, for N > 1, all-gathers the samples over
RCCL.  value = games of all ranks / time.  The timed region runs the engine as it ships (searches replayed as HIP
graphs, no event recording); the per-kernel times of the roofline come from one extra, separately profiled step.

At N = 1 the same JSON line also carries the two single-GPU BASELINE configs at their LITERAL sizes:
  config2 : Othello 8x8, 4096 concurrent games, 100 sims/move           (BASELINE.json configs[1])
  config4 : Connect4 6x7, 8192 concurrent games, 200 sims/move          (BASELINE.json configs[3])
each with games/s, examples/s and its own roofline object, and `cpu_baseline`: the CPU oracle (C port of the
reference's self-play loop) timed on the host cores with the protocol of BASELINE.md section 3.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver (before any HIP call)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
PROF_NAMES = ["k_trunk2", "k_gemm fc1", "k_gemm fc2", "k_heads"]
# the Python reference and the C oracle on ONE core of the build container (Xeon @ 2.10 GHz, 8 vCPU): BASELINE.md section 2
# (8.38 s per Othello 8x8 game at 100 sims) and `python oracle/selfplay_worker.py` there (4 games: 6.33 +- 0.73 s per game)
REF_S_PER_GAME_BUILD_CONTAINER = 8.38
ORACLE_S_PER_GAME_BUILD_CONTAINER = 6.33


def stage_flops(CH, CW, F1, F2, A):
    """algorithmic FLOPs (2 x MAC) per board of the four network stages"""
    p1, p3, p4 = CH * CW, (CH - 2) * (CW - 2), (CH - 4) * (CW - 4)
    conv = 2 * (p1 * 9 * 32 + p1 * 9 * 32 * 32 + p3 * 9 * 32 * 32 + p4 * 9 * 32 * 32)
    return [conv, 2 * 32 * p4 * F1, 2 * F1 * F2, 2 * F2 * (A + 1)]


def executed_conv_flops(CH, CW):
    """FLOPs the trunk kernel EXECUTES per board on the matrix cores: conv2 of 8x8 and 7x6 planes runs in the Winograd F(2x2,3x3)
    form (tiles x 16 frequencies x 32 x 32 MACs instead of positions x 9 x 32 x 32; AZ_WINOGRAD=0 switches it off) -- the SURVEY's
    algorithmic figure stays the numerator of `frac`, this one is reported beside it"""
    p1, p3, p4 = CH * CW, (CH - 2) * (CW - 2), (CH - 4) * (CW - 4)
    wino = (CH, CW) in ((8, 8), (7, 6)) and os.environ.get("AZ_WINOGRAD", "") != "0"
    conv2 = ((CH + 1) // 2) * ((CW + 1) // 2) * 16 * 32 * 32 if wino else p1 * 9 * 32 * 32
    return 2 * (p1 * 9 * 32 + conv2 + p3 * 9 * 32 * 32 + p4 * 9 * 32 * 32), wino


def algorithmic_bytes(CH, CW, F1, F2, A):
    """algorithmic HBM bytes per board of the four network stages (inputs + outputs; weights stay cache resident)"""
    fin = 32 * (CH - 4) * (CW - 4)
    return [4 * (CH * CW + fin), 4 * (fin + F1), 4 * (F1 + F2), 4 * (F2 + A + 1)]


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(state_dict, n_sim=100, games_per_proc=3):
    """BASELINE.md section 3 on this box's host cores, with the CPU oracle (the checker; C port of the reference's loop):
    one process, then one process per core (<= 16), >= 3 games each, every game timed (SelfPlayTimer idiom,
    timers.py:53-76); then the Arena path (AlphaZeroPlayer @100 vs rollout MCTSPlayer @100, 2 rounds, arena.py:119-185).
    Runs BEFORE this process touches the GPU: the workers are plain child processes that never see HIP."""
    from oracle import oracle as O
    from oracle import selfplay_worker as W
    weights = {k: v.cpu().numpy() for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
    t_all = time.perf_counter()
    one = W.play(weights, games_per_proc, n_sim, 0)
    s1 = np.array(one["seconds_per_game"])
    nproc = max(1, min(os.cpu_count() or 1, 16))
    many = None
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "w.npz")
        np.savez(path, **weights)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "selfplay_worker.py"), path, str(games_per_proc),
                                   str(n_sim), str(1000 * (i + 1))], stdout=subprocess.PIPE, text=True,
                                  env=dict(os.environ, OMP_NUM_THREADS="1")) for i in range(nproc)]
        outs = [p.communicate()[0] for p in procs]
        wall = time.perf_counter() - t0
        if all(p.returncode == 0 for p in procs):
            per = np.concatenate([json.loads(o.strip().splitlines()[-1])["seconds_per_game"] for o in outs])
            plies = sum(json.loads(o.strip().splitlines()[-1])["plies"] for o in outs)
            many = {"processes": nproc, "games": int(len(per)), "wall_s": wall, "games_per_sec": len(per) / wall, "examples_per_sec": plies / wall,
                    "s_per_game_mean": float(per.mean()), "s_per_game_std": float(per.std())}
    net = O.ConvNet(O.OTHELLO, 8, 8, weights)
    t0 = time.perf_counter()
    moves, winners, _, _ = O.arena_games((O.OTHELLO, 8, 8), ("conv", net), n_sim, "mcts", n_sim, seed=0, n_rounds=2)
    t_arena = time.perf_counter() - t0
    return {"value": 1.0 / float(s1.mean()), "unit": "games/s", "cores": 1, "kind": "port",
            "sample": f"{games_per_proc} full Othello 8x8 self-play games at {n_sim} sims/move ({one['plies']} plies, {one['net_evals']} net evals) "
                      f"on 1 host core, oracle/liboracle.so; then {nproc} processes x {games_per_proc} games; then 2 Arena rounds",
            "cpu_model": cpu_model(), "nproc": os.cpu_count(),
            "one_process": {"games": games_per_proc, "s_per_game_mean": float(s1.mean()), "s_per_game_std": float(s1.std()),
                            "games_per_sec": 1.0 / float(s1.mean()), "examples_per_sec": one["plies"] / float(s1.sum())},
            "all_cores": many,
            "arena": {"rounds": 2, "player1": f"AlphaZeroPlayer({n_sim} sims)", "player2": f"MCTSPlayer(rollout, {n_sim} sims)",
                      "seconds": t_arena, "s_per_game": t_arena / 2, "plies": [len(m) for m in moves], "winners": winners},
            "reference_python": {"s_per_game": REF_S_PER_GAME_BUILD_CONTAINER, "games_per_sec": 1.0 / REF_S_PER_GAME_BUILD_CONTAINER,
                                 "where": "build container, Xeon @ 2.10 GHz, 1 core (BASELINE.md section 2)",
                                 "oracle_s_per_game_same_core": ORACLE_S_PER_GAME_BUILD_CONTAINER,
                                 "oracle_speedup_over_reference_same_core": REF_S_PER_GAME_BUILD_CONTAINER / ORACLE_S_PER_GAME_BUILD_CONTAINER},
            "seconds": time.perf_counter() - t_all}


# ------------------------------------------------------------------------------------------------ one workload
class Workload:
    """one BASELINE config on this rank's GPU: engine + network + the bookkeeping of the roofline"""

    def __init__(self, name, game, G, sims, seed=0, waves=1):
        from alphazero_amd import engine as E
        from alphazero_amd.games.connect4 import Connect4Net
        from alphazero_amd.games.othello import OthelloNet
        self.name, self.game, self.G, self.sims = name, game, G, sims
        self.per_step = waves * G  # games per step: G stay resident, a finished game's slot is refilled until per_step are started
        torch.manual_seed(0)
        if game == "othello":
            self.gid, self.H, self.W, self.model = 0, 8, 8, OthelloNet(n=8).eval()
            self.geom = (8, 8, 1024, 512, 65)
            self.desc = f"Othello 8x8, {G} concurrent self-play games per GPU, {sims} sims/move, random-init OthelloNet(n=8) seed 0"
        else:
            self.gid, self.H, self.W, self.model = 1, 6, 7, Connect4Net(7, 6).eval()
            self.geom = (7, 6, 64, 32, 7)  # the (6,7) grid is view-ed as a 7x6 plane (connect4.py:399)
            self.desc = f"Connect4 6x7, {G} concurrent self-play games per GPU, {sims} sims/move, random-init Connect4Net(7,6) seed 0"
        self.desc += ", Dirichlet 0.03/0.25, tau linear(4,4), tree reuse"
        if waves > 1:
            self.desc += f"; a step plays {waves} x {G} games through the {G} resident slots (finished slots are refilled: steady state)"
        self.hnet = self.model.to_hip(max_batch=G)
        plies = 128 if game == "othello" else 43
        self.eng = E.SelfPlayEngine(self.gid, self.H, self.W, n_slots=G, n_sim=sims, net=self.hnet, dirichlet_alpha=0.03,
                                    dirichlet_epsilon=0.25, temp_max_step=4, temp_min_step=4, tie_mode=E.TIE_RANDOM,
                                    noise_mode=E.NOISE_PHILOX, seed=seed, max_plies=plies,
                                    sample_capacity=self.per_step * (72 if game == "othello" else 43))

    def wave(self, first_game_id):
        """one step on this rank: per_step games to the end; returns (samples dict of device views, engine stats)"""
        self.eng.run(self.per_step, first_game_id=first_game_id)
        return self.eng.samples(copy=False), self.eng.stats()

    def profiled_wave(self, first_game_id):
        """one more step with HIP events around every network kernel launch (az_net_profile; kernel-by-kernel launches
        instead of graph replays).  -> roofline object of the dominant kernel + the step's time split"""
        self.hnet.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, st = self.wave(first_game_id)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = self.hnet.profile_read()
        ov = self.hnet.profile_overhead_ms()  # an event-to-event interval costs this much with no kernel in it
        self.hnet.profile(False)
        fl, by = stage_flops(*self.geom), algorithmic_bytes(*self.geom)
        if self.game == "connect4":  # fc1 + fc2 + heads are one fused launch, booked in the fc1 slot
            fl = [fl[0], fl[1] + fl[2] + fl[3], 0, 0]
        evals = st["net_evals"]
        raw_ms = [prof[k][0] for k in PROF_NAMES]
        raw_ms[0] += prof["k_trunk"][0]  # small-batch launches of the one-board-per-wave trunk kernel
        launches = [prof[k][1] for k in PROF_NAMES]
        launches[0] += prof["k_trunk"][1]
        tot_ms = [max(raw_ms[i] - ov * launches[i], 1e-9) for i in range(4)]  # kernel time: what rocprofv3 reports per dispatch
        dom = int(np.argmax(tot_ms))
        ach = fl[dom] * evals / (tot_ms[dom] * 1e-3) / 1e12
        key = f"{self.game}_{self.G}"
        traffic, counters = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes per full-batch launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        if os.path.exists(tfile):
            t = json.load(open(tfile)).get(key)
            if t:
                traffic = t.get(["k_trunk", "k_gemm_fc1", "k_gemm_fc2", "k_heads"][dom])
        cfile = os.path.join(ROOT, "profiles", "mfma_counters.json")  # SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES passes
        if os.path.exists(cfile):
            counters = json.load(open(cfile)).get(key)
        full = [self.hnet.time_stage(s, self.G, iters=20) for s in range(4)]
        net_ms = sum(tot_ms)
        roof = {"bound": "mfma", "kernel": PROF_NAMES[dom] + (" (+k_trunk at small batches)" if dom == 0 and prof["k_trunk"][1] else ""),
                "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                "traffic": traffic,
                "traffic_is_for": f"one launch at the full batch of {self.G} boards (tools/prof_net.py under rocprofv3 --pmc); algorithmic bytes of that launch: {self.G * by[dom]}",
                "mfma_busy": (counters or {}).get(["k_trunk", "k_gemm_fc1", "k_gemm_fc2", "k_heads"][dom]),
                # the kernel's own launches (what rocprofv3 lists under its name); for the trunk, `achieved` also covers the few
                # small-batch launches of k_trunk at the end of a wave (their boards and their time)
                "launches": prof[PROF_NAMES[dom]][1],
                "avg_launch_ms": max(prof[PROF_NAMES[dom]][0] - ov * prof[PROF_NAMES[dom]][1], 0.0) / max(1, prof[PROF_NAMES[dom]][1]),
                "avg_launch_ms_with_event_overhead": prof[PROF_NAMES[dom]][0] / max(1, prof[PROF_NAMES[dom]][1]), "event_overhead_ms_per_interval": ov,
                "small_batch_trunk_launches": prof["k_trunk"][1] if dom == 0 else 0,
                "achieved_uncorrected": fl[dom] * evals / (raw_ms[dom] * 1e-3) / 1e12,
                "avg_boards_per_launch": evals / max(1, launches[dom]),
                "algorithmic_flops_per_board": fl[dom], "boards_evaluated": evals,
                **({"executed_flops_per_board": executed_conv_flops(*self.geom[:2])[0],
                    "achieved_executed": executed_conv_flops(*self.geom[:2])[0] * evals / (tot_ms[0] * 1e-3) / 1e12,
                    "frac_executed": executed_conv_flops(*self.geom[:2])[0] * evals / (tot_ms[0] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "conv2_form": "Winograd F(2x2,3x3): `frac` counts the SURVEY's algorithmic FLOPs of the direct form, `frac_executed` "
                                  "the multiplications the kernel issues" if executed_conv_flops(*self.geom[:2])[1] else "direct"} if dom == 0 else {}),
                "measured_on": "one separately profiled step after the timed region: HIP events on the engine's stream around every launch, "
                               "minus the calibrated cost of an empty event interval per launch",
                "profiled_step_ms": 1e3 * dt,
                "profiled_step_kernel_ms": {k: prof[k][0] for k in prof}, "profiled_step_launches": {k: prof[k][1] for k in prof},
                "stage_tflops": {PROF_NAMES[i]: (fl[i] * evals / (tot_ms[i] * 1e-3) / 1e12 if tot_ms[i] > 0.05 * raw_ms[i] else None) for i in range(4)},
                "full_batch_launch_ms": dict(zip(PROF_NAMES, full)),
                "full_batch_tflops": {PROF_NAMES[i]: fl[i] * self.G / (full[i] * 1e-3) / 1e12 for i in range(4)},
                "forward_tflops": sum(fl) * evals / (net_ms * 1e-3) / 1e12,
                "network_share_of_profiled_step": sum(raw_ms) / (1e3 * dt),
                "tree_and_host_share_of_profiled_step": 1.0 - sum(raw_ms) / (1e3 * dt)}
        if self.game == "connect4":
            roof["note"] = "Connect4Net: fc1 + fc2 + heads run as ONE fused launch (k_tail_small), booked under 'k_gemm fc1'"
        return roof

    def close(self):
        self.eng.close()
        self.hnet.close()


def run_single(name, game, G, sims, steps, warmup, waves):
    """a BASELINE config at its literal size on one GPU (G concurrent games; SURVEY 8d: "run >= 2 waves and report the
    steady-state rate"): `steps` timed steps of waves x G games (graphs on, no profiling) + one profiled step"""
    w = Workload(name, game, G, sims, waves=waves)
    G = w.per_step  # game ids advance by the games of a step
    for i in range(warmup):
        w.wave(i * G)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_samples, evals = 0, 0
    for k in range(steps):
        smp, st = w.wave((warmup + k) * G)
        n_samples += smp["z"].shape[0]
        evals += st["net_evals"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = w.eng.stats()
    games = steps * G
    out = {"workload": w.desc, "concurrent_games": w.G, "games_per_step": G, "value": games / dt, "unit": "games/s", "examples_per_sec": n_samples / dt, "sims_per_sec": n_samples * sims / dt,
           "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "plies_per_game": n_samples / games,
           "net_evals_per_step": evals / steps, "lockstep_iters_last_step": st["lockstep_iters"], "graph_replays": st["graph_replays"],
           "max_tree_nodes_per_game": st["max_nodes_used"], "dtype": "f32"}
    out["roofline"] = w.profiled_wave((warmup + steps) * G)
    w.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=32768, help="concurrent games (engine slots) per GPU of the headline run")
    ap.add_argument("--waves", type=int, default=1, help="games per step per GPU = waves x games (finished slots are refilled)")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-literal-configs", action="store_true", help="skip the config2 / config4 objects (N = 1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    cpu = None
    if world == 1 and not args.no_cpu_baseline:  # before HIP is initialised: the worker processes are forked from a GPU-free parent
        from alphazero_amd.games.othello import OthelloNet
        torch.manual_seed(0)
        cpu = cpu_baseline(OthelloNet(n=8).eval().state_dict())

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

    from alphazero_amd.dist import all_gather_samples, rank_game_range

    w = Workload("headline", "othello", args.games, args.sims, waves=args.waves)
    G = w.per_step
    gather_s = [0.0]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def step(wave):
        first, cnt = rank_game_range(rank, world, G, wave)
        smp, st = w.wave(first)
        if world > 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            smp = all_gather_samples({k: smp[k] for k in ("state", "pi", "z", "meta")})
            torch.cuda.synchronize()
            gather_s[0] += time.perf_counter() - t0
        return smp["z"].shape[0], st["net_evals"]

    for i in range(args.warmup):
        step(i)
    gather_s[0] = 0.0
    sync()
    t0 = time.perf_counter()
    n_samples_total, evals_total = 0, 0
    for k in range(args.steps):
        s, e = step(args.warmup + k)
        n_samples_total += s
        evals_total += e
    t_local = time.perf_counter() - t0  # this rank's own time, before it waits for the others
    sync()
    dt = time.perf_counter() - t0
    dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = torch.tensor([t_local, gather_s[0]], dtype=torch.float64, device=dev)
    per_rank_all = per_rank.clone().view(1, 2)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_rank_all = torch.zeros(world * 2, dtype=torch.float64, device=dev)  # flat: the one layout every backend concatenates alike
        dist.all_gather_into_tensor(per_rank_all, per_rank)
    dt = float(t.item())
    per_rank_all = per_rank_all.view(-1, 2).cpu().numpy()
    st = w.eng.stats()

    if rank == 0:
        games = args.steps * G * world
        samples = n_samples_total  # after the all-gather every rank holds all ranks' samples
        out = {
            "metric": "self-play games/sec (whole node), Othello 8x8 @100 sims/move", "value": games / dt, "unit": "games/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w.desc, "games_per_gpu_per_step": G, "concurrent_games_per_gpu": w.G, "sims_per_move": args.sims,
                       "parallelism": f"game-sharded x{world}", "timed_region": "engine as shipped: searches replayed as HIP graphs, no event recording"},
            "examples_per_sec": samples / dt, "sims_per_sec": samples * args.sims / dt,
            "plies_per_game": samples / games, "net_evals_last_step": st["net_evals"], "lockstep_iters_last_step": st["lockstep_iters"],
            "graph_replays": st["graph_replays"], "max_tree_nodes_per_game": st["max_nodes_used"],
            # diagnosis of a scaling run: every rank's own step time (before waiting for the others) and its share spent in the RCCL gather
            "per_rank_ms_per_step": [1e3 * float(x) / args.steps for x in per_rank_all[:, 0]],
            "per_rank_gather_ms_per_step": [1e3 * float(x) / args.steps for x in per_rank_all[:, 1]],
        }
    if world > 1:
        dist.barrier()
    # the roofline's per-kernel times: one extra step, profiled, outside the timed region (rank 0; the others idle at the barrier)
    roof = w.profiled_wave((args.warmup + args.steps) * G * world + rank * G) if rank == 0 else None
    w.close()
    if rank == 0:
        out["roofline"] = roof
        sims_per_gpu = samples * args.sims / dt / world
        tree = {"algorithmic_bytes_per_sim": 919, "achieved": sims_per_gpu * 919 / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": sims_per_gpu * 919 / 8e12, "note": "per GPU, whole path: the tree kernels are a few % of a step, the path is bound by the network's MFMA work"}
        kfile = os.path.join(ROOT, "profiles", "kstep_counters.json")  # k_step<true,true>: duration + FETCH_SIZE / WRITE_SIZE passes (tools/refresh_profiles.sh)
        if os.path.exists(kfile):
            tree["k_step"] = json.load(open(kfile)).get(f"othello_{w.G}")
        out["tree_hbm"] = tree
        if world == 1 and not args.no_literal_configs:
            # Othello games all last 60-65 plies: one synchronised wave per step keeps 92 % of the leaf rows filled (the rest are
            # terminal leaves, which need no evaluation).  Connect4 games last 18-42 plies: finished slots are refilled and a
            # step plays 8 x 8192 games, so that the drain at the end of a step (its length is one game) is amortised
            out["config2"] = run_single("config2", "othello", 4096, 100, steps=3, warmup=1, waves=1)
            out["config4"] = run_single("config4", "connect4", 8192, 200, steps=2, warmup=1, waves=8)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
