#!/usr/bin/env python3
"""Headline benchmark: self-play games/s, Othello 8x8 @ 100 sims/move, 4096 concurrent games per GPU (BASELINE.json configs[1];
at N GPUs: configs[2] = the same per GPU, sharded, + the RCCL all-gather of the samples).

  python bench.py --gpus N --steps K --warmup W
Started plainly with N > 1 it launches its own N ranks (one child process per GPU, spawned BEFORE the parent makes any HIP call; the
parent only relays rank 0's JSON line and fails if any rank fails); under torch.distributed.run (WORLD_SIZE set) it is one rank.

A "step" is one self-play wave at BASELINE's LITERAL size: every rank keeps `--games` (4096) Othello 8x8 games resident and plays them
from the start position to the end at 100 MCTS simulations per move through the HIP engine (random-init OthelloNet(n=8) under
torch.manual_seed(0), Dirichlet noise 0.03/0.25, tau linear(4,4), tree reuse) and, for N > 1, all-gathers the samples over RCCL.
value = games of all ranks / time; K steps = K waves (">= 2 waves, steady-state rate").  The timed region runs the engine as
it ships (searches replayed as HIP graphs, no event recording); the per-kernel times of the roofline come from one extra, separately
profiled step in which every network launch carries its own start / stop events (the dispatch's begin / end timestamps).

The LAST line of stdout is the line of record and holds scalars only (`record_line`: the contract's top-level fields + flat `config`,
`roofline`, `cpu_baseline`; < 6 KB, strict JSON, printed ONCE, nothing on stdout after it).  Every nested object -- `saturated`, `config4`,
`config5`, `config1`, `config3`, `latency`, `tree_hbm`, the per-kernel tables of the roofline, the opt-in `dense_i8_prototype` -- goes to
bench_detail.json beside this file (and gpurun_out/bench_detail.json); the line names it in `detail`.  Flat copies in the line:
  config.saturated_*   the same workload at 32768 concurrent games per GPU (where the rate has flattened: the chip is full)
  config.config4_*     BASELINE.json configs[3]: Connect4 6x7, 8192 concurrent games, 200 sims/move, at its literal size
  config.config5_10_epochs_*  BASELINE.json configs[4] at the reference's own hyper-parameters (10 epochs of batch 64)
  config.config3_*     N > 1 and 32768 / N != 4096: configs[2] at its literal TOTAL (32768 games sharded N ways)
Order of the run: CPU baseline (before HIP is touched) -> headline (warm-up, timed steps, one profiled step) -> saturated -> config4 ->
config5 (ten-epoch variant) -> THE LINE -> config1, the other config5 variants, latency (detail file only).  On one GPU a failing optional
leg is recorded in `errors` and never takes the line down.

How to read `roofline` (every field can be recomputed from profiles/ + the fields beside it):
  kernel            the kernel with the largest share of the profiled step's network time; fc1 and fc2 are ONE kernel (two launches
                    per forward), the trunk is k_trunk2 (+ the few small-batch launches of k_trunk at the end of a wave)
  achieved          flops_per_board x boards_evaluated / kernel_ms_total                     [TFLOP/s]
                    dense layers: flops_per_board = 2 (FIN F1 + F1 F2) algorithmic = executed;
                    trunk: the FLOPs the kernel ISSUES (conv2 in the Winograd form issues fewer multiplications than the direct
                    form) -- the direct-form figure is in frac_algorithmic_direct_form
  frac              achieved / peak (157.3 TFLOP/s, f32-input MFMA): how busy the matrix pipe is, never above 1
  avg_launch_ms     time / launches of the kernel the stage is named after (k_gemm, k_trunk2, ...): what rocprofv3 --stats lists as that
                    kernel's average duration; the last plies of a wave run on the small-batch kernels (k_dense_frag, k_trunk, k_trunk_q),
                    whose time is in kernel_ms_total (all boards went through one or the other) and whose launches are booked apart
  forward_frac      all four stages: algorithmic FLOPs of a forward x boards / network time / peak
  end_to_end_frac   boards the TIMED region evaluated x algorithmic FLOPs of a forward / timed seconds / peak (per GPU)
  traffic, mfma_busy  PMC figures of one full-batch launch (profiles/traffic.json, mfma_counters.json), used only when the files'
                    `csrc_sha` equals the hash of the sources this build was made from (alphazero_amd._lib.csrc_tree_hash), else null
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver (before any HIP call)

import numpy as np  # noqa: E402

torch = dist = None  # imported by _imports(): the self-launching parent never loads torch, let alone HIP


def _imports():
    global torch, dist
    import torch as _t
    import torch.distributed as _d
    torch, dist = _t, _d


PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
# az_net_profile_read's counters, one per kernel family (the small-batch kernels serve the last plies of a wave, when few games are left)
SLOTS = ["k_trunk2", "k_gemm fc1", "k_gemm fc2", "k_heads", "k_trunk", "small fc1", "small fc2", "k_trunk_q"]
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
    """FLOPs the trunk kernel ISSUES per board on the matrix cores: conv2 of 8x8 and 7x6 planes runs in the Winograd F(2x2,3x3)
    form (tiles x 16 frequencies x 32 x 32 MACs instead of positions x 9 x 32 x 32; AZ_WINOGRAD=0 switches it off)"""
    p1, p3, p4 = CH * CW, (CH - 2) * (CW - 2), (CH - 4) * (CW - 4)
    wino = (CH, CW) in ((8, 8), (7, 6)) and os.environ.get("AZ_WINOGRAD", "") != "0"
    conv2 = ((CH + 1) // 2) * ((CW + 1) // 2) * 16 * 32 * 32 if wino else p1 * 9 * 32 * 32
    return 2 * (p1 * 9 * 32 + conv2 + p3 * 9 * 32 * 32 + p4 * 9 * 32 * 32), wino


def algorithmic_bytes(CH, CW, F1, F2, A):
    """algorithmic HBM bytes per board of the four network stages (inputs + outputs; weights stay cache resident)"""
    fin = 32 * (CH - 4) * (CW - 4)
    return [4 * (CH * CW + fin), 4 * (fin + F1), 4 * (F1 + F2), 4 * (F2 + A + 1)]


def stamped(name):
    """profiles/<name> if it was measured on the sources this build was made from, else None (+ why)"""
    from alphazero_amd import _lib
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no such file"
    d = json.load(open(path))
    comp = d.get("csrc_component")  # which sources the measurement depends on (None: the whole tree)
    have, want = d.get("csrc_sha"), _lib.csrc_tree_hash(comp)
    if have != want:
        return None, f"profiles/{name} was measured on csrc {comp or 'tree'} {have}, this build is {want}: dropped"
    return d, f"profiles/{name}, csrc {comp or 'tree'} {have}"


# ------------------------------------------------------------------------------------------------ self-launch
def self_launch(n):
    """`python bench.py --gpus N` started plainly: N fresh child processes, one per GPU, each with the environment
    torch.distributed.run would give it.  The parent has made no HIP call (torch is not even imported) and makes none: it waits,
    relays rank 0's stdout (the JSON line) and exits non-zero as soon as any rank fails.  Whatever ends the parent -- a failing rank,
    SIGTERM / SIGINT from a harness time limit, an exception -- the ranks it started are terminated and, after a grace period, killed
    (exactly those PIDs): a rank blocked in a collective must not outlive the run holding its GPU."""
    import signal
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []

    def stop_all(grace=float(os.environ.get("AZ_BENCH_KILL_GRACE_S", "10"))):
        live = [q for q in procs if q.poll() is None]
        for q in live:
            q.terminate()
        t_end = time.time() + grace
        for q in live:
            try:
                q.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    def on_signal(signum, _frame):
        raise KeyboardInterrupt(f"signal {signum}")

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AZ_BENCH_SELF_LAUNCHED="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else sys.stderr))  # rank 0 inherits stdout: its JSON line is ours
        alive = set(range(n))
        while alive and rc == 0:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
            if alive and rc == 0:
                time.sleep(0.2)
    except KeyboardInterrupt as e:
        print(f"bench.py: interrupted ({e}); stopping the ranks", file=sys.stderr, flush=True)
        rc = 130
    finally:
        stop_all()
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """how many host cores this process may use, and where the figure comes from: the cgroup CPU quota (v2 cpu.max / v1
    cfs_quota_us) when one is set, else the scheduler affinity mask; -> (cores, source, affinity count, quota or None)"""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < aff:
        return max(1, int(quota)), "cgroup cpu quota", aff, quota
    return aff, "sched_getaffinity", aff, quota


def cpu_baseline(state_dict, n_sim=100, games_per_proc=3):
    """BASELINE.md section 3 on this box's host cores, with the CPU oracle (the checker; C port of the reference's loop):
    one process, then one process per core of the lease's CPU share, >= 3 games each, every game timed (SelfPlayTimer idiom,
    timers.py:53-76); then the Arena path (AlphaZeroPlayer @100 vs rollout MCTSPlayer @100, 2 rounds, arena.py:119-185).
    Runs BEFORE this process touches the GPU: the workers are plain child processes that never see HIP."""
    from oracle import oracle as O
    from oracle import selfplay_worker as W
    weights = {k: v.cpu().numpy() for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
    t_all = time.perf_counter()
    one = W.play(weights, games_per_proc, n_sim, 0)
    s1 = np.array(one["seconds_per_game"])
    usable, share_src, aff, quota = cpu_share()
    # the multi-process leg: one process per core this process may use, capped at 16 -- the pool's documented CPU share of a one-GPU
    # lease (worker pools there are to be sized to 16; its host's 256 hardware threads serve 8 leases) -- unless AZ_BENCH_CPU_PROCS
    # says otherwise.  Which of the three set the number is in `processes_set_by`
    cap = int(os.environ.get("AZ_BENCH_CPU_PROCS", "16"))
    nproc = max(1, min(usable, cap))
    set_by = share_src if usable <= cap else ("AZ_BENCH_CPU_PROCS" if "AZ_BENCH_CPU_PROCS" in os.environ else "documented share of a one-GPU lease (16)")
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
                    "s_per_game_mean": float(per.mean()), "s_per_game_std": float(per.std()),
                    "note": f"{nproc} processes: {set_by}; usable cores {usable} ({share_src}), not the whole host"}
    net = O.ConvNet(O.OTHELLO, 8, 8, weights)
    t0 = time.perf_counter()
    moves, winners, _, _ = O.arena_games((O.OTHELLO, 8, 8), ("conv", net), n_sim, "mcts", n_sim, seed=0, n_rounds=2)
    t_arena = time.perf_counter() - t0
    # BASELINE config 1 on the oracle: 2 TicTacToe self-play games, rollout-mode MCTS at 100 simulations, temp 0 (reference 0.159 / 0.171 s)
    c1 = []
    for g in range(2):
        t0 = time.perf_counter()
        r = O.selfplay(O.TICTACTOE, 3, 3, 1, 100, alpha=-1, eps=-1, temp_max_step=-1, temp_min_step=0, tie_mode=O.TIE_RANDOM,
                       noise_mode=O.NOISE_OFF, seed=11, first_game_id=g, eval_method=O.EVAL_ROLLOUT)
        c1.append({"seconds": time.perf_counter() - t0, "plies": int(len(r["z"]))})
    return {"value": 1.0 / float(s1.mean()), "unit": "games/s", "cores": 1, "kind": "port",
            "sample": f"{games_per_proc} Othello 8x8 self-play games @{n_sim} sims on 1 core (oracle/liboracle.so); then {nproc} procs; arena; config 1",
            "cpu_model": cpu_model(), "host_hardware_threads": os.cpu_count(), "usable_cores": usable, "usable_cores_source": share_src,
            "affinity_cores": aff, "cgroup_cpu_quota": quota, "multi_process_count": nproc, "multi_process_count_set_by": set_by,
            "multi_process_games_per_sec": many["games_per_sec"] if many else None,
            "config1_oracle_seconds_game0": c1[0]["seconds"], "config1_oracle_seconds_game1": c1[1]["seconds"],
            "config1_reference_seconds": "0.159 / 0.171 (BASELINE.md section 2, build container)",
            "sample_detail": f"{games_per_proc} full games: {one['plies']} plies, {one['net_evals']} net evals",
            "config1_oracle": c1,
            "one_process": {"games": games_per_proc, "s_per_game_mean": float(s1.mean()), "s_per_game_std": float(s1.std()),
                            "games_per_sec": 1.0 / float(s1.mean()), "examples_per_sec": one["plies"] / float(s1.sum())},
            f"{nproc}_processes": many,
            "arena": {"rounds": 2, "player1": f"AlphaZeroPlayer({n_sim} sims)", "player2": f"MCTSPlayer(rollout, {n_sim} sims)",
                      "seconds": t_arena, "s_per_game": t_arena / 2, "plies": [len(m) for m in moves], "winners": winners},
            "reference_python": {"s_per_game": REF_S_PER_GAME_BUILD_CONTAINER, "games_per_sec": 1.0 / REF_S_PER_GAME_BUILD_CONTAINER,
                                 "where": "build container, Xeon @ 2.10 GHz, 1 core (BASELINE.md section 2)",
                                 "oracle_s_per_game_same_core": ORACLE_S_PER_GAME_BUILD_CONTAINER,
                                 "oracle_speedup_over_reference_same_core": REF_S_PER_GAME_BUILD_CONTAINER / ORACLE_S_PER_GAME_BUILD_CONTAINER},
            "seconds": time.perf_counter() - t_all}


# ------------------------------------------------------------------------------------------------ one workload
# the driver's record of the line keeps the scalar fields of `config` / `roofline` / `cpu_baseline` (nested objects dropped, long
# strings cut): what has to survive there comes first and is a scalar
ROOF_FIRST = ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "end_to_end_frac", "forward_frac", "mfma_busy", "avg_launch_ms",
              "launches", "flops_per_board", "boards_evaluated", "kernel_ms_total", "traffic_algorithmic", "share_of_network_time", "trunk_frac",
              "dense_frac", "heads_frac", "end_to_end_tflops", "avg_boards_per_launch", "network_share_of_profiled_step"]


def ordered(d, first):
    """d with the keys of `first` in front (then scalars, then strings, then nested objects)"""
    rest = [k for k in d if k not in first]
    rank = lambda k: 2 if isinstance(d[k], (dict, list)) else (1 if isinstance(d[k], str) else 0)
    return {k: d[k] for k in [f for f in first if f in d] + sorted(rest, key=rank)}


def compact(r):
    """the flat summary of a run_single / timed_waves result that goes into `config` of the headline line"""
    ro = r.get("roofline") or {}
    return {"games_per_sec": r["value"], "examples_per_sec": r.get("examples_per_sec"), "ms_per_step": r.get("ms_per_step"),
            "end_to_end_frac": ro.get("end_to_end_frac"), "dominant_kernel": (ro.get("kernel") or "")[:40], "dominant_frac": ro.get("frac"),
            "forward_frac": ro.get("forward_frac")}


# ------------------------------------------------------------------------------------------------ the line of record
# The driver parses the LAST line of stdout and stores an 8 KB tail of it: the line of record holds scalars only and stays far below that
# (round 4's 23 KB line, nested objects included, came back unparsed).  Everything nested goes to bench_detail.json beside this file.
LINE_BUDGET = 6144
DETAIL_NAME = "bench_detail.json"
TOP_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
TOP_EXTRA = ["examples_per_sec", "sims_per_sec", "plies_per_game"]
CONFIG_KEYS = ["workload", "concurrent_games_per_gpu", "games_per_gpu_per_step", "sims_per_move", "parallelism", "examples_per_sec", "us_per_lockstep",
               "end_to_end_frac", "samples_to_host_ms_per_step", "games_per_sec_incl_host_copy",
               "saturated_concurrent_games_per_gpu", "saturated_games_per_sec", "saturated_end_to_end_frac",
               "config4_workload", "config4_games_per_sec", "config4_end_to_end_frac", "config4_us_per_lockstep",
               "config3_games_per_sec", "config3_concurrent_games_per_gpu", "config3_end_to_end_frac_per_gpu",
               "config5_10_epochs_iteration_seconds", "config5_10_epochs_sgd_share", "config5_10_epochs_sgd_ms_per_step",
               "config1_device_seconds_game0", "config1_device_seconds_game1", "timed_region", "dense_layers"]
ROOF_KEYS = ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_algorithmic", "end_to_end_frac", "forward_frac", "mfma_busy",
             "avg_launch_ms", "launches", "flops_per_board", "boards_evaluated", "kernel_ms_total", "share_of_network_time", "trunk_frac", "dense_frac",
             "heads_frac", "saturated_frac", "saturated_end_to_end_frac", "saturated_kernel", "config4_frac", "config4_end_to_end_frac", "config4_kernel",
             "k_step_us", "k_step_traffic", "k_step_traffic_algorithmic"]
CPU_KEYS = ["value", "unit", "cores", "kind", "sample", "cpu_model", "usable_cores", "usable_cores_source", "multi_process_count", "multi_process_games_per_sec",
            "reference_s_per_game", "reference_where", "config1_oracle_seconds_game0", "config1_oracle_seconds_game1", "seconds"]


def _scalar(v, cut=110):
    """what may go into the line of record: None / bool / int / finite float (6 significant digits) / short string; else the marker _DROP"""
    if v is None or isinstance(v, bool):
        return v
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        v = float(v)
        return float(f"{v:.6g}") if np.isfinite(v) else None
    if isinstance(v, str):
        return v[:cut]
    return _scalar  # a nested object: never in the line


def _pick(d, keys):
    out = {}
    for k in keys:
        if d and k in d:
            v = _scalar(d[k])
            if v is not _scalar:
                out[k] = v
    return out


def record_line(full, detail=DETAIL_NAME):
    """the ONE line of stdout the driver keeps: the contract's top-level scalars + flat `config`, `roofline`, `cpu_baseline` (scalars only,
    strings cut, no NaN / Infinity) + the name of the file with everything else.  Raises if the result would not fit the budget or lacks the
    contract's fields -- a bench that cannot be recorded must fail loudly here, not in the driver's parser."""
    line = _pick(full, TOP_KEYS)
    missing = [k for k in TOP_KEYS if k not in line]
    if missing:
        raise ValueError(f"bench result lacks {missing}")
    line["config"] = _pick(full.get("config"), CONFIG_KEYS)
    line["roofline"] = _pick(full.get("roofline"), ROOF_KEYS) if full.get("roofline") else None
    line["cpu_baseline"] = _pick(full.get("cpu_baseline"), CPU_KEYS) if full.get("cpu_baseline") else None
    line.update(_pick(full, TOP_EXTRA))
    if full.get("errors"):
        line["errors"] = _scalar("; ".join(f"{k}: {v}" for k, v in full["errors"].items()), cut=300)
    line["detail"] = detail
    s = json.dumps(line, allow_nan=False)
    if len(s) > LINE_BUDGET or "\n" in s:
        raise ValueError(f"line of record is {len(s)} bytes (budget {LINE_BUDGET})")
    return s


def write_detail(full):
    """everything the run measured, nested objects included: bench_detail.json beside bench.py and, where the directory exists or can be made
    (a gpurun box merges it back), under gpurun_out/.  Never fails the bench."""
    def clean(o):
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (float, np.floating)):
            return float(o) if np.isfinite(o) else None
        if isinstance(o, np.integer):
            return int(o)
        return o
    body = json.dumps(clean(full), indent=1, allow_nan=False)
    only = os.environ.get("AZ_BENCH_DETAIL_DIR")  # a test that runs beside other bench runs keeps its detail file to itself
    for d in ((only,) if only else (ROOT, os.path.join(ROOT, "gpurun_out"))):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, DETAIL_NAME), "w") as f:
                f.write(body + "\n")
        except OSError as e:
            print(f"bench.py: could not write {d}/{DETAIL_NAME}: {e}", file=sys.stderr)


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

    def forward_flops(self):
        return sum(stage_flops(*self.geom))

    def wave(self, first_game_id):
        """one step on this rank: per_step games to the end; returns (samples dict of device views, engine stats)"""
        self.eng.run(self.per_step, first_game_id=first_game_id)
        return self.eng.samples(copy=False), self.eng.stats()

    def profiled_wave(self, first_game_id, timed_evals=None, timed_seconds=None):
        """one more step with HIP events around every network kernel launch (az_net_profile; kernel-by-kernel launches
        instead of graph replays).  -> roofline object of the dominant kernel + the step's time split"""
        self.hnet.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, st = self.wave(first_game_id)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = self.hnet.profile_read()
        ov = self.hnet.profile_overhead_ms()  # 0 since round 4: every launch carries its own start / stop events (dispatch begin / end)
        self.hnet.profile(False)
        fl, by = stage_flops(*self.geom), algorithmic_bytes(*self.geom)
        fused_tail = self.game == "connect4"  # fc1 + fc2 + heads are one fused launch (k_tail_mfma), booked in the fc1 slot
        evals = st["net_evals"]
        raw = {k: prof[k][0] for k in SLOTS}
        cnt = {k: prof[k][1] for k in SLOTS}
        ms = {k: max(raw[k] - ov * cnt[k], 0.0) for k in SLOTS}  # kernel time: what rocprofv3 reports per dispatch
        exe, wino = executed_conv_flops(*self.geom[:2])
        # the kernels of a forward.  fc1 and fc2 are two launches of ONE kernel; the trunk's few small-batch launches (k_trunk, end of a
        # wave) are booked with k_trunk2: their boards are in `evals` too
        # `ms`: all launches of the stage (every board of `evals` went through one of them); `own_ms` / `launches`: the launches of the
        # kernel the stage is NAMED after -- their mean is the average duration rocprofv3 --stats lists for that kernel.  The few
        # launches of a wave's last plies run on the small-batch kernels (k_trunk / k_trunk_q, k_dense_frag) and are booked apart
        def own(*slots):  # the slot(s) of the kernel the stage is named after = whichever family served the full-size launches
            return max(slots, key=lambda ks: sum(cnt[k] for k in ks))
        t_own, d_own = own(("k_trunk2",), ("k_trunk",), ("k_trunk_q",)), own(("k_gemm fc1", "k_gemm fc2"), ("small fc1", "small fc2"))
        kern = {
            "trunk": {"name": self.hnet.stage_kernel(0, self.G), "ms": ms["k_trunk2"] + ms["k_trunk"] + ms["k_trunk_q"], "launches": sum(cnt[k] for k in t_own),
                      "own_ms": sum(ms[k] for k in t_own), "flops": exe, "flops_algorithmic": fl[0], "bytes": by[0], "pmc": ["k_trunk"]},
            "dense": {"name": self.hnet.stage_kernel(1, self.G), "ms": ms["k_gemm fc1"] + ms["k_gemm fc2"] + ms["small fc1"] + ms["small fc2"],
                      "launches": sum(cnt[k] for k in d_own),
                      "own_ms": sum(ms[k] for k in d_own), "flops": fl[1] + fl[2] + (fl[3] if fused_tail else 0),
                      "flops_algorithmic": fl[1] + fl[2] + (fl[3] if fused_tail else 0), "bytes": by[1] + by[2], "pmc": ["k_gemm_fc1", "k_gemm_fc2"] if not fused_tail else ["k_tail"]},
            "heads": {"name": self.hnet.stage_kernel(3, self.G), "ms": ms["k_heads"], "launches": cnt["k_heads"], "own_ms": ms["k_heads"],
                      "flops": fl[3], "flops_algorithmic": fl[3], "bytes": by[3], "pmc": ["k_heads"]},
        }
        if not fused_tail and self.hnet.stage_kernel(2, self.G) != kern["dense"]["name"]:
            kern["dense"]["name"] += " (fc1) / " + self.hnet.stage_kernel(2, self.G) + " (fc2)"
        net_ms = sum(k["ms"] for k in kern.values())
        dom = max(kern, key=lambda k: kern[k]["ms"])
        K = kern[dom]
        ach = K["flops"] * evals / (K["ms"] * 1e-3) / 1e12
        key = f"{self.game}_{self.G}"
        tfile, twhy = stamped("traffic.json")  # HBM bytes per full-batch launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        cfile, cwhy = stamped("mfma_counters.json")  # SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE passes
        traffic = None
        if tfile and tfile.get(key) and all(p in tfile[key] for p in K["pmc"]):
            traffic = sum(tfile[key][p] for p in K["pmc"])
        busy = None
        if cfile and cfile.get(key):
            b = [cfile[key].get(p, {}).get("mfma_busy_frac") for p in K["pmc"]]
            busy = float(np.mean(b)) if all(x is not None for x in b) else None
        full = [self.hnet.time_stage(s, self.G, iters=20) for s in range(4)]
        roof = {"bound": "mfma", "kernel": K["name"] + (" (fc1 + fc2: two launches per forward)" if dom == "dense" and not fused_tail else ""),
                "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                "flops_per_board": K["flops"], "boards_evaluated": evals, "kernel_ms_total": K["ms"], "launches": K["launches"],
                "avg_launch_ms": K["own_ms"] / max(1, K["launches"]),
                "avg_boards_per_launch": evals / max(1, cnt["k_trunk2"] + cnt["k_trunk"] + cnt["k_trunk_q"]),
                "small_batch_launches": {k: cnt[k] for k in ("k_trunk", "k_trunk_q", "small fc1", "small fc2")},
                "share_of_network_time": K["ms"] / net_ms,
                "traffic": traffic, "traffic_algorithmic": self.G * K["bytes"], "traffic_source": twhy,
                "traffic_is_for": f"one launch of every stage of the kernel at the full batch of {self.G} boards (tools/prof_net.py under rocprofv3 --pmc)",
                "mfma_busy": busy, "mfma_busy_source": cwhy,
                "event_overhead_ms_per_interval": ov,
                "measured_on": "one separately profiled step after the timed region: every network launch carries a start and a stop event "
                               "(hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps, what rocprofv3 reports)",
                "profiled_step_ms": 1e3 * dt,
                "kernels": {k: {"name": v["name"], "ms": v["ms"], "launches": v["launches"], "flops_per_board": v["flops"],
                                "tflops": v["flops"] * evals / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else None,
                                "frac": v["flops"] * evals / (v["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS if v["ms"] > 0 else None,
                                "share_of_network_time": v["ms"] / net_ms} for k, v in kern.items() if v["launches"]},
                "profiled_step_kernel_ms_raw": raw, "profiled_step_launches": cnt,
                "full_batch_launch_ms": dict(zip(["trunk", "fc1", "fc2", "heads"], full)),
                "forward_flops_per_board": sum(fl),
                "forward_frac": sum(fl) * evals / (net_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "network_share_of_profiled_step": sum(raw.values()) / (1e3 * dt)}
        if dom == "trunk":
            roof["conv2_form"] = "Winograd F(2x2,3x3)" if wino else "direct"
            roof["small_batch_trunk_launches"] = cnt["k_trunk"] + cnt["k_trunk_q"]
            roof["frac_algorithmic_direct_form"] = fl[0] * evals / (K["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS
            roof["flops_per_board_direct_form"] = fl[0]
        if timed_evals is not None and timed_seconds:
            roof["end_to_end_frac"] = timed_evals * sum(fl) / timed_seconds / 1e12 / PEAK_F32_MFMA_TFLOPS
            roof["end_to_end_tflops"] = timed_evals * sum(fl) / timed_seconds / 1e12
        if fused_tail:
            roof["note"] = "Connect4Net: fc1 + fc2 + heads run as ONE fused launch (k_tail_mfma), booked under 'dense'"
        for k, v in roof["kernels"].items():  # flat copies: the driver's record keeps scalars only
            roof[f"{k}_frac"], roof[f"{k}_share_of_network_time"] = v["frac"], v["share_of_network_time"]
        return ordered(roof, ROOF_FIRST)

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
           "us_per_lockstep": 1e6 * dt / steps / max(1, st["lockstep_iters"]),
           "max_tree_nodes_per_game": st["max_nodes_used"], "dtype": "f32"}
    out["roofline"] = w.profiled_wave((warmup + steps) * G, timed_evals=evals, timed_seconds=dt)
    w.close()
    return out


# ------------------------------------------------------------------------------------------------ distributed helpers
class Job:
    """the torch.distributed job this rank belongs to (world 1: no group)"""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # AZ_BENCH_BACKEND=gloo + AZ_BENCH_ONE_DEVICE=1: rehearsal of the N > 1 control flow on a one-GPU box (never a fallback:
        # with the default backend a failing RCCL init is an error)
        self.backend = os.environ.get("AZ_BENCH_BACKEND", "nccl")
        if os.environ.get("AZ_BENCH_ONE_DEVICE") == "1":
            self.local_rank = 0
        self.dev = "cuda" if self.backend == "nccl" else "cpu"

    def init(self):
        torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend)

    def sync(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(self, x):
        t = torch.tensor([x], dtype=torch.float64, device=self.dev)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_rows(self, values):
        """[world, len(values)] float64 array of every rank's values"""
        t = torch.tensor(values, dtype=torch.float64, device=self.dev)
        if self.world == 1:
            return t.view(1, -1).cpu().numpy()
        out = torch.zeros(self.world * len(values), dtype=torch.float64, device=self.dev)  # flat: the one layout every backend concatenates alike
        dist.all_gather_into_tensor(out, t)
        return out.view(self.world, -1).cpu().numpy()


def timed_waves(job, w, steps, warmup, first_wave=0):
    """`steps` self-play waves of w on every rank + (N > 1) the all-gather of the samples, bracketed by barrier + synchronize on
    both sides.  -> dict with the max-over-ranks time and every rank's own time / gather time"""
    from alphazero_amd.dist import all_gather_samples, rank_game_range
    G = w.per_step
    gather_s = [0.0]

    def step(wave):
        first, _ = rank_game_range(job.rank, job.world, G, wave)
        smp, st = w.wave(first)
        if job.world > 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            smp = all_gather_samples({k: smp[k] for k in ("state", "pi", "z", "meta")})
            torch.cuda.synchronize()
            gather_s[0] += time.perf_counter() - t0
        return smp["z"].shape[0], st["net_evals"]

    for i in range(warmup):
        step(first_wave + i)
    gather_s[0] = 0.0
    job.sync()
    t0 = time.perf_counter()
    samples, evals = 0, 0
    for k in range(steps):
        s, e = step(first_wave + warmup + k)
        samples += s  # after the all-gather every rank holds all ranks' samples
        evals += e
    t_local = time.perf_counter() - t0  # this rank's own time, before it waits for the others
    job.sync()
    dt = job.max_over_ranks(time.perf_counter() - t0)
    per_rank = job.gather_rows([t_local, gather_s[0], float(evals)])
    return {"dt": dt, "samples": samples, "evals_rank0": evals, "evals_all": float(per_rank[:, 2].sum()),
            "per_rank_ms_per_step": [1e3 * float(x) / steps for x in per_rank[:, 0]],
            "per_rank_gather_ms_per_step": [1e3 * float(x) / steps for x in per_rank[:, 1]]}


def run_config3(job, total_games, sims, steps, warmup):
    """BASELINE.json configs[2] at its literal size: `total_games` (32768) concurrent Othello games sharded over the N ranks
    (4096 per GPU at N = 8), RCCL all-gather of the samples after every wave"""
    per = total_games // job.world
    w = Workload("config3", "othello", per, sims)
    r = timed_waves(job, w, steps, warmup)
    games = steps * per * job.world
    out = {"workload": f"Othello 8x8, {total_games} concurrent self-play games sharded {job.world} ways ({per} per GPU), {sims} sims/move, "
                       f"RCCL all-gather of the samples after every wave", "concurrent_games_per_gpu": per, "n_gpus": job.world,
           "value": games / r["dt"], "unit": "games/s", "examples_per_sec": r["samples"] / r["dt"], "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * r["dt"] / steps, "per_rank_ms_per_step": r["per_rank_ms_per_step"],
           "per_rank_gather_ms_per_step": r["per_rank_gather_ms_per_step"], "dtype": "f32",
           "end_to_end_frac_per_gpu": r["evals_all"] / job.world * w.forward_flops() / r["dt"] / 1e12 / PEAK_F32_MFMA_TFLOPS}
    w.close()
    return out


def run_config5(job, episodes, sims, eval_episodes=64, variants=None, only=None):
    """BASELINE.json configs[4]: AlphaZeroTrainer's loop on Othello 8x8 (trainer.py:475-572: self-play -> optimize_network ->
    update_network -> evaluate against the PREVIOUS network), two iterations per variant with per-phase wall times.  At N > 1 the
    episodes and the evaluation rounds are sharded over the ranks, rank 0 runs the SGD, the weights are broadcast."""
    from alphazero_amd import base
    from alphazero_amd.games.othello import OthelloConfig
    from alphazero_amd.trainer import AlphaZeroTrainer
    if variants is None:
        # the reference's hyper-parameters are batch_size 64, 10 epochs (games/othello.py:34-35): the `_10_epochs` variant; the others run
        # ONE epoch (stated), the batch-64 ones on a smaller episode count so that all variants do a comparable number of SGD steps
        variants = [("reference_batch_64", dict(episodes=max(64, episodes // 8), batch_size=64, epochs=1), "hip"),
                    # the reference's own hyper-parameters (games/othello.py:34-35: 10 epochs of batch 64): what the SGD share of an iteration is
                    ("reference_batch_64_10_epochs", dict(episodes=max(64, episodes // 8), batch_size=64, epochs=10), "hip"),
                    ("batch_512", dict(episodes=episodes, batch_size=512, epochs=1), "hip"),
                    # the same loop on the stock PyTorch step (MIOpen convolutions, replayed as a HIP graph): the checker, timed beside it
                    ("reference_batch_64_stock_pytorch", dict(episodes=max(64, episodes // 8), batch_size=64, epochs=1), "torch")]
    if only:
        variants = [v for v in variants if v[0] in only]
    base.DEFAULT_MODELS_PATH = tempfile.mkdtemp() + "/"
    out = {"workload": f"Othello 8x8 trainer loop: self-play ({sims} sims/move) + symmetry augmentation + SGD (momentum 0.9, weight decay 1e-4, "
                       f"ExponentialLR 0.9) + weight hand-off + {eval_episodes} arena games against the previous network", "n_gpus": job.world,
           "note": "iteration 0 carries one-off costs (engine creation, graph capture, MIOpen's algorithm search when the stock PyTorch step runs); "
                   "iteration 1 is the steady state", "variants": {}}
    for name, v, backend in variants:
        torch.manual_seed(0)
        tr = AlphaZeroTrainer(verbose=False, engine_slots=min(v["episodes"], 32768), seed=0, materialize_memory=False)
        tr.sgd_backend = backend
        tr.game = "othello"
        tr.config = OthelloConfig(board_size=8, simulations=sims, episodes=v["episodes"], epochs=v["epochs"], batch_size=v["batch_size"], iterations=2,
                                  device="cuda", eval_opponent="previous", eval_episodes=eval_episodes, do_eval=True, save=False, save_checkpoints=False)
        tr.setup()
        its = []
        for it in range(2):
            phases = {}

            def timed(label, fn):
                job.sync()
                t0 = time.perf_counter()
                fn()
                job.sync()
                phases[label] = job.max_over_ranks(time.perf_counter() - t0)

            timed("self_play_and_augmentation", lambda: tr.self_play(it))
            n = int(tr.device_memory["z"].shape[0])
            timed("optimize_network", lambda: tr.optimize_network(it))
            timed("update_network", lambda: tr.update_network(it))
            timed("evaluate", lambda: tr.evaluate(it))
            steps = v["epochs"] * (n // v["batch_size"])
            res = tr.eval_results["results"].get(it) if job.rank == 0 else None
            its.append({"iteration": it, "samples_with_twins": n, "sgd_steps": steps, "seconds": phases, "iteration_seconds": sum(phases.values()),
                        "sgd_ms_per_step": 1e3 * phases["optimize_network"] / max(1, steps),
                        "sgd_share": phases["optimize_network"] / sum(phases.values()),
                        "last_losses": ({k: tr.loss_values[it][v["epochs"] - 1][k][-1] for k in ("pi", "v")} if job.rank == 0 else None),
                        "eval_results": res})
        out["variants"][name] = {**v, "sgd_step": {"hip": "hand-written HIP step (csrc/az_train.hip)", "torch": "stock PyTorch"}[tr.sgd_backend_used or backend], "iterations": its,
                                 "games_per_sec_whole_loop": v["episodes"] / its[1]["iteration_seconds"],
                                 "examples_per_sec_whole_loop": its[1]["samples_with_twins"] / its[1]["iteration_seconds"]}
        if tr._engine is not None:
            tr._engine.close()
        if tr._hipnet is not None:
            tr._hipnet.close()
        if tr._hip_step is not None:
            tr._hip_step[1].close()
        del tr
        torch.cuda.empty_cache()
    a, b = out["variants"].get("reference_batch_64"), out["variants"].get("reference_batch_64_stock_pytorch")
    if a and b:
        out["sgd_ms_per_step_batch_64"] = {"hand_written": a["iterations"][1]["sgd_ms_per_step"], "stock_pytorch": b["iterations"][1]["sgd_ms_per_step"],
                                           "ratio": b["iterations"][1]["sgd_ms_per_step"] / a["iterations"][1]["sgd_ms_per_step"]}
    return out


def run_latency(sims):
    """The reference's own shapes of use, where nothing fills the chip and a lock-step is a chain of five dependent launches: one game's
    search (MCT.search behind AlphaZeroPlayer.get_move, players.py:158-191), small self-play waves, and an evaluation arena of 64 games
    against another network (trainer.py:408-446) with the two players' searches overlapped."""
    import numpy as np
    from alphazero_amd import engine as E
    from alphazero_amd.arena import BatchedArena
    from alphazero_amd.games.othello import OthelloNet
    torch.manual_seed(0)
    net, net2 = OthelloNet(n=8).cuda().eval(), OthelloNet(n=8).cuda().eval()
    out = {"workload": f"Othello 8x8, {sims} sims/move, random-init OthelloNet", "self_play_waves": {}}
    for G in (1, 64, 512):
        hnet = net.to_hip(max_batch=G)
        eng = E.SelfPlayEngine(0, 8, 8, n_slots=G, n_sim=sims, net=hnet, seed=0)
        eng.run(G)  # warm-up: kernels loaded, graph captured
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        smp = eng.run(G, first_game_id=G)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = eng.stats()
        out["self_play_waves"][str(G)] = {"games_per_sec": G / dt, "seconds": dt, "us_per_lockstep": 1e6 * dt / max(1, st["lockstep_iters"]),
                                          "ms_per_move": 1e3 * dt / (int(smp["z"].shape[0]) / G)}
        eng.close()
        hnet.close()
    for overlap in (True, False):
        ar = BatchedArena("othello", net, opponent=net2, n_sim=sims, seed=1, board_size=8)
        ar.overlap = overlap
        ar.play_games(64, shard=False)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ar.play_games(64, shard=False)
        torch.cuda.synchronize()
        out["arena_64_games_vs_network_seconds" + ("" if overlap else "_one_player_after_the_other")] = time.perf_counter() - t0
    return out


def run_config1(sims=100, games=2):
    """BASELINE.json configs[0]: TicTacToe, rollout-mode MCTSPlayer(n_sim=100) on both sides, temp 0, 2 self-play games, timed game by
    game in the SelfPlayTimer idiom (timers.py:42-76) through the reference's plugin surface -- Board + Player objects, the tree
    on the GPU (one engine slot, k_rollout_step: one launch per simulation).  Reference: 0.159 / 0.171 s."""
    from alphazero_amd.games.tictactoe import TicTacToeBoard
    from alphazero_amd.players import MCTSPlayer
    np.random.seed(0)
    board, player = TicTacToeBoard(), MCTSPlayer(n_sim=sims)
    out = []
    for g in range(games + 1):  # game 0 is the warm-up (engine creation, kernels loaded), not reported
        board.reset()
        player.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plies = 0
        while not board.is_game_over():
            move = player.get_move(board)[0]
            board.play_move(move)
            player.apply_move(move, player=-board.player)
            plies += 1
        torch.cuda.synchronize()
        if g:
            out.append({"seconds": time.perf_counter() - t0, "plies": plies, "winner": int(board.get_winner())})
    if player.mct._engine is not None:
        player.mct._engine.close()
    return {"workload": f"TicTacToe, MCTSPlayer(n_sim={sims}, rollout mode) on both sides, temp 0, {games} self-play games (SelfPlayTimer idiom)",
            "games": out, "seconds_per_game_mean": float(np.mean([g["seconds"] for g in out])),
            "reference_seconds": [0.159, 0.171], "reference_where": "build container, 1 core (BASELINE.md section 2)"}


def run_dense_i8_prototype(sims, sizes=(4096, 32768)):
    """the headline workload once more with OthelloNet's dense layers on the int8 matrix pipe (AZ_DENSE_I8=1, DESIGN section 10: prototype,
    default off).  Product and oracle read the switch once per process, so every size runs in a child process of this one (two
    timed waves after one warm-up wave); a failing child is recorded, it does not fail the bench.  NOT the line of record."""
    import subprocess
    out = {"what": "AZ_DENSE_I8=1: fc1 / fc2 as exact block-fixed-point integer GEMMs (k_q_rows + k_qgemm / k_qdense_small); child processes of this bench",
           "sizes": {}}
    env = dict(os.environ, AZ_DENSE_I8="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for g in sizes:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--games", str(g), "--sims", str(sims), "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline", "--no-literal-configs"]
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                out["sizes"][str(g)] = {"error": (p.stderr or p.stdout)[-300:]}
                continue
            d = json.loads(lines[-1])
            out["sizes"][str(g)] = {"games_per_sec": d["value"], "us_per_lockstep": d["config"]["us_per_lockstep"], "examples_per_sec": d["examples_per_sec"],
                                    "dense_layers": d["config"].get("dense_layers")}
        except Exception as e:  # noqa: BLE001 -- an extra must never take the line of record down
            out["sizes"][str(g)] = {"error": repr(e)[:300]}
    return out


def run_saturated(job, games, sims, steps, warmup, first_wave):
    """the headline's workload at `games` concurrent games per GPU (weak scaling at N > 1, all-gather included): where the
    rate-against-batch curve has flattened"""
    w = Workload("saturated", "othello", games, sims)
    r = timed_waves(job, w, steps, warmup, first_wave=first_wave)
    total = steps * games * job.world
    out = {"workload": w.desc, "concurrent_games": games, "n_gpus": job.world, "value": total / r["dt"], "unit": "games/s",
           "examples_per_sec": r["samples"] / r["dt"], "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * r["dt"] / steps,
           "per_rank_ms_per_step": r["per_rank_ms_per_step"], "per_rank_gather_ms_per_step": r["per_rank_gather_ms_per_step"], "dtype": "f32"}
    if job.world > 1:
        dist.barrier()
    if job.rank == 0:
        out["roofline"] = w.profiled_wave((first_wave + warmup + steps) * games * job.world, timed_evals=r["evals_all"] / job.world, timed_seconds=r["dt"])
    w.close()
    if job.world > 1:
        dist.barrier()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games (engine slots) per GPU of the headline run: BASELINE configs[1] / [2]")
    ap.add_argument("--waves", type=int, default=1, help="games per step per GPU = waves x games (finished slots are refilled)")
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--saturated-games", type=int, default=32768, help="concurrent games per GPU of the `saturated` extra (0: skip)")
    ap.add_argument("--config3-total", type=int, default=32768, help="N > 1: concurrent games of config3, sharded over the ranks")
    ap.add_argument("--config4-games", type=int, default=8192)
    ap.add_argument("--config5-episodes", type=int, default=4096, help="episodes per iteration of the config5 trainer loop")
    ap.add_argument("--config5-eval-episodes", type=int, default=64)
    ap.add_argument("--config5-variants", default="", help="comma-separated subset of the config5 variants (default: all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense-i8", action="store_true", help="also run the opt-in AZ_DENSE_I8=1 prototype (child processes, after the line of record; detail file only)")
    ap.add_argument("--no-literal-configs", action="store_true", help="skip the saturated / config1 / config3 / config4 / config5 / latency objects")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))  # before torch is imported: the parent never touches HIP

    t_start = time.perf_counter()
    phases = {}  # wall seconds of this rank per part of the run (rank 0's go into the line as `bench_seconds`)

    def lap(name, since):
        phases[name] = round(time.perf_counter() - since, 2)
        return time.perf_counter()

    _imports()
    t_lap = lap("imports", t_start)
    job = Job()
    if job.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={job.world}")
    dry = os.environ.get("AZ_BENCH_DRYRUN")
    if dry:
        # launcher rehearsal without a GPU (tests/test_dist.py): the ranks only rendezvous over gloo, reduce one number and rank 0 prints
        # a stub line; "fail:<rank>" makes that rank exit non-zero after the rendezvous (the parent must then stop the others);
        # "hang:<rank>" makes every OTHER rank wait forever at a barrier that rank never reaches (the parent must kill them)
        dist.init_process_group("gloo")
        t = torch.tensor([float(job.rank + 1)])
        dist.all_reduce(t)
        if dry.startswith("fail:") and int(dry.split(":")[1]) == job.rank:
            sys.exit(7)
        if dry.startswith("hang:"):
            if int(dry.split(":")[1]) == job.rank:
                sys.exit(7)
            import signal
            signal.signal(signal.SIGTERM, signal.SIG_IGN)  # a rank stuck in a device wait does not react to SIGTERM either
            time.sleep(3600)
        dist.barrier()
        if job.rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": job.world, "sum_of_ranks_plus_one": float(t.item())}), flush=True)
        dist.destroy_process_group()
        return

    cpu = None
    if job.world == 1 and not args.no_cpu_baseline:  # before HIP is initialised: the worker processes are forked from a GPU-free parent
        from alphazero_amd.games.othello import OthelloNet
        torch.manual_seed(0)
        cpu = cpu_baseline(OthelloNet(n=8).eval().state_dict())
        t_lap = lap("cpu_baseline", t_lap)

    job.init()
    rank, world = job.rank, job.world

    w = Workload("headline", "othello", args.games, args.sims, waves=args.waves)
    G = w.per_step
    r = timed_waves(job, w, args.steps, args.warmup)
    dt, samples = r["dt"], r["samples"]
    st = w.eng.stats()
    t_lap = lap("headline_build_warmup_timed", t_lap)

    if rank == 0:
        games = args.steps * G * world
        literal = args.games == 4096 and args.sims == 100 and args.waves == 1
        out = {
            "metric": "self-play games/sec (whole node), Othello 8x8 @100 sims/move", "value": games / dt, "unit": "games/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{1 if world == 1 else 2}] literal: " if literal else "") +
                                   f"Othello 8x8, {w.G} concurrent games/GPU, {args.sims} sims/move, {world} MI355X",
                       "concurrent_games_per_gpu": w.G, "games_per_gpu_per_step": G, "sims_per_move": args.sims,
                       "parallelism": f"game-sharded x{world}" + (", RCCL sample all-gather per wave" if world > 1 else ""),
                       "examples_per_sec": samples / dt, "us_per_lockstep": 1e3 * 1e3 * dt / args.steps / max(1, st["lockstep_iters"])},
            "examples_per_sec": samples / dt, "sims_per_sec": samples * args.sims / dt,
            "plies_per_game": samples / games, "net_evals_last_step": st["net_evals"], "lockstep_iters_last_step": st["lockstep_iters"],
            "graph_replays": st["graph_replays"], "max_tree_nodes_per_game": st["max_nodes_used"],
            # diagnosis of a scaling run: every rank's own step time (before waiting for the others) and its share spent in the RCCL gather
            "per_rank_ms_per_step": r["per_rank_ms_per_step"], "per_rank_gather_ms_per_step": r["per_rank_gather_ms_per_step"],
        }
        # the boundary hands over DEVICE buffers (az_engine_samples; the trainer's memory stays in HBM).  What a caller who wants the
        # samples of a step in host memory would pay on top, measured outside the timed region on the last step's samples:
        smp = w.eng.samples(copy=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = [smp[k].cpu() for k in ("state", "pi", "z")]
        t_host = time.perf_counter() - t0
        out["config"]["samples_to_host_ms_per_step"] = 1e3 * t_host
        out["config"]["samples_to_host_mbytes_per_step"] = sum(t.numel() * t.element_size() for t in host) / 1e6
        out["config"]["games_per_sec_incl_host_copy"] = games / (dt + args.steps * t_host * world)  # every rank would copy the gathered set
        del host, smp
    if world > 1:
        dist.barrier()
    # the roofline's per-kernel times: one extra step, profiled, outside the timed region (rank 0; the others idle at the barrier)
    roof = w.profiled_wave((args.warmup + args.steps) * G * world + rank * G, timed_evals=r["evals_all"] / world, timed_seconds=dt) if rank == 0 else None
    kstep_key = f"othello_{w.G}"
    w.close()
    if world > 1:
        dist.barrier()
    t_lap = lap("headline_profiled_step", t_lap)
    saturated = config1 = config3 = config4 = config5 = latency = dense_i8 = None
    errors = {}

    def leg(name, fn):
        """one optional leg.  On a single GPU a failing leg is recorded (`errors`) and never takes the line of record down; inside a
        torch.distributed job it propagates: a rank that leaves a collective sequence would hang the others"""
        nonlocal t_lap
        try:
            r = fn()
        except Exception as e:  # noqa: BLE001
            if world > 1:
                raise
            import traceback
            traceback.print_exc(file=sys.stderr)
            errors[name], r = repr(e)[:200], None
        t_lap = lap(name, t_lap)
        return r

    only5 = [v for v in args.config5_variants.split(",") if v] or None
    first5 = ["reference_batch_64_10_epochs"]  # the reference's own hyper-parameters: its scalars are in the line of record
    if not args.no_literal_configs:
        if args.saturated_games and args.saturated_games != args.games:
            saturated = leg("saturated", lambda: run_saturated(job, args.saturated_games, args.sims, steps=2, warmup=1, first_wave=1000))
        if world > 1 and args.config3_total // world != args.games:
            config3 = leg("config3", lambda: run_config3(job, args.config3_total, args.sims, steps=2, warmup=1))
        if world == 1:
            # Othello games all last 60-65 plies: one synchronised wave per step keeps 92 % of the leaf rows filled (the rest are
            # terminal leaves, which need no evaluation).  Connect4 games last 18-42 plies: finished slots are refilled and a
            # step plays 8 x 8192 games, so that the drain at the end of a step (its length is one game) is amortised
            config4 = leg("config4", lambda: run_single("config4", "connect4", args.config4_games, 200, steps=2, warmup=1, waves=8))
            config1 = leg("config1", run_config1)
        config5 = leg("config5", lambda: run_config5(job, args.config5_episodes, args.sims, eval_episodes=args.config5_eval_episodes,
                                                      only=[v for v in first5 if only5 is None or v in only5] if world == 1 else only5))
    full = None
    if rank == 0:
        cfg = out["config"]
        cfg["end_to_end_frac"] = roof.get("end_to_end_frac")
        if saturated is not None:
            c = compact(saturated)
            cfg.update({"saturated_concurrent_games_per_gpu": args.saturated_games, "saturated_games_per_sec": c["games_per_sec"],
                        "saturated_end_to_end_frac": c["end_to_end_frac"], "saturated_dominant_kernel": c["dominant_kernel"],
                        "saturated_dominant_frac": c["dominant_frac"], "saturated_forward_frac": c["forward_frac"]})
        if config4 is not None:
            c = compact(config4)
            cfg.update({"config4_games_per_sec": c["games_per_sec"], "config4_end_to_end_frac": c["end_to_end_frac"],
                        "config4_dominant_kernel": c["dominant_kernel"], "config4_dominant_frac": c["dominant_frac"],
                        "config4_us_per_lockstep": config4.get("us_per_lockstep"),
                        "config4_workload": f"Connect4 6x7, {args.config4_games} concurrent games, 200 sims/move"})
        if config3 is not None:
            cfg.update({"config3_games_per_sec": config3["value"], "config3_concurrent_games_per_gpu": config3["concurrent_games_per_gpu"],
                        "config3_end_to_end_frac_per_gpu": config3["end_to_end_frac_per_gpu"]})
        if config1 is not None:
            cfg.update({"config1_device_seconds_game0": config1["games"][0]["seconds"], "config1_device_seconds_game1": config1["games"][1]["seconds"]})
        if config5 is not None:
            v10 = config5["variants"].get("reference_batch_64_10_epochs")
            if v10:
                cfg.update({"config5_10_epochs_iteration_seconds": v10["iterations"][1]["iteration_seconds"],
                            "config5_10_epochs_sgd_share": v10["iterations"][1]["sgd_share"],
                            "config5_10_epochs_sgd_ms_per_step": v10["iterations"][1]["sgd_ms_per_step"]})
        cfg["timed_region"] = "engine as shipped: HIP-graph replays, no event recording"
        cfg["dense_layers"] = ("exact block-fixed-point on the int8 matrix pipe (AZ_DENSE_I8=1: the roofline's f32 MFMA peak does not price fc1 / fc2)"
                               if os.environ.get("AZ_DENSE_I8") == "1" else "float32 fma chains on the f32-input MFMA")
        if saturated is not None and saturated.get("roofline"):
            roof["saturated_frac"], roof["saturated_end_to_end_frac"] = saturated["roofline"]["frac"], saturated["roofline"].get("end_to_end_frac")
            roof["saturated_kernel"] = saturated["roofline"]["kernel"][:40]
        if config4 is not None:
            roof["config4_frac"], roof["config4_end_to_end_frac"] = config4["roofline"]["frac"], config4["roofline"].get("end_to_end_frac")
            roof["config4_kernel"] = config4["roofline"]["kernel"][:40]
        sims_per_gpu = samples * args.sims / dt / world
        tree = {"algorithmic_bytes_per_sim": 919, "achieved": sims_per_gpu * 919 / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": sims_per_gpu * 919 / 8e12, "note": "per GPU, whole path: the tree kernels are a few % of a step, the path is bound by the network's MFMA work"}
        kfile, kwhy = stamped("kstep_counters.json")  # k_step<true,true>: duration + FETCH_SIZE / WRITE_SIZE passes (tools/refresh_profiles.sh)
        tree["k_step"] = kfile.get(kstep_key) if kfile else None
        tree["k_step_source"] = kwhy
        if tree["k_step"]:  # flat copies for the line of record
            ks = tree["k_step"]
            roof["k_step_us"], roof["k_step_traffic_algorithmic"] = ks.get("duration_us_under_pmc"), ks.get("algorithmic_bytes_per_launch")
            roof["k_step_traffic"] = ks.get("hbm_bytes_per_launch_lower")
        out["roofline"] = ordered(roof, ROOF_FIRST + ["saturated_frac", "saturated_end_to_end_frac", "config4_frac", "config4_end_to_end_frac"])
        out["tree_hbm"] = tree
        if cpu is not None:
            cpu["reference_s_per_game"], cpu["reference_where"] = REF_S_PER_GAME_BUILD_CONTAINER, "Python reference, build container, 1 core (BASELINE.md 2)"
        full = dict(out)
        for name, obj in (("saturated", saturated), ("config1", config1), ("config3", config3), ("config4", config4), ("config5", config5), ("cpu_baseline", cpu)):
            if obj is not None:
                full[name] = obj
        full["errors"] = errors
        phases["until_line_of_record"] = round(time.perf_counter() - t_start, 2)
        full["bench_seconds"] = phases  # where the run's wall time went (the timed region is `ms_per_step` x `steps` of "headline_build_warmup_timed")
        write_detail(full)
        # THE line of record: printed once, as soon as everything it carries is measured; nothing is written to stdout after it
        print(record_line(full), flush=True)
        sys.stdout.flush()
        os.dup2(2, 1)  # whatever the remaining legs (or a library under them) print goes to stderr
    # the remaining legs only feed bench_detail.json
    if not args.no_literal_configs and world == 1:
        rest5 = leg("config5_other_variants", lambda: run_config5(job, args.config5_episodes, args.sims, eval_episodes=args.config5_eval_episodes,
                                                                   only=[v for v in ("reference_batch_64", "batch_512", "reference_batch_64_stock_pytorch")
                                                                         if only5 is None or v in only5]))
        if rest5 is not None and config5 is not None:
            config5["variants"].update(rest5["variants"])
            a, b = config5["variants"].get("reference_batch_64"), config5["variants"].get("reference_batch_64_stock_pytorch")
            if a and b:
                config5["sgd_ms_per_step_batch_64"] = {"hand_written": a["iterations"][1]["sgd_ms_per_step"], "stock_pytorch": b["iterations"][1]["sgd_ms_per_step"],
                                                       "ratio": b["iterations"][1]["sgd_ms_per_step"] / a["iterations"][1]["sgd_ms_per_step"]}
        latency = leg("latency", lambda: run_latency(args.sims))
        if os.environ.get("AZ_DENSE_I8") != "1" and args.dense_i8:
            dense_i8 = leg("dense_i8_prototype", lambda: run_dense_i8_prototype(args.sims, sizes=(args.games, args.saturated_games) if args.saturated_games else (args.games,)))
        if rank == 0:
            for name, obj in (("latency", latency), ("dense_i8_prototype", dense_i8)):
                if obj is not None:
                    full[name] = obj
            phases["total"] = round(time.perf_counter() - t_start, 2)
            write_detail(full)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
