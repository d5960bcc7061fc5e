#!/usr/bin/env python3
"""Randomised differential test of the HIP self-play engine against the CPU oracle (test infrastructure; needs a GPU): random game /
board size / slot count / game count / simulations / Dirichlet parameters / temperature schedule (incl. fractional temperatures) /
tie and noise modes / evaluation method / seed per trial; every sample array must be bit-equal.
    python tools/fuzz_engine.py [trials] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd import engine as E  # noqa: E402
from oracle import oracle as O  # noqa: E402


def sort_samples(d):
    """by (game id as uint32, ply): the engine emits samples in lock-step order"""
    d = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in d.items() if k != "n_evals"}
    order = np.lexsort((d["meta"][:, 1], d["meta"][:, 0].astype(np.int64) & 0xFFFFFFFF))
    return {k: v[order] for k, v in d.items()}


def run(trials, seed, verbose=True):
    """-> the configurations whose samples differ between engine and oracle (empty: all equal)"""
    rng = np.random.default_rng(seed)
    bad = []
    for t in range(trials):
        game = int(rng.choice([0, 0, 1, 1, 2]))
        if game == 0:
            H = W = int(rng.choice([4, 6, 6, 8]))
        elif game == 1:
            H, W = int(rng.integers(4, 9)), int(rng.integers(4, 9))
        else:
            H = W = 3
        slots = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 31, 48, 64, 70]))
        n_games = int(rng.integers(1, 3 * slots + 1))
        rollout = bool(rng.random() < 0.15)
        n_sim = int(rng.integers(3, 50 if not rollout else 25))
        plies = {0: H * W, 1: H * W * 2 // 3, 2: 8}[game]
        n_games = max(1, min(n_games, 60000 // (n_sim * plies)))  # the oracle plays these on one CPU core: ~60 K simulations per trial
        noisy = (not rollout) and rng.random() < 0.7
        alpha, eps = (float(rng.choice([0.03, 0.3, 1.0])), float(rng.choice([0.25, 0.5, 1.0]))) if noisy else (-1.0, -1.0)
        tmax = int(rng.integers(-1, 8))
        tmin = tmax + int(rng.choice([0, 1, 2, 3, 5, 9]))
        tie = int(rng.choice([O.TIE_LOWEST, O.TIE_RANDOM]))
        noise = (int(rng.choice([O.NOISE_PHILOX, O.NOISE_HASH])) if noisy else O.NOISE_OFF)
        seed, first = int(rng.integers(0, 2**32)), int(rng.integers(0, 10**6))
        if rng.random() < 0.06:
            first = 2**32 - int(rng.integers(1, 6))  # the game ids of the trial wrap around 2^32
        elif rng.random() < 0.06:
            first = 2**31 - int(rng.integers(1, 6))  # ... or cross the sign bit of the int32 meta column
        cfg = dict(game=game, H=H, W=W, slots=slots, n_games=n_games, n_sim=n_sim, alpha=alpha, eps=eps, tmax=tmax, tmin=tmin, tie=tie, noise=noise,
                   rollout=rollout, seed=seed, first=first)
        try:
            eng = E.SelfPlayEngine(game, H, W, n_slots=slots, n_sim=n_sim, dirichlet_alpha=alpha if noisy else None, dirichlet_epsilon=eps if noisy else None,
                                   temp_max_step=tmax, temp_min_step=tmin, tie_mode=tie, noise_mode=noise,
                                   evaluator=E.EVAL_ROLLOUT if rollout else E.EVAL_FAKE, seed=seed, node_capacity=1 << 16,
                                   sample_capacity=n_games * (2 * H * W + 8), max_plies=2 * H * W + 8)
            got = sort_samples(eng.run(n_games, first_game_id=first))
            st = eng.stats()
            eng.close()
            ref = O.selfplay(game, H, W, n_games, n_sim, ("fake", None), alpha=alpha, eps=eps, temp_max_step=tmax, temp_min_step=tmin, tie_mode=tie,
                             noise_mode=noise, seed=seed, first_game_id=first, eval_method=O.EVAL_ROLLOUT if rollout else O.EVAL_NEURAL)
            n_evals = ref["n_evals"]
            ref = sort_samples(ref)
            ok = len(got["z"]) == len(ref["z"]) and all(np.array_equal(got[k], ref[k]) for k in ("state", "z", "meta", "visits", "pi"))
            ok = ok and st["games_done"] == n_games and (rollout or st["net_evals"] == n_evals)
        except Exception as e:  # noqa: BLE001
            ok = False
            cfg["exception"] = repr(e)[:200]
        if not ok:
            bad.append(cfg)
            if verbose:
                print("MISMATCH", cfg, flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    mism = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"fuzz: {n} trials, {len(mism)} mismatches")
    sys.exit(1 if mism else 0)
