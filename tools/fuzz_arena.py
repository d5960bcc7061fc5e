#!/usr/bin/env python3
"""Randomised differential test of BatchedArena (two device trees per game, side-masked searches, device random / greedy movers) against
the oracle's arena (test infrastructure; needs a GPU): random game / board size / players (closed-form network, rollout MCTS) /
opponents (random, greedy, rollout MCTS, another tree on the closed-form network) / rounds / start player / simulations / tie mode /
seed per trial; every move of every game, the winners and the stats dict must be equal.
    python tools/fuzz_arena.py [trials] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd import engine as E  # noqa: E402
from alphazero_amd.arena import BatchedArena  # noqa: E402
from oracle import oracle as O  # noqa: E402


def run(trials, seed, verbose=True):
    rng = np.random.default_rng(seed)
    bad = []
    for t in range(trials):
        game = str(rng.choice(["othello", "othello", "connect4", "connect4", "tictactoe"]))
        kw, dims = {}, None
        if game == "othello":
            n = int(rng.choice([4, 6, 6, 8]))
            kw, dims = dict(board_size=n), (O.OTHELLO, n, n)
        elif game == "connect4":
            w, h = int(rng.integers(4, 9)), int(rng.integers(4, 9))
            kw, dims = dict(board_width=w, board_height=h), (O.CONNECT4, h, w)
        else:
            dims = (O.TICTACTOE, 3, 3)
        opponent = str(rng.choice(["random", "greedy", "mcts", "fake"]))
        start = [None, 1, 2][int(rng.integers(0, 3))]
        n_rounds = int(rng.integers(1, 13)) * (2 if start is None else 1)  # without a start player the reference wants an even number
        n_sim, opp_sim = int(rng.integers(2, 26)), int(rng.integers(2, 26))
        tie = int(rng.choice([O.TIE_LOWEST, O.TIE_RANDOM]))
        sd = int(rng.integers(0, 10**6))
        cfg = dict(game=game, dims=dims, opponent=opponent, start=start, n_rounds=n_rounds, n_sim=n_sim, opp_sim=opp_sim, tie=tie, seed=sd)
        try:
            ar = BatchedArena(game, "fake", opponent=opponent, n_sim=n_sim, opponent_n_sim=opp_sim, seed=sd, **kw)
            ar.tie_mode = E.TIE_LOWEST if tie == O.TIE_LOWEST else None
            ar.overlap = bool(rng.random() < 0.5)
            stats = ar.play_games(n_rounds, start_player=start, return_stats=True, record_moves=True, shard=False)
            got = [[int(m[g]) for m in ar.moves if m[g] >= 0] for g in range(n_rounds)]
            opp = ("fake", None) if opponent == "fake" else opponent
            moves, winners, scores, ost = O.arena_games(dims, ("fake", None), n_sim, opp, opp_sim, sd, n_rounds, start_player=start, tie_mode=tie)
            ok = got == moves and stats["draw"] == ost["draw"] and sorted(stats["player1"]) == sorted(ost["player1"]) \
                and sorted(stats["player2"]) == sorted(ost["player2"]) and dict(stats["player1_starts"]) == dict(ost["player1_starts"]) \
                and dict(stats["player2_starts"]) == dict(ost["player2_starts"])
        except Exception as e:  # noqa: BLE001
            ok = False
            cfg["exception"] = repr(e)[:300]
        if not ok:
            bad.append(cfg)
            if verbose:
                print("MISMATCH", cfg, flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    mism = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"arena fuzz: {n} trials, {len(mism)} mismatches")
    sys.exit(1 if mism else 0)
