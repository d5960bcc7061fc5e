#!/usr/bin/env python3
"""Randomised differential test of the single-game plugin surface -- alphazero_amd.mcts.MCT (tree on the GPU, one engine slot) driven
like a Player drives it (one search per ply, change_root with the move played: players.py:158-199, arena.py:70-99) -- against the oracle's
MCT (test infrastructure; needs a GPU).  Per trial a random game / network / noise parameters / seed; per ply a random number of
simulations and a random LEGAL move (often one the tree holds, sometimes one it does not: change_root then starts a fresh root,
mcts.py:124-125).  After every search the root's visit counts and priors must be the oracle's.
    python tools/fuzz_mct.py [trials] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd.mcts import MCT, _action_of  # noqa: E402
from oracle import oracle as O  # noqa: E402


def _np_sd(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items() if not k.endswith("num_batches_tracked")}


def run(trials, seed, verbose=True):
    from alphazero_amd.games.connect4 import Connect4Board, Connect4Net
    from alphazero_amd.games.othello import OthelloBoard, OthelloNet
    from alphazero_amd.games.tictactoe import TicTacToeBoard, TicTacToeNet
    rng = np.random.default_rng(seed)
    bad = []
    for t in range(trials):
        game = str(rng.choice(["tictactoe", "connect4", "othello"]))
        rollout = bool(rng.random() < 0.3)
        torch.manual_seed(int(rng.integers(0, 10**6)))
        if game == "tictactoe":
            board, dims, net = TicTacToeBoard(), (O.TICTACTOE, 3, 3), None if rollout else TicTacToeNet().eval()
            ev = None if rollout else ("mlp", O.MlpNet(_np_sd(net)))
        elif game == "connect4":
            board, dims, net = Connect4Board(width=7, height=6), (O.CONNECT4, 6, 7), None if rollout else Connect4Net(7, 6).eval()
            ev = None if rollout else ("conv", O.ConvNet(O.CONNECT4, 6, 7, _np_sd(net)))
        else:
            n = int(rng.choice([6, 8])) if not rollout else int(rng.choice([4, 6, 8]))
            board, dims, net = OthelloBoard(n=n), (O.OTHELLO, n, n), None if rollout else OthelloNet(n=n).eval()
            ev = None if rollout else ("conv", O.ConvNet(O.OTHELLO, n, n, _np_sd(net)))
        noisy = (not rollout) and rng.random() < 0.6
        alpha, eps = (0.3, 0.25) if noisy else (None, None)
        sd, npseed = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        info = dict(game=game, dims=dims, rollout=rollout, noisy=noisy, seed=sd)
        try:
            np.random.seed(npseed)
            gid = int(np.random.randint(0, 2**31 - 1))  # the game id MCT draws at its first search
            np.random.seed(npseed)
            mct = MCT(eval_method="rollout" if rollout else "neural", nn=net, dirichlet_alpha=alpha, dirichlet_epsilon=eps, seed=sd)
            ref = O.MCT(ev if ev else ("fake", None), eval_method=O.EVAL_ROLLOUT if rollout else O.EVAL_NEURAL, alpha=alpha if noisy else -1.0,
                        eps=eps if noisy else -1.0, tie_mode=O.TIE_RANDOM, noise_mode=O.NOISE_PHILOX if noisy else O.NOISE_OFF, seed=sd, game_id=gid)
            ob = O.new_board(*dims)
            ok, ply = True, 0
            while not board.is_game_over() and ply < 40:
                n_sim = int(rng.integers(2, 30))
                mct.search(board, n_sim=n_sim)
                ref.set_ply(ply)
                ref.search(ob, n_sim)
                a, N, Q, P = ref.root_children()
                _, visits = mct.get_action_probs(board, temp=1)
                want = {int(x): int(c) for x, c in zip(a, N)}
                have = {_action_of(board, m): int(c) for m, c in visits.items()}
                ok = have == want
                if ok and not rollout:
                    pri = mct.get_prior_probs()
                    ok = {_action_of(board, m): float(p) for m, p in pri.items()} == {int(x): float(p) for x, p in zip(a, P)}
                if not ok:
                    info["ply"] = ply
                    break
                legal = board.get_moves()
                move = legal[int(rng.integers(0, len(legal)))] if rng.random() < 0.5 else max(visits, key=visits.get)
                act = _action_of(board, move)
                mct.change_root(move)
                board.play_move(move)
                ref.change_root(act)
                if O.lib().orc_play(O.C.byref(ob), act) != 0:
                    raise RuntimeError("oracle refused a move the mirror board accepted")
                ply += 1
            if mct._engine is not None:
                mct._engine.close()
        except Exception as e:  # noqa: BLE001
            ok = False
            info["exception"] = repr(e)[:300]
        if not ok:
            bad.append(info)
            if verbose:
                print("MISMATCH", info, flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    mism = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"MCT fuzz: {n} trials, {len(mism)} mismatches")
    sys.exit(1 if mism else 0)
