#!/usr/bin/env python3
"""Randomised differential test of the trainer loop (trainer.py:475-572 on the mirror: self-play on the engine -> device augmentation ->
SGD -> weight hand-off -> evaluation in the batched arena) against the oracle (test infrastructure; needs a GPU).  Per trial a random
game / board size / simulations / episodes / batch size / temperature schedule / engine slot count / evaluation opponent / seed; two
iterations.  The samples of iteration i must equal the oracle's self-play with the weights iteration i played with (iteration 1: the
TRAINED weights), and every evaluation the oracle's arena between the networks involved.
    python tools/fuzz_trainer.py [trials] [seed]"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd import base  # noqa: E402
from alphazero_amd.trainer import AlphaZeroTrainer  # noqa: E402
from oracle import oracle as O  # noqa: E402


def _np_sd(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items() if not k.endswith("num_batches_tracked")}


def run(trials, seed, verbose=True):
    from alphazero_amd.games.connect4 import Connect4Config
    from alphazero_amd.games.othello import OthelloConfig
    from alphazero_amd.games.tictactoe import TicTacToeConfig
    rng = np.random.default_rng(seed)
    base.DEFAULT_MODELS_PATH = tempfile.mkdtemp() + "/"
    bad = []
    for t in range(trials):
        game = str(rng.choice(["tictactoe", "connect4", "othello"]))
        sims = int(rng.integers(3, 14))
        tmax = int(rng.integers(0, 5))
        common = dict(simulations=sims, epochs=1, iterations=2, do_eval=True, device="cuda", save=False, save_checkpoints=False,
                      temp_max_step=tmax, temp_min_step=tmax + int(rng.choice([0, 1, 3])), data_augmentation=bool(rng.random() < 0.8),
                      eval_opponent=str(rng.choice(["random", "greedy", "mcts", "previous"])), eval_episodes=2 * int(rng.integers(1, 4)),
                      batch_size=int(rng.choice([16, 32])))
        if game == "tictactoe":
            cfg, dims, kind = TicTacToeConfig(episodes=int(rng.integers(12, 30)), **common), (O.TICTACTOE, 3, 3), "mlp"
        elif game == "connect4":
            cfg, dims, kind = Connect4Config(episodes=int(rng.integers(4, 12)), **common), (O.CONNECT4, 6, 7), "conv"
        else:
            n = int(rng.choice([6, 8]))
            cfg, dims, kind = OthelloConfig(board_size=n, episodes=int(rng.integers(3, 8)), **common), (O.OTHELLO, n, n), "conv"
        slots = int(rng.integers(1, cfg.episodes + 4))
        sd = int(rng.integers(0, 10**6))
        info = dict(game=game, dims=dims, sims=sims, episodes=cfg.episodes, slots=slots, seed=sd, opp=cfg.eval_opponent, temps=(cfg.temp_max_step, cfg.temp_min_step),
                    aug=cfg.data_augmentation, batch=cfg.batch_size)

        def onet(sdict):
            return (kind, O.MlpNet(sdict) if kind == "mlp" else O.ConvNet(dims[0], dims[1], dims[2], sdict))
        try:
            torch.manual_seed(int(rng.integers(0, 10**6)))
            tr = AlphaZeroTrainer(verbose=False, engine_slots=slots, seed=sd, materialize_memory=False)
            tr.game, tr.config = game, cfg
            tr.setup()
            ok = True
            for it in range(2):
                played_with = _np_sd(tr.nn)
                tr.self_play(it)
                got = {k: v.cpu().numpy() for k, v in tr.device_samples.items()}
                ref = O.selfplay(dims[0], dims[1], dims[2], cfg.episodes, sims, onet(played_with), seed=sd, first_game_id=cfg.episodes * it,
                                 temp_max_step=cfg.temp_max_step, temp_min_step=cfg.temp_min_step)
                ok = ok and len(got["z"]) == len(ref["z"]) and all(np.array_equal(got[k], ref[k]) for k in ("state", "pi", "z", "visits"))
                ok = ok and np.array_equal(got["meta"][:, 0] + cfg.episodes * it, ref["meta"][:, 0]) and np.array_equal(got["meta"][:, 1:], ref["meta"][:, 1:])
                tr.optimize_network(it)
                tr.update_network(it)
                tr.evaluate(it)
                opp = onet(played_with) if cfg.eval_opponent == "previous" else cfg.eval_opponent
                _, _, _, ost = O.arena_games(dims, onet(_np_sd(tr.nn)), sims, opp, sims, seed=sd + it, n_rounds=cfg.eval_episodes)
                want = {k: dict(v) for k, v in ost.items() if k not in ("player1", "player2", "draw")}
                ok = ok and tr.eval_results["results"][it] == want
                if not ok:
                    info["failed_at_iteration"] = it
                    break
            for h in (tr._engine, tr._hipnet):
                if h is not None:
                    h.close()
            if tr._hip_step is not None:
                tr._hip_step[1].close()
        except Exception as e:  # noqa: BLE001
            ok = False
            info["exception"] = repr(e)[:300]
        if not ok:
            bad.append(info)
            if verbose:
                print("MISMATCH", info, flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    mism = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"trainer fuzz: {n} trials, {len(mism)} mismatches")
    sys.exit(1 if mism else 0)
