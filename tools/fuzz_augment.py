#!/usr/bin/env python3
"""Randomised differential test of the device symmetry augmentation (csrc/az_augment.hip) against the host mirror of the reference's
augmentation (alphazero_amd.trainer.augment: trainer.py:80-118, 275-284; pinned to the reference's memory by golden G4 on the CPU):
random game / board size / sample count / states / policies / move indices (samples below move 2 have no twins); the twins must come
out in the same order with the same states, policies, outcomes and transformation tags.  Needs a GPU.
    python tools/fuzz_augment.py [trials] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd import engine as E  # noqa: E402
from alphazero_amd.games.registers import DATA_AUGMENT_STRATEGIES  # noqa: E402
from alphazero_amd.trainer import Sample, augment  # noqa: E402


def run(trials, seed, verbose=True):
    from alphazero_amd.games.connect4 import Connect4Net
    from alphazero_amd.games.othello import OthelloNet
    from alphazero_amd.games.tictactoe import TicTacToeNet
    rng = np.random.default_rng(seed)
    nets = {}
    bad = []
    for t in range(trials):
        game = str(rng.choice(["othello", "connect4", "tictactoe"]))
        if game == "othello":
            n = int(rng.choice([6, 8]))
            gid, H, W, key = 0, n, n, ("othello", n)
            make = lambda: OthelloNet(n=n)  # noqa: E731
        elif game == "connect4":
            w, h = int(rng.integers(5, 9)), int(rng.integers(5, 9))
            gid, H, W, key = 1, h, w, ("connect4", w, h)
            make = lambda: Connect4Net(w, h)  # noqa: E731
        else:
            gid, H, W, key = 2, 3, 3, ("tictactoe",)
            make = TicTacToeNet
        if key not in nets:
            nets[key] = make()
        nn = nets[key]
        A = nn.action_size
        S = int(rng.choice([0, 1, 2, 7, 33, 64, 129, 400]))
        state = rng.integers(-1, 2, (S, H, W)).astype(np.int8)
        pi = rng.random((S, A)).astype(np.float32)
        z = rng.integers(-1, 2, S).astype(np.int8)
        meta = np.stack([rng.integers(0, 50, S), rng.integers(0, 6, S), np.ones(S, np.int64), rng.integers(0, A, S)], axis=1).astype(np.int32)
        cfg = dict(game=game, H=H, W=W, S=S)
        try:
            dev = lambda a: torch.as_tensor(a, device="cuda")  # noqa: E731
            tw = E.augment_samples(gid, H, W, {"state": dev(state), "pi": dev(pi), "z": dev(z), "meta": dev(meta)})
            mem = [Sample(state=state[i].astype(np.float64), pi=pi[i].astype(np.float64), player=1, outcome=int(z[i]), episode_idx=int(meta[i, 0]),
                          move_idx=int(meta[i, 1])) for i in range(S)]
            host = augment(mem, nn, DATA_AUGMENT_STRATEGIES[game])
            ok = len(host) == tw["z"].shape[0]
            if ok and host:
                ts, tp, tz, tm = (tw[k].cpu().numpy() for k in ("state", "pi", "z", "meta"))
                ok = np.array_equal(ts, np.array([s.state for s in host]).astype(np.int8)) \
                    and np.array_equal(tp, np.array([s.pi for s in host]).astype(np.float32)) \
                    and np.array_equal(tz, np.array([s.outcome for s in host]).astype(np.int8)) \
                    and [E.TRANSFORM_NAMES[c] for c in tm[:, 3]] == [s.transformation for s in host] \
                    and np.array_equal(tm[:, 0], [s.episode_idx for s in host]) and np.array_equal(tm[:, 1], [s.move_idx for s in host])
        except Exception as e:  # noqa: BLE001
            ok = False
            cfg["exception"] = repr(e)[:300]
        if not ok:
            bad.append(cfg)
            if verbose:
                print("MISMATCH", cfg, flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    mism = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"augmentation fuzz: {n} trials, {len(mism)} mismatches")
    sys.exit(1 if mism else 0)
