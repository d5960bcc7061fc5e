"""CPU baseline worker (test infrastructure; started by bench.py's cpu_baseline leg only): plays n Othello 8x8 self-play
games with the C oracle (one thread), timing every game like SelfPlayTimer.timeit (timers.py:53-76), prints one JSON line.
usage: python oracle/selfplay_worker.py WEIGHTS.npz N_GAMES N_SIM FIRST_GAME_ID"""
import json
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != _HERE]  # run as a script, oracle/ itself would shadow the package
sys.path.insert(0, os.path.dirname(_HERE))
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402


def play(weights, n_games, n_sim, first_id):
    net = O.ConvNet(O.OTHELLO, 8, 8, weights)
    secs, plies, evals = [], 0, 0
    for g in range(n_games):
        t0 = time.perf_counter()
        r = O.selfplay(O.OTHELLO, 8, 8, 1, n_sim, ("conv", net), seed=0, first_game_id=first_id + g)
        secs.append(time.perf_counter() - t0)
        plies += len(r["z"]); evals += r["n_evals"]
    return {"seconds_per_game": secs, "plies": plies, "net_evals": evals}


if __name__ == "__main__":
    w = dict(np.load(sys.argv[1]))
    print(json.dumps(play(w, int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))), flush=True)
