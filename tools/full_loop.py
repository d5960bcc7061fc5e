#!/usr/bin/env python3
"""BASELINE config 5: one full AlphaZeroTrainer iteration on one GPU with per-phase wall times
(self-play on the engine -> symmetry augmentation on the device -> SGD on device-resident batches -> weight hand-off
-> batched arena evaluation).   python tools/full_loop.py [episodes] [epochs] [batch_size] [eval_opponent] [eval_episodes] [iterations]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from alphazero_amd import base
from alphazero_amd.games.othello import OthelloConfig
from alphazero_amd.trainer import AlphaZeroTrainer

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 512
opp = sys.argv[4] if len(sys.argv) > 4 else "mcts"
eval_eps = int(sys.argv[5]) if len(sys.argv) > 5 else 64
base.DEFAULT_MODELS_PATH = tempfile.mkdtemp() + "/"
torch.manual_seed(0)
tr = AlphaZeroTrainer(verbose=False, engine_slots=min(episodes, 32768), seed=0, materialize_memory=False)
tr.game = "othello"
tr.config = OthelloConfig(board_size=8, simulations=100, episodes=episodes, epochs=epochs, batch_size=batch, iterations=1,
                          device="cuda", eval_opponent=opp, eval_episodes=eval_eps, do_eval=True, save=False, save_checkpoints=False)
tr.setup()
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 2
report = []
for it in range(iters):
    phases = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        phases[name] = round(time.perf_counter() - t0, 3)

    timed("self_play+augmentation", lambda: tr.self_play(it))
    n_samples = int(tr.device_memory["z"].shape[0])
    timed("optimize_network", lambda: tr.optimize_network(it))
    timed("update_network", lambda: tr.update_network(it))
    timed("evaluate", lambda: tr.evaluate(it))
    report.append({"iteration": it, "samples_with_twins": n_samples, "sgd_steps": epochs * (n_samples // batch), "seconds": phases,
                   "iteration_seconds": round(sum(phases.values()), 3), "eval_results": tr.eval_results["results"][it]})
print(json.dumps({"config": "othello 8x8 full loop (iteration 0 includes MIOpen's one-off algorithm search)", "episodes": episodes, "epochs": epochs,
                  "batch_size": batch, "eval": f"{eval_eps} games vs {opp}", "iterations": report}))
