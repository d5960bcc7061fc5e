#!/usr/bin/env python3
"""Runs one BASELINE.json config through the engine and prints throughput + workload shape (not the headline bench).
  python tools/run_config.py connect4   # configs[3]: Connect4 6x7, 8192 concurrent games, 200 sims/move
  python tools/run_config.py othello    # configs[1]
  python tools/run_config.py connect4 8192 4   # 4 x 8192 games through 8192 slots (finished slots are refilled)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from alphazero_amd import engine as E
from alphazero_amd.games.connect4 import Connect4Net
from alphazero_amd.games.othello import OthelloNet

which = sys.argv[1] if len(sys.argv) > 1 else "connect4"
torch.manual_seed(0)
if which == "connect4":
    gid, H, W, G, sims, net = 1, 6, 7, 8192, 200, Connect4Net(7, 6).eval()
else:
    gid, H, W, G, sims, net = 0, 8, 8, 4096, 100, OthelloNet(n=8).eval()
if len(sys.argv) > 2:
    G = int(sys.argv[2])
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # games of the timed run = waves x G through the G slots (finished slots are refilled)
hnet = net.to_hip(max_batch=G)
eng = E.SelfPlayEngine(gid, H, W, n_slots=G, n_sim=sims, net=hnet, seed=0, sample_capacity=waves * G * (72 if which != "connect4" else 43))
eng.run(G)  # warm-up wave
torch.cuda.synchronize()
t0 = time.perf_counter()
smp = eng.run(waves * G, first_game_id=G)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = eng.stats()
z = smp["z"].cpu().numpy(); meta = smp["meta"].cpu().numpy()
first = meta[:, 1] == 0
w = z[first] * meta[first, 2]
G *= waves
print(json.dumps({"config": which, "games": G, "sims": sims, "seconds": dt, "games_per_s": G / dt, "examples_per_s": len(z) / dt,
                  "plies_per_game": len(z) / G, "net_evals": st["net_evals"], "max_tree_nodes": st["max_nodes_used"],
                  "winner_+1/-1/draw": [int((w == 1).sum()), int((w == -1).sum()), int((w == 0).sum())]}))
