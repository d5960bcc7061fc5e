#!/usr/bin/env python3
"""Self-play rate and time per lock-step at small engine sizes (1 .. 4096 games), with HIP-graph replay of the searches
(default) or kernel-by-kernel launches (AZ_ENGINE_GRAPHS=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphazero_amd import engine as E
from alphazero_amd.games.othello import OthelloNet
torch.manual_seed(0)
net = OthelloNet(n=8).eval()
for G in (1, 64, 512, 4096):
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=G, n_sim=100, net=net.to_hip(max_batch=G), seed=0)
    eng.run(G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run(G, first_game_id=G)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = eng.stats()
    print("graphs", os.environ.get("AZ_ENGINE_GRAPHS", "1"), "G", G, "%.3f s" % dt, "%.1f games/s" % (G / dt), "us per lock-step %.1f" % (dt / st["lockstep_iters"] * 1e6), "replays", st["graph_replays"], flush=True)
