#!/usr/bin/env python3
"""us per launch of every network stage at several batch sizes (az_net_time_stage: back-to-back launches, HIP events).
usage: stage_times.py [othello|connect4] [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from alphazero_amd.games.connect4 import Connect4Net  # noqa: E402
from alphazero_amd.games.othello import OthelloNet  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "othello"
sizes = [int(x) for x in sys.argv[2:]] or [512, 2048, 3584, 4096, 8192, 16384, 32768]
torch.manual_seed(0)
net = (OthelloNet(n=8) if game == "othello" else Connect4Net(7, 6)).eval().to_hip(max_batch=max(sizes))
names = ["trunk", "fc1", "fc2", "heads"]
for B in sizes:
    t = [net.time_stage(s, B, 30) * 1e3 for s in range(4)]
    whole = net.time_stage(-1, B, 30) * 1e3
    print(f"{game} B={B:6d}  " + "  ".join(f"{n} {x:7.1f}" for n, x in zip(names, t)) + f"   forward {whole:7.1f} us", flush=True)
