#!/usr/bin/env python3
"""Runs each stage of the HIP network forward a few times at one batch size (for rocprofv3 --kernel-trace / --pmc runs).
usage: prof_net.py [othello|connect4] [BATCH] [ITERS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from alphazero_amd.games.connect4 import Connect4Net  # noqa: E402
from alphazero_amd.games.othello import OthelloNet  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "othello"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
net = (OthelloNet(n=8) if game == "othello" else Connect4Net(7, 6)).eval().to_hip(max_batch=B)
cells = 64 if game == "othello" else 42
x = torch.randint(-1, 2, (B, cells), device="cuda").float()
for _ in range(iters):
    p, v = net.forward(x)
torch.cuda.synchronize()
print({s: round(net.time_stage(s, B, 20) * 1e3, 1) for s in range(4)}, "us per stage")
