#!/usr/bin/env python3
"""stage times of fc1 / fc2 at a few batch sizes (AZ_DENSE_I8, AZ_QG_CFG from the environment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd.games.othello import OthelloNet
torch.manual_seed(0)
net = OthelloNet(n=8).eval()
hip = net.to_hip(max_batch=32768)
out = []
for B in (1, 16, 64, 256, 512, 1024, 4096, 8192, 32768):
    t = [1e3 * hip.time_stage(s, B, iters=30) for s in (1, 2)]
    out.append(f"{B}: {t[0]:.1f}/{t[1]:.1f}")
print(f"I8={os.environ.get('AZ_DENSE_I8','0')} cfg={os.environ.get('AZ_QG_CFG','0')} fc1/fc2 us  " + "  ".join(out))
