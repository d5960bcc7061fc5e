#!/usr/bin/env python3
"""a few forwards at one batch size (for rocprofv3 --kernel-trace --stats): python tools/qdense_prof.py B [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd.games.othello import OthelloNet
B = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.manual_seed(0)
net = OthelloNet(n=8).eval()
hip = net.to_hip(max_batch=B)
x = torch.randint(-1, 2, (B, 64), device="cuda").float()
for _ in range(iters):
    hip.forward(x)
torch.cuda.synchronize()
