#!/usr/bin/env python3
"""Race screen of the hand-synchronised LDS pipeline of k_qgemm (AZ_DENSE_I8=1): the same batch forwarded again and again must give the
same bits -- LDS-DMA data read before it has landed shows up as a rare wrong tile, not as a crash.
    AZ_DENSE_I8=1 python tools/qdense_race_screen.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_amd.games.othello import OthelloNet
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
assert os.environ.get("AZ_DENSE_I8") == "1"
torch.manual_seed(0)
bad = 0
for n, B in ((8, 700), (8, 4096), (8, 8200), (6, 5000), (8, 32768)):
    net = OthelloNet(n=n).eval()
    hip = net.to_hip(max_batch=B)
    x = torch.randint(-1, 2, (B, n * n), device="cuda").float()
    p0, v0 = hip.forward(x)
    p0, v0 = p0.clone(), v0.clone()
    r = reps if B < 20000 else max(20, reps // 10)
    same = 0
    for _ in range(r):
        p1, v1 = hip.forward(x)
        same += int(torch.equal(p0, p1) and torch.equal(v0, v1))
    print(f"othello{n} B {B}: {same} of {r} repeated forwards bit-equal to the first ({hip.stage_kernel(1, B)})", flush=True)
    bad += r - same
    hip.close()
print("race screen:", "clean" if bad == 0 else f"{bad} DIFFERENT RESULTS")
sys.exit(1 if bad else 0)
