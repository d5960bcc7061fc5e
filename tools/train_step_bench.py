#!/usr/bin/env python3
"""ms per optimisation step of the hand-written training step (csrc/az_train.hip) on synthetic device-resident samples:
    python tools/train_step_bench.py [tag] [batch] [steps]        (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import check_train_step as C  # noqa: E402
from alphazero_amd.train_step import HipTrainStep  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "othello8"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
net = C.make_net(tag).cuda()
S = 1 << 16
state, pi, z = (t.cuda() for t in C.make_samples(net, S))
perm = torch.randint(0, S, (steps * B,), device="cuda", dtype=torch.int64)
hip = HipTrainStep(net, max_batch=B)
hip.load(net)
hip.begin(0.01, 0.9, 1e-4, 0.3, seed=1)
lp, lv = torch.zeros(steps, device="cuda"), torch.zeros(steps, device="cuda")
hip.steps(state, pi, z, perm, 20, B, lp, lv)  # warm-up (first step eager, then the captured graph)
torch.cuda.synchronize()
t0 = time.perf_counter()
hip.steps(state, pi, z, perm, steps, B, lp, lv)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fl = {"othello8": 4339712}.get(tag, 0)
print(f"{tag} batch {B}: {1e3 * dt / steps:.4f} ms per step ({steps} steps, host enqueue {1e3 * t_host / steps:.4f} ms per step)"
      + (f", {3 * fl * B * steps / dt / 1e12:.2f} TFLOP/s (3 x forward FLOPs)" if fl else "") + f"; last losses {lp[-1].item():.4f} {lv[-1].item():.4f}")
