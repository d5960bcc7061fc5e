#!/usr/bin/env python3
"""phase stamps of the training-step kernels (library built with `make -C alphazero_amd/csrc TPROBE=1`): us between the stamps of
workgroup 0 in the LAST launch of each stamped kernel.   python tools/probe_train.py [tag] [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import check_train_step as C  # noqa: E402
from alphazero_amd.train_step import HipTrainStep  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "othello8"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
net = C.make_net(tag).cuda()
state, pi, z = (t.cuda() for t in C.make_samples(net, 4096))
perm = torch.randint(0, 4096, (8 * B,), device="cuda", dtype=torch.int64)
hip = HipTrainStep(net, max_batch=B)
hip.load(net)
hip.begin(0.01, 0.9, 1e-4, 0.3, seed=1)
lp, lv = torch.zeros(8, device="cuda"), torch.zeros(8, device="cuda")
hip.steps(state, pi, z, perm, 8, B, lp, lv)
torch.cuda.synchronize()
raw = hip.debug("probe").view(torch.int32).cpu().numpy().view(np.uint64).reshape(32, 16)
ids = {0: "conv1_fwd", 1: "conv2_fwd", 2: "conv3_fwd", 3: "conv4_fwd", 4: "fc1_fwd", 5: "fc2_fwd", 6: "heads_fwd", 7: "heads_bwd (block 1)", 8: "fc_dgrad", 9: "mix1 wgrad (block 0)",
       10: "mix1 dgrad (first)", 11: "mix2 wgrad (block 0)", 12: "mix2 conv4 bwd (first)", 13: "conv2_bwd", 14: "conv3_bwd", 15: "conv1_bwd", 16: "update (block 0)", 17: "update (last block)"}
ev = sorted((int(raw[k][14]), int(raw[k][15]), nm) for k, nm in ids.items() if raw[k][14] > 0)
t0 = ev[0][0]
prev_end = t0
print("timeline of the last step (workgroup stamps, us): start, in-kernel time, gap to the previous end")
for b, e, nm in ev:
    print(f"  {nm:26s} start {(b - t0) / 100.0:8.2f}  body {(e - b) / 100.0:7.2f}  gap {(b - prev_end) / 100.0:7.2f}")
    prev_end = max(prev_end, e)
names = {1: "k_conv_fwd phases (last = conv4)", 2: "k_fc_fwd phases (last = fc2)", 3: "conv_bwd_body phases (last = conv2)"}
for k, nm in names.items():
    st = raw[k][:13]
    n = int((st > 0).sum())
    print(nm, " ".join(f"{(int(st[i + 1]) - int(st[i])) / 100.0:.2f}" for i in range(n - 1) if st[i + 1] >= st[i]), "us")
