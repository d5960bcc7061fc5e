#!/usr/bin/env python3
"""Diagnostic (needs `make -C alphazero_amd/csrc -B PROBE=1`): phase shares inside k_step (simulation 50 of the last ply played)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alphazero_amd import _lib, engine as E
from alphazero_amd.games.othello import OthelloNet

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
net = OthelloNet(n=8).eval().to_hip(max_batch=G)
eng = E.SelfPlayEngine(0, 8, 8, n_slots=G, n_sim=100, net=net, seed=0, max_plies=128, sample_capacity=G * 72)
eng.set_roots(np.tile(np.array([[0]*27+[1,-1]+[0]*6+[-1,1]+[0]*27], np.int8), (G, 1)), np.ones(G, np.int8))
for ply in range(12):
    eng.search(100)
    eng.advance()
L = _lib.lib()
L.az_debug_read_step_probe.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(1024 * 8, dtype=np.uint64)
assert L.az_debug_read_step_probe(buf.ctypes.data, buf.size) == 0
b = buf.reshape(1024, 8).astype(np.int64)
d = np.diff(b[:, :7], axis=1)
names = ["top loads", "create children", "backprop+fence", "root+noise", "walk", "status+leaf write"]
print("mean cycles per phase (wave 0..3 of every block):")
for i, n in enumerate(names):
    print("  %-18s mean %7.0f  p90 %7.0f  max %7.0f" % (n, d[:, i].mean(), np.percentile(d[:, i], 90), d[:, i].max()))
print("  total mean %.0f  max %.0f   depth(lane0 game) mean %.2f" % ((b[:, 6] - b[:, 0]).mean(), (b[:, 6] - b[:, 0]).max(), b[:, 7].mean()))
os._exit(0)
