#!/usr/bin/env python3
"""Diagnostic (needs `make -C alphazero_amd/csrc -B PROBE=1`): shader-clock shares of the phases of k_trunk2 (per board pair)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alphazero_amd import _lib
from alphazero_amd.games.connect4 import Connect4Net
from alphazero_amd.games.othello import OthelloNet

game = sys.argv[1] if len(sys.argv) > 1 else "othello"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
torch.manual_seed(0)
net = (OthelloNet(n=8) if game == "othello" else Connect4Net(7, 6)).eval().to_hip(max_batch=B)
L = _lib.lib()
L.az_debug_read_probe.argtypes = [C.c_void_p, C.c_int]
print("trunk us", net.time_stage(0, B, 10) * 1e3)
buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
assert L.az_debug_read_probe(buf.ctypes.data, buf.size) == 0
b = buf.reshape(256 * 8, 8).astype(np.float64)
b = b[b[:, 5] > 0]
names = ["input staging", "conv1", "conv2", "conv3", "conv4 + store"]
per = b[:, :5] / b[:, 5:6]
tot = per.sum(1)
print("waves with work:", len(b), " pairs per wave: mean %.1f" % b[:, 5].mean())
for i, n in enumerate(names):
    print("  %-14s %8.0f cycles per pair  %5.1f %%" % (n, per[:, i].mean(), 100 * per[:, i].mean() / tot.mean()))
print("  total          %8.0f cycles per pair" % tot.mean())
os._exit(0)
