#!/usr/bin/env python3
"""Diagnostic (needs `make -C alphazero_amd/csrc -B PROBE=1`): k_gemm_solo's own clocks -- shader cycles per K tile of a wave
(ideal: 256 MFMAs x 64 = 16384) and the shader clock the kernel really ran at (s_memtime against the 100 MHz real-time counter)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alphazero_amd import _lib
from alphazero_amd.games.othello import OthelloNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
torch.manual_seed(0)
net = OthelloNet(n=8).eval().to_hip(max_batch=B)
L = _lib.lib()
L.az_debug_read_probe.argtypes = [C.c_void_p, C.c_int]
for stage, name in ((1, "fc1"), (2, "fc2")):
    us = net.time_stage(stage, B, 10) * 1e3
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert L.az_debug_read_probe(buf.ctypes.data, buf.size) == 0
    b = buf.reshape(-1, 8).astype(np.float64)
    b = b[b[:, 3] == (15 if stage == 1 else 31)]  # K / 32 - 1 loop tiles: drops what the other layer left in the buffer
    ghz = b[:, 0] / (b[:, 2] * 10.0)  # s_memtime ticks per ns of the 100 MHz real-time counter
    print("%s: %.1f us per launch; %d blocks; block %.0f cycles, loop %.0f cycles = %.0f per K tile (ideal 16384: %.3f); "
          "s_memtime / realtime = %.3f ticks per ns" % (name, us, len(b), b[:, 0].mean(), b[:, 1].mean(), (b[:, 1] / b[:, 3]).mean(),
                                                      16384.0 / (b[:, 1] / b[:, 3]).mean(), ghz.mean()))
os._exit(0)
