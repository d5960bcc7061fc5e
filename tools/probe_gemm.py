#!/usr/bin/env python3
"""Diagnostic (needs `make -C alphazero_amd/csrc -B PROBE=1`): per-phase shader-clock shares inside k_gemm."""
import ctypes as C
import os
import sys

os.environ.setdefault("AZ_GEMM_SOLO", "0")  # this probe reads k_gemm's stamps (k_gemm_solo: tools/probe_gemm_solo.py)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alphazero_amd import _lib
from alphazero_amd.games.othello import OthelloNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
net = OthelloNet(n=8).eval().to_hip(max_batch=B)
L = _lib.lib()
L.az_debug_read_probe.argtypes = [C.c_void_p, C.c_int]
for stage, nblk in ((1, min(8192, (B // 128) * 8 if B >= 16384 else 512)), (2, min(8192, (B // 128) * 4 if B >= 16384 else 512))):
    print("stage", stage, "time us", net.time_stage(stage, B, 10) * 1e3)
    buf = np.zeros(nblk * 8, dtype=np.uint64)
    assert L.az_debug_read_probe(buf.ctypes.data, buf.size) == 0
    b = buf.reshape(nblk, 8)
    names = ["issue", "compute", "store", "barrier", "total"]
    print("  mean cycles per block:", {n: int(b[:, i].astype(np.float64).mean()) for i, n in enumerate(names)})
    st, en = b[:, 5].astype(np.int64), b[:, 6].astype(np.int64)
    t0 = st.min()
    print("  starts (cycles after first): p50 %d p90 %d max %d | ends: min %d p50 %d max %d" % (
        np.percentile(st - t0, 50), np.percentile(st - t0, 90), (st - t0).max(), (en - t0).min(), np.percentile(en - t0, 50), (en - t0).max()))
    rt = b[:, 7].astype(np.float64)
    print("  block wall us (100 MHz counter): mean %.1f max %.1f -> shader clock %.2f GHz" % (rt.mean() / 100, rt.max() / 100, (b[:, 4].astype(np.float64) / rt).mean() * 0.1))
os._exit(0)
