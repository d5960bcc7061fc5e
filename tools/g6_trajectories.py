#!/usr/bin/env python3
"""Golden G6 on the GPU, step by step: |hand-written step - float64|, |reference float32 (fixture) - float64| and, for the stock
PyTorch step on the GPU, |torch GPU - float64| -- the data behind the bound in tests/test_gpu_train_step.py.  Needs a GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train_step as T  # noqa: E402

for tag in sys.argv[1:] or ["othello6", "othello8"]:
    out = {}
    for backend in ("hip", "torch"):
        tr, fx = T._fixture_trainer(tag, backend)
        tr.graph_sgd = False
        tr.optimize_network(0)
        out[backend] = tr.loss_values[0]
    f64 = T._float64_trajectory(tag, fx)
    for k in ("pi", "v"):
        for e in range(int(fx["epochs"])):
            ex = np.array(f64[e][k])
            print(tag, k, "epoch", e)
            print("  ref f32 - f64 :", " ".join(f"{x:.1e}" for x in np.abs(fx[f"{k}_loss_{e}"] - ex)))
            print("  hip     - f64 :", " ".join(f"{x:.1e}" for x in np.abs(np.array(out["hip"][e][k]) - ex)))
            print("  torchGPU- f64 :", " ".join(f"{x:.1e}" for x in np.abs(np.array(out["torch"][e][k]) - ex)))
