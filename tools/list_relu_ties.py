#!/usr/bin/env python3
"""Which (case, seed) pairs of tests/test_gpu_train_step.py::test_training_step_equals_torch_autograd are off, and whether the float64
run has a ReLU input on the kink there (tools/check_train_step.py::relu_ties).  Needs a GPU.  Prints one JSON line per run."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import check_train_step as C  # noqa: E402

CASES = [("tictactoe", 16, 3, 0.0), ("tictactoe", 64, 3, 0.0), ("tictactoe", 250, 2, 0.0), ("othello8", 64, 3, 0.0), ("othello8", 64, 2, 0.3), ("othello6", 32, 2, 0.0),
         ("connect4", 32, 3, 0.0), ("connect4", 128, 2, 0.3), ("othello8", 256, 1, 0.0), ("othello8", 512, 1, 0.0), ("connect4", 512, 1, 0.0), ("othello8", 16, 2, 0.0),
         ("othello6", 48, 2, 0.3), ("connect4_8x5", 32, 2, 0.0), ("connect4_5x8", 64, 2, 0.3), ("othello8", 48, 2, 0.3), ("othello6", 80, 2, 0.0),
         ("connect4", 144, 2, 0.3), ("othello8", 320, 1, 0.3)]

if __name__ == "__main__":
    for case in CASES:
        for seed in (0, 1, 2):
            rows = C.report(*case, verbose=False, seed=seed)
            bad = [(n, e, s) for n, e, s in rows if e > 2e-4 * max(s, 1e-3) + 1e-6]
            worst = max(rows, key=lambda r: r[1] / max(r[2], 1e-3))
            print(json.dumps({"case": case, "seed": seed, "bad": len(bad), "first_bad": bad[:2], "ties": C.report.ties[:4], "worst": worst}), flush=True)
