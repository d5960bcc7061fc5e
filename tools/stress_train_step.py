#!/usr/bin/env python3
"""Randomised stress of the hand-written training step: 10 network / batch / dropout configurations x 6 data seeds against torch autograd in
float64 (tools/check_train_step.py).  A configuration above the bound is printed with its first buffers; a lone ReLU tie between float32
and float64 shows as ONE element of one dy buffer off at step 0 and errors everywhere from step 1 on (run check_train_step.py on the case
to see where the error sits).  Last run (round 4, MI355X, profiles/r04_stress_train.txt): 60 cases, 3 above the bound, each a tie in a conv layer -- the one on the row-split
path (othello8, batch 512, seed 6) shows the same five buffers with the same errors unsplit and with row blocks of 64 / 128 (AZ_TRAIN_RB)."""
import sys, os
sys.path.insert(0, "tools")
import check_train_step as C
bad_total = 0
cases = [("othello8", 64, 2, 0.3), ("othello8", 128, 2, 0.0), ("othello6", 32, 2, 0.3), ("connect4", 64, 2, 0.3), ("connect4_8x5", 48, 2, 0.0),
         ("othello8", 256, 1, 0.3), ("othello8", 512, 1, 0.3), ("tictactoe", 32, 3, 0.0), ("othello8", 16, 2, 0.3), ("connect4", 512, 1, 0.0)]
for seed in range(3, 9):
    for tag, B, steps, p in cases:
        rows = C.report(tag, B, steps, p, verbose=False, seed=seed)
        bad = [(n, e, s) for n, e, s in rows if e > 2e-4 * max(s, 1e-3) + 1e-6]
        if bad:
            bad_total += 1
            print("BAD", tag, B, steps, p, "seed", seed, len(bad), bad[:3], flush=True)
print("cases", 6 * len(cases), "with errors above the bound:", bad_total)
