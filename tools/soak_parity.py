#!/usr/bin/env python3
"""One-off parity soak at the BASELINE simulation counts (longer than the unit tests allow: the oracle needs ~2.5 s per
Othello game): whole self-play runs with the real network, HIP engine vs CPU oracle, every sample compared bit for bit.
  python tools/soak_parity.py [othello_games] [connect4_games]       (last run: 48 / 96 games, BIT-EQUAL)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402  (test infrastructure: this tool is a checker, not product code)
from alphazero_amd import engine as E  # noqa: E402
from alphazero_amd.games.connect4 import Connect4Net  # noqa: E402
from alphazero_amd.games.othello import OthelloNet  # noqa: E402
from test_gpu_engine import sort_samples  # noqa: E402

n_oth = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n_c4 = int(sys.argv[2]) if len(sys.argv) > 2 else 96
for name, gid, H, W, make, games, sims in (("othello8", 0, 8, 8, lambda: OthelloNet(n=8), n_oth, 100),
                                           ("connect4", 1, 6, 7, lambda: Connect4Net(7, 6), n_c4, 200)):
    torch.manual_seed(1)
    net = make().eval()
    sd = {k: v.numpy() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    onet = O.ConvNet(gid, H, W, sd)
    eng = E.SelfPlayEngine(gid, H, W, n_slots=games, n_sim=sims, net=net.to_hip(max_batch=games), seed=77)
    got = sort_samples(eng.run(games, first_game_id=123))
    t0 = time.time()
    ref = O.selfplay(gid, H, W, games, sims, ("conv", onet), seed=77, first_game_id=123)
    ok = all(np.array_equal(got[k], ref[k]) for k in ("state", "z", "meta", "visits", "pi"))
    print(name, "games", games, "sims", sims, "samples", len(ref["z"]), "oracle %.0f s" % (time.time() - t0),
          "BIT-EQUAL" if ok else "MISMATCH", flush=True)
    if not ok:
        sys.exit(1)
