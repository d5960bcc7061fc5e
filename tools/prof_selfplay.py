#!/usr/bin/env python3
"""A bounded piece of self-play for counter passes (rocprofv3 --pmc serialises the dispatches): G games from the start
position, `plies` plies of search + move, production mode.  usage: prof_selfplay.py [othello|connect4] G SIMS PLIES"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from alphazero_amd import engine as E  # noqa: E402
from alphazero_amd.games.connect4 import Connect4Board, Connect4Net  # noqa: E402
from alphazero_amd.games.othello import OthelloBoard, OthelloNet  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "othello"
G, sims, plies = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(0)
if game == "othello":
    gid, H, W, net, board = 0, 8, 8, OthelloNet(n=8).eval(), OthelloBoard(n=8)
else:
    gid, H, W, net, board = 1, 6, 7, Connect4Net(7, 6).eval(), Connect4Board(width=7, height=6)
hnet = net.to_hip(max_batch=G)
eng = E.SelfPlayEngine(gid, H, W, n_slots=G, n_sim=sims, net=hnet, seed=0)
eng.set_roots(np.tile(board.grid.astype(np.int8)[None], (G, 1, 1)), np.ones(G, np.int8))
for _ in range(plies):
    eng.search(sims)
    eng.advance()
torch.cuda.synchronize()
print("done", eng.stats())
