#!/usr/bin/env python3
"""Experiment: P independent engines (own stream, own network scratch) on one GPU, driven from P host threads, against
one engine with the same total number of concurrent games.   python tools/two_engines.py TOTAL_GAMES P"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from alphazero_amd import engine as E
from alphazero_amd.games.othello import OthelloNet

total, P = int(sys.argv[1]), int(sys.argv[2])
G = total // P
torch.manual_seed(0)
model = OthelloNet(n=8).eval()
engs, streams = [], []
for p in range(P):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        hnet = model.to_hip(max_batch=G)
        engs.append(E.SelfPlayEngine(0, 8, 8, n_slots=G, n_sim=100, net=hnet, seed=0))
    streams.append(st)


def work(p, first):
    with torch.cuda.stream(streams[p]):
        engs[p].run(G, first_game_id=first + p * G)


for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(p, rep * total)) for p in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"P={P} x {G} games: {dt:.3f} s, {total / dt:.1f} games/s")
os._exit(0)
