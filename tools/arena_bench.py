"""Times BatchedArena.play_games (AlphaZero player vs another network / rollout MCTS) with the two players' searches overlapped and
one after the other.  usage: python tools/arena_bench.py [game] [rounds] [n_sim] [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from alphazero_amd.arena import BatchedArena
from alphazero_amd.games.registers import CONFIGS_REGISTER, NETWORKS_REGISTER


def main():
    game = sys.argv[1] if len(sys.argv) > 1 else "othello"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n_sim = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    cfg = CONFIGS_REGISTER[game](**({"board_size": 8} if game == "othello" else {}))  # BASELINE's Othello is 8x8 (the class default is 6x6)
    nets = []
    for seed in (1, 2):
        torch.manual_seed(seed)
        nets.append(NETWORKS_REGISTER[game](config=cfg).to("cuda").eval())
    for opp in (nets[1], "mcts"):
        stats = {}
        for overlap in (True, False) * (int(sys.argv[4]) if len(sys.argv) > 4 else 2):
            ar = BatchedArena(game, nets[0], opponent=opp, n_sim=n_sim, seed=3)
            ar.overlap = overlap
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = ar.play_games(rounds, shard=False)
            dt = time.perf_counter() - t0
            stats[overlap] = (len(st["player1"]), len(st["player2"]), st["draw"])
            print(f"{game} {rounds} rounds @{n_sim} vs {'mcts' if isinstance(opp, str) else 'network'} overlap={overlap}: {dt:.3f} s  {stats[overlap]}", flush=True)
        assert stats[True] == stats[False]


if __name__ == "__main__":
    main()
