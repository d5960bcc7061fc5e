"""Wall time of the reference's own call pattern on the GPU path: Arena.play_game with two AlphaZeroPlayers (one game, one tree each,
players.py:158-191 / arena.py:61-117) -- per get_move, everything included (tree sync, search, move choice, re-rooting both trees)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from alphazero_amd.arena import Arena
from alphazero_amd.games.othello import OthelloBoard, OthelloConfig, OthelloNet
from alphazero_amd.players import AlphaZeroPlayer


def main():
    n_sim = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    torch.manual_seed(0)
    nets = [OthelloNet(n=8).to("cuda").eval() for _ in range(2)]
    players = [AlphaZeroPlayer(n_sim=n_sim, nn=nets[i]) for i in range(2)]
    for rep in range(3):
        board = OthelloBoard(n=8)
        arena = Arena(players[0], players[1], board)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        arena.play_game()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        plies = int((board.grid != 0).sum()) - 4  # stones placed (passes not counted)
        print(f"game {rep}: {dt:.3f} s, {1e3 * dt / max(1, plies):.2f} ms per move ({plies} moves, {n_sim} sims each)", flush=True)


if __name__ == "__main__":
    main()
