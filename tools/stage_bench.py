"""Stage times (az_net_time_stage) of a conv net over row counts.  usage: python tools/stage_bench.py othello8|othello6|connect4 [rows ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from alphazero_amd.games.connect4 import Connect4Net
from alphazero_amd.games.othello import OthelloNet


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "connect4"
    sizes = [int(a) for a in sys.argv[2:]] or [1, 64, 512, 2048, 4096, 5573, 8192]
    torch.manual_seed(0)
    net = {"othello8": lambda: OthelloNet(n=8), "othello6": lambda: OthelloNet(n=6), "connect4": lambda: Connect4Net(7, 6)}[tag]().eval()
    hnet = net.to_hip(max_batch=max(sizes))
    print(tag, {k: v for k, v in os.environ.items() if k.startswith("AZ_")})
    print(f"{'rows':>6} {'trunk':>8} {'fc1/tail':>8} {'forward':>8}   us per launch; kernels")
    for B in sizes:
        t = [1e3 * hnet.time_stage(s, B, 200) for s in (0, 1, -1)]
        print(f"{B:>6} " + " ".join(f"{x:8.1f}" for x in t) + "   " + ", ".join(hnet.stage_kernel(s, B) for s in (0, 1)), flush=True)


if __name__ == "__main__":
    main()
