"""Stage times (HIP events around back-to-back launches, az_net_time_stage) of the OthelloNet 8x8 forward over row counts: which dense
kernel serves which size.  usage: python tools/dense_bench.py [rows ...]      AZ_DENSE_FRAG_MAX=0 shows the kernels without k_dense_frag"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from alphazero_amd.games.othello import OthelloNet


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1, 16, 64, 128, 256, 512, 1024, 2048, 3072, 4096, 8192]
    torch.manual_seed(0)
    net = OthelloNet(n=8).eval()
    hnet = net.to_hip(max_batch=max(sizes))
    print(f"AZ_DENSE_FRAG_MAX={os.environ.get('AZ_DENSE_FRAG_MAX', '(default)')}")
    print(f"{'rows':>6} {'trunk':>8} {'fc1':>8} {'fc2':>8} {'heads':>8} {'forward':>8}   us per launch; kernels")
    for B in sizes:
        t = [1e3 * hnet.time_stage(s, B, 200) for s in (0, 1, 2, 3, -1)]
        names = [hnet.stage_kernel(s, B) for s in (1, 2, 3)]
        print(f"{B:>6} " + " ".join(f"{x:8.1f}" for x in t) + "   " + ", ".join(names), flush=True)


if __name__ == "__main__":
    main()
