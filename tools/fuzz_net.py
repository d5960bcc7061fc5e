#!/usr/bin/env python3
"""Randomised test of the HIP network forward across its kernel variants (test infrastructure; needs a GPU).  A board's outputs must
not depend on the batch it is evaluated in: a pool of boards is evaluated once at a small batch (that result is checked bit for bit
against the CPU oracle), then at random batch sizes -- around every row count where the dispatch switches kernels (128, 256, 512, 1024,
2048, 4096, 8192, 16384, 32768: +-1) and in between --, in random order, with random device-side row counts (az_net_forward_dyn): the
bits must be those of the pool.
    python tools/fuzz_net.py [trials] [seed]"""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from alphazero_amd import engine as E  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tools import closed_form as cf  # noqa: E402

TAGS = {"othello8": (0, 8, 8), "othello6": (0, 6, 6), "connect4": (1, 6, 7), "tictactoe": (2, 3, 3)}
EDGES = [128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]


def run(trials, seed, verbose=True):
    rng = np.random.default_rng(seed)
    bad = []
    pools = {}
    for t in range(trials):
        tag = str(rng.choice(["othello8", "othello8", "othello6", "connect4", "tictactoe"]))
        gid, H, W = TAGS[tag]
        top = 33000 if tag == "othello8" else 9000
        if tag not in pools:
            fx = np.load(os.path.join(ROOT, "tests", "golden", f"net_{tag}.npz"), allow_pickle=False)
            shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(fx["shape_keys"], fx["shape_vals"])}
            sd = {k: v for k, v in cf.closed_form_state_dict(shapes).items() if not k.endswith("num_batches_tracked")}
            onet = O.MlpNet(sd) if tag == "tictactoe" else O.ConvNet(gid, H, W, sd)
            hnet = E.HipNet(gid, H, W, sd, max_batch=top)
            grids, players, _ = O.random_positions(gid, H, W, 23, 40, 900)
            canon = (grids * players[:, None]).astype(np.float32)
            dev = torch.as_tensor(canon, device="cuda")
            p_ref, v_ref = hnet.forward(dev[:100].contiguous())  # small batch first: the oracle's bits
            op, ov = onet.forward(canon[:100])
            assert np.array_equal(p_ref.cpu().numpy(), op) and np.array_equal(v_ref.cpu().numpy(), ov), tag
            p_all, v_all = hnet.forward(dev)
            assert torch.equal(p_all[:100], p_ref) and torch.equal(v_all[:100], v_ref), tag
            pools[tag] = (hnet, dev, p_all, v_all)
        hnet, dev, p_all, v_all = pools[tag]
        n0 = dev.shape[0]
        edge = int(rng.choice([e for e in EDGES if e < top]))
        B = int(np.clip(edge + int(rng.integers(-1, 2)) if rng.random() < 0.7 else int(rng.integers(1, top)), 1, top))
        idx = torch.as_tensor(rng.integers(0, n0, B), device="cuda")
        x = dev[idx].contiguous()
        cfg = dict(tag=tag, B=B)
        try:
            if rng.random() < 0.5:
                p, v = hnet.forward(x)
                ok = torch.equal(p, p_all[idx]) and torch.equal(v, v_all[idx])
            else:
                count = int(rng.integers(0, B + 1)) if rng.random() < 0.7 else B + int(rng.integers(0, 50))
                cfg["count"] = count
                c = torch.tensor([count], dtype=torch.int32, device="cuda")
                p = torch.full((B, p_all.shape[1]), -7.0, device="cuda")
                v = torch.full((B,), -7.0, device="cuda")
                hnet.forward_dyn(x, c, p, v)
                m = min(count, B)
                ok = torch.equal(p[:m], p_all[idx[:m]]) and torch.equal(v[:m], v_all[idx[:m]]) and bool((p[m:] == -7.0).all()) and bool((v[m:] == -7.0).all())
        except Exception as e:  # noqa: BLE001
            ok = False
            cfg["exception"] = repr(e)[:300]
        if not ok:
            bad.append(cfg)
            if verbose:
                print("MISMATCH", cfg, flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    mism = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"network fuzz: {n} trials, {len(mism)} mismatches")
    sys.exit(1 if mism else 0)
