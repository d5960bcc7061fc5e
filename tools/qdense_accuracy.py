#!/usr/bin/env python3
"""How far each arithmetic of OthelloNet's forward is from float64 -- the evidence VERDICT r4 item 2(b) asks for before the int8 route may
carry a headline: >= 1e5 positions drawn from REAL self-play, random-init (seed 0) and post-SGD weights, torch float64 on the GPU as the
yardstick (not the oracle, which restates whichever arithmetic the switch selects).

    python tools/qdense_accuracy.py make FILE     self-play (engine, random-init OthelloNet seed 0) -> boards; a few hundred SGD steps on
                                                  those samples (hand-written step) -> post-SGD weights; both saved to FILE (npz)
    python tools/qdense_accuracy.py eval FILE     this process's arithmetic (AZ_DENSE_I8 unset: float32 fma chains on the f32 MFMA;
                                                  AZ_DENSE_I8=1: fc1 / fc2 as exact block-fixed-point int8 GEMMs) against torch float64 on
                                                  FILE's boards and both weight sets -> one JSON line
The switch is read once per process, so `eval` runs once per arithmetic (tests/test_gpu_qdense.py starts both)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alphazero_amd import engine as E
from alphazero_amd.games.othello import OthelloNet
from alphazero_amd.train_step import HipTrainStep


def make(path, games=2048, sims=16, sgd_steps=400, batch=64):
    torch.manual_seed(0)
    net = OthelloNet(n=8).eval()
    hnet = net.to_hip(max_batch=games)
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=games, n_sim=sims, net=hnet, dirichlet_alpha=0.03, dirichlet_epsilon=0.25, temp_max_step=4,
                           temp_min_step=4, tie_mode=E.TIE_RANDOM, noise_mode=E.NOISE_PHILOX, seed=0)
    smp = eng.run(games)
    state, pi, z = smp["state"].clone().contiguous(), smp["pi"].clone().contiguous(), smp["z"].clone().contiguous()
    eng.close()
    hnet.close()
    S = state.shape[0]
    assert S >= 100000, S
    # post-SGD weights: the trainer's optimisation (momentum 0.9, weight decay 1e-4, dropout 0.3, lr 0.1: games/othello.py:31-36) on these samples
    train = net.clone().cuda().train()
    step = HipTrainStep(train, max_batch=batch)
    step.load(train)
    step.begin(0.1, 0.9, 1e-4, 0.3, seed=0)
    g = torch.Generator(device="cuda").manual_seed(0)
    perm = torch.randint(0, S, (sgd_steps * batch,), dtype=torch.int64, device="cuda", generator=g)
    lp, lv = torch.zeros(sgd_steps, device="cuda"), torch.zeros(sgd_steps, device="cuda")
    step.steps(state.view(S, 8, 8), pi, z, perm, sgd_steps, batch, lp, lv)
    step.check()
    step.store(train)
    step.close()
    out = {"boards": state.view(S, 64).cpu().numpy(), "loss_pi_first_last": np.array([lp[0].item(), lp[-1].item()])}
    for tag, m in (("init", net), ("sgd", train)):
        for k, v in m.state_dict().items():
            if v.dtype == torch.float32:
                out[f"{tag}/{k}"] = v.detach().cpu().numpy()
    np.savez_compressed(path, **out)
    print(f"{S} self-play positions ({games} games at {sims} sims), {sgd_steps} SGD steps at batch {batch}: policy loss {lp[0].item():.3f} -> {lp[-1].item():.3f}; saved {path}")


def evaluate(path, chunk=16384):
    d = np.load(path)
    boards = torch.tensor(d["boards"].astype(np.float32), device="cuda")
    S = boards.shape[0]
    res = {"arithmetic": "int8 block-fixed-point fc1 / fc2 (AZ_DENSE_I8=1)" if os.environ.get("AZ_DENSE_I8") == "1" else "float32 fma chains (default)",
           "positions": int(S)}
    for tag in ("init", "sgd"):
        sd = {k.split("/", 1)[1]: torch.tensor(d[k]) for k in d.files if k.startswith(tag + "/")}
        net = OthelloNet(n=8)
        net.load_state_dict(sd, strict=False)
        net.eval()
        hnet = net.to_hip(max_batch=chunk)
        ref = net.clone().double().cuda().eval()
        ref.device = torch.device("cuda")
        f32 = net.clone().cuda().eval()
        dp, dv, tp, tv = [], [], [], []
        with torch.no_grad():
            for i in range(0, S, chunk):
                x = boards[i:i + chunk]
                p, v = hnet.forward(x)
                lp64, v64 = ref(x.view(-1, 8, 8).double())
                p64, v64 = torch.exp(lp64), v64.view(-1)
                dp.append((p.double() - p64).abs()); dv.append((v.double() - v64).abs())
                lp32, v32 = f32(x.view(-1, 8, 8))  # stock PyTorch float32 (rocBLAS / MIOpen) on the same boards, for scale
                tp.append((torch.exp(lp32).double() - p64).abs()); tv.append((v32.view(-1).double() - v64).abs())
        dp, dv, tp, tv = torch.cat(dp), torch.cat(dv), torch.cat(tp), torch.cat(tv)
        res[tag] = {"pi_max": dp.max().item(), "pi_mean": dp.mean().item(), "v_max": dv.max().item(), "v_mean": dv.mean().item(),
                    "torch_f32_pi_max": tp.max().item(), "torch_f32_pi_mean": tp.mean().item(), "torch_f32_v_max": tv.max().item(), "torch_f32_v_mean": tv.mean().item(),
                    "dense_kernels": f"{hnet.stage_kernel(1, chunk)} / {hnet.stage_kernel(2, chunk)}"}
        hnet.close()
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) != 3 or sys.argv[1] not in ("make", "eval"):
        sys.exit(__doc__)
    (make if sys.argv[1] == "make" else evaluate)(sys.argv[2])
