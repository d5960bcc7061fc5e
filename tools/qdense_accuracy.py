#!/usr/bin/env python3
"""How far each arithmetic of OthelloNet's forward is from float64 -- the evidence VERDICT r4 item 2(b) asks for before the int8 route may
carry a headline: >= 1e5 positions drawn from REAL self-play, random-init (seed 0) and post-SGD weights, torch float64 on the GPU as the
yardstick (not the oracle, which restates whichever arithmetic the switch selects).

    python tools/qdense_accuracy.py make FILE     self-play (engine, random-init OthelloNet seed 0) -> boards; a few hundred SGD steps on
                                                  those samples (hand-written step) -> post-SGD weights; torch float64 outputs; all saved to
                                                  FILE (npz); then `eval` of this process's own arithmetic
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

CHUNK = 16384


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
    # the yardstick, once: torch float64 on the GPU (and stock torch float32 beside it, for scale) -- every `eval` process reads them from the file
    boards = state.view(S, 64).float()
    pad = (-S) % CHUNK  # one batch shape for MIOpen (its algorithm search runs per shape and per dtype)
    padded = torch.cat([boards, boards[:pad]]) if pad else boards
    for tag, m in (("init", net), ("sgd", train)):
        for k, v in m.state_dict().items():
            if v.dtype == torch.float32:
                out[f"{tag}/{k}"] = v.detach().cpu().numpy()
        ref, f32 = m.clone().double().cuda().eval(), m.clone().float().cuda().eval()
        p64, v64, p32, v32 = [], [], [], []
        with torch.no_grad():
            for i in range(0, padded.shape[0], CHUNK):
                x = padded[i:i + CHUNK].view(-1, 8, 8)
                a, b = ref(x.double()); p64.append(torch.exp(a)); v64.append(b.view(-1))
                a, b = f32(x); p32.append(torch.exp(a)); v32.append(b.view(-1))
        p64, v64, p32, v32 = torch.cat(p64)[:S], torch.cat(v64)[:S], torch.cat(p32)[:S], torch.cat(v32)[:S]
        out[f"{tag}.p64"], out[f"{tag}.v64"] = p64.cpu().numpy(), v64.cpu().numpy()
        dp, dv = (p32.double() - p64).abs(), (v32.double() - v64).abs()
        out[f"{tag}.torch_f32"] = np.array([dp.max().item(), dp.mean().item(), dv.max().item(), dv.mean().item()])
    np.savez(path, **out)
    print(f"{S} self-play positions ({games} games at {sims} sims), {sgd_steps} SGD steps at batch {batch}: policy loss {lp[0].item():.3f} -> {lp[-1].item():.3f}; saved {path}")
    evaluate(path)  # this process's own arithmetic right away: one process start less for the caller


def evaluate(path):
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
        hnet = net.to_hip(max_batch=CHUNK)
        p64, v64 = torch.tensor(d[f"{tag}.p64"], device="cuda"), torch.tensor(d[f"{tag}.v64"], device="cuda")
        dp, dv = [], []
        for i in range(0, S, CHUNK):
            p, v = hnet.forward(boards[i:i + CHUNK])
            dp.append((p.double() - p64[i:i + CHUNK]).abs()); dv.append((v.double() - v64[i:i + CHUNK]).abs())
        dp, dv = torch.cat(dp), torch.cat(dv)
        t32 = d[f"{tag}.torch_f32"]
        res[tag] = {"pi_max": dp.max().item(), "pi_mean": dp.mean().item(), "v_max": dv.max().item(), "v_mean": dv.mean().item(),
                    "torch_f32_pi_max": float(t32[0]), "torch_f32_pi_mean": float(t32[1]), "torch_f32_v_max": float(t32[2]), "torch_f32_v_mean": float(t32[3]),
                    "dense_kernels": f"{hnet.stage_kernel(1, CHUNK)} / {hnet.stage_kernel(2, CHUNK)}"}
        hnet.close()
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) != 3 or sys.argv[1] not in ("make", "eval"):
        sys.exit(__doc__)
    (make if sys.argv[1] == "make" else evaluate)(sys.argv[2])
