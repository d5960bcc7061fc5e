"""GPU, opt-in (AZ_SOAK=1; ~3 minutes: the oracle needs ~2.5 s per Othello game): whole self-play runs at the BASELINE
simulation counts with the real network, HIP engine vs CPU oracle, every sample compared bit for bit.
The last green run of every round is recorded under profiles/ (r03_soak.txt: 48 Othello 8x8 games at 100 sims/move and 96 Connect4
games at 200: equal).  A trimmed version (8 / 16 games) runs in the default set
(tests/test_gpu_paths.py::test_selfplay_at_baseline_simulation_counts_equals_oracle)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from alphazero_amd import engine as E
from test_gpu_engine import sort_samples

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("AZ_SOAK") != "1", reason="opt-in soak (AZ_SOAK=1)")]


@pytest.mark.parametrize("name,gid,H,W,games,sims", [("othello8", 0, 8, 8, 48, 100), ("connect4", 1, 6, 7, 96, 200)])
def test_selfplay_at_baseline_simulation_counts_equals_oracle(name, gid, H, W, games, sims):
    from alphazero_amd.games.connect4 import Connect4Net
    from alphazero_amd.games.othello import OthelloNet
    torch.manual_seed(1)
    net = (OthelloNet(n=8) if gid == 0 else Connect4Net(7, 6)).eval()
    sd = {k: v.numpy() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    onet = O.ConvNet(gid, H, W, sd)
    eng = E.SelfPlayEngine(gid, H, W, n_slots=games, n_sim=sims, net=net.to_hip(max_batch=games), seed=77)
    got = sort_samples(eng.run(games, first_game_id=123))
    ref = O.selfplay(gid, H, W, games, sims, ("conv", onet), seed=77, first_game_id=123)
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k
