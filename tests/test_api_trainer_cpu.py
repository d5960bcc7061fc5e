"""CPU: Sample / symmetry augmentation of the trainer mirror against the reference's recorded memory (G4),
argument checks and the baseline players (the search trees, rollout and neural, live on the GPU: test_gpu_api.py)."""
import numpy as np
import pytest

from conftest import TAGS, golden
from alphazero_amd.games.registers import BOARDS_REGISTER, DATA_AUGMENT_STRATEGIES, NETWORKS_REGISTER
from alphazero_amd.mcts import MCT
from alphazero_amd.players import GreedyPlayer, MCTSPlayer, RandomPlayer
from alphazero_amd.arena import Arena
from alphazero_amd.trainer import Sample, augment


@pytest.mark.parametrize("tag", ["othello8", "othello6", "connect4", "tictactoe"])
def test_augmentation_matches_reference_memory(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"selfplay_{tag}.npz")
    names = [str(x) for x in fx["transf_names"]]
    orig = fx["transformation"] == names.index("None")
    net = {"othello": lambda: NETWORKS_REGISTER[game](n=n), "connect4": lambda: NETWORKS_REGISTER[game](7, 6),
           "tictactoe": lambda: NETWORKS_REGISTER[game]()}[game]()
    mem = [Sample(state=fx["state"][i].astype(np.float64), pi=fx["pi"][i].copy(), player=1, outcome=int(fx["outcome"][i]),
                  episode_idx=int(fx["episode_idx"][i]), move_idx=int(fx["move_idx"][i])) for i in np.flatnonzero(orig)]
    extra = augment(mem, net, DATA_AUGMENT_STRATEGIES[game])
    ref = np.flatnonzero(~orig)
    assert len(extra) == len(ref)
    for s, j in zip(extra, ref):  # same order as the reference builds them (trainer.py:275-284)
        assert np.array_equal(s.state.astype(np.int8), fx["state"][j]) and np.array_equal(s.pi, fx["pi"][j])
        assert s.outcome == fx["outcome"][j] and s.move_idx == fx["move_idx"][j] and s.transformation == names[fx["transformation"][j]]


def test_sample_normalize():
    s = Sample(state=np.array([[1., -1.], [0., 1.]]), pi=np.array([1., 0.]), player=-1, outcome=1)
    s.normalize()
    assert s.player == 1 and s.outcome == -1 and np.array_equal(s.state, [[-1., 1.], [0., -1.]])


def test_baseline_players_and_argument_errors():
    with pytest.raises(ValueError):
        MCTSPlayer()
    with pytest.raises(ValueError):
        MCTSPlayer(n_sim=10, compute_time=1.0)
    with pytest.raises(ValueError):
        MCT(eval_method=None, nn=object())
    fx = golden("stats.npz")
    assert abs(int(fx["ttt_rollout_draw"]) / int(fx["ttt_rollout_games"]) - 0.625) < 0.01
    stats = Arena(GreedyPlayer(), RandomPlayer(), BOARDS_REGISTER["othello"](n=6)).play_games(40, return_stats=True)
    assert len(stats["player1"]) > len(stats["player2"])  # report Table 3: greedy beats random
    assert sum(stats["player1_starts"].values()) == 20 and sum(stats["player2_starts"].values()) == 20
