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
    np.random.seed(5)  # the reference's players draw from the global numpy state: pinned, so that the outcome is the same every run
    stats = Arena(GreedyPlayer(), RandomPlayer(), BOARDS_REGISTER["othello"](n=6)).play_games(120, return_stats=True)
    assert len(stats["player1"]) > len(stats["player2"])  # report Table 3: greedy beats random (60 : 35 over 10 K games)
    assert sum(stats["player1_starts"].values()) == 60 and sum(stats["player2_starts"].values()) == 60


@pytest.mark.parametrize("tag", ["tictactoe", "connect4", "othello6", "othello8"])
def test_optimize_network_matches_reference_losses(tag):
    """SURVEY 8f rank 3 / golden G6: AlphaZeroTrainer.optimize_network of the reference (trainer.py:320-381: loss
    -sum(pi log p)/B + sum((v - z)^2)/B, SGD momentum 0.9, weight decay 1e-4, ExponentialLR 0.9, train-mode BatchNorm) run by
    tools/gen_golden.py on the G4 memory with closed-form initial weights, dropout 0 and a pinned batch shuffle -> the
    mirror's loop must log the same per-batch losses over two epochs and end with the same weights (1e-5)."""
    import ast
    import torch
    from tools import closed_form as cf
    from alphazero_amd.games.registers import CONFIGS_REGISTER
    from alphazero_amd.trainer import AlphaZeroTrainer
    torch.set_num_threads(1)
    game, gid, H, W, A, n = TAGS[tag]
    fx, mem_fx, net_fx = golden(f"sgd_{tag}.npz"), golden(f"selfplay_{tag}.npz"), golden(f"net_{tag}.npz")
    extra = {"board_size": n} if game == "othello" else {}
    cfg = CONFIGS_REGISTER[game](epochs=int(fx["epochs"]), batch_size=int(fx["batch_size"]), **extra)
    assert cfg.learning_rate == float(fx["lr"])
    tr = AlphaZeroTrainer(verbose=False)
    tr.config, tr.game = cfg, game
    net = NETWORKS_REGISTER[game](config=cfg)
    shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(net_fx["shape_keys"], net_fx["shape_vals"])}
    net.load_state_dict({k: torch.tensor(v) for k, v in cf.closed_form_state_dict(shapes).items()})
    if hasattr(net, "dropout"):
        net.dropout = 0.0
    tr.nn = net
    tr.memory = [Sample(state=mem_fx["state"][i].astype(np.float64), pi=mem_fx["pi"][i].copy(), player=1, outcome=int(mem_fx["outcome"][i]),
                        episode_idx=int(mem_fx["episode_idx"][i]), move_idx=int(mem_fx["move_idx"][i])) for i in range(len(mem_fx["outcome"]))]
    assert len(tr.memory) == int(fx["n_samples"])
    tr.loss_values = {}
    np.random.seed(int(fx["shuffle_seed"]))
    tr.optimize_network(0)
    for e in range(int(fx["epochs"])):
        got_pi, got_v = np.array(tr.loss_values[0][e]["pi"]), np.array(tr.loss_values[0][e]["v"])
        assert got_pi.shape == fx[f"pi_loss_{e}"].shape
        assert np.abs(got_pi - fx[f"pi_loss_{e}"]).max() < 1e-5 and np.abs(got_v - fx[f"v_loss_{e}"]).max() < 1e-5
    sd = tr.nn_twin.state_dict()
    assert np.abs(sd["fc1.weight"].numpy()[:64] - fx["fc1_weight"]).max() < 1e-5
    assert np.abs(sd["fc_value.weight"].numpy() - fx["fc_value_weight"]).max() < 1e-5
    bn = ("fc_bn1" if game != "tictactoe" else "bn1") + ".running_mean"
    assert np.abs(sd[bn].numpy() - fx["bn_running_mean"]).max() < 1e-5
    # a wrong momentum / weight-decay constant must not pass: the fixture separates them
    if int(fx["epochs"]) > 1:
        assert np.abs(fx["pi_loss_0"] - fx["pi_loss_1"][: len(fx["pi_loss_0"])]).max() > 1e-3


def test_print_config_and_freeze_config(tmp_path, capsys, monkeypatch):
    """trainer.py:149-154, 580-587"""
    import json
    from alphazero_amd import base
    from alphazero_amd.games.othello import OthelloConfig
    from alphazero_amd.trainer import AlphaZeroTrainer, freeze_config
    AlphaZeroTrainer.print_config(OthelloConfig())
    out = capsys.readouterr().out
    assert "- game: othello" in out and "- simulations: 100" in out and "- batch_size: 64" in out
    AlphaZeroTrainer.print_config(OthelloConfig(), verbose=False)
    assert capsys.readouterr().out == ""
    monkeypatch.setattr(base, "DEFAULT_CONFIGS_PATH", str(tmp_path / "configs"))
    freeze_config()
    for game in ("othello", "connect4", "tictactoe"):
        d = json.load(open(tmp_path / "configs" / f"{game}.json"))
        assert d["game"] == game and d == dict(AlphaZeroTrainer.load_config_from_json(game, None).to_dict())


def test_arena_play_game_binds_positionally_like_the_reference():
    """arena.py:36-45: play_game(player2_starts, display, save_frames, return_results, ...): the fourth positional argument asks for results"""
    board = BOARDS_REGISTER["tictactoe"]()
    arena = Arena(RandomPlayer(), RandomPlayer(), board)
    np.random.seed(1)
    res = arena.play_game(False, False, False, True)
    assert set(res) == {"winner", "score"} and res["winner"] in (0, 1, 2)
    assert arena.play_game(True, False, False, False, True, False, False) is None
