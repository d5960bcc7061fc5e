"""The host-side Board / Net / scheduler mirror of the reference's plugin surface against the golden fixtures."""
import ast

import numpy as np
import pytest
import torch

from conftest import TAGS, golden, unpack_mask
from tools import closed_form as cf
from alphazero_amd.games.registers import BOARDS_REGISTER, CONFIGS_REGISTER, DATA_AUGMENT_STRATEGIES, GAMES_SET, NETWORKS_REGISTER
from alphazero_amd.schedulers import TEMP_SCHEDULERS


def make_board(tag, grid=None, player=1):
    game, gid, H, W, A, n = TAGS[tag]
    kw = {"othello": dict(n=n), "connect4": dict(width=7, height=6), "tictactoe": {}}[game]
    cls = BOARDS_REGISTER[game]
    return cls(grid=np.array(grid, dtype=np.float64), player=player, **kw) if grid is not None else cls(**kw)


def mask(board, tag, player=None):
    game, gid, H, W, A, n = TAGS[tag]
    m = np.zeros(A, bool)
    for mv in board.get_moves(player):
        m[cf.move_to_action(game, mv, n)] = True
    return m


@pytest.mark.parametrize("tag", list(TAGS))
def test_board_playouts(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"rules_{tag}.npz")
    legal, other, off = unpack_mask(fx["legal"], A), unpack_mask(fx["legal_other"], A), fx["offsets"]
    for g in range(0, len(off) - 1, 3):
        b = make_board(tag)
        for i in range(off[g], off[g + 1]):
            assert not b.is_game_over() and b.player == fx["players"][i]
            assert np.array_equal(mask(b, tag), legal[i])
            if game == "othello":
                assert np.array_equal(mask(b, tag, -b.player), other[i])
            bad = np.flatnonzero(~legal[i])
            if len(bad):
                with pytest.raises(ValueError):
                    b.clone().play_move(cf.action_to_move(game, int(bad[0]), n))
            b.play_move(cf.action_to_move(game, int(fx["actions"][i]), n))
        assert b.is_game_over() and b.get_winner() == fx["winners"][g]
        sc = b.get_score()
        assert (32767 if sc == float("inf") else int(sc)) == fx["scores"][g]
        assert np.array_equal(b.grid.astype(np.int8), fx["final_grids"][g]) and b.player == fx["final_players"][g]


@pytest.mark.parametrize("tag", list(TAGS))
def test_board_positions(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"rules_{tag}.npz")
    legal = unpack_mask(fx["pos_legal"], A)
    for i in range(len(fx["pos_players"])):
        b = make_board(tag, fx["pos_grids"][i], int(fx["pos_players"][i]))
        assert np.array_equal(mask(b, tag), legal[i])
        assert b.is_game_over() == bool(fx["pos_over"][i])
        if fx["pos_over"][i]:
            assert b.get_winner() == fx["pos_winner"][i]
        else:
            with pytest.raises(ValueError, match="not over"):
                b.get_winner()
        a = int(fx["pos_action"][i])
        if a >= 0:
            b.play_move(cf.action_to_move(game, a, n))
            assert np.array_equal(b.grid.astype(np.int8), fx["pos_result"][i])


def test_constructors_and_registers():
    assert GAMES_SET == set(CONFIGS_REGISTER) == set(BOARDS_REGISTER) == set(NETWORKS_REGISTER) == set(DATA_AUGMENT_STRATEGIES)
    with pytest.raises(ValueError, match="even"):
        BOARDS_REGISTER["othello"](n=5)
    with pytest.raises(ValueError, match="4x4"):
        BOARDS_REGISTER["connect4"](width=3, height=6)
    b = BOARDS_REGISTER["othello"](config=CONFIGS_REGISTER["othello"]())
    assert b.n == 6 and b.pass_move == (6, 6) and b.max_moves == 32 and str(b) == "OthelloBoard6"
    assert b.clone().grid is not b.grid and b.get_action_size() == 37
    assert sorted(BOARDS_REGISTER["othello"](n=8).get_moves()) == [(2, 4), (3, 5), (4, 2), (5, 3)]
    c = BOARDS_REGISTER["connect4"](width=7, height=6)
    assert c.pass_move is None and c.get_action_size() == 7 and c.max_moves == 42
    lin = TEMP_SCHEDULERS["linear"](4, 4, 60)
    assert [lin[s] for s in (0, 4, 5, 60)] == [1, 1, 0, 0]
    lin = TEMP_SCHEDULERS["linear"](2, 6, 60)
    assert lin[4] == 0.5
    with pytest.raises(ValueError):
        TEMP_SCHEDULERS["linear"](5, 4, 60)
    assert TEMP_SCHEDULERS["constant"](3, 3, 9)[0] == 0


@pytest.mark.parametrize("tag", ["othello8", "othello6", "connect4", "tictactoe"])
def test_net_mirror_matches_reference_forward(tag):
    """same state_dict keys / shapes as the reference and the same forward (golden G2, tolerance 1e-5)"""
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"net_{tag}.npz")
    shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(fx["shape_keys"], fx["shape_vals"])}
    net = {"othello": lambda: NETWORKS_REGISTER[game](n=n), "connect4": lambda: NETWORKS_REGISTER[game](7, 6),
           "tictactoe": lambda: NETWORKS_REGISTER[game]()}[game]()
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == shapes
    assert net.get_parameters_count() == int(fx["n_params"])
    net.load_state_dict({k: torch.tensor(v) for k, v in cf.closed_form_state_dict(shapes).items()})
    canon = fx["grids"].astype(np.float32) * fx["players"].astype(np.float32)[:, None, None]
    p, v = net.predict(torch.tensor(canon))
    assert np.abs(p.numpy() - fx["probs"]).max() < 1e-5 and np.abs(v.numpy().reshape(-1) - fx["v"]).max() < 1e-5
    b = make_board(tag, fx["grids"][0], int(fx["players"][0]))
    pr, val = net.evaluate(b)
    assert np.abs(pr - fx["eval_probs"][0]).max() < 1e-5 and abs(val - fx["eval_v"][0]) < 1e-5
    legal = b.get_moves()
    norm = net.get_normalized_probs(pr, legal)
    assert abs(sum(norm.values()) - 1) < 1e-5 and set(norm) == set(legal)
    pi = net.to_neural_output(norm)
    assert pi.shape == (A,) and abs(pi.sum() - 1) < 1e-5


@pytest.mark.parametrize("H,W", [(4, 4), (5, 6), (8, 8), (6, 8)])
def test_connect4_other_sizes_mirror_equals_oracle(H, W):
    """closes the loop for non-default Connect4 boards: mirror == reference (tests/test_live_reference.py, where the
    reference is mounted), HIP == oracle (tests/test_gpu_rules.py), and here mirror == oracle"""
    from oracle import oracle as O
    from alphazero_amd.games.connect4 import Connect4Board
    grids, players, actions = O.random_positions(O.CONNECT4, H, W, 99, 60, 5000)
    legal = O.batch_legal(O.CONNECT4, H, W, grids, players)
    og, op, _ = O.batch_play(O.CONNECT4, H, W, grids, players, actions)
    over, win, score = O.batch_status(O.CONNECT4, H, W, og, op)
    for i in range(0, len(players), 3):
        b = Connect4Board(width=W, height=H, grid=grids[i].reshape(H, W).astype(np.float64), player=int(players[i]))
        assert sorted(int(m) for m in b.get_moves()) == np.flatnonzero(legal[i]).tolist()
        b.play_move(np.int64(actions[i]))
        assert np.array_equal(b.grid.astype(np.int8).reshape(-1), og[i]) and b.player == op[i]
        assert b.is_game_over() == bool(over[i]) and b.get_score() == score[i]
        if over[i]:
            assert b.get_winner() == win[i]


def test_timers_keep_the_reference_surface():
    """timers.py:11-151: constructors, get_fake_batch shapes and the optimisation timing loop (on the CPU: no engine involved)"""
    import torch
    from alphazero_amd.games.connect4 import Connect4Config
    from alphazero_amd.games.tictactoe import TicTacToeConfig
    from alphazero_amd.timers import NeuralTimer, SelfPlayTimer
    spt = SelfPlayTimer("tictactoe")
    assert spt.az_player.n_sim == spt.config.simulations and spt.board.get_action_size() == 9 and spt.nn.get_parameters_count() == 316
    nt = NeuralTimer("connect4", Connect4Config(batch_size=8, device="cpu"))
    x, pi, v = nt.get_fake_batch()
    assert tuple(x.shape) == (8, 6, 7) and tuple(pi.shape) == (8, 7) and tuple(v.shape) == (8,)
    before = [p.detach().clone() for p in nt.nn.parameters()]
    assert nt.timeit(n_batches=2) > 0
    assert any(not torch.equal(a, b) for a, b in zip(before, nt.nn.parameters()))  # the steps did update the network
    nt = NeuralTimer("tictactoe", TicTacToeConfig(batch_size=4))
    assert tuple(nt.get_fake_batch()[0].shape) == (4, 3, 3)
