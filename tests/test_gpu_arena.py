"""GPU parity of the batched arena (arena.py:36-185, trainer.py:408-446): every game BatchedArena plays
is replayed by the CPU oracle driven exactly as Arena.play_game drives its players -- each side owns a tree, BOTH trees
receive every move (arena.py:98-99), the AlphaZero side searches without noise at temperature 0 (trainer.py:421-425),
the opponent is RandomPlayer / GreedyPlayer (players.py:76-123), rollout MCTSPlayer, or another network -- and must
produce the same move at every ply, the same winners and the same stats dict."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from alphazero_amd.arena import BatchedArena

pytestmark = pytest.mark.gpu


def _np_sd(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items() if not k.endswith("num_batches_tracked")}


def _nets(game, seed):
    from alphazero_amd.games.connect4 import Connect4Net
    from alphazero_amd.games.othello import OthelloNet
    from alphazero_amd.games.tictactoe import TicTacToeNet
    torch.manual_seed(seed)
    if game == "othello":
        net = OthelloNet(n=6).eval()
        return net, ("conv", O.ConvNet(O.OTHELLO, 6, 6, _np_sd(net))), (O.OTHELLO, 6, 6)
    if game == "connect4":
        net = Connect4Net(7, 6).eval()
        return net, ("conv", O.ConvNet(O.CONNECT4, 6, 7, _np_sd(net))), (O.CONNECT4, 6, 7)
    net = TicTacToeNet().eval()
    return net, ("mlp", O.MlpNet(_np_sd(net))), (O.TICTACTOE, 3, 3)


oracle_arena = O.arena_games  # Arena.play_games restated on the oracle (oracle/oracle.py)


def _check(game, opponent_kind, n_rounds, n_sim, opp_sim, seed):
    net, ev1, dims = _nets(game, 20 + seed)
    if opponent_kind == "network":
        net2, ev2, _ = _nets(game, 40 + seed)
        opp_engine, opp_oracle = net2, ev2
    else:
        opp_engine = opp_oracle = opponent_kind
    arena = BatchedArena(game, net, opponent=opp_engine, n_sim=n_sim, opponent_n_sim=opp_sim, seed=seed, board_size=6)
    stats = arena.play_games(n_rounds, return_stats=True, record_moves=True)
    got = [[int(m[g]) for m in arena.moves if m[g] >= 0] for g in range(n_rounds)]
    moves, winners, scores, ostats = oracle_arena(dims, ev1, n_sim, opp_oracle, opp_sim, seed, n_rounds)
    for g in range(n_rounds):
        assert got[g] == moves[g], (game, opponent_kind, g, got[g], moves[g])
    assert stats["draw"] == ostats["draw"] and sorted(stats["player1"]) == sorted(ostats["player1"])
    assert sorted(stats["player2"]) == sorted(ostats["player2"])
    assert dict(stats["player1_starts"]) == dict(ostats["player1_starts"]) and dict(stats["player2_starts"]) == dict(ostats["player2_starts"])
    return stats


@pytest.mark.parametrize("game", ["othello", "connect4", "tictactoe"])
@pytest.mark.parametrize("opponent", ["random", "greedy", "mcts"])
def test_batched_arena_equals_oracle_arena(game, opponent):
    stats = _check(game, opponent, n_rounds=32, n_sim=16, opp_sim=24, seed=3)
    assert len(stats["player1"]) + len(stats["player2"]) + stats["draw"] == 32


@pytest.mark.parametrize("game", ["othello", "connect4"])
def test_batched_arena_network_vs_network_equals_oracle(game):
    """evaluation against another network (eval_opponent = "previous"): two device trees per game, both re-rooted at every move"""
    _check(game, "network", n_rounds=16, n_sim=20, opp_sim=12, seed=5)


@pytest.mark.parametrize("game,opponent,n_rounds,n_sim,picks", [("othello", "greedy", 8192, 20, (0, 1, 4097, 8191)),
                                                                ("connect4", "network", 4096, 24, (0, 2049, 4095))])
def test_large_batched_arena_sampled_rounds_equal_oracle(game, opponent, n_rounds, n_sim, picks):
    """the arena at evaluation scale: thousands of rounds at once put the side-masked search on the large-batch network kernels
    (two boards per wave, tiled GEMMs, device row counter).  Rounds are independent given (seed, round), so a few of them are
    replayed alone on the oracle: same moves at every ply, same winners."""
    seed = 11
    net, ev1, dims = _nets(game, 20 + seed)
    if opponent == "network":
        net2, ev2, _ = _nets(game, 40 + seed)
        opp_engine, opp_oracle = net2, ev2
    else:
        opp_engine = opp_oracle = opponent
    arena = BatchedArena(game, net, opponent=opp_engine, n_sim=n_sim, opponent_n_sim=16, seed=seed, board_size=6)
    stats = arena.play_games(n_rounds, return_stats=True, record_moves=True)
    assert len(stats["player1"]) + len(stats["player2"]) + stats["draw"] == n_rounds
    moves, winners, scores, _ = oracle_arena(dims, ev1, n_sim, opp_oracle, 16, seed, n_rounds, rounds=picks)
    for i, g in enumerate(picks):
        got = [int(m[g]) for m in arena.moves if m[g] >= 0]
        assert got == moves[i], (game, opponent, g, got, moves[i])


def test_trainer_evaluates_against_the_previous_network(tmp_path):
    """BASELINE config 5 "arena eval vs prev net": eval_opponent = "previous" plays the trained network against the one it
    replaces; "alphazero" keeps the reference's refusal (trainer.py:399-400)"""
    import json
    import os
    from alphazero_amd import base
    from alphazero_amd.games.othello import OthelloConfig
    from alphazero_amd.trainer import AlphaZeroTrainer
    base.DEFAULT_MODELS_PATH = str(tmp_path) + "/"
    tr = AlphaZeroTrainer(verbose=False, engine_slots=16, seed=1, materialize_memory=False)
    tr.game = "othello"
    tr.config = OthelloConfig(board_size=6, simulations=8, episodes=16, epochs=1, batch_size=32, iterations=1, do_eval=True,
                              eval_opponent="alphazero", eval_episodes=8)
    with pytest.raises(ValueError, match="not yet implemented"):
        tr.setup()
    tr.config.eval_opponent = "previous"
    tr.setup()
    first = tr.nn
    tr.self_play(0); tr.optimize_network(0); tr.update_network(0)
    assert tr.prev_nn is first and tr.nn is not first
    tr.evaluate(0)
    res = tr.eval_results["results"][0]
    assert sum(sum(v.values()) for v in res.values()) == 8 and tr.eval_results["eval_opponent"] == "previous"
    # the same games, replayed by the oracle with the two weight sets
    ev_new = ("conv", O.ConvNet(O.OTHELLO, 6, 6, _np_sd(tr.nn)))
    ev_old = ("conv", O.ConvNet(O.OTHELLO, 6, 6, _np_sd(first)))
    _, _, _, ostats = oracle_arena((O.OTHELLO, 6, 6), ev_new, 8, ev_old, 8, tr.seed + 0, 8)
    assert res == {k: dict(v) for k, v in ostats.items() if k.endswith("_starts")}
    tr.save_training_stats("prev-test")
    assert json.load(open(os.path.join(tmp_path, "prev-test", "eval.json")))["eval_opponent"] == "previous"


@pytest.mark.parametrize("tag", ["tictactoe", "connect4", "othello6", "othello8"])
def test_batched_arena_equals_the_reference_arena_fixture(tag):
    """golden G7 directly (no oracle in between): the reference's Arena.play_games -- AlphaZeroPlayer on the closed-form fake network
    against GreedyPlayer / another AlphaZeroPlayer under the deterministic fair_max -- move for move, winners, scores, stats dict,
    on Othello 8x8 too"""
    from conftest import TAGS
    from test_oracle_arena import arena_fixture, same_stats
    from alphazero_amd import engine as E
    game, gid, H, W, A, n = TAGS[tag]
    for p in arena_fixture(tag):
        arena = BatchedArena(game, "fake", opponent=p["opponent"], n_sim=p["sims1"], opponent_n_sim=p["sims2"] or None, seed=0, board_size=n)
        arena.tie_mode = E.TIE_LOWEST
        stats = arena.play_games(p["n_rounds"], start_player=p["start_player"], return_stats=True, record_moves=True)
        got = [[int(m[g]) for m in arena.moves if m[g] >= 0] for g in range(p["n_rounds"])]
        assert got == p["moves"], (tag, p["opponent"])
        assert same_stats(stats, p["stats"]), (tag, stats, p["stats"])


def test_batched_arena_equals_oracle_arena_on_othello8():
    """the BASELINE board: real OthelloNet(n=8) against greedy and against another network, move for move"""
    from alphazero_amd.games.othello import OthelloNet
    for opp_kind in ("greedy", "network"):
        torch.manual_seed(61)
        net = OthelloNet(n=8).eval()
        ev1 = ("conv", O.ConvNet(O.OTHELLO, 8, 8, _np_sd(net)))
        if opp_kind == "network":
            torch.manual_seed(62)
            net2 = OthelloNet(n=8).eval()
            opp_engine, opp_oracle = net2, ("conv", O.ConvNet(O.OTHELLO, 8, 8, _np_sd(net2)))
        else:
            opp_engine = opp_oracle = "greedy"
        arena = BatchedArena("othello", net, opponent=opp_engine, n_sim=12, opponent_n_sim=10, seed=9, board_size=8)
        stats = arena.play_games(8, return_stats=True, record_moves=True)
        got = [[int(m[g]) for m in arena.moves if m[g] >= 0] for g in range(8)]
        moves, winners, scores, ostats = oracle_arena((O.OTHELLO, 8, 8), ev1, 12, opp_oracle, 10, 9, 8)
        assert got == moves, opp_kind
        assert stats["draw"] == ostats["draw"] and stats["player1"] == ostats["player1"] and stats["player2"] == ostats["player2"]


@pytest.mark.parametrize("opponent", ["network", "mcts"])
def test_two_players_thinking_at_once_play_the_same_games(opponent):
    """BatchedArena queues both tree players' searches before it waits for either (az_engine_search_begin / _end, the second engine
    on a stream of its own priority: az_engine_pair); one search after the other must give the same moves, ply for ply"""
    net, _, _ = _nets("othello", 7)
    opp = _nets("othello", 47)[0] if opponent == "network" else "mcts"
    runs = []
    for overlap in (True, False):
        arena = BatchedArena("othello", net, opponent=opp, n_sim=20, opponent_n_sim=16, seed=4, board_size=6)
        arena.overlap = overlap
        stats = arena.play_games(24, return_stats=True, record_moves=True)
        runs.append(([m.tolist() for m in arena.moves], stats["draw"], stats["player1"], stats["player2"]))
        assert all(st["error_flags"] == 0 for st in arena.engine_stats)
    assert runs[0] == runs[1]


def test_search_begin_and_end_come_in_pairs():
    from alphazero_amd import engine as E
    net, _, _ = _nets("othello", 8)
    arena = BatchedArena("othello", net, opponent="random", n_sim=4, board_size=6)
    eng = arena._engine(net, 4, 4, 0)
    try:
        board = np.zeros((4, 6, 6), np.int8)
        board[:, 2, 2] = board[:, 3, 3] = -1
        board[:, 2, 3] = board[:, 3, 2] = 1
        eng.set_roots(board, np.ones(4, np.int8))
        from alphazero_amd._lib import AzError
        with pytest.raises(AzError):
            eng.search_end()  # nothing begun
        eng.search_begin(4)
        with pytest.raises(AzError):
            eng.search_begin(4)  # the first one is still open
        # ... and so is every other entry point of this engine until the search has been ended: they would reorder the
        # host-side state of the search in flight and its error flags would never be read
        for call in (lambda: eng.search(4), eng.advance, lambda: eng.play([0, 0, 0, 0]), lambda: eng.set_roots(board, np.ones(4, np.int8)),
                     lambda: eng.root_children(0), lambda: eng.nodes_used(0), lambda: eng.grow_pools(1 << 15), eng.best_moves, lambda: eng.run(4),
                     lambda: eng.samples()):
            with pytest.raises(AzError, match="has not been ended"):
                call()
        eng.search_end()
        a, n, _, _, rootn = eng.root_children(0)  # the search itself was not disturbed
        assert rootn == 4 == int(np.sum(n))
        with pytest.raises(ValueError):
            eng.pair_with(eng)
    finally:
        eng.close()


def test_randomised_arenas_equal_oracle():
    """tools/fuzz_arena.py: 80 random arenas (game, board size, opponent kind, rounds, start player, simulations, tie mode, overlapped
    or serial searches, seeds up to 10^6 -- a seed above 42 949 used to overflow the uint32 game ids on the host) against the oracle's
    arena: every move, the winners and the stats dict"""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_arena
    assert fuzz_arena.run(80, seed=31, verbose=False) == []
