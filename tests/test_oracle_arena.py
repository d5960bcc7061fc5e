"""Golden G7 pins the oracle's arena loop (oracle.arena_games) to the reference's Arena.play_games (arena.py:36-185): who starts,
both players receive every move (arena.py:98-99), winners, scores and the stats dict.  tools/gen_golden.py::gen_arena ran the
reference with AlphaZeroPlayer (closed-form fake network, no noise, temperature 0) against GreedyPlayer / another AlphaZeroPlayer
under the deterministic fair_max (lowest action among the maxima), logging every move of the arena's own board."""
import numpy as np
import pytest

from conftest import TAGS, golden
from oracle import oracle as O


def arena_fixture(tag):
    """-> list of pairings: dict(opponent, sims1, sims2, n_rounds, start_player, moves[list per round], p2_starts, winner_colour, score, stats)"""
    fx = golden(f"arena_{tag}.npz")
    out = []
    for i in range(len(fx["n_rounds"])):
        r0, r1 = int(fx["round_off"][i]), int(fx["round_off"][i + 1])
        enc = lambda x: float("inf") if tag == "tictactoe" and x == 32767 else int(x)  # noqa: E731
        st = fx["starts"][i]
        out.append({"opponent": "greedy" if fx["opponent"][i] == 0 else "fake", "sims1": int(fx["sims1"][i]), "sims2": int(fx["sims2"][i]),
                    "n_rounds": int(fx["n_rounds"][i]), "start_player": int(fx["start_player"][i]) or None,
                    "moves": [[int(a) for a in fx["moves"][fx["move_off"][r]:fx["move_off"][r + 1]]] for r in range(r0, r1)],
                    "p2_starts": [bool(x) for x in fx["p2_starts"][r0:r1]], "winner_colour": [int(x) for x in fx["winner_colour"][r0:r1]],
                    "score": [enc(x) for x in fx["score"][r0:r1]],
                    "stats": {"player1": [enc(x) for x in fx["p1_scores"][fx["p1_off"][i]:fx["p1_off"][i + 1]]],
                              "player2": [enc(x) for x in fx["p2_scores"][fx["p2_off"][i]:fx["p2_off"][i + 1]]], "draw": int(fx["draws"][i]),
                              "player1_starts": {k: int(v) for k, v in zip(("win", "loss", "draw"), st[:3]) if v},
                              "player2_starts": {k: int(v) for k, v in zip(("win", "loss", "draw"), st[3:]) if v}}})
    return out


def same_stats(got, want):
    return (got["draw"] == want["draw"] and list(got["player1"]) == list(want["player1"]) and list(got["player2"]) == list(want["player2"])
            and dict(got["player1_starts"]) == want["player1_starts"] and dict(got["player2_starts"]) == want["player2_starts"])


@pytest.mark.parametrize("tag", ["tictactoe", "connect4", "othello6", "othello8"])
def test_oracle_arena_equals_reference_arena(tag):
    game, gid, H, W, A, n = TAGS[tag]
    for p in arena_fixture(tag):
        opp = "greedy" if p["opponent"] == "greedy" else ("fake", None)
        moves, winners, scores, stats = O.arena_games((gid, H, W), ("fake", None), p["sims1"], opp, p["sims2"], seed=0, n_rounds=p["n_rounds"],
                                                      start_player=p["start_player"], tie_mode=O.TIE_LOWEST)
        assert moves == p["moves"], (tag, p["opponent"])
        assert winners == p["winner_colour"] and scores == p["score"]
        assert same_stats(stats, p["stats"]), (stats, p["stats"])
        # the fixture really alternates the starting player / honours start_player
        want = [{1: False, 2: True}.get(p["start_player"], bool(r % 2)) for r in range(p["n_rounds"])]
        assert p["p2_starts"] == want
