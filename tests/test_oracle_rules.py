"""G1: the CPU oracle's board rules against the reference's recorded playouts and positions (bit-exact)."""
import ctypes as C

import numpy as np
import pytest

from conftest import TAGS, golden, unpack_mask
from oracle import oracle as O


def mask_of(b, A, player=0):
    m = np.zeros(A, dtype=bool)
    m[O.legal_moves(b, player)] = True
    return m


@pytest.mark.parametrize("tag", list(TAGS))
def test_playouts(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"rules_{tag}.npz")
    L = O.lib()
    legal = unpack_mask(fx["legal"], A)
    legal_other = unpack_mask(fx["legal_other"], A)
    off = fx["offsets"]
    passes = 0
    for g in range(len(off) - 1):
        b = O.new_board(gid, H, W)
        for i in range(off[g], off[g + 1]):
            assert not L.orc_is_over(C.byref(b))
            assert b.player == fx["players"][i]
            assert np.array_equal(mask_of(b, A), legal[i]), (tag, g, i)
            assert np.array_equal(mask_of(b, A, -b.player), legal_other[i]), (tag, g, i)
            a = int(fx["actions"][i])
            passes += int(game == "othello" and a == A - 1)
            # an illegal action is refused and leaves the board untouched (reference: ValueError)
            bad = np.flatnonzero(~legal[i])
            if len(bad):
                c = O.Board.from_buffer_copy(b)
                assert L.orc_play(C.byref(c), int(bad[i % len(bad)])) == -1
                assert bytes(c) == bytes(b)
            assert L.orc_play(C.byref(b), a) == 0
        assert L.orc_is_over(C.byref(b))
        w = C.c_int()
        assert L.orc_winner(C.byref(b), C.byref(w)) == 0
        assert w.value == fx["winners"][g]
        assert L.orc_score(C.byref(b)) == fx["scores"][g]
        assert b.player == fx["final_players"][g]
        assert np.array_equal(b.grid_np(), fx["final_grids"][g])
    if tag == "othello8":
        assert passes > 0  # forced passes are covered


@pytest.mark.parametrize("tag", list(TAGS))
def test_positions(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"rules_{tag}.npz")
    L = O.lib()
    legal = unpack_mask(fx["pos_legal"], A)
    legal_other = unpack_mask(fx["pos_legal_other"], A)
    n_over = 0
    for i in range(len(fx["pos_players"])):
        b = O.new_board(gid, H, W)
        b.set_grid(fx["pos_grids"][i], fx["pos_players"][i])
        assert np.array_equal(mask_of(b, A), legal[i]), (tag, i)
        assert np.array_equal(mask_of(b, A, -b.player), legal_other[i]), (tag, i)
        over = bool(L.orc_is_over(C.byref(b)))
        assert over == bool(fx["pos_over"][i])
        w = C.c_int()
        rc = L.orc_winner(C.byref(b), C.byref(w))
        if over:
            n_over += 1
            assert rc == 0 and w.value == fx["pos_winner"][i]
        else:
            assert rc == -1  # reference raises ValueError("Game is not over yet...")
        a = int(fx["pos_action"][i])
        if a >= 0:
            assert L.orc_play(C.byref(b), a) == 0
            assert np.array_equal(b.grid_np(), fx["pos_result"][i])
            assert b.player == -fx["pos_players"][i]
    assert n_over > 0


def test_start_position_moves():
    # start-position legal moves (2,4), (3,5), (4,2), (5,3)
    b = O.new_board(O.OTHELLO, 8, 8)
    assert O.legal_moves(b) == [2 * 8 + 4, 3 * 8 + 5, 4 * 8 + 2, 5 * 8 + 3]


def test_random_playout_envelope_matches_the_reference_measurements():
    """SURVEY Appendix C (measured on the reference with 60 random Othello 8x8 playouts): 60.6 plies per game (60-63),
    0.94 % forced passes, mean branching factor 8.34 (max 22).  The oracle's rules over 300 random playouts must sit in
    the same envelope (statistical: seeds fixed, bounds wide enough for the sample sizes on both sides)."""
    n_games = 300
    grids, players, actions = O.random_positions(O.OTHELLO, 8, 8, 2024, n_games, 30000)
    start = O.new_board(O.OTHELLO, 8, 8).grid_np().reshape(-1)
    is_start = (grids == start).all(1) & (players == 1)
    assert is_start.sum() == n_games                       # every game was recorded from its first position to its last move
    plies = np.diff(np.append(np.flatnonzero(is_start), len(players)))
    assert 60.0 <= plies.mean() <= 61.5 and plies.min() >= 50 and plies.max() <= 68, (plies.mean(), plies.min(), plies.max())  # a few games end with empty cells
    passes = (actions == 64)
    assert 0.002 < passes.mean() < 0.02, passes.mean()
    legal = O.batch_legal(O.OTHELLO, 8, 8, grids, players)[:, :64].sum(1)
    assert (legal[passes] == 0).all() and (legal[~passes] > 0).all()
    branching = legal[~passes]
    assert 7.8 < branching.mean() < 8.9 and branching.max() <= 33, (branching.mean(), branching.max())
