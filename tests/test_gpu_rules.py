"""GPU parity, board rules (K1/K2): the HIP bitboard kernels through the C ABI against the golden
fixtures and, at scale, against the CPU oracle on the same positions.  Bit-exact."""
import numpy as np
import pytest
import torch

from conftest import TAGS, golden, unpack_mask
from oracle import oracle as O
from alphazero_amd import engine as E

pytestmark = pytest.mark.gpu


def dev(a, dt):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")


def replay_positions(tag):
    """positions along the fixture's playouts, reconstructed with the oracle"""
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"rules_{tag}.npz")
    off = fx["offsets"]
    grids, players = [], []
    for g in range(len(off) - 1):
        b = O.new_board(gid, H, W)
        for i in range(off[g], off[g + 1]):
            grids.append(b.grid_np().reshape(-1)); players.append(b.player)
            assert O.lib().orc_play(O.C.byref(b), int(fx["actions"][i])) == 0
    return fx, np.array(grids, np.int8), np.array(players, np.int8)


@pytest.mark.parametrize("tag", list(TAGS))
def test_fixture_playouts(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx, grids, players = replay_positions(tag)
    g, p = dev(grids, torch.int8), dev(players, torch.int8)
    legal = E.legal_batch(gid, H, W, g, p).cpu().numpy().astype(bool)
    assert np.array_equal(legal, unpack_mask(fx["legal"], A))
    other = E.legal_batch(gid, H, W, g, p, for_player=-p).cpu().numpy().astype(bool)
    assert np.array_equal(other, unpack_mask(fx["legal_other"], A))
    og, op, st = E.play_batch(gid, H, W, g, p, dev(fx["actions"].astype(np.int32), torch.int32))
    assert int(st.abs().sum()) == 0
    rg, rp, rs = O.batch_play(gid, H, W, grids, players, fx["actions"].astype(np.int32))
    assert np.array_equal(og.cpu().numpy(), rg) and np.array_equal(op.cpu().numpy(), rp)
    # an illegal action is refused, board unchanged (reference: ValueError)
    bad = np.array([np.flatnonzero(~legal[i])[0] if (~legal[i]).any() else -1 for i in range(len(players))], np.int32)
    og2, op2, st2 = E.play_batch(gid, H, W, g, p, dev(bad, torch.int32))
    assert bool((st2 == -5).all())
    assert torch.equal(og2, g) and torch.equal(op2, p)
    # final positions: over, winner, score
    fg = fx["final_grids"].reshape(len(fx["winners"]), -1)
    over, win, score = E.status_batch(gid, H, W, dev(fg, torch.int8), dev(fx["final_players"], torch.int8))
    assert bool(over.all())
    assert np.array_equal(win.cpu().numpy(), fx["winners"])
    if game != "tictactoe":
        assert np.array_equal(score.cpu().numpy(), fx["scores"])
    over, win, _ = E.status_batch(gid, H, W, g, p)
    assert int(over.sum()) == 0 and bool((win == 2).all())


@pytest.mark.parametrize("tag", list(TAGS))
def test_fixture_positions(tag):
    """random, possibly unreachable positions.  Connect4/TicTacToe: restricted to the engine's domain
    (at most one side aligned); the oracle covers the rest on the CPU."""
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"rules_{tag}.npz")
    grids = fx["pos_grids"].reshape(len(fx["pos_players"]), -1)
    players = fx["pos_players"]
    keep = np.ones(len(players), bool)
    if game != "othello":
        for i in range(len(players)):
            w = []
            for s in (1, -1):
                b = O.new_board(gid, H, W)
                b.set_grid(np.where(grids[i] == s, s, 0), 1)
                w.append(O.batch_status(gid, H, W, b.grid_np().reshape(1, -1), np.array([1], np.int8))[1][0] == s)
            keep[i] = not (w[0] and w[1])
    assert keep.sum() > 50
    grids, players = grids[keep], players[keep]
    g, p = dev(grids, torch.int8), dev(players, torch.int8)
    assert np.array_equal(E.legal_batch(gid, H, W, g, p).cpu().numpy().astype(bool), unpack_mask(fx["pos_legal"], A)[keep])
    over, win, score = E.status_batch(gid, H, W, g, p)
    assert np.array_equal(over.cpu().numpy().astype(bool), fx["pos_over"][keep])
    assert np.array_equal(win.cpu().numpy(), fx["pos_winner"][keep])
    act = fx["pos_action"][keep].astype(np.int32)
    ok = act >= 0
    og, op, st = E.play_batch(gid, H, W, g[torch.as_tensor(ok, device="cuda")].contiguous(), p[torch.as_tensor(ok, device="cuda")].contiguous(),
                              dev(act[ok], torch.int32))
    assert int(st.abs().sum()) == 0
    assert np.array_equal(og.cpu().numpy(), fx["pos_result"].reshape(len(fx["pos_players"]), -1)[keep][ok])


@pytest.mark.parametrize("tag,n_games", [("othello8", 20000), ("othello6", 8000), ("connect4", 20000), ("tictactoe", 4000)])
def test_million_positions_vs_oracle(tag, n_games):
    """>= 1e6 reachable Othello 8x8 positions: legal sets, flips, status vs the oracle"""
    game, gid, H, W, A, n = TAGS[tag]
    grids, players, actions = O.random_positions(gid, H, W, 1234, n_games, n_games * (H * W + 8))
    if tag == "othello8":
        assert len(players) >= 1_000_000
    g, p, a = dev(grids, torch.int8), dev(players, torch.int8), dev(actions, torch.int32)
    legal = E.legal_batch(gid, H, W, g, p).cpu().numpy()
    assert np.array_equal(legal, O.batch_legal(gid, H, W, grids, players))
    legal_o = E.legal_batch(gid, H, W, g, p, for_player=-p).cpu().numpy()
    assert np.array_equal(legal_o, O.batch_legal(gid, H, W, grids, players, for_player=-players))
    og, op, st = E.play_batch(gid, H, W, g, p, a)
    rg, rp, rs = O.batch_play(gid, H, W, grids, players, actions)
    assert np.array_equal(og.cpu().numpy(), rg) and np.array_equal(op.cpu().numpy(), rp) and np.array_equal(st.cpu().numpy(), rs)
    over, win, score = E.status_batch(gid, H, W, og, op)
    ro, rw, rsc = O.batch_status(gid, H, W, rg, rp)
    assert np.array_equal(over.cpu().numpy(), ro) and np.array_equal(win.cpu().numpy(), rw) and np.array_equal(score.cpu().numpy(), rsc)
    assert ro.sum() == n_games  # every playout ends exactly once


def test_empty_and_error_inputs():
    g = torch.empty((0, 64), dtype=torch.int8, device="cuda"); p = torch.empty(0, dtype=torch.int8, device="cuda")
    assert E.legal_batch(0, 8, 8, g, p).shape == (0, 65)
    with pytest.raises(ValueError, match="even"):
        E.legal_batch(0, 7, 7, g, p)


@pytest.mark.parametrize("H,W", [(4, 4), (8, 8), (5, 6), (6, 8), (8, 5)])
def test_connect4_other_board_sizes(H, W):
    """Connect4 boards from the minimum 4x4 (connect4.py:90-91) to the 8x8 maximum of the bitboards, non-square included"""
    gid = O.CONNECT4
    grids, players, actions = O.random_positions(gid, H, W, 3 + H * 8 + W, 3000, 200000)
    assert len(players) > 15000
    dg = torch.as_tensor(grids.reshape(-1, H, W), device="cuda")
    dp = torch.as_tensor(players, device="cuda")
    assert np.array_equal(E.legal_batch(gid, H, W, dg, dp).cpu().numpy(), O.batch_legal(gid, H, W, grids, players))
    og, op, ost = O.batch_play(gid, H, W, grids, players, actions)
    g, p, st = E.play_batch(gid, H, W, dg, dp, torch.as_tensor(actions, device="cuda"))
    assert np.array_equal(g.cpu().numpy().reshape(-1, H * W), og) and np.array_equal(p.cpu().numpy(), op)
    assert np.array_equal(st.cpu().numpy(), ost)
    over, win, score = E.status_batch(gid, H, W, g, p)
    oo, ow, osc = O.batch_status(gid, H, W, og, op)
    assert np.array_equal(over.cpu().numpy(), oo) and np.array_equal(score.cpu().numpy(), osc)
    assert np.array_equal(win.cpu().numpy()[oo == 1], ow[oo == 1])
    assert oo.sum() > 1000  # finished games (wins in every direction, full boards) are in the sample
