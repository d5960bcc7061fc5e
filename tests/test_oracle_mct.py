"""G3 / G4: the oracle's tree search and self-play loop against the reference's MCT and
AlphaZeroTrainer.self_play under the closed-form fake net and deterministic tie-breaks.
Visit counts exact, Q within 1e-12, priors within 1e-12 (dyadic fake priors make them exact)."""
import numpy as np
import pytest

from conftest import TAGS, golden
from oracle import oracle as O

MCT_TAGS = ["othello8", "othello6", "connect4", "tictactoe"]


def run_case(gid, H, W, grid, player, noise):
    b = O.new_board(gid, H, W)
    b.set_grid(grid, player)
    t = O.MCT(("fake", None), alpha=0.03 if noise else -1.0, eps=0.25 if noise else -1.0, tie_mode=O.TIE_LOWEST,
              noise_mode=O.NOISE_HASH if noise else O.NOISE_OFF)
    out = []
    for s in (1, 1, 8, 90):
        t.search(b, s)
        out.append((t.root_children(), t.root_n(), -1))
    act, pi, vis = t.choose(b, 0.0)
    assert O.lib().orc_play(O.C.byref(b), act) == 0
    t.change_root(act)
    if not O.lib().orc_is_over(O.C.byref(b)):
        t.search(b, 100)
        out.append((t.root_children(), t.root_n(), act))
    return out


@pytest.mark.parametrize("tag", MCT_TAGS)
def test_mct_fixture(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"mct_{tag}.npz")
    ro = fx["row_off"]
    rec = 0
    n_rec = len(fx["stage"])
    checked = 0
    while rec < n_rec:
        assert fx["stage"][rec] == 0
        grid, player, noise = fx["grids"][rec], int(fx["players"][rec]), int(fx["noise"][rec])
        res = run_case(gid, H, W, grid, player, noise)
        k = 0
        while rec < n_rec and (k == 0 or fx["stage"][rec] != 0):
            (a, N, Q, P), rootn, moved = res[k]
            sl = slice(ro[rec], ro[rec + 1])
            assert np.array_equal(a, fx["action"][sl]), (tag, rec)
            assert np.array_equal(N, fx["N"][sl]), (tag, rec)
            assert rootn == fx["rootN"][rec]
            assert moved == fx["moved"][rec]
            assert np.abs(Q - fx["Q"][sl]).max() <= 1e-12
            assert np.abs(P - fx["P"][sl]).max() <= 1e-12
            rec += 1
            k += 1
            checked += 1
        assert k == len(res)
    assert checked == n_rec and checked > 100


@pytest.mark.parametrize("name", ["selfplay", "selfplay_frac"])
@pytest.mark.parametrize("tag", MCT_TAGS)
def test_selfplay_fixture(tag, name):
    """AlphaZeroTrainer.self_play (trainer.py:215-273) sample stream, un-augmented part.  `selfplay_frac`: the same under
    temp_max_step = 2, temp_min_step = 6 -- plies 3, 4, 5 are played at tau = 0.75, 0.5, 0.25, where the reference's pi is
    N ** (1 / tau) / sum (mcts.py:114-116, schedulers.py:33-40) and the oracle's is orc_det_pow."""
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"{name}_{tag}.npz")
    if name == "selfplay_frac":
        assert (int(fx["temp_max_step"]), int(fx["temp_min_step"])) == (2, 6)
        frac = (fx["move_idx"] > 2) & (fx["move_idx"] < 6) & (fx["transformation"] == [str(x) for x in fx["transf_names"]].index("None"))
        # the fixture does hold fractional-temperature policies: neither one-hot nor proportional to the visits alone
        assert frac.sum() >= 3 * int(fx["episodes"]) - 3 and ((fx["pi"][frac] > 0) & (fx["pi"][frac] < 1)).sum() > frac.sum()
    names = [str(x) for x in fx["transf_names"]]
    orig = fx["transformation"] == names.index("None")
    r = O.selfplay(gid, H, W, int(fx["episodes"]), int(fx["sims"]), ("fake", None), alpha=float(fx["alpha"]),
                   eps=float(fx["eps"]), temp_max_step=int(fx["temp_max_step"]), temp_min_step=int(fx["temp_min_step"]),
                   tie_mode=O.TIE_LOWEST, noise_mode=O.NOISE_HASH, seed=int(fx["seed"]))
    S = int(orig.sum())
    assert len(r["z"]) == S
    assert np.array_equal(r["state"], fx["state"][orig])
    assert np.array_equal(r["z"], fx["outcome"][orig])
    assert np.array_equal(r["meta"][:, 0], fx["episode_idx"][orig])
    assert np.array_equal(r["meta"][:, 1], fx["move_idx"][orig])
    assert np.all(fx["player"] == 1)
    assert np.abs(r["pi"].astype(np.float64) - fx["pi"][orig]).max() < 1e-7  # pi is stored as float32


def test_det_pow_against_libm():
    """orc_det_pow (= az_det_pow of the HIP engine, bit for bit) against the pow the reference calls (Python float ** float =
    libm): relative distance below 1e-13 for every visit count up to 2000 and every temperature a linear schedule produces
    with up to 64 steps between temp_max_step and temp_min_step; exact at the edges (0 visits -> 0, tau = 1 is not routed here)"""
    L = O.lib()
    worst = 0.0
    for tmax, tmin in [(2, 6), (15, 20), (0, 3), (1, 50), (0, 64)]:
        for step in range(tmax + 1, tmin):
            t = 1 - (step - tmax) / (tmin - tmax)
            for n in list(range(1, 2000, 7)) + [1, 2, 3, 99, 100, 199, 200, 1999]:
                try:
                    want = n ** (1. / t)
                except OverflowError:
                    continue
                worst = max(worst, abs(L.orc_det_pow(float(n), 1. / t) - want) / want)
    assert worst < 1e-13, worst
    assert L.orc_det_pow(0.0, 4.0) == 0.0 and L.orc_det_pow(1.0, 1. / 0.75) == 1.0


def test_tree_invariants_random_mode():
    """N(root) = sum N(children) on a fresh root; production (random) mode runs."""
    b = O.new_board(O.OTHELLO, 8, 8)
    t = O.MCT(("fake", None), alpha=0.03, eps=0.25, tie_mode=O.TIE_RANDOM, noise_mode=O.NOISE_PHILOX, seed=3)
    t.search(b, 100)
    a, N, Q, P = t.root_children()
    assert t.root_n() == 100 == N.sum()
    assert abs(P.sum() - 1.0) < 1e-6
    r1 = O.selfplay(O.OTHELLO, 8, 8, 2, 20, seed=5)
    r2 = O.selfplay(O.OTHELLO, 8, 8, 2, 20, seed=5)
    r3 = O.selfplay(O.OTHELLO, 8, 8, 2, 20, seed=6)
    assert np.array_equal(r1["pi"], r2["pi"]) and np.array_equal(r1["state"], r2["state"])
    assert not np.array_equal(r1["meta"][:, 3][:len(r3["meta"])], r3["meta"][:, 3][:len(r1["meta"])])


def test_rollout_tictactoe_stats():
    """BASELINE config 1 (TicTacToe, rollout MCTS, 100 sims, temp 0): outcome mix near the reference's."""
    fx = golden("stats.npz")
    n = 400
    r = O.selfplay(O.TICTACTOE, 3, 3, n, 100, alpha=-1, eps=-1, temp_max_step=-1, temp_min_step=0,
                   tie_mode=O.TIE_RANDOM, noise_mode=O.NOISE_OFF, seed=11, eval_method=O.EVAL_ROLLOUT)
    last = np.flatnonzero(np.r_[r["meta"][1:, 1] == 0, True])
    first = np.flatnonzero(r["meta"][:, 1] == 0)
    assert len(first) == n
    winner = r["z"][first] * r["meta"][first, 2]  # z is normalised by the player to move
    draw = np.mean(winner == 0)
    ref_draw = int(fx["ttt_rollout_draw"]) / int(fx["ttt_rollout_games"])
    assert abs(draw - ref_draw) < 0.12, (draw, ref_draw)
    plies = (last - first + 1).mean()
    assert abs(plies - float(fx["ttt_rollout_mean_plies"])) < 0.5


def test_games_and_arena_rounds_are_independent_given_their_ids():
    """what the sampled large-batch GPU tests rest on (tests/test_gpu_paths.py, test_gpu_arena.py): a game of a self-play wave and a
    round of an arena depend on (seed, game id / round) only, so replaying one of them alone gives exactly its rows of the whole run"""
    gid, H, W = O.TICTACTOE, 3, 3
    whole = O.selfplay(gid, H, W, 6, 12, ("fake", None), seed=3, first_game_id=40)
    for g in (40, 43, 45):
        alone = O.selfplay(gid, H, W, 1, 12, ("fake", None), seed=3, first_game_id=g)
        rows = whole["meta"][:, 0] == g
        assert rows.sum() == len(alone["meta"]) > 0
        for k in ("state", "z", "meta", "visits", "pi"):
            assert np.array_equal(whole[k][rows], alone[k]), (g, k)
    dims = (O.CONNECT4, 4, 4)
    for opponent in ("random", "greedy", "mcts"):
        moves, winners, scores, stats = O.arena_games(dims, ("fake", None), 10, opponent, 8, 5, 6)
        picks = (4, 1, 5)
        m2, w2, s2, st2 = O.arena_games(dims, ("fake", None), 10, opponent, 8, 5, 6, rounds=picks)
        assert m2 == [moves[r] for r in picks] and w2 == [winners[r] for r in picks] and s2 == [scores[r] for r in picks]
        assert len(st2["player1"]) + len(st2["player2"]) + st2["draw"] == len(picks)
