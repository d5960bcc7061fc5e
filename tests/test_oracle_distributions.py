"""CPU: the production-mode randomness of the engine/oracle pair against numpy's distributions.

GPU == oracle is bit-exact in production mode (tests/test_gpu_engine.py, test_gpu_paths.py), so the statistics are
checked once, on the oracle side: the Philox-driven draws must have the LAW of the reference's global-numpy draws --
  * root noise eta ~ np.random.dirichlet([alpha] * k)            (mcts.py:235-240): marginals Beta(alpha, (k-1) alpha)
  * fair_max ties ~ np.random.choice over the maxima, uniform     (utils.py:28-34)
  * tau = 1 move ~ np.random.choice(moves, p = N / sum N)          (mcts.py:114-116, players.py:185-189)
Seeds are fixed: the tests are deterministic; the thresholds are p > 1e-3 on exact distributions."""
import ctypes as C

import numpy as np
import pytest
from scipy import stats

from oracle import oracle as O

ALPHA = 0.03
MANY = np.array([[0, 0, 0, 0, 0, 0, 0, 0], [0, -1, -1, -1, 1, -1, -1, 0], [0, 1, 1, 0, -1, 1, 0, 1], [0, -1, -1, 1, 0, 0, -1, 0],
                 [0, 0, 1, 0, 1, 0, -1, 0], [0, -1, 0, 0, -1, 1, 1, 1], [0, -1, -1, 0, -1, 1, -1, 0], [0, -1, 1, 0, 0, 0, 0, 0]], np.int8)


def _position_with_k_moves(k):
    """(board, k) with exactly k legal moves for the side to move"""
    if k == 4:
        return O.new_board(O.OTHELLO, 8, 8)
    if k == 34:
        b = O.new_board(O.OTHELLO, 8, 8)
        b.set_grid(MANY, 1)
        return b
    if k == 1:  # TicTacToe with one empty cell, nobody aligned
        b = O.new_board(O.TICTACTOE, 3, 3)
        b.set_grid(np.array([[1, -1, 1], [1, -1, -1], [-1, 1, 0]], np.int8), 1)
        return b
    rng = np.random.default_rng(k)
    while True:
        b = O.new_board(O.OTHELLO, 8, 8)
        for _ in range(int(rng.integers(8, 40))):
            if O.lib().orc_is_over(C.byref(b)):
                break
            O.lib().orc_play(C.byref(b), int(rng.choice(O.legal_moves(b))))
        lm = O.legal_moves(b)
        if len(lm) == k and lm[0] != 64 and not O.lib().orc_is_over(C.byref(b)):
            return b


def _dirichlet_draws(b, n):
    """n independent root-noise vectors of the oracle for position b: with epsilon = 1 the children's priors after the
    noise step ARE eta ((1 - eps) P vanishes), read after the second simulation of a fresh tree (mcts.py:235-240)"""
    out = []
    for g in range(n):
        t = O.MCT(("fake", None), alpha=ALPHA, eps=1.0, tie_mode=O.TIE_RANDOM, noise_mode=O.NOISE_PHILOX, seed=77, game_id=g)
        t.search(b, 2)
        out.append(t.root_children()[3])
    return np.array(out)


@pytest.mark.parametrize("k", [1, 4, 12, 34])
def test_philox_dirichlet_has_numpys_law(k):
    b = _position_with_k_moves(k)
    assert len(O.legal_moves(b)) == k
    n = 4000 if k <= 12 else 1500
    eta = _dirichlet_draws(b, n)
    assert eta.shape == (n, k) and np.abs(eta.sum(1) - 1).max() < 1e-12 and (eta >= 0).all()
    if k == 1:
        assert (eta == 1.0).all()
        return
    ref = np.random.RandomState(5).dirichlet([ALPHA] * k, size=n)
    for j in sorted({0, 1, k // 2, k - 1}):
        # exact marginal: Beta(alpha, (k - 1) alpha)
        p_exact = stats.kstest(eta[:, j], stats.beta(ALPHA, (k - 1) * ALPHA).cdf).pvalue
        # two-sample against numpy's own generator (values below 1e-300 underflow on both sides: clip)
        p_np = stats.ks_2samp(np.clip(eta[:, j], 1e-300, 1), np.clip(ref[:, j], 1e-300, 1)).pvalue
        assert p_exact > 1e-3 and p_np > 1e-3, (k, j, p_exact, p_np)
    # the joint structure numpy's Dirichlet has at alpha = 0.03: almost all mass on one child, every child equally often
    top = np.bincount(eta.argmax(1), minlength=k)
    assert stats.chisquare(top).pvalue > 1e-3, top
    assert abs(np.mean(eta.max(1)) - np.mean(ref.max(1))) < 0.02 and abs(np.var(eta[:, 0]) - np.var(ref[:, 0])) < 0.02


@pytest.mark.parametrize("k", [4, 12, 34])
def test_fair_max_tie_choice_is_uniform(k):
    """first simulation of a fresh root: every PUCT is 0 (root N = 0), fair_max picks uniformly among ALL children
     -- the child holding the single visit after one simulation is the pick"""
    b = _position_with_k_moves(k)
    n = 6000
    picks = np.zeros(k, np.int64)
    for g in range(n):
        t = O.MCT(("fake", None), tie_mode=O.TIE_RANDOM, noise_mode=O.NOISE_OFF, seed=3, game_id=g)
        t.search(b, 1)
        N = t.root_children()[1]
        assert N.sum() == 1
        picks[int(N.argmax())] += 1
    assert stats.chisquare(picks).pvalue > 1e-3, picks
    ref = np.bincount(np.random.RandomState(1).choice(k, size=n), minlength=k)  # what np.random.choice does on n ties
    assert stats.chi2_contingency(np.stack([picks, ref]))[1] > 1e-3


def test_temperature_one_move_sampling_follows_the_visit_distribution():
    """tau = 1: the move is drawn with probability N / sum N (mcts.py:114-116, players.py:188-189); deterministic tree
    (fake network, lowest-index ties, no noise), the draw keyed by the game id"""
    b = O.new_board(O.OTHELLO, 8, 8)
    counts, pi_ref = None, None
    n = 8000
    for g in range(n):
        t = O.MCT(("fake", None), tie_mode=O.TIE_LOWEST, noise_mode=O.NOISE_OFF, seed=9, game_id=g)
        t.search(b, 60)
        act, pi, vis = t.choose(b, 1.0)
        if counts is None:
            counts, pi_ref = np.zeros(65, np.int64), pi.copy()
            assert abs(pi.sum() - 1) < 1e-12 and np.array_equal(pi > 0, vis > 0)
        assert np.array_equal(pi, pi_ref)
        counts[act] += 1
    sup = pi_ref > 0
    assert counts[~sup].sum() == 0 and sup.sum() == 4
    assert stats.chisquare(counts[sup], n * pi_ref[sup]).pvalue > 1e-3, (counts[sup], n * pi_ref[sup])
    ref = np.bincount(np.random.RandomState(2).choice(65, size=n, p=pi_ref), minlength=65)
    assert stats.chi2_contingency(np.stack([counts[sup], ref[sup]]))[1] > 1e-3


@pytest.mark.parametrize("tau", [0.75, 0.5, 0.25])
def test_fractional_temperature_move_sampling_follows_the_powered_visit_distribution(tau):
    """0 < tau < 1 (linear schedule between temp_max_step and temp_min_step, schedulers.py:33-40): the move is drawn with
    probability N ** (1 / tau) / sum (mcts.py:114-116, players.py:188-189)"""
    b = O.new_board(O.OTHELLO, 8, 8)
    n = 8000
    counts, pi_ref = np.zeros(65, np.int64), None
    for g in range(n):
        t = O.MCT(("fake", None), tie_mode=O.TIE_LOWEST, noise_mode=O.NOISE_OFF, seed=9, game_id=g)
        t.search(b, 60)
        act, pi, vis = t.choose(b, tau)
        if pi_ref is None:
            pi_ref = pi.copy()
            want = vis.astype(np.float64) ** (1.0 / tau)
            assert np.abs(pi - want / want.sum()).max() < 1e-13 and abs(pi.sum() - 1) < 1e-12
        assert np.array_equal(pi, pi_ref)
        counts[act] += 1
    sup = pi_ref > 0
    assert counts[~sup].sum() == 0 and sup.sum() == 4
    assert stats.chisquare(counts[sup], n * pi_ref[sup]).pvalue > 1e-3, (counts[sup], n * pi_ref[sup])
    ref = np.bincount(np.random.RandomState(4).choice(65, size=n, p=pi_ref), minlength=65)
    seen = sup & (counts + ref > 0)  # at tau = 0.25 the rarest move has p ~ 1e-5: a column neither generator ever draws
    assert stats.chi2_contingency(np.stack([counts[seen], ref[seen]]))[1] > 1e-3


def test_gamma_sampler_exhaustion_is_reported():
    """the 64-attempt cap of the Marsaglia-Tsang loop cannot be reached in practice (p < 1e-38 per draw): it is an error
    flag on both sides (oracle: search fails; engine: ERR_RNG -> AZ_ESTATE), never a silent g = d.  Here: the draws of
    100k noise vectors never trip it."""
    b = O.new_board(O.OTHELLO, 8, 8)
    for g in range(2000):
        t = O.MCT(("fake", None), alpha=ALPHA, eps=0.25, tie_mode=O.TIE_RANDOM, noise_mode=O.NOISE_PHILOX, seed=123, game_id=g)
        t.search(b, 2)  # raises on failure
