"""GPU parity, tree search + self-play (K3/K4/K7/K8/K9) through the C ABI.

  * G3 golden trees of the reference's MCT (fake net, deterministic ties, +/- closed-form noise):
    visit counts exact, Q and P within 1e-12.
  * G4 golden AlphaZeroTrainer.self_play sample stream.
  * engine == CPU oracle on whole self-play runs in PRODUCTION mode (random ties, Philox Dirichlet
    noise, temperature sampling): every sample (state, pi, z, visits) bit-equal.
"""
import numpy as np
import pytest

from conftest import TAGS, golden
from oracle import oracle as O
from alphazero_amd import engine as E

pytestmark = pytest.mark.gpu
MCT_TAGS = ["othello8", "othello6", "connect4", "tictactoe"]


def sort_samples(d):
    """order by (game id, move idx) -- the engine emits samples in lock-step order"""
    meta = d["meta"] if isinstance(d["meta"], np.ndarray) else d["meta"].cpu().numpy()
    order = np.lexsort((meta[:, 1], meta[:, 0]))
    return {k: (v if isinstance(v, np.ndarray) else v.cpu().numpy())[order] for k, v in d.items() if k != "n_evals"}


@pytest.mark.parametrize("tag", MCT_TAGS)
def test_mct_fixture(tag):
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"mct_{tag}.npz")
    ro, stage = fx["row_off"], fx["stage"]
    starts = np.flatnonzero(stage == 0)
    for noise in (0, 1):
        cases = [s for s in starts if fx["noise"][s] == noise]
        eng = E.SelfPlayEngine(gid, H, W, n_slots=len(cases), n_sim=100, dirichlet_alpha=0.03 if noise else None,
                               dirichlet_epsilon=0.25 if noise else None, temp_max_step=-1, temp_min_step=0,
                               tie_mode=E.TIE_LOWEST, noise_mode=E.NOISE_HASH if noise else E.NOISE_OFF,
                               evaluator=E.EVAL_FAKE, node_capacity=8192)
        eng.set_roots(np.array([fx["grids"][s] for s in cases]), np.array([fx["players"][s] for s in cases]))

        def check(rec_of_case):
            for slot, rec in enumerate(rec_of_case):
                if rec is None:
                    continue
                a, N, Q, P, rootn = eng.root_children(slot)
                sl = slice(ro[rec], ro[rec + 1])
                assert np.array_equal(a, fx["action"][sl]), (tag, rec)
                assert np.array_equal(N, fx["N"][sl]), (tag, rec, N, fx["N"][sl])
                assert rootn == fx["rootN"][rec]
                assert np.abs(Q - fx["Q"][sl]).max() <= 1e-12
                assert np.abs(P - fx["P"][sl]).max() <= 1e-12
        for k, sims in enumerate((1, 1, 8, 90)):
            eng.search(sims)
            check([s + k for s in cases])
        eng.advance()  # tau = 0 move (lowest-index tie-break), tree reuse
        eng.search(100)
        last = []
        for s in cases:
            r = s + 4
            last.append(r if r < len(stage) and stage[r] == 4 else None)
        check(last)
        smp = sort_samples(eng.samples())
        moved = [fx["moved"][r] for r in last if r is not None]
        got = [smp["meta"][i, 3] for i, r in enumerate(last) if r is not None]
        assert moved == got
        eng.close()


@pytest.mark.parametrize("name", ["selfplay", "selfplay_frac"])
@pytest.mark.parametrize("tag", MCT_TAGS)
def test_selfplay_fixture(tag, name):
    """G4 on the GPU; `selfplay_frac` = the reference's self_play under temp_max_step 2 / temp_min_step 6: plies 3-5 at
    tau = 0.75 / 0.5 / 0.25 (N ** (1 / tau), mcts.py:114-116 -> az_det_pow in k_move)"""
    game, gid, H, W, A, n = TAGS[tag]
    fx = golden(f"{name}_{tag}.npz")
    names = [str(x) for x in fx["transf_names"]]
    orig = fx["transformation"] == names.index("None")
    eng = E.SelfPlayEngine(gid, H, W, n_slots=2, n_sim=int(fx["sims"]), dirichlet_alpha=float(fx["alpha"]),
                           dirichlet_epsilon=float(fx["eps"]), temp_max_step=int(fx["temp_max_step"]),
                           temp_min_step=int(fx["temp_min_step"]), tie_mode=E.TIE_LOWEST, noise_mode=E.NOISE_HASH,
                           evaluator=E.EVAL_FAKE, seed=int(fx["seed"]), node_capacity=16384, sample_capacity=4096)
    r = sort_samples(eng.run(int(fx["episodes"])))  # 2 slots, >2 episodes: exercises slot refill
    assert np.array_equal(r["state"], fx["state"][orig])
    assert np.array_equal(r["z"], fx["outcome"][orig])
    assert np.array_equal(r["meta"][:, 0], fx["episode_idx"][orig])
    assert np.array_equal(r["meta"][:, 1], fx["move_idx"][orig])
    assert np.abs(r["pi"].astype(np.float64) - fx["pi"][orig]).max() < 1e-7
    assert eng.stats()["games_done"] == int(fx["episodes"])


@pytest.mark.parametrize("tag,n_games,n_sim,slots", [("othello8", 96, 40, 64), ("othello6", 128, 30, 50), ("othello4", 200, 25, 64),
                                                     ("connect4", 160, 50, 64), ("tictactoe", 256, 30, 100)])
def test_production_mode_equals_oracle_fakenet(tag, n_games, n_sim, slots):
    game, gid, H, W, A, n = TAGS[tag]
    eng = E.SelfPlayEngine(gid, H, W, n_slots=slots, n_sim=n_sim, evaluator=E.EVAL_FAKE, seed=9, node_capacity=32768,
                           sample_capacity=n_games * (2 * H * W))
    got = sort_samples(eng.run(n_games, first_game_id=1000))
    ref = O.selfplay(gid, H, W, n_games, n_sim, ("fake", None), seed=9, first_game_id=1000)
    st = eng.stats()
    assert st["games_done"] == n_games and st["samples"] == len(ref["z"]) and st["net_evals"] == ref["n_evals"]
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.parametrize("tag,n_games,n_sim,tmax,tmin", [("othello8", 48, 40, 2, 8), ("othello6", 64, 30, 15, 20), ("connect4", 96, 50, 1, 7),
                                                         ("tictactoe", 128, 30, 0, 5)])
def test_production_mode_fractional_temperatures_equal_oracle(tag, n_games, n_sim, tmax, tmin):
    """temp_max_step < temp_min_step - 1 (the reference's base defaults are 15 / 20, base.py:70-72): the plies in between are
    played at fractional temperatures -- pi = N ** (1 / tau) / sum through az_det_pow on the GPU and orc_det_pow on the CPU, the
    move drawn from it: samples, policies and moves bit-equal, random ties and Philox noise on"""
    game, gid, H, W, A, n = TAGS[tag]
    eng = E.SelfPlayEngine(gid, H, W, n_slots=32, n_sim=n_sim, evaluator=E.EVAL_FAKE, seed=13, node_capacity=32768,
                           temp_max_step=tmax, temp_min_step=tmin, sample_capacity=n_games * (2 * H * W))
    got = sort_samples(eng.run(n_games, first_game_id=500))
    ref = O.selfplay(gid, H, W, n_games, n_sim, ("fake", None), seed=13, first_game_id=500, temp_max_step=tmax, temp_min_step=tmin)
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k
    ply = got["meta"][:, 1]
    frac = (ply > tmax) & (ply < tmin)
    vis = got["visits"][frac].astype(np.float64)
    tau = 1.0 - (ply[frac] - tmax) / (tmin - tmax)
    want = vis ** (1.0 / tau)[:, None]
    want /= want.sum(1, keepdims=True)
    assert frac.sum() > n_games and np.abs(got["pi"][frac] - want).max() < 1e-7  # pi is stored as float32
    many = (vis > 0).sum(1) > 1
    assert many.any() and not np.allclose(got["pi"][frac][many], (vis / vis.sum(1, keepdims=True))[many])


def test_capacity_errors_are_loud():
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=8, n_sim=50, evaluator=E.EVAL_FAKE, node_capacity=256)
    with pytest.raises(E._lib.AzError, match="node pool"):
        eng.run(8)
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=8, n_sim=10, evaluator=E.EVAL_FAKE, sample_capacity=20)
    with pytest.raises(E._lib.AzError, match="sample buffer"):
        eng.run(8)
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=8, n_sim=10, evaluator=E.EVAL_FAKE, max_plies=12, sample_capacity=4096)
    with pytest.raises(E._lib.AzError, match="max_plies"):
        eng.run(8)


@pytest.mark.parametrize("tag,n_games,n_sim", [("othello8", 12, 25), ("connect4", 24, 40), ("tictactoe", 64, 25)])
def test_production_mode_equals_oracle_real_net(tag, n_games, n_sim):
    """end to end with the HIP network: engine samples == oracle samples (same weights, same accumulation order)"""
    from test_gpu_net import nets
    game, gid, H, W, A, n = TAGS[tag]
    fx, sd, onet, hnet = nets(tag)
    eng = E.SelfPlayEngine(gid, H, W, n_slots=8, n_sim=n_sim, net=hnet, seed=21, node_capacity=32768,
                           sample_capacity=n_games * (2 * H * W))
    got = sort_samples(eng.run(n_games))
    kind = "mlp" if game == "tictactoe" else "conv"
    ref = O.selfplay(gid, H, W, n_games, n_sim, (kind, onet), seed=21)
    assert len(got["z"]) == len(ref["z"])
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.parametrize("tag,n_games,n_sim,slots", [("othello8", 24, 60, 16), ("othello6", 40, 50, 40),
                                                     ("connect4", 48, 80, 32), ("tictactoe", 128, 100, 64)])
def test_rollout_mode_equals_oracle(tag, n_games, n_sim, slots):
    """TreeEval.ROLLOUT (UCT + random playouts, no network) on the device == the oracle, sample for sample"""
    game, gid, H, W, A, n = TAGS[tag]
    eng = E.SelfPlayEngine(gid, H, W, n_slots=slots, n_sim=n_sim, evaluator=E.EVAL_ROLLOUT, seed=5, node_capacity=16384,
                           sample_capacity=n_games * (2 * H * W))
    got = sort_samples(eng.run(n_games, first_game_id=77))
    ref = O.selfplay(gid, H, W, n_games, n_sim, ("fake", None), seed=5, first_game_id=77, eval_method=O.EVAL_ROLLOUT)
    assert eng.stats()["games_done"] == n_games and len(got["z"]) == len(ref["z"])
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(got[k], ref[k]), k


def _replay_check(gid, H, W, smp, n_games, n_sim, tmax):
    """size-independent properties of a self-play run, checked for every sample against the CPU rules oracle:
    consecutive states of a game are linked by its recorded action (legal, oracle play), the first state is the start
    position, z is the final winner seen from the side to move, pi is the visit distribution (tau = 1) or the one-hot
    of a most-visited move (tau = 0), and every search ran exactly n_sim simulations on top of the reused subtree"""
    st, pi, z, meta, vis = (smp[k] for k in ("state", "pi", "z", "meta", "visits"))
    assert len(np.unique(meta[:, 0])) == n_games
    first = np.flatnonzero(meta[:, 1] == 0)
    last = np.r_[first[1:], len(z)] - 1
    assert np.array_equal(meta[first, 0], np.unique(meta[:, 0]))
    assert np.array_equal(meta[:, 1], np.arange(len(z)) - np.repeat(first, last - first + 1))  # move_idx 0, 1, 2, ...
    player = meta[:, 2].astype(np.int8)
    grids = (st.reshape(len(z), -1) * player[:, None]).astype(np.int8)  # undo the normalisation
    b0 = O.new_board(gid, H, W)
    start = np.array([b0.grid[i] for i in range(H * W)], np.int8) if hasattr(b0, "grid") else None
    if start is not None:
        assert (grids[first] == start[None]).all() and (player[first] == 1).all()
    legal = O.batch_legal(gid, H, W, grids, player)
    assert legal[np.arange(len(z)), meta[:, 3]].all()  # every recorded action was legal
    ng, npl, status = O.batch_play(gid, H, W, grids, player, meta[:, 3])
    assert (status == 0).all()
    inner = np.ones(len(z), bool)
    inner[last] = False
    nxt = np.flatnonzero(inner) + 1
    assert np.array_equal(ng[inner], grids[nxt]) and np.array_equal(npl[inner], player[nxt])
    over, win, _ = O.batch_status(gid, H, W, ng[last], npl[last])
    assert over.all()  # the game ends exactly after its last sample
    assert np.array_equal(z, np.repeat(win, last - first + 1) * player)
    over_mid, _, _ = O.batch_status(gid, H, W, grids, player)
    assert not over_mid.any()
    # visit statistics
    assert (vis[~legal.astype(bool)] == 0).all()
    tot = vis.sum(1)
    assert (tot[first] == n_sim).all() and (tot >= n_sim).all()  # every simulation descends into one root child; reuse adds more
    tau1 = meta[:, 1] <= tmax
    assert np.allclose(pi[tau1], vis[tau1] / tot[tau1, None], atol=1e-6)
    arg = pi[~tau1].argmax(1)
    assert np.allclose(pi[~tau1].sum(1), 1) and (pi[~tau1].max(1) == 1).all()
    assert (vis[~tau1][np.arange(len(arg)), arg] == vis[~tau1].max(1)).all()
    assert (pi[~tau1].argmax(1) == meta[~tau1, 3]).all()  # tau = 0 plays the move it reports


def test_full_size_othello_selfplay_properties():
    """BASELINE config 2 at full size (4096 concurrent Othello 8x8 games, 100 simulations per move, real network)"""
    from alphazero_amd.games.othello import OthelloNet
    import torch
    torch.manual_seed(0)
    G, n_sim = 4096, 100
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=G, n_sim=n_sim, net=OthelloNet(n=8).eval().to_hip(max_batch=G), seed=0)
    smp = sort_samples(eng.run(G))
    stt = eng.stats()
    assert stt["games_done"] == G and stt["samples"] == len(smp["z"])
    plies = len(smp["z"]) / G
    assert 58.5 < plies < 62.0, plies  # reference: 60.3 plies per game
    assert 0.85 < stt["net_evals"] / (len(smp["z"]) * n_sim) <= 1.0  # ~0.92 evaluations per simulation
    _replay_check(0, 8, 8, smp, G, n_sim, tmax=4)


def test_full_size_connect4_selfplay_properties():
    """BASELINE config 4 at full size (8192 concurrent Connect4 6x7 games, 200 simulations per move)"""
    from alphazero_amd.games.connect4 import Connect4Net
    import torch
    torch.manual_seed(0)
    G, n_sim = 8192, 200
    eng = E.SelfPlayEngine(1, 6, 7, n_slots=G, n_sim=n_sim, net=Connect4Net(7, 6).eval().to_hip(max_batch=G), seed=0)
    smp = sort_samples(eng.run(G))
    assert eng.stats()["games_done"] == G
    plies = len(smp["z"]) / G
    assert 24 < plies < 31, plies  # reference: 27 plies per game
    _replay_check(1, 6, 7, smp, G, n_sim, tmax=4)


def test_results_do_not_depend_on_the_batch_shape():
    """the same game ids played 4096 at a time (two-board trunk kernel, 128-wide GEMM tiles, slot = game) and 384 at a
    time (one-board trunk kernel, small tiles, slots refilled ~11 times) give the same samples bit for bit: a game's
    trajectory depends on its id only (Philox keyed by game id; every network row depends on its board only)"""
    from alphazero_amd.games.othello import OthelloNet
    import torch
    torch.manual_seed(0)
    model = OthelloNet(n=8).eval()
    G, n_sim = 4096, 30
    big = E.SelfPlayEngine(0, 8, 8, n_slots=G, n_sim=n_sim, net=model.to_hip(max_batch=G), seed=3)
    a = sort_samples(big.run(G, first_game_id=500))
    small = E.SelfPlayEngine(0, 8, 8, n_slots=384, n_sim=n_sim, net=model.to_hip(max_batch=384), seed=3, sample_capacity=G * 70)
    b = sort_samples(small.run(G, first_game_id=500))
    assert len(a["z"]) == len(b["z"])
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(a[k], b[k]), k


def test_searches_are_replayed_as_hip_graphs():
    """a search (1 + 5 n_sim launches) is captured the second time it is issued with the same shape and replayed from
    then on; the samples do not depend on it"""
    import os
    import subprocess
    import sys
    game, gid, H, W, A, n = TAGS["othello6"]
    eng = E.SelfPlayEngine(gid, H, W, n_slots=32, n_sim=20, evaluator=E.EVAL_FAKE, seed=1, node_capacity=8192)
    a = sort_samples(eng.run(32))
    st = eng.stats()
    assert st["graph_replays"] >= st["plies"] // 32 - 2 > 10  # every ply but the first two
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import numpy as np\n"
            "from alphazero_amd import engine as E\nfrom test_gpu_engine import sort_samples\n"
            "eng = E.SelfPlayEngine(%d, %d, %d, n_slots=32, n_sim=20, evaluator=E.EVAL_FAKE, seed=1, node_capacity=8192)\n"
            "s = sort_samples(eng.run(32)); assert eng.stats()['graph_replays'] == 0\n"
            "np.savez(sys.argv[1], **s)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), gid, H, W)
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "az_nograph_%d.npz" % os.getpid())
    subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, AZ_ENGINE_GRAPHS="0"))
    b = np.load(out)
    os.remove(out)
    for k in ("state", "z", "meta", "visits", "pi"):
        assert np.array_equal(a[k], b[k]), k


def test_randomised_configurations_equal_oracle():
    """tools/fuzz_engine.py: 150 random configurations -- game, board size (Othello 4 / 6 / 8, Connect4 4x4 .. 8x8, TicTacToe), slot and
    game counts (slot refill, slot counts that do not fill a wavefront), simulations, Dirichlet alpha / epsilon, temperature schedule
    (incl. fractional temperatures), tie and noise modes, network-less rollout mode, seed: every sample array bit-equal to the oracle"""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_engine
    assert fuzz_engine.run(150, seed=2024, verbose=False) == []
